#!/bin/bash
# Reduced counter set (write path): run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/pmc2
rm -rf $OUT; mkdir -p $OUT
CMD="${1:-./tools/kbench pmc}"
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- $CMD > $OUT/p$i.log 2>&1
done <<'SETS'
TCC_REQ_sum TCC_WRITE_sum TCC_HIT_sum TCC_MISS_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_BUSY_sum TCC_EA0_WRREQ_STALL_sum
TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TOTAL_WRITE_sum TCP_TCC_READ_REQ_sum
TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum
GRBM_GUI_ACTIVE GRBM_TC_BUSY
SETS
python3 tools/pmc_summary.py $OUT > $OUT/summary.md 2>&1
cat $OUT/summary.md
