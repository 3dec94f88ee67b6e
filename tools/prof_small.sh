cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_small
rm -rf $O; mkdir -p $O
for R in 8 32 64; do
  rocprofv3 --kernel-trace --stats -d $O/kt$R -o kt -- python tools/small_fwd_cmd.py $R 1 2 > $O/kt$R.log 2>&1
done
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TA_BUSY_sum TA_TA_BUSY_sum" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p$i -- python tools/small_fwd_cmd.py 32 1 > $O/p$i.log 2>&1
done
python3 tools/pmc_summary.py $O > $O/pmc_summary.md 2>&1
for R in 8 32 64; do echo "== R=$R"; find $O/kt$R -name "*kernel_stats.csv" | head -1 | xargs cat | cut -d, -f1-8 | head -8; done
cat $O/pmc_summary.md | head -60
