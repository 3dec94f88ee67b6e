# write-path counters of the merging form against the SHIFT form on the C = 64 training shapes -> gpurun_out/merge_pmc/*.md
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/merge_pmc
rm -rf $O; mkdir -p $O
for m in 1 0; do
  mkdir -p $O/m$m
  for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_WRITE_sum TCC_REQ_sum"; do
    d=$O/m$m/$(echo $c | tr ' ' '_')
    RROI_MERGE=$m timeout 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python tools/merge_pmc_cmd.py > $d.log 2>&1
  done
  python3 tools/pmc_summary.py $O/m$m > $O/merge$m.md 2>&1
done
cat $O/merge1.md $O/merge0.md
