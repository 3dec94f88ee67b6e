cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_bwd_builds
rm -rf $O; mkdir -p $O
for w in prev new; do for R in ${RS:-512 32}; do
  rocprofv3 --kernel-trace -d $O/$w$R -o kt -- python tools/bwd_prev_cmd.py $w $R > $O/$w$R.log 2>&1
  echo "== $w R=$R"; python3 tools/rocpd_summary.py $O/$w$R/kt_results.db | cut -c1-150 | grep rroi
done; done
