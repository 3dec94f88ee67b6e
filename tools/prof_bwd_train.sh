cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_bwd_train
rm -rf $O; mkdir -p $O
for R in 512 32; do
  rocprofv3 --kernel-trace -d $O/kt$R -o kt -- python tools/bwd_train_paths_cmd.py $R 96 > $O/kt$R.log 2>&1
  echo "== R=$R"; python3 tools/rocpd_summary.py $O/kt$R/kt_results.db | cut -c1-150 | grep "rroi"
done
