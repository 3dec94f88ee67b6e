# rocprofv3 kernel traces of the reference's training shapes (C = 64, two 120 x 160 maps): forward (tools/train_fwd_profile.py)
# and backward (tools/bwd_train_paths_cmd.py, R = 512 and 32) -> gpurun_out/r05_train/*.md
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_train
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/fwd -o kt -- python tools/train_fwd_profile.py > $O/fwd.log 2>&1
python3 tools/rocpd_summary.py $O/fwd/kt_results.db > $O/train_fwd_kernel_stats.md
for R in 512 32; do
  rocprofv3 --kernel-trace -d $O/bwd$R -o kt -- python tools/bwd_train_paths_cmd.py $R 96 > $O/bwd$R.log 2>&1
  python3 tools/rocpd_summary.py $O/bwd$R/kt_results.db > $O/train_bwd_R${R}_kernel_stats.md
done
cut -c1-160 $O/train_fwd_kernel_stats.md | head -12; cut -c1-160 $O/train_bwd_R512_kernel_stats.md | head -14
