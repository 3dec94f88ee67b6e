cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_bwd0
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o kt -- python tools/train_bwd_profile.py > $O/kt.log 2>&1
python3 tools/rocpd_summary.py $O/kt/kt_results.db | cut -c1-200 | head -30
