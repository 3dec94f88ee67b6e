cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/bwdround
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bwd -- python tools/bwd_profile.py > $OUT/times.txt 2>&1
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/bwd_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  RROI_BWD_ONLY=1 timeout 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/traffic/$c -o $c -- python tools/bwd_profile.py > $OUT/traffic_$c.log 2>&1
done
python3 tools/pmc_summary.py $OUT/traffic > $OUT/pmc_bwd_traffic.md 2>&1
RROI_BWD_ONLY=1 bash tools/run_pmc.sh "python tools/bwd_profile.py" > /dev/null 2>&1
cp gpurun_out/pmc/summary.md $OUT/pmc_bwd_kernels.md
rm -rf $OUT/stats $OUT/traffic gpurun_out/pmc
python3 tools/bwd_traffic_json.py $OUT/pmc_bwd_traffic.md 6 > $OUT/bwd_traffic.json
cat $OUT/pmc_bwd_traffic.md | head -40
