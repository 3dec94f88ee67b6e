#!/bin/bash
# Everything profiles/ holds for one round, in one gpurun call (run on the GPU box from the repo
# root): the bench line, the rocprofv3 kernel-trace stats of the same command, the HBM-traffic
# counters (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, never with a trace domain other
# than --kernel-trace) and the full PMC sweep of the product kernels.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
RROI_BENCH_TRAFFIC=0 RROI_BENCH_E2E=0 RROI_BENCH_SENSITIVITY=0 RROI_BENCH_TRAIN=0 RROI_BENCH_BIG=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- \
    python bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
# the driver's flags (K = 20 after W = 5), for the record
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line_driver_flags.json 2>> $OUT/bench.err
cp $OUT/stats/*kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null || \
    find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/traffic/$c -o $c -- \
      ./tools/kbench pmc product > $OUT/traffic_$c.log 2>&1
done
python3 tools/pmc_summary.py $OUT/traffic > $OUT/pmc_traffic.md 2>&1
bash tools/run_pmc.sh "./tools/kbench pmc product" > /dev/null 2>&1
cp gpurun_out/pmc/summary.md $OUT/pmc_product_kernels.md
cat $OUT/bench_line.json; head -8 $OUT/bench_kernel_stats.csv; cat $OUT/pmc_traffic.md
