#!/bin/bash
# Round 6, on the GPU box (through gpurun) from the repo root: raw rocprofv3 output -> gpurun_out/prof_r06*
#  1) kernel trace + stats of bench.py (headline + configs), 2) PMC passes of the apply kernel (FETCH_SIZE, WRITE_SIZE separately),
#  3) kernel stats of C4 alone (run_configs c4), 4) kernel stats of C5 alone (run_configs c5)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r06
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
    env CHX_BENCH_NO_PMC=1 python $REPO/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-scaling-legs > $OUT/bench_under_profiler.json 2> $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o probe -- \
    python $REPO/profiles/traffic_probe.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o probe -- \
    python $REPO/profiles/traffic_probe.py > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -o c4 -- python $REPO/benchmarks/run_configs.py c4 > $OUT/c4_bench.log 2> $OUT/c4_trace.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -o c5 -- python $REPO/benchmarks/run_configs.py c5 > $OUT/c5_bench.log 2> $OUT/c5_trace.log
cp $(find $OUT/c4 -name "*kernel_stats.csv" | head -1) $OUT/c4_kernel_stats.csv
cp $(find $OUT/c5 -name "*kernel_stats.csv" | head -1) $OUT/c5_kernel_stats.csv
# keep what the summaries need, drop the bulky traces
find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls -la $OUT $OUT/trace | head -40
tail -1 $OUT/c4_bench.log; tail -1 $OUT/c5_bench.log
