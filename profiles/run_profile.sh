#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root. Writes raw rocprofv3 output to gpurun_out/.
# 1) kernel trace + stats of the benchmark command; 2) PMC passes (FETCH_SIZE, WRITE_SIZE separately,
#    as MI355X_MICROARCH.md prescribes: TCC has 4 slots, FETCH_SIZE costs 3 and WRITE_SIZE 2).
set -x
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${1:-r01}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- \
    python $REPO/bench.py --steps 200 --warmup 10 --no-cpu-baseline > $OUT/bench_under_profiler.json 2> $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o probe -- \
    python $REPO/profiles/traffic_probe.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o probe -- \
    python $REPO/profiles/traffic_probe.py > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -40
