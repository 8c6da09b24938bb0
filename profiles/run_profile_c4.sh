#!/bin/bash
# Kernel trace + stats of config C4 (50-element linac with 10 SpaceChargeKicks at 128^3, 1e6 particles):
#   prof_<round>_c4         as shipped (Green-function chain on the side stream, overlapping the deposit)
#   prof_<round>_c4_serial  the same with everything on one stream: per-kernel durations in isolation
set -x
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${1:-r01}_c4
mkdir -p $OUT ${OUT}_serial
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c4 -- \
    python $REPO/benchmarks/run_configs.py c4 > $OUT/bench.log 2> $OUT/trace.log
rocprofv3 --kernel-trace --stats --output-format csv -d ${OUT}_serial/trace -o c4 -- \
    python $REPO/benchmarks/c4_serial.py > ${OUT}_serial/bench.log 2> ${OUT}_serial/trace.log
