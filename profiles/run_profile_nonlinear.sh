#!/bin/bash
# Kernel trace + stats of the non-linear tracking kernels (benchmarks/nonlinear_bench.py) -> gpurun_out/prof_<tag>_nl
set -x
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${1:-r01}_nl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o nl -- \
    python $REPO/benchmarks/nonlinear_bench.py > $OUT/bench.log 2> $OUT/trace.log
find $OUT -name "*.csv" | head
