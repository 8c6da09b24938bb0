#!/bin/bash
# Kernel trace + stats of config C4 (round 5: chain of tile-ordered kicks with riders) -> gpurun_out/prof_r05_c4
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${1:-r05}_c4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c4 -- \
    python $REPO/benchmarks/run_configs.py c4 > $OUT/bench.log 2> $OUT/trace.log
