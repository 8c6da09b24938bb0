#!/bin/bash
# two-stream timeline of the last chain kicks of C4 (rocprofv3 kernel trace -> profiles/c4_timeline.py)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/c4_tl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o c4 -- python $REPO/benchmarks/run_configs.py c4 > $OUT/bench.log 2> $OUT/trace.log
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python $REPO/profiles/c4_timeline.py $f 3 > $OUT/timeline.txt
tail -3 $OUT/bench.log
cat $OUT/timeline.txt
