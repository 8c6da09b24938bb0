#!/bin/bash
# Round 6: C5 (forward + backward through [Drift, Quad(k1), Drift, Screen], 1e6 particles): host stages and the kernel stats
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/c5_prof
rm -rf $OUT; mkdir -p $OUT
python benchmarks/c5_stages.py 2>&1 | grep -E "host us|whole step" | tee $OUT/stages.txt
python benchmarks/c5_graph.py 2>&1 | tail -2 | tee -a $OUT/stages.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c5 -- python $REPO/benchmarks/run_configs.py c5 > $OUT/bench.log 2> $OUT/trace.log
python $REPO/benchmarks/_show_stats.py $OUT/trace 14 | tee $OUT/kernels.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/c5_kernel_stats.csv
rm -rf $OUT/trace
tail -1 $OUT/bench.log
