#!/bin/bash
# parameter_lattice_kernel with / without the staged tables: kernel duration over bench.py's DIAGNOSTICS_LATTICES + the control steps
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  OUT=$REPO/gpurun_out/param_prof_$m; rm -rf $OUT; mkdir -p $OUT
  CHX_TUNE_PARAMETER_STAGED=$m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python $REPO/benchmarks/run_bench_config.py DIAGNOSTICS_LATTICES > $OUT/bench.log 2> $OUT/trace.log
  grep -o '"parameter_beam_us": [0-9.]*' $OUT/bench.log | tr '\n' ' '; echo
done
