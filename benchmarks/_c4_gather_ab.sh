#!/bin/bash
# Round 6: C4 with the float32 particle step of the gather pass (default) against the float64 step of rounds 1-5
# (CHX_SC_GATHER_FP64=1): time per track, the kick's error against the reference's float64 run, kernel stats of both.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/c4_gather_ab
rm -rf $OUT; mkdir -p $OUT
for mode in 1 0 1 0; do
  CHX_SC_GATHER_FP64=$mode python benchmarks/run_configs.py c4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp64step=$mode C4 track %.3f ms, single kick %.3f ms' % (d['track_ms'], d['single_kick_ms']))"
done | tee $OUT/ab.txt
for mode in 1 0; do
  echo "fp64step=$mode" | tee -a $OUT/ab.txt
  CHX_SC_GATHER_FP64=$mode python benchmarks/sc_fp32_error.py 2>&1 | grep f32 | tee -a $OUT/ab.txt
done
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  CHX_SC_GATHER_FP64=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$mode -o c4 -- \
      python $REPO/benchmarks/run_configs.py c4 > $OUT/bench$mode.log 2> $OUT/trace$mode.log
  f=$(find $OUT/trace$mode -name "*kernel_stats.csv" | head -1)
  echo "fp64step=$mode kernel stats" | tee -a $OUT/ab.txt
  python $REPO/benchmarks/_show_stats.py $OUT/trace$mode 16 2>/dev/null | head -16 | tee -a $OUT/ab.txt
  cp $f $OUT/kernel_stats_fp64step$mode.csv
  rm -rf $OUT/trace$mode
done
