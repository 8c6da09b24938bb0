#!/bin/bash
# Round 6: C4 against the deposit's share size (slots of a tile per workgroup; CHX_TUNE_DEPOSIT_SHARE, default 2048)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/c4_deposit_share
rm -rf $OUT; mkdir -p $OUT
for share in 2048 1024 512 256 128 2048 512; do
  CHX_TUNE_DEPOSIT_SHARE=$share python benchmarks/run_configs.py c4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('share=$share C4 track %.3f ms, single kick %.3f ms' % (d['track_ms'], d['single_kick_ms']))"
done | tee $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
for share in 2048 512 256; do
  CHX_TUNE_DEPOSIT_SHARE=$share timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$share -o c4 -- \
      python $REPO/benchmarks/run_configs.py c4 > $OUT/bench$share.log 2> $OUT/trace$share.log
  echo "share=$share kernel stats" | tee -a $OUT/ab.txt
  python $REPO/benchmarks/_show_stats.py $OUT/trace$share 3 2>/dev/null | tee -a $OUT/ab.txt
  rm -rf $OUT/trace$share
done
