#!/bin/bash
# kernel durations of the second-order lattice benchmark per tuning mode (rocprofv3 kernel trace)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for m in ${MODES:-0 1 3}; do
  rm -rf /tmp/so_$m
  CHX_TUNE_SO_MODE=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/so_$m -o m -- python $REPO/benchmarks/second_order_lattice.py > /tmp/so_$m.log 2>&1
  echo "== mode $m"; python3 $REPO/benchmarks/_show_stats.py /tmp/so_$m 4 | cut -c1-70,100-150
done
