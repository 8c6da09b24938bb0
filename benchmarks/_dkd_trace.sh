#!/bin/bash
# kernel durations of the drift-kick-drift kernels (rocprofv3 kernel trace around benchmarks/second_order_lattice.py)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dkd_t
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dkd_t -o m -- python $REPO/benchmarks/second_order_lattice.py > /tmp/dkd_t.log 2>&1
python3 $REPO/benchmarks/_show_stats.py /tmp/dkd_t 12 | grep -i "dkd\|apply_tile\|second" | cut -c1-110,100-150
