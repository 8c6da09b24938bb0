#!/bin/bash
# kernel durations of the one-element drift-kick-drift kernels (rocprofv3 kernel trace around benchmarks/dkd_single.py)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dkd_t
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dkd_t -o m -- python $REPO/benchmarks/dkd_single.py > /tmp/dkd_t.log 2>&1
python3 $REPO/benchmarks/_show_stats.py /tmp/dkd_t 12 | grep -i "dkd" | cut -c1-90,100-150
