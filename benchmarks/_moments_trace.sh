#!/bin/bash
# kernel durations of the moments kernels per workgroup count (rocprofv3 kernel trace); run on the GPU box through gpurun
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for w in ${WGS:-256 512 1024}; do
  rm -rf /tmp/mt_$w
  CHX_TUNE_MOMENTS_WGS=$w rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mt_$w -o m -- python $REPO/benchmarks/moments_bench.py > /tmp/mt_$w.log 2>&1
  echo "== WGS=$w"
  f=$(find /tmp/mt_$w -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "moments" in r["Name"]:
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us min {float(r["MinNs"])/1e3:8.2f} max {float(r["MaxNs"])/1e3:8.2f}')
P
done
