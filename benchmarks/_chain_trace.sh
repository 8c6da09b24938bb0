#!/bin/bash
# kernel durations of the in-register chains (second-order and drift-kick-drift FODO100, 1e6 particles): rocprofv3 kernel trace
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for b in second_order_lattice dkd_lattice; do
  rm -rf /tmp/ch_$b
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ch_$b -o m -- python $REPO/benchmarks/$b.py > /tmp/ch_$b.log 2>&1
  echo "== $b"; grep -v "^[WE]2026\|amdgpu.ids" /tmp/ch_$b.log | head -6
  python3 - "$b" <<'P'
import csv, glob, sys
f = sorted(glob.glob(f"/tmp/ch_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True))
if not f:
    print("no kernel_stats.csv"); raise SystemExit
for r in csv.DictReader(open(f[0])):
    if "chain" in r["Name"] or "path_length" in r["Name"]:
        print(f'{r["Name"][:110]:110s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.2f} us min {float(r["MinNs"])/1e3:9.2f} max {float(r["MaxNs"])/1e3:9.2f}')
P
done
