#!/bin/bash
# per-kernel durations of bench.py's DIAGNOSTICS_LATTICES entries (rocprofv3 kernel stats)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/diag_prof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o diag -- python $REPO/benchmarks/run_bench_config.py DIAGNOSTICS_LATTICES > $OUT/bench.log 2> $OUT/trace.log
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:22]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
