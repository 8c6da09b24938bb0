#!/bin/bash
# kernel durations of C1 (README segment + screen reading): rocprofv3 kernel stats
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/c1_prof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python $REPO/benchmarks/run_bench_config.py C1 > $OUT/bench.log 2> $OUT/trace.log
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:6]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:7.1f}")
PY
grep -o '"ms_[a-z_]*": [0-9.]*' $OUT/bench.log | tr '\n' ' '; echo
