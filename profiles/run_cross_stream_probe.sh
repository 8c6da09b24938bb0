#!/bin/bash
# Timeline of benchmarks/cross_stream_probe.hip (events vs hipStreamWaitValue32) under the kernel tracer. Needs build/cross_stream_probe.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
  timeout 120 $R/build/cross_stream_probe $mode
  timeout 180 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/xs_$mode -o x -- $R/build/cross_stream_probe $mode > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, statistics
for mode in (0, 1):
    try:
        rows = list(csv.DictReader(open(f"gpurun_out/xs_{mode}/x_kernel_trace.csv")))
    except FileNotFoundError:
        print("mode", mode, "no trace"); continue
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    pw, pc, wa = [], [], []
    last_p = None
    for i, r in enumerate(rows):
        n = r["Kernel_Name"]
        if "busy_P" in n:
            last_p = int(r["End_Timestamp"]); seen_c = False
            if i: wa.append((int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1000)
        elif last_p is not None and "busy_W" in n:
            pw.append((int(r["Start_Timestamp"]) - last_p) / 1000)
        elif last_p is not None and "busy_C" in n and not seen_c:
            pc.append((int(r["Start_Timestamp"]) - last_p) / 1000); seen_c = True
    med = lambda v: round(statistics.median(v[20:]), 2) if len(v) > 40 else None
    print("mode", "events    " if mode == 0 else "wait-value", "P.end -> W.start", med(pw), "us   P.end -> C.start", med(pc), "us   previous kernel -> P (join)", med(wa), "us")
PY
