#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV per (kernel, grid size): calls, mean duration."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if flt in n:
        short = n.split("(anonymous namespace)::")[-1].split("(")[0][:48]
        agg[(short, r["Grid_Size_X"], r["Grid_Size_Y"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items()):
    print(f"{k[0]:50s} grid=({k[1]},{k[2]}) calls={len(v):4d} mean={sum(v) / len(v) / 1e3:9.1f} us")
