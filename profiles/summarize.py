#!/usr/bin/env python3
"""Condense raw rocprofv3 CSVs (gpurun_out/prof_<round>/) into the tracked summaries under profiles/:
   profiles/<round>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py` (libchx kernels + top torch kernels)
   profiles/<round>_pmc_apply.md       FETCH_SIZE / WRITE_SIZE per launch of the apply kernel, with the gfx950
                                       corrections of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests at 64 B
                                       for wide coalesced reads -> x2; unit KiB)
   profiles/apply_traffic.json         read by bench.py for roofline.traffic
usage: python profiles/summarize.py r01
"""
import collections
import csv
import json
import os
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}")
out = os.path.join(ROOT, "profiles")

rows = list(csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_stats.csv"))))
with open(os.path.join(out, f"{rnd}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows[:45]:
        name = r["Name"]
        if len(name) > 160:
            name = name[:157] + "..."
        w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])

agg = collections.defaultdict(list)
for which in ("fetch", "write"):
    for r in csv.DictReader(open(os.path.join(src, f"pmc_{which}", "probe_counter_collection.csv"))):
        if "apply_tile_kernel" in r["Kernel_Name"] or "apply_wave_kernel" in r["Kernel_Name"]:
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            # rows per lane = PPT template argument: apply_tile_kernel<T, PPT, MODE>
            ppt = int(r["Kernel_Name"].split("<")[1].split(",")[1])
            agg[(int(r["Grid_Size"]), r["Counter_Name"], ppt)].append((float(r["Counter_Value"]), dur))
lines = ["| grid (threads) | particles | counter | raw per launch (KiB) | corrected bytes per launch | algorithmic bytes | ratio |",
         "|---|---|---|---|---|---|---|"]
traffic = {}
for (grid, cname, ppt), vals in sorted(agg.items()):
    n_part = {(500224, 2): 1_000_000}.get((grid, ppt), grid * ppt)  # 1e6 rows fill 1954 tiles of 512 (last partial)
    raw = sum(v for v, _ in vals) / len(vals)
    corr = raw * 1024 * (2.0 if cname == "FETCH_SIZE" else 1.0)
    algo = 28.0 * n_part
    lines.append(f"| {grid} | {n_part} | {cname} | {raw:.2f} | {corr:.4e} | {algo:.4e} | {corr / algo:.4f} |")
    traffic.setdefault(n_part, {})[cname] = corr
with open(os.path.join(out, f"{rnd}_pmc_apply.md"), "w") as f:
    f.write(f"# HBM traffic of apply_tile_kernel<float,2,0> (1e6 rows) and apply_wave_kernel<float,1,64> (1.6e7 rows) ({rnd})\n\n"
            "Source: `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) over\n"
            "`profiles/traffic_probe.py` (20 launches per size). Corrections per MI355X_MICROARCH.md section HBM: counters are\n"
            "in KiB; on gfx950 FETCH_SIZE reports exactly half of a wide (16 B/lane) coalesced read stream -> x2.\n"
            "The 1.6e7-particle launches (448 MB in + 448 MB out) cannot be Infinity-Cache resident and calibrate the\n"
            "counters: corrected/algorithmic = 1.00 for both.\n\n" + "\n".join(lines) + "\n")
t = traffic.get(1_000_000, {})
if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
    big = traffic.get(16_000_000, {})
    json.dump({"kernel": "apply_tile_kernel<float,2,0>", "particles": 1_000_000,
               "hbm_bytes_per_launch": t["FETCH_SIZE"] + t["WRITE_SIZE"],
               "streaming": {"kernel": "apply_wave_kernel<float,1,64>", "particles": 16_000_000,
                             "hbm_bytes_per_launch": big.get("FETCH_SIZE", 0.0) + big.get("WRITE_SIZE", 0.0)},
               "fetch_bytes": t["FETCH_SIZE"], "write_bytes": t["WRITE_SIZE"], "round": rnd,
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 (gfx950), KiB units"},
              open(os.path.join(out, "apply_traffic.json"), "w"), indent=1)
print(open(os.path.join(out, f"{rnd}_pmc_apply.md")).read())
