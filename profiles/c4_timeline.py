"""Timeline of the last chain kicks from a rocprofv3 kernel trace: python profiles/c4_timeline.py <kernel_trace.csv> [n_kicks]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nk = int(sys.argv[2]) if len(sys.argv) > 2 else 2
idx = [i for i, r in enumerate(rows) if "sc_tile_particle" in r["Kernel_Name"]]
first = idx[-nk - 1]
t0 = int(rows[first]["Start_Timestamp"])
import re
for r in rows[first: idx[-1] + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:64]
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} q{r['Queue_Id']} vgpr{r['VGPR_Count']:>4} lds{r['LDS_Block_Size']:>6} grid{r['Grid_Size_X']:>8} {name}")
