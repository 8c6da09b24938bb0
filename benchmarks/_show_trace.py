"""Durations of the kernels whose name contains argv[2] from a rocprofv3 kernel-trace directory argv[1], in launch order."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    n = r["Kernel_Name"]
    if sys.argv[2] in n:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"{d:8.1f} us grid {r['Grid_Size_X']:>9} x {r['Grid_Size_Y']:>6} wg {r['Workgroup_Size_X']:>4} {n[:80]}")
