cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/strong
mkdir -p $O
python $GRAFT_REPO_ROOT/benchmarks/strong_leg_trace.py
rocprofv3 --kernel-trace --output-format csv -d $O/t -o s -- python $GRAFT_REPO_ROOT/benchmarks/strong_leg_trace.py > $O/log.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/t/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ap = [r for r in rows if "apply_tile_kernel" in r["Kernel_Name"] or "apply_wave" in r["Kernel_Name"]]
# a window of 100 consecutive apply kernels in the middle
mid = len(ap) // 3
w = ap[mid:mid + 200]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in w]
g = [(int(w[i + 1]["Start_Timestamp"]) - int(w[i]["End_Timestamp"])) / 1e3 for i in range(len(w) - 1)]
g = [x for x in g if x < 50]
print("apply kernels:", len(ap), "mean duration us", sum(d) / len(d), "mean gap us", sum(g) / len(g), "name", w[0]["Kernel_Name"][:60], "grid", w[0]["Grid_Size_X"])
PY
