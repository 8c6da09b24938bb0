cd /tmp && export TMPDIR=/tmp
W=${1:-control}
O=$GRAFT_REPO_ROOT/gpurun_out/graph_$W
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o s -- python $GRAFT_REPO_ROOT/benchmarks/graph_modes.py $W > $O/log.txt 2>&1
python - <<EOF
import csv, glob, re
rows = list(csv.DictReader(open(glob.glob("$O/t/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-40:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:6.1f} " + re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:90])
EOF
