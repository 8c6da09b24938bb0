cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/c5_graph
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o s -- python $GRAFT_REPO_ROOT/benchmarks/c5_graph.py > $O/log.txt 2>&1
python - <<EOF
import csv, glob, re
f = glob.glob("$O/t/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the eager loop runs last: take a window in the middle of the replay loop instead = rows before the last eager section. Find replays by
# the periodic pattern: locate occurrences of the moment_entry_mapped_bwd kernel and print one period in the replay section
idx = [i for i, r in enumerate(rows) if "moment_entry_mapped_bwd" in r["Kernel_Name"]]
mid = idx[len(idx) // 3]
prev = idx[len(idx) // 3 - 1]
t0 = int(rows[prev + 1]["Start_Timestamp"])
for r in rows[prev + 1: mid + 6]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:70]
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {name}")
EOF
