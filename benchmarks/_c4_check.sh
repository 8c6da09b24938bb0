#!/bin/bash
# chain tests, un-overlapped kernel durations of C4's last track (rocprofv3 kernel trace), one- vs two-stream track time
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_gpu_sc_chain.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
bash benchmarks/_c4_serial.sh > /dev/null 2>&1
python - <<EOF
import csv
rows=list(csv.DictReader(open("gpurun_out/c4_serial/t/s_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def last(name,n=10):
    v=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if name in r["Kernel_Name"]]
    return [round(x,1) for x in v[-n:]]
for k in ["sc_tile_deposit","sc_tile_merge","sc_tile_particle","sc_geometry_partials","igf_table_near","igf_compact_far"]:
    print(k,last(k))
EOF
grep header gpurun_out/c4_serial/log.txt
python benchmarks/c4_stream_probe.py
