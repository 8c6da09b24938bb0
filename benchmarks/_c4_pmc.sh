#!/bin/bash
# PMC counters of the chain kernels (serial C4): bash benchmarks/_c4_pmc.sh "CTR1 CTR2 ..." tag
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/c4_pmc_$2
mkdir -p $O
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $O/t -o s -- python $R/benchmarks/c4_serial_trace.py > $O/log.txt 2>&1
python - <<EOF
import csv, glob, collections
f = glob.glob("$O/t/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    for key in ("sc_tile_deposit", "sc_tile_particle", "sc_tile_schedule"):
        if key in n: acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: [round(x) for x in v[-10:]] for c, v in d.items()})
EOF
