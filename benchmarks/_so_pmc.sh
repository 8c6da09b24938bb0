#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the non-linear particle kernels at 1e6 particles (separate passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  O=$R/gpurun_out/so_pmc_$C
  mkdir -p $O
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/t -o s -- python $R/benchmarks/nonlinear_bench.py 1e6 > $O/log.txt 2>&1
  python - <<EOF
import csv, glob, collections, re
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob("$O/t/*counter_collection.csv")[0])):
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])
    if "dkd_kernel" in n or "second_order" in n:
        acc[n[:60]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("$C", k, len(v), round(sum(v) / len(v)))
EOF
done
