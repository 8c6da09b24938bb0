#!/bin/bash
# Kernel trace of the control loop (benchmarks/rl_stages.py): which launches a control step is made of and what they cost.
# Run on the GPU box from the repo root: bash profiles/run_profile_rl.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rl_prof -o rl -- python $R/benchmarks/rl_stages.py > /dev/null 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/rl_prof/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:100], r['Calls'], r['AverageNs'], r['Percentage'])
PY
