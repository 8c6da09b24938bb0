#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r03_final
S=$(date +%s)
python bench.py > gpurun_out/r03_final/bench.json 2> gpurun_out/r03_final/bench.err
echo "bench.py default run: rc $? wall $(( $(date +%s) - S )) s"
python - <<EOF
import json
for l in open("gpurun_out/r03_final/bench.json"):
    if l.startswith("{"):
        d=json.loads(l)
        print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_vs_copy_ceiling"])
        print({k:v.get("ms_per_track", v.get("ms_fwd_bwd")) for k,v in d["configs"].items() if isinstance(v,dict)})
        print(d["scaling_legs"]["c4_particle_shard"]["ms_per_track"], d["cpu_baseline"]["value"])
EOF
