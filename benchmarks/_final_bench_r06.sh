#!/bin/bash
# Round 6: the default bench.py run (what the driver runs), and the driver's own --steps 20 --warmup 5 form
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=gpurun_out/r06_final
mkdir -p $OUT
S=$(date +%s)
python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench.py default run: rc $? wall $(( $(date +%s) - S )) s"
tail -3 $OUT/bench.err
python - <<PY
import json
for l in open("$OUT/bench.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print("value", d["value"], "ms", d["ms_per_step"], "long", d.get("headline_long"), "frac", d["roofline"]["frac"])
        c = d["configs"]
        print({k: (v.get("ms_per_track", v.get("ms_fwd_bwd", v.get("ms_per_step"))) if isinstance(v, dict) else v) for k, v in c.items()})
        print("C4", {k: c["C4"].get(k) for k in ("ms_per_track", "fp32_kick_error")}, c["C4"]["roofline"].get("particle_kernels"))
        print("C5", c["C5"].get("ms_fwd_bwd"), c["C5"].get("graph_replay", {}).get("ms_fwd_bwd"))
        print("ARES", c.get("ARES_EA_SPEED_GUARD"))
        print("graph modes", {k: v for k, v in d["modes"]["graph_replay"].items() if k != "note"})
        print("legs", {k: (v.get("ms_per_track", v.get("ms_per_step")) if isinstance(v, dict) else v) for k, v in d["scaling_legs"].items()})
        print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("cores"))
PY
