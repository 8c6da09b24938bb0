#!/usr/bin/env python3
"""The strong-scaling leg of bench.py (C2, 1e6 particles in total) at the sizes a rank holds on 1 / 8 GPUs, eager against a
replayed device graph of the 100 apply launches. Result on ROCm 7.2 / MI355X: the graph LOSES at both sizes (a replayed kernel
node costs ~9 us of scheduling, a launch issued from libchx's C loop ~2.5 us), so the leg stays eager and is launch-bound below
~2e5 particles per rank. Prints one JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cheetah_amd as ca  # noqa: E402
from cheetah_amd import _ops, graph  # noqa: E402

seg = bench.build_fodo(ca, torch, "cuda", torch.float32)
out = {}
for n in (1_000_000, 125_000):
    torch.manual_seed(4321)
    beam = ca.ParticleBeam.from_parameters(num_particles=n, dtype=torch.float32, device="cuda")
    with torch.no_grad():
        replay = graph.capture(lambda: seg.track_elementwise(beam, fused=False))

    def eager():
        o = seg.track_elementwise(beam, fused=False)
        _ops.moments(o.particles, o.survival_probabilities)

    def replayed():
        o = replay()
        _ops.moments(o.particles, o.survival_probabilities)

    res = {}
    for name, fn in (("eager", eager), ("graph_replay", replayed)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        res[name + "_ms_per_step"] = (time.perf_counter() - t0) / 50 * 1e3
    res["graph_equals_eager"] = bool(torch.equal(replay.outputs.particles, seg.track_elementwise(beam, fused=False).particles))
    out[str(n)] = res
print(json.dumps({"strong_leg_probe": out}))
