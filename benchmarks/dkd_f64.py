#!/usr/bin/env python3
"""Drift-kick-drift FODO100 on a float64 beam of 1e6 particles: Segment.track with the particles in registers vs element passes
(CHX_DKD_CHAIN_FUSED=0)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402

kw = {"dtype": torch.float64, "device": "cuda"}
tt = lambda v: torch.tensor(v, **kw)  # noqa: E731
dkd = {"tracking_method": "drift_kick_drift"}
els = []
for _ in range(25):
    els += [ca.Quadrupole(tt(0.2), k1=tt(4.2), **dkd, **kw), ca.Drift(tt(0.8), **dkd, **kw),
            ca.Quadrupole(tt(0.2), k1=tt(-4.2), **dkd, **kw), ca.Drift(tt(0.8), **dkd, **kw)]
seg = ca.Segment(els)
beam = ca.ParticleBeam.from_parameters(num_particles=1_000_000, **kw)
with torch.no_grad():
    for _ in range(3):
        seg.track(beam)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            seg.track(beam)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
print(f"drift-kick-drift FODO100, float64, 1e6 particles: {best:.3f} ms per track")
