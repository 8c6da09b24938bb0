#!/usr/bin/env python3
"""cProfile of Segment.track on a lattice with an active BPM in every cell (100 elements, 1e5 particles)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)
els = []
for i in range(25):
    els += [ca.Quadrupole(t(0.2), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.8), **kw), ca.BPM(is_active=True, **kw),
            ca.Drift(t(0.2), **kw)]
seg = ca.Segment(els)
beam = ca.ParticleBeam.from_parameters(num_particles=100_000, **kw)
with torch.no_grad():
    for _ in range(5):
        seg.track(beam)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        seg.track(beam)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats(os.environ.get("SORT", "cumulative")).print_stats(40)
