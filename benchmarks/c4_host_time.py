#!/usr/bin/env python3
"""C4: host enqueue time of Segment.track (no synchronisation inside the timed region) against its wall time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)
g = 128
els = []
for i in range(10):
    els += [ca.Drift(t(0.1)), ca.SpaceChargeKick(t(0.2), grid_shape=(g, g, g), **kw), ca.Drift(t(0.1)),
            ca.Quadrupole(t(0.1), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.1))]
seg = ca.Segment(els)
beam = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=1_000_000, total_charge=t(1e-9), energy=t(2.5e8), radius_x=t(1e-3), radius_y=t(1e-3), radius_tau=t(1e-3), sigma_px=t(1e-6), sigma_py=t(1e-6), sigma_p=t(1e-6), **kw)
for _ in range(3): seg.track(beam)
torch.cuda.synchronize()
host, wall = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); seg.track(beam); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
print("host enqueue ms", [round(v, 3) for v in host])
print("wall ms", [round(v, 3) for v in wall])
# back to back (the GPU never idles between tracks)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): seg.track(beam)
torch.cuda.synchronize(); print("back-to-back ms/track", round((time.perf_counter() - t0) * 100, 3))
