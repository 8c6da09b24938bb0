#!/usr/bin/env python3
"""Host-side (Python) cost of one SpaceChargeKick.track at the default 32^3 grid (GPU work is ~0.15 ms there)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402

dt = torch.float32
t = lambda v: torch.tensor(v, dtype=dt, device="cuda")  # noqa: E731
beam = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=100_000, total_charge=t(1e-9), energy=t(2.5e8), radius_x=t(1e-3),
                                            radius_y=t(1e-3), radius_tau=t(1e-3), sigma_px=t(1e-6), sigma_py=t(1e-6),
                                            sigma_p=t(1e-6), dtype=dt, device="cuda")
sc = ca.SpaceChargeKick(t(0.2), dtype=dt, device="cuda")
for _ in range(20):
    sc.track(beam)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    sc.track(beam)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / 200:.3f} ms per kick, with GPU drain {1e3 * (t2 - t0) / 200:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    sc.track(beam)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
