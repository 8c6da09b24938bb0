#!/usr/bin/env python3
"""cProfile of the host side of one merged Segment.track (C1-sized: the kernel is ~5 us, the rest is Python)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402
from benchmarks.run_configs import ares_subcell, t  # noqa: E402

dt = torch.float64
seg = ares_subcell(dt, t(8.2, dt))
beam = ca.ParticleBeam.from_twiss(beta_x=t(3.14, dt), beta_y=t(42.0, dt), num_particles=10_000, dtype=dt, device="cuda")
for _ in range(100):
    seg.track(beam)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    seg.track(beam)
torch.cuda.synchronize()
print("track us:", (time.perf_counter() - t0) / 2000 * 1e6)
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    seg.track(beam)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
