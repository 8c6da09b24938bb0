#!/usr/bin/env python3
"""Host profile of the README-style control step (five settings ASSIGNED as new tensors, track, screen reading): where the time
between the caller's own tensor ops and the two launches goes. cProfile of 2000 steps + wall-clock splits."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import run_configs as rc  # noqa: E402
import cheetah_amd as ca  # noqa: E402

dt = torch.float32
seg = rc.ares_subcell(dt, rc.t(8.2, dt))
seg.AREABSCR1.is_active = True
beam = ca.ParticleBeam.from_twiss(beta_x=rc.t(3.14, dt), beta_y=rc.t(42.0, dt), num_particles=10_000, dtype=dt, device="cuda")
actions = torch.randn(300, 5, device="cuda", dtype=dt)
vals = [[actions[i, j] * (10 if j in (0, 1, 3) else 1e-4) for j in range(5)] for i in range(300)]   # the caller's ops, done ahead
torch.cuda.synchronize()
counter = [0]


def assign_only():
    v = vals[counter[0] % 300]
    counter[0] += 1
    seg.AREAMQZM1.k1, seg.AREAMQZM2.k1, seg.AREAMCVM1.angle, seg.AREAMQZM3.k1, seg.AREAMCHM1.angle = v


def step():
    assign_only()
    seg.track(beam)
    return seg.AREABSCR1.reading


def timeit(fn, reps=2000, warm=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


with torch.no_grad():
    print("assign five settings only us", round(timeit(assign_only), 1))
    print("assign + track + reading us", round(timeit(step), 1))
    print("track + reading (no assignment) us", round(timeit(lambda: (seg.track(beam), seg.AREABSCR1.reading)), 1))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)
