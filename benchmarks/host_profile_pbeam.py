#!/usr/bin/env python3
"""cProfile of the control step with a ParameterBeam (README segment: five settings written in place, track, screen reading)."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import run_configs as rc
import cheetah_amd as ca
dt = torch.float32
seg = rc.ares_subcell(dt, rc.t(8.2, dt))
seg.AREABSCR1.is_active = True
beam = ca.ParameterBeam.from_twiss(beta_x=rc.t(3.14, dt), beta_y=rc.t(42.0, dt), dtype=dt, device="cuda")


def step():
    seg.track(beam)
    return seg.AREABSCR1.reading


with torch.no_grad():
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        step()
    torch.cuda.synchronize()
    print("track + reading, us:", (time.perf_counter() - t0) / 300 * 1e6)
    t0 = time.perf_counter()
    for _ in range(300):
        seg.track(beam)
    torch.cuda.synchronize()
    print("track only, us:", (time.perf_counter() - t0) / 300 * 1e6)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        step()
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
