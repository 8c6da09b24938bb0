#!/usr/bin/env python3
"""The control loop on the ARES EA section (assign 5 magnet settings, track 1e4 particles, read the screen image) by stage:
host wall time of each stage, run back to back without device synchronisation (the loop is host-bound)."""
import os
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
vals = [torch.tensor(v, device="cuda", dtype=dt) for v in (10.0, -9.0, 1e-4, 8.0, -1e-4)] * 2
N = 3000


def loop(fn):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6


state = {"i": 0}


def assign():
    i = state["i"] = (state["i"] + 1) % 2
    seg.AREAMQZM1.k1, seg.AREAMQZM2.k1, seg.AREAMCVM1.angle = vals[i], vals[i + 1], vals[i + 2]
    seg.AREAMQZM3.k1, seg.AREAMCHM1.angle = vals[i + 3], vals[i + 4]


def assign_track():
    assign()
    seg.track(beam)


def full():
    assign()
    seg.track(beam)
    return seg.AREABSCR1.reading


def inplace_full():
    i = state["i"] = (state["i"] + 1) % 2
    seg.AREAMQZM1.k1.copy_(vals[i])
    seg.AREAMQZM2.k1.copy_(vals[i + 1])
    seg.AREAMCVM1.angle.copy_(vals[i + 2])
    seg.AREAMQZM3.k1.copy_(vals[i + 3])
    seg.AREAMCHM1.angle.copy_(vals[i + 4])
    seg.track(beam)
    return seg.AREABSCR1.reading


a = loop(assign)
b = loop(assign_track)
c = loop(full)
d = loop(lambda: seg.track(beam))
e = loop(inplace_full)
print(f"assign 5 settings        {a:7.1f} us")
print(f"+ Segment.track          {b:7.1f} us   (track alone, unchanged settings: {d:.1f} us)")
print(f"+ Screen.reading         {c:7.1f} us   = one control step")
print(f"same, settings written in place (copy_)  {e:7.1f} us")

if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile
    import pstats

    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3000):
        (assign_track if len(sys.argv) > 2 else full)()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(32)
