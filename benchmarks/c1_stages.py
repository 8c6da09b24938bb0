#!/usr/bin/env python3
"""Host time of the pieces of the README step (C1: 13-element ARES section, 1e4 particles): merged run, Screen.track (snapshot),
Screen.reading (cloud-in-cell image) — wall time per piece with the GPU kept busy (no sync inside the loop)."""
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
scr = seg.AREABSCR1
run_only = ca.Segment(list(seg.elements)[:-1])


def timed(fn, n=2000):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


mid = run_only.track(beam)
with torch.no_grad():
    print(f"merged run (12 elements):       {timed(lambda: run_only.track(beam)):6.1f} us")
    print(f"Screen.track (snapshot):        {timed(lambda: scr.track(mid)):6.1f} us")

    def read():
        scr.__dict__["_cached_reading"] = None
        return scr.reading
    print(f"Screen.reading (CIC image):     {timed(read):6.1f} us")
    print(f"segment.track:                  {timed(lambda: seg.track(beam)):6.1f} us")
    print(f"segment.track + reading:        {timed(lambda: (seg.track(beam), scr.reading)):6.1f} us")
