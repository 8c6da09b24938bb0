#!/usr/bin/env python3
"""Screen at 1e6 particles, 2448 x 2040, fp32: `track` (snapshot for the read beam + the independent outgoing beam, like the
reference's two clones) and `reading` (cloud-in-cell / histogram) timed apart."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402
from benchmarks.run_configs import timeit  # noqa: E402

dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
beam = ca.ParticleBeam.from_parameters(num_particles=1_000_000, **kw)
for method in ("cloud-in-cell", "histogram"):
    scr = ca.Screen(resolution=(2448, 2040), pixel_size=torch.tensor((3.5488e-6, 2.5003e-6), **kw), method=method, is_active=True, **kw)
    t_track = timeit(lambda: scr.track(beam), 50, 30)

    def read():
        scr.__dict__["_cached_reading"] = None
        return scr.reading

    scr.track(beam)
    t_read = timeit(read, 20, 3)
    seg = ca.Segment([ca.Drift(torch.tensor(1.0, **kw)), scr])
    t_seg = timeit(lambda: seg.track(beam), 20, 3)
    print(f"{method:14s} track {t_track * 1e3:7.1f} us   reading {t_read * 1e3:7.1f} us   [Drift, Screen].track {t_seg * 1e3:7.1f} us")
