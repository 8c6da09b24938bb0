#!/usr/bin/env python3
"""Measured error of a float32 SpaceChargeKick (C4 size: 1e6 particles, 128^3) against the reference's float64 run
(tests/golden/fullsize_c4.npz, first kick): max |kick error| per momentum coordinate as a fraction of the kick amplitude —
the number behind the 2 % bound of tests/test_gpu_fullsize.py."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cheetah_amd as ca  # noqa: E402
import fullsize_inputs as fi  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_c4.npz"))
for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    sp = ca.Species("electron", **kw)
    beam = ca.ParticleBeam(torch.from_numpy(fi.c4_particles()).to(dt).cuda(), t(fi.C4_ENERGY),
                           particle_charges=torch.from_numpy(fi.c4_charges()).to(dt).cuda(), species=sp)
    b1 = ca.Drift(t(0.1), **kw).track(beam)
    b2 = ca.SpaceChargeKick(t(0.2), grid_shape=fi.C4_GRID, **kw).track(b1)
    sl = slice(None, None, fi.C4_SAMPLE_STRIDE)
    got = (b2.particles[sl].double() - b1.particles[sl].double()).cpu().numpy()
    ref = g["kick1_out_sample"] - g["kick1_in_sample"]
    kick = np.max(np.abs(ref), axis=0)
    err = np.max(np.abs(got - ref), axis=0)
    rms = np.sqrt(np.mean((got - ref) ** 2, axis=0))
    print(tag, "max error / kick amplitude (px, py, delta):", [f"{err[c] / kick[c]:.2e}" for c in (1, 3, 5)],
          " rms:", [f"{rms[c] / kick[c]:.2e}" for c in (1, 3, 5)])
