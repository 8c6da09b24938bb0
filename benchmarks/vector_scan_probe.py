#!/usr/bin/env python3
"""Vectorised scans on the ARES subcell: all five magnet settings as (B,) tensors (what Bayesian optimisation / a batched
environment evaluates), ParameterBeam and ParticleBeam of 1e3 particles: ms per track + sigma_x, with NEW setting tensors every step
and with the settings written in place."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import run_configs as rc
import cheetah_amd as ca
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for B in (64, 4096):
    seg = rc.ares_subcell(dt, torch.zeros(B, **kw))
    mags = [seg.AREAMQZM1, seg.AREAMQZM2, seg.AREAMCVM1, seg.AREAMQZM3, seg.AREAMCHM1]
    names = ["k1", "k1", "angle", "k1", "angle"]
    for m, n in zip(mags, names):
        setattr(m, n, torch.randn(B, **kw) * (5.0 if n == "k1" else 1e-4))
    pb = ca.ParameterBeam.from_twiss(beta_x=rc.t(3.14, dt), beta_y=rc.t(42.0, dt), **kw)
    beam = ca.ParticleBeam.from_twiss(beta_x=rc.t(3.14, dt), beta_y=rc.t(42.0, dt), num_particles=1000, **kw)
    actions = torch.randn(5, B, **kw)

    def step_assign(b):
        for i, (m, n) in enumerate(zip(mags, names)):
            setattr(m, n, actions[i] * (5.0 if n == "k1" else 1e-4))
        return seg.track(b).sigma_x

    def step_inplace(b):
        for i, (m, n) in enumerate(zip(mags, names)):
            getattr(m, n).copy_(actions[i])
        return seg.track(b).sigma_x

    with torch.no_grad():
        print(f"B = {B:5d}: ParameterBeam assign {timeit(lambda: step_assign(pb)):7.3f} ms, in place {timeit(lambda: step_inplace(pb)):7.3f} ms, unchanged {timeit(lambda: seg.track(pb).sigma_x):7.3f} ms;"
              f"  ParticleBeam 1e3 assign {timeit(lambda: step_assign(beam)):7.3f} ms, in place {timeit(lambda: step_inplace(beam)):7.3f} ms", flush=True)
