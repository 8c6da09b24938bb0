#!/usr/bin/env python3
"""Segment.track against the number of elements of one mergeable run (a persistent device plan holds at most 192 of them), and of
a long lattice with screens / cavities in between."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


beam = ca.ParticleBeam.from_parameters(num_particles=100_000, **kw)
for E in (100, 192, 193, 400, 1000, 5000):
    els = []
    for i in range(E // 2):
        els += [ca.Quadrupole(t(0.2), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.8), **kw)]
    seg = ca.Segment(els)
    with torch.no_grad():
        us = timeit(lambda: seg.track(beam))
    print(f"one run of {E:5d} elements: {us:9.1f} us per track ({us / E:6.2f} us per element)", flush=True)
for E in (100, 1000):
    els = []
    for i in range(E // 4):
        els += [ca.Quadrupole(t(0.2), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.8), **kw), ca.BPM(is_active=True, **kw),
                ca.Drift(t(0.2), **kw)]
    seg = ca.Segment(els)
    with torch.no_grad():
        us = timeit(lambda: seg.track(beam), reps=5)
    print(f"{E:5d} elements with an active BPM every fourth: {us:9.1f} us per track ({us / (E // 4):6.2f} us per BPM)", flush=True)
