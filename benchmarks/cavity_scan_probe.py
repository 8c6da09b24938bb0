#!/usr/bin/env python3
"""A PHASE scan of cavities in a 16-cell linac: the phase (and voltage) of one / all cavities a (64,) tensor; ParameterBeam and one
shared ParticleBeam of 1e4 particles: us per Segment.track."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def linac(scanned, B=64):
    els = []
    for i in range(16):
        ph = torch.linspace(-30.0, 30.0, B, **kw) if i in scanned else t(-10.0)
        els += [ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw),
                ca.Cavity(t(1.0377), voltage=t(18e6), phase=ph, frequency=t(1.3e9), **kw)]
    return ca.Segment(els)


pb = ca.ParameterBeam.from_parameters(energy=t(1e8), **kw)
beam = ca.ParticleBeam.from_parameters(num_particles=10_000, energy=t(1e8), **kw)
with torch.no_grad():
    for name, scanned in (("no cavity", ()), ("cavity 3", (3,)), ("cavity 15 (the last)", (15,)), ("every cavity", tuple(range(16)))):
        seg = linac(scanned)
        print(f"phase scan of {name:22s}: ParameterBeam {timeit(lambda: seg.track(pb)):8.1f} us   ParticleBeam 1e4 {timeit(lambda: seg.track(beam)):8.1f} us", flush=True)
