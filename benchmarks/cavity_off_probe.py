#!/usr/bin/env python3
"""A 16-cell cavity linac with some cavities switched off (voltage 0: a skippable, drift-like element, cavity.py:253-262), 1e5
particles and a ParameterBeam: us per Segment.track against the same linac with every cavity on."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def linac(off):
    els = []
    for i in range(16):
        els += [ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw),
                ca.Cavity(t(1.0377), voltage=t(0.0 if i in off else 18e6), phase=t(-10.0), frequency=t(1.3e9), **kw)]
    return ca.Segment(els)


beam = ca.ParticleBeam.from_parameters(num_particles=100_000, energy=t(1e8), **kw)
pbeam = ca.ParameterBeam.from_parameters(energy=t(1e8), **kw)
with torch.no_grad():
    for off in ((), (5,), (5, 9), (1, 5, 9, 13)):
        seg = linac(off)
        print(f"cavities off {off}: ParticleBeam {timeit(lambda: seg.track(beam)):8.1f} us   ParameterBeam {timeit(lambda: seg.track(pbeam)):8.1f} us", flush=True)
    seg = linac((5, 9))
    # RL-style: the voltage of one cavity toggled between tracks
    v_on, v_off = t(18e6), t(0.0)
    state = [0]

    def step():
        state[0] ^= 1
        seg.elements[8].voltage = v_off if state[0] else v_on
        seg.track(beam)

    print(f"one cavity toggled on / off between tracks: {timeit(step):8.1f} us", flush=True)
