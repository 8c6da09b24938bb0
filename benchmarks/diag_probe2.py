#!/usr/bin/env python3
"""More lattices with diagnostics: a vectorised magnet in front of BPM cells, ParameterBeams through BPM / cavity lattices."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)


def timeit(fn, reps=20, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def bpm_cells(cells, first_k1=None):
    els = []
    for i in range(cells):
        k1 = t(4.2 if i % 2 == 0 else -4.2) if (i or first_k1 is None) else first_k1
        els += [ca.Quadrupole(t(0.2), k1=k1, **kw), ca.Drift(t(0.8), **kw), ca.BPM(is_active=True, **kw), ca.Drift(t(0.2), **kw)]
    return ca.Segment(els)


def linac(cells):
    els = []
    for i in range(cells):
        els += [ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw),
                ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(-10.0), frequency=t(1.3e9), **kw)]
    return ca.Segment(els)


beam = ca.ParticleBeam.from_parameters(num_particles=10_000, energy=t(1e8), **kw)
pbeam = ca.ParameterBeam.from_parameters(energy=t(1e8), **kw)
with torch.no_grad():
    print(f"25 BPM cells, scalar settings        : ParticleBeam {timeit(lambda: bpm_cells25.track(beam)) if (bpm_cells25 := bpm_cells(25)) else 0:8.1f} us   ParameterBeam {timeit(lambda: bpm_cells25.track(pbeam)):8.1f} us", flush=True)
    seg = bpm_cells(25, first_k1=torch.linspace(-5, 5, 64, **kw))
    print(f"25 BPM cells, first quad 64 strengths: ParticleBeam {timeit(lambda: seg.track(beam), reps=5):8.1f} us   ParameterBeam {timeit(lambda: seg.track(pbeam), reps=5):8.1f} us", flush=True)
    lin = linac(16)
    print(f"16-cell cavity linac                 : ParticleBeam {timeit(lambda: lin.track(beam)):8.1f} us   ParameterBeam {timeit(lambda: lin.track(pbeam), reps=5):8.1f} us", flush=True)
