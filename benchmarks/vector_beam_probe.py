#!/usr/bin/env python3
"""Vectorised BEAMS (B beams of N particles in one ParticleBeam) and vectorised settings through lattices with cavities / BPMs:
us per track."""
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


def linac(cells):
    els = []
    for i in range(cells):
        els += [ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw),
                ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(-10.0), frequency=t(1.3e9), **kw)]
    return ca.Segment(els)


def bpm_lattice(cells):
    els = []
    for i in range(cells):
        els += [ca.Quadrupole(t(0.2), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.8), **kw), ca.BPM(is_active=True, **kw),
                ca.Drift(t(0.2), **kw)]
    return ca.Segment(els)


one = ca.ParticleBeam.from_parameters(num_particles=10_000, energy=t(1e8), **kw)
many = ca.ParticleBeam(one.particles.unsqueeze(0).repeat(16, 1, 1).contiguous(), one.energy, particle_charges=one.particle_charges,
                       survival_probabilities=one.survival_probabilities, **kw)
for name, seg in (("16-cell cavity linac", linac(16)), ("25-cell BPM lattice", bpm_lattice(25)), ("plain FODO 100", ca.Segment([e for i in range(50) for e in (ca.Quadrupole(t(0.2), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.8), **kw))]))):
    with torch.no_grad():
        a = timeit(lambda: seg.track(one))
        b = timeit(lambda: seg.track(many), reps=5)
    print(f"{name:24s}: one beam of 1e4 {a:9.1f} us   16 beams of 1e4 in one ParticleBeam {b:9.1f} us", flush=True)
