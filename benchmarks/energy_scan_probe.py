#!/usr/bin/env python3
"""A scan of the BEAM ENERGY: beams whose energy is a (B,) tensor through a 100-element FODO and a 16-cell cavity linac (scalar
lattice settings), ParameterBeam and ParticleBeam of 1e4 particles: us per Segment.track."""
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


fodo, linac = [], []
for i in range(50):
    fodo += [ca.Quadrupole(t(0.2), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.8), **kw)]
for i in range(16):
    linac += [ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw),
              ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(-10.0), frequency=t(1.3e9), **kw)]
with torch.no_grad():
    for B in (1, 64):
        energy = t(1e8) if B == 1 else torch.linspace(8e7, 1.2e8, B, **kw)
        pb = ca.ParameterBeam.from_parameters(energy=energy, **kw)
        beam = ca.ParticleBeam.from_parameters(num_particles=10_000, energy=energy, **kw)
        for name, els in (("FODO100", fodo), ("linac16", linac)):
            seg = ca.Segment(els)
            a = timeit(lambda: seg.track(pb))
            b = timeit(lambda: seg.track(beam))
            # the same energies with a NEW tensor every track (a scan loop that draws its energies per step)
            def fresh():
                e2 = energy + 1.0
                seg.track(ca.ParameterBeam(pb.mu, pb.cov, e2, **kw))
            c = timeit(fresh)
            print(f"B = {B:3d} energies, {name}: ParameterBeam {a:8.1f} us   ParticleBeam 1e4 {b:8.1f} us   ParameterBeam, new energy tensor per track {c:8.1f} us", flush=True)
