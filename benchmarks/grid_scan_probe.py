#!/usr/bin/env python3
"""A 2-D GRID scan by broadcasting: one quadrupole's strength of shape (8, 1), another's (1, 8) — 64 lattice settings — in a 100-element
FODO with 5 monitors (and the same with both strengths flattened to (64,)): us per Segment.track."""
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


def lattice(ka, kb, monitors):
    els = []
    for i in range(50):
        k1 = ka if i == 10 else kb if i == 30 else t(4.2 if i % 2 == 0 else -4.2)
        els += [ca.Quadrupole(t(0.2), k1=k1, **kw), ca.Drift(t(0.8), **kw)]
        if monitors and i % 10 == 9:
            els += [ca.BPM(is_active=True, **kw)]
    return ca.Segment(els)


a, b = torch.linspace(-5, 5, 8, **kw), torch.linspace(-4, 4, 8, **kw)
A, Bm = torch.meshgrid(a, b, indexing="ij")
pb = ca.ParameterBeam.from_parameters(energy=t(1e8), **kw)
beam = ca.ParticleBeam.from_parameters(num_particles=10_000, energy=t(1e8), **kw)
if __name__ == "__main__":
    with torch.no_grad():
        for monitors in (False, True):
            for name, ka, kb in (("broadcast (8,1) x (1,8)", a.reshape(8, 1).contiguous(), b.reshape(1, 8).contiguous()),
                                 ("flattened (64,) and (64,)", A.reshape(-1).contiguous(), Bm.reshape(-1).contiguous()),
                                 ("meshgrid (8,8) and (8,8)", A.contiguous(), Bm.contiguous())):
                seg = lattice(ka, kb, monitors)
                print(f"monitors {monitors!s:5s} {name:28s}: ParameterBeam {timeit(lambda: seg.track(pb)):8.1f} us   ParticleBeam 1e4 {timeit(lambda: seg.track(beam)):8.1f} us", flush=True)
