#!/usr/bin/env python3
"""get_beam_attrs_along_segment (beta functions and beam sizes after every element: what plots and matching routines call)
against lattice length and structure, 1e5 particles / ParameterBeam, ms per call."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def cell(i):
    return [ca.Quadrupole(t(0.2), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.8), **kw)]


beam = ca.ParticleBeam.from_parameters(num_particles=100_000, **kw)
pbeam = ca.ParameterBeam.from_parameters(**kw)
names = ("beta_x", "beta_y", "sigma_x", "s")
cases = {"FODO 100": ca.Segment([e for i in range(50) for e in cell(i)]),
         "FODO 400": ca.Segment([e for i in range(200) for e in cell(i)]),
         "25 nested cells of 4": ca.Segment([ca.Segment(cell(2 * i) + cell(2 * i + 1)) for i in range(25)]),
         "FODO 100 with 10 active BPMs": ca.Segment([e for i in range(50) for e in cell(i) + ([ca.BPM(is_active=True, **kw)] if i % 5 == 0 else [])])}
for name, seg in cases.items():
    with torch.no_grad():
        a = timeit(lambda: seg.get_beam_attrs_along_segment(names, beam))
        b = timeit(lambda: seg.get_beam_attrs_along_segment(names, pbeam))
    print(f"{name:32s}: ParticleBeam {a:8.3f} ms   ParameterBeam {b:8.3f} ms", flush=True)
