#!/usr/bin/env python3
"""Segment.track on 100-element FODO-like lattices with things real lattice files contain: nested sub-segments, active apertures
and screens, a CustomTransferMap, one vectorised setting, markers between all elements. 1e5 particles, fp32, us per track."""
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


def cell(i):
    return [ca.Quadrupole(t(0.2), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.8), **kw)]


beam = ca.ParticleBeam.from_parameters(num_particles=100_000, **kw)
pbeam = ca.ParameterBeam.from_parameters(**kw)
cases = {}
cases["plain FODO, 100 elements"] = ca.Segment([e for i in range(50) for e in cell(i)])
cases["25 nested sub-segments of 4 elements"] = ca.Segment([ca.Segment(cell(2 * i) + cell(2 * i + 1)) for i in range(25)])
cases["a Marker between all elements (200 elements)"] = ca.Segment([e for i in range(50) for c in cell(i) for e in (c, ca.Marker(**kw))])
cases["10 active apertures"] = ca.Segment([e for i in range(50) for e in cell(i) + ([ca.Aperture(x_max=t(5e-3), y_max=t(5e-3), **kw)] if i % 5 == 0 else [])])
cases["10 active screens"] = ca.Segment([e for i in range(50) for e in cell(i) + ([ca.Screen(resolution=(64, 64), is_active=True, **kw)] if i % 5 == 0 else [])])
cases["one CustomTransferMap in the middle"] = ca.Segment([e for i in range(25) for e in cell(i)] + [ca.CustomTransferMap(torch.eye(7, **kw), length=t(0.1), **kw)] + [e for i in range(25) for e in cell(i)])
vec = [e for i in range(50) for e in cell(i)]
vec[10] = ca.Quadrupole(t(0.2), k1=torch.linspace(-5, 5, 64, **kw), **kw)
cases["one quadrupole with 64 strengths (vectorised)"] = ca.Segment(vec)
cases["a Solenoid and a Dipole per cell"] = ca.Segment([e for i in range(25) for e in cell(i) + [ca.Solenoid(t(0.1), k=t(0.2), **kw), ca.Dipole(t(0.2), angle=t(1e-3), **kw)]])
for name, seg in cases.items():
    with torch.no_grad():
        try:
            us = timeit(lambda: seg.track(beam))
        except Exception as exc:  # noqa: BLE001
            us = float("nan"); print("  ParticleBeam failed:", type(exc).__name__, str(exc)[:120])
        try:
            usp = timeit(lambda: seg.track(pbeam))
        except Exception as exc:  # noqa: BLE001
            usp = float("nan"); print("  ParameterBeam failed:", type(exc).__name__, str(exc)[:120])
    print(f"{name:50s}: ParticleBeam {us:9.1f} us   ParameterBeam {usp:9.1f} us", flush=True)
