#!/usr/bin/env python3
"""`lattice_apply_kernel<float, 2>` (two particles per lane, from 1e6 particle rows on): ms per Segment.track of one beam of 1e6 / 4e6
particles through (a) a 16-cell linac [Drift, Quadrupole, active Cavity], (b) 24 FODO cells with an active BPM behind every cell
(25 maps, 24 monitors), (c) the same with apertures."""
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
    return (time.perf_counter() - t0) / reps * 1e3


def linac():
    els = []
    for i in range(16):
        els += [ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw),
                ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(-10.0), frequency=t(1.3e9), **kw)]
    return ca.Segment(els)


def fodo(apertures):
    els = []
    for i in range(24):
        els += [ca.Quadrupole(t(0.2), k1=t(4.0 if i % 2 else -4.0), **kw), ca.Drift(t(0.5), **kw), ca.BPM(is_active=True, **kw)]
        if apertures and i % 4 == 0:
            els.append(ca.Aperture(x_max=t(5e-3), y_max=t(5e-3), shape="rectangular", is_active=True, **kw))
    return ca.Segment(els)


with torch.no_grad():
    for n in (1_000_000, 4_000_000):
        beam = ca.ParticleBeam.from_parameters(num_particles=n, energy=t(1e8), **kw)
        for name, seg in (("16-cavity linac", linac()), ("24 cells, 24 monitors", fodo(False)), ("24 cells, monitors + apertures", fodo(True))):
            print(f"{n:>8} particles, {name:32s}: {timeit(lambda: seg.track(beam)):.3f} ms", flush=True)
