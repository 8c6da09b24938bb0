#!/usr/bin/env python3
"""Orbit response: B settings of the correctors of a 25-cell lattice (each corrector's angle a (B,) tensor), all BPM readings, for a
ParameterBeam and a ParticleBeam of 1e4 particles: ms per track."""
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


for B in (64, 4096):
    els, bpms = [], []
    for i in range(25):
        bpm = ca.BPM(is_active=True, **kw)
        bpms.append(bpm)
        els += [ca.Quadrupole(t(0.2), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.HorizontalCorrector(t(0.05), angle=1e-5 * torch.randn(B, **kw), **kw),
                ca.Drift(t(0.8), **kw), bpm]
    seg = ca.Segment(els)
    pb = ca.ParameterBeam.from_parameters(**kw)
    beam = ca.ParticleBeam.from_parameters(num_particles=10_000, **kw)
    with torch.no_grad():
        a = timeit(lambda: seg.track(pb))
        r = torch.stack([b.reading for b in bpms])
        b_ = timeit(lambda: seg.track(beam), reps=2) if B <= 64 else float("nan")
    print(f"B = {B:5d} corrector settings, 25 BPMs: ParameterBeam {a:8.3f} ms (readings {tuple(r.shape)})   ParticleBeam 1e4 {b_:8.3f} ms", flush=True)
