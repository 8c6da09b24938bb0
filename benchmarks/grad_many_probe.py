#!/usr/bin/env python3
"""Gradient-based tuning of MANY magnets: all 50 quadrupole strengths of a 100-element FODO trainable, loss = sigma_x + sigma_y of the
outgoing beam, forward + backward; ParticleBeam 1e5 and ParameterBeam; also with 25 active BPMs whose readings enter the loss."""
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
    return (time.perf_counter() - t0) / reps * 1e3


def lattice(bpms):
    els, ks, ms = [], [], []
    for i in range(50):
        k = torch.nn.Parameter(t(4.2 if i % 2 == 0 else -4.2))
        ks.append(k)
        els += [ca.Quadrupole(t(0.2), k1=k, **kw), ca.Drift(t(0.8), **kw)]
        if bpms and i % 2 == 1:
            m = ca.BPM(is_active=True, **kw)
            ms.append(m)
            els.append(m)
    return ca.Segment(els), ks, ms


beam = ca.ParticleBeam.from_parameters(num_particles=100_000, mu_x=t(1e-4), **kw)
pbeam = ca.ParameterBeam.from_parameters(mu_x=t(1e-4), **kw)
for bpms in (False, True):
    seg, ks, ms = lattice(bpms)
    for name, b in (("ParticleBeam 1e5", beam), ("ParameterBeam", pbeam)):
        def step():
            for k in ks:
                k.grad = None
            out = seg.track(b)
            loss = out.sigma_x + out.sigma_y
            if ms:
                loss = loss + sum((m.reading ** 2).sum() for m in ms)
            loss.backward()
        try:
            ms_t = timeit(step)
            print(f"50 trainable quadrupoles{', 25 active BPMs in the loss' if bpms else '':30s} {name:18s}: {ms_t:8.3f} ms per forward + backward, "
                  f"grad ok: {all(k.grad is not None and torch.isfinite(k.grad) for k in ks)}", flush=True)
        except Exception as exc:  # noqa: BLE001
            print(f"{name} bpms={bpms}: {type(exc).__name__}: {str(exc)[:160]}", flush=True)
