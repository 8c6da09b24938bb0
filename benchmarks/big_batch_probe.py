#!/usr/bin/env python3
"""Vectorised scans with very many lattice settings: ParticleBeam (B settings x N particles) and ParameterBeam through the ARES
subcell; does every stage take batches beyond 65 535 rows, and what does it cost?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import run_configs as rc
import cheetah_amd as ca

dt = torch.float32


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for B, N in [(4096, 1000), (100_000, 64), (1_000_000, 16)]:
    k1 = torch.linspace(-10.0, 10.0, B, dtype=dt, device="cuda")
    seg = rc.ares_subcell(dt, k1)
    beam = ca.ParticleBeam.from_twiss(beta_x=rc.t(3.14, dt), beta_y=rc.t(42.0, dt), num_particles=N, dtype=dt, device="cuda")
    try:
        with torch.no_grad():
            ms = timeit(lambda: seg.track(beam).sigma_x)
            out = seg.track(beam)
            sx = out.sigma_x
        ref = out.particles[..., 0].double().std(dim=-1)
        err = float(((sx.double() - ref).abs() / ref).max())
        print(f"ParticleBeam  B {B:8d} N {N:5d}: track + sigma_x {ms:8.3f} ms, sigma_x vs torch {err:.1e}", flush=True)
    except Exception as exc:  # noqa: BLE001
        print(f"ParticleBeam  B {B:8d} N {N:5d}: FAILED {type(exc).__name__}: {str(exc)[:200]}", flush=True)
    pb = ca.ParameterBeam.from_twiss(beta_x=rc.t(3.14, dt), beta_y=rc.t(42.0, dt), dtype=dt, device="cuda")
    try:
        with torch.no_grad():
            ms = timeit(lambda: seg.track(pb).sigma_x)
            sx2 = seg.track(pb).sigma_x
        print(f"ParameterBeam B {B:8d}        : track + sigma_x {ms:8.3f} ms, shape {tuple(sx2.shape)}", flush=True)
    except Exception as exc:  # noqa: BLE001
        print(f"ParameterBeam B {B:8d}        : FAILED {type(exc).__name__}: {str(exc)[:200]}", flush=True)
