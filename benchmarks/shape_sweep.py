#!/usr/bin/env python3
"""Achieved bytes/s of the particle kernels over unusual shapes (looking for shapes a kernel's launch geometry suits badly):
apply (`particles @ R.mT`) for shared / per-row beams and maps, moments, cloud-in-cell screen readings."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
from cheetah_amd import _ops


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


print("== apply_map: x (Bx, N, 7) @ R (BR, 7, 7)")
for dt in (torch.float32, torch.float64):
    es = 4 if dt == torch.float32 else 8
    for Bx, BR, N in [(1, 1, 1000), (1, 1, 100_000), (1, 1, 10_000_000), (16, 16, 1_000_000), (1, 16, 1_000_000), (16, 1, 1_000_000),
                      (4096, 4096, 1000), (65536, 65536, 100), (65536, 1, 100), (1, 65536, 100), (1000, 1000, 9999), (3, 3, 3_333_333),
                      (1, 64, 100_001), (256, 256, 50_001)]:
        B = max(Bx, BR)
        if B * N * 7 * es > 20e9:
            continue
        x = torch.randn(Bx, N, 7, dtype=dt, device="cuda")
        R = torch.randn(BR, 7, 7, dtype=dt, device="cuda")
        with torch.no_grad():
            s = timeit(lambda: _ops.apply_map(x, R))
        by = (Bx * N * 7 + B * N * 7) * es
        print(f"{str(dt)[6:]:8s} Bx {Bx:6d} BR {BR:6d} N {N:9d}: {s * 1e6:9.1f} us  {by / s / 1e12:6.2f} TB/s", flush=True)
        del x, R

print("== moments: x (B, N, 7), w (B, N)")
for dt in (torch.float32, torch.float64):
    es = 4 if dt == torch.float32 else 8
    for B, N in [(1, 1000), (1, 100_000), (1, 10_000_000), (16, 1_000_000), (4096, 1000), (65536, 100), (1000, 9999), (3, 3_333_333)]:
        x = torch.randn(B, N, 7, dtype=dt, device="cuda")
        w = torch.rand(B, N, dtype=dt, device="cuda")
        with torch.no_grad():
            s = timeit(lambda: _ops._moments_raw(x, w, B, N))
        by = B * N * 8 * es
        print(f"{str(dt)[6:]:8s} B {B:6d} N {N:9d}: {s * 1e6:9.1f} us  {by / s / 1e12:6.2f} TB/s", flush=True)
        del x, w

print("== screen readings (cloud-in-cell unless noted), fp32")
kw = {"dtype": torch.float32, "device": "cuda"}
for res, n, method in [((2448, 2040), 10_000, "cloud-in-cell"), ((2448, 2040), 1_000_000, "cloud-in-cell"), ((2448, 2040), 10_000_000, "cloud-in-cell"),
                       ((64, 48), 1_000_000, "cloud-in-cell"), ((1024, 1024), 1_000_000, "histogram"), ((64, 48), 10_000_000, "histogram"),
                       ((8192, 8192), 1_000_000, "cloud-in-cell")]:
    scr = ca.Screen(resolution=res, pixel_size=torch.tensor([2e-6, 2e-6], **kw), is_active=True, method=method, **kw)
    beam = ca.ParticleBeam.from_parameters(num_particles=n, sigma_x=torch.tensor(3e-4, **kw), sigma_y=torch.tensor(3e-4, **kw), **kw)

    def f():
        scr.track(beam)
        return scr.reading
    with torch.no_grad():
        s = timeit(f, reps=10)
    print(f"{method:14s} {str(res):14s} particles {n:9d}: {s * 1e6:9.1f} us", flush=True)
