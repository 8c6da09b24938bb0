#!/usr/bin/env python3
"""A/B/C of the Poisson solve in SpaceChargeKick (same result, different cost): in-place hipFFT plans inside libchx with
the Green-function chain on a side stream (default), torch.fft full (2g)^3 transforms, torch.fft axis-by-axis pruned."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402

dt = torch.float32
t = lambda v: torch.tensor(v, dtype=dt, device="cuda")  # noqa: E731
torch.manual_seed(0)
beam = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=1_000_000, total_charge=t(1e-9), energy=t(2.5e8),
                                            radius_x=t(1e-3), radius_y=t(1e-3), radius_tau=t(1e-3), sigma_px=t(1e-6),
                                            sigma_py=t(1e-6), sigma_p=t(1e-6), dtype=dt, device="cuda")
for g in (32, 64, 128):
    sc = ca.SpaceChargeKick(t(0.2), grid_shape=(g, g, g), dtype=dt, device="cuda")
    outs = {}
    for pruned in ("pruned", "hipfft", False, True):
        ca.SpaceChargeKick.fft_backend = pruned if isinstance(pruned, str) else "torch"
        ca.SpaceChargeKick.pruned_fft = pruned is True
        for _ in range(3):
            o = sc.track(beam)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            o = sc.track(beam)
        torch.cuda.synchronize()
        outs[pruned] = o.particles
        label = {"pruned": "libchx pruned line FFTs + side stream", "hipfft": "libchx hipFFT in-place + side stream",
                 False: "torch.fft full", True: "torch.fft pruned"}[pruned]
        print(f"g={g:4d} {label:38s} {(time.perf_counter() - t0) / 10 * 1e3:8.3f} ms")
    kick = (outs[False] - beam.particles).abs().amax(dim=0)
    for other in (True, "hipfft", "pruned"):
        diff = (outs[other] - outs[False]).abs().amax(dim=0)
        print(f"   max |{other} - torch full| / kick amplitude:", [f"{float(d / k):.1e}" for d, k in zip(diff[[1, 3, 5]], kick[[1, 3, 5]])])
