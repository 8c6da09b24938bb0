#!/usr/bin/env python3
"""Per-kick time of a 10-kick segment against grid shape and beam size; run with CHX_SC_CHAIN=on and =off to compare the
tile-ordered chain with kick-by-kick tracking (the chain deposits one workgroup per 8^3 tile: few tiles and many particles suit it badly)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)


def timeit(fn, reps=30, warm=6):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


grids = [(32, 32, 32), (64, 64, 64), (32, 32, 128), (64, 64, 32), (128, 128, 128)]
for g in grids:
    for n in (100_000, 300_000, 1_000_000, 3_000_000):
        beam = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=n, total_charge=t(1e-9), energy=t(2.5e8), radius_x=t(1e-3),
                                                    radius_y=t(1e-3), radius_tau=t(1e-3), sigma_px=t(1e-6), sigma_py=t(1e-6),
                                                    sigma_p=t(1e-6), **kw)
        els = []
        for i in range(10):
            els += [ca.Drift(t(0.1)), ca.SpaceChargeKick(t(0.2), grid_shape=g, **kw), ca.Drift(t(0.1)),
                    ca.Quadrupole(t(0.1), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.1))]
        seg = ca.Segment(els)
        with torch.no_grad():
            tr = timeit(lambda: seg.track(beam))
        nt = (g[0] // 8) * (g[1] // 8) * (g[2] // 8)
        print(f"CHX_SC_CHAIN={os.environ.get('CHX_SC_CHAIN', 'auto'):4s} grid {str(g):16s} tiles {nt:5d} particles {n:>8d} per tile {n / nt:9.0f}: {tr / 10:7.1f} us per kick", flush=True)
