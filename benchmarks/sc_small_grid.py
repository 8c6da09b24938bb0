#!/usr/bin/env python3
"""SpaceChargeKick at the reference's default grid (32^3) and small beams: time per isolated kick and per kick inside a Segment."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)


def timeit(fn, reps=200, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for g in (32, 64):
    for n in (10_000, 100_000, 1_000_000):
        beam = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=n, total_charge=t(1e-9), energy=t(2.5e8), radius_x=t(1e-3),
                                                    radius_y=t(1e-3), radius_tau=t(1e-3), sigma_px=t(1e-6), sigma_py=t(1e-6),
                                                    sigma_p=t(1e-6), **kw)
        kick = ca.SpaceChargeKick(t(0.2), grid_shape=(g, g, g), **kw)
        els = []
        for i in range(10):
            els += [ca.Drift(t(0.1)), ca.SpaceChargeKick(t(0.2), grid_shape=(g, g, g), **kw), ca.Drift(t(0.1)),
                    ca.Quadrupole(t(0.1), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.1))]
        seg = ca.Segment(els)
        with torch.no_grad():
            one = timeit(lambda: kick.track(beam))
            tr = timeit(lambda: seg.track(beam), reps=50)
        print(f"grid {g}^3, {n:>8d} particles: isolated kick {one:7.1f} us, in a 10-kick segment {tr / 10:7.1f} us per kick", flush=True)
