#!/usr/bin/env python3
"""100-element FODO tracked with second-order maps (bench.py's SECOND_ORDER_FODO100) and the two kernels per element kind:
ms per track and us per element pass at 1e6 particles; A/B of builds inside one gpurun call."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402

dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
tt = lambda v: torch.tensor(v, **kw)  # noqa: E731
so = {"tracking_method": "second_order"}
els = []
for _ in range(25):
    els += [ca.Quadrupole(tt(0.2), k1=tt(4.2), **so, **kw), ca.Drift(tt(0.8), **so, **kw),
            ca.Quadrupole(tt(0.2), k1=tt(-4.2), **so, **kw), ca.Drift(tt(0.8), **so, **kw)]
seg = ca.Segment(els)
beam = ca.ParticleBeam.from_parameters(num_particles=1_000_000, **kw)
lin = ca.Segment([ca.Quadrupole(tt(0.2), k1=tt(4.2), **kw) if i % 2 == 0 else ca.Drift(tt(0.8), **kw) for i in range(100)])
with torch.no_grad():
    for _ in range(3):
        lin.track_elementwise(beam)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        lin.track_elementwise(beam)
    torch.cuda.synchronize()
    print(f"calibration: linear apply, 100 passes: {(time.perf_counter() - t0) / 10 / 100 * 1e6:.2f} us per pass on this box")
with torch.no_grad():
    for _ in range(3):
        seg.track(beam)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            seg.track(beam)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
    print(f"second-order FODO100, 1e6 particles: {best:.3f} ms per track = {best * 10:.2f} us per element pass "
          f"({56e6 * 100 / (best * 1e-3) / 1e12:.2f} TB/s)")
    dkd = {"tracking_method": "drift_kick_drift"}
    for prec in ("double", "mixed", "storage"):
        els2 = []
        for _ in range(25):
            els2 += [ca.Quadrupole(tt(0.2), k1=tt(4.2), **dkd, **kw), ca.Drift(tt(0.8), **dkd, **kw),
                     ca.Quadrupole(tt(0.2), k1=tt(-4.2), **dkd, **kw), ca.Drift(tt(0.8), **dkd, **kw)]
        for e in els2:
            e.dkd_precision = prec
        seg2 = ca.Segment(els2)
        for _ in range(3):
            seg2.track(beam)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            seg2.track(beam)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f"drift-kick-drift FODO100 [{prec}]: {ms:.3f} ms per track ({56e6 * 100 / (ms * 1e-3) / 1e12:.2f} TB/s)")
    for name, e in (("quadrupole", els[0]), ("drift", els[1]), ("dipole", ca.Dipole(tt(0.5), angle=tt(0.03), **so, **kw)),
                    ("sextupole", ca.Sextupole(tt(0.15), k2=tt(25.0), **so, **kw))):
        for _ in range(3):
            e.track(beam)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        T = e.second_order_transfer_map(beam.energy, beam.species)
        from cheetah_amd import _ops
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            _ops.apply_second_order(beam.particles, T)
        e1.record()
        torch.cuda.synchronize()
        nnz = int((T != 0).sum())
        print(f"  {name:10s}: {e0.elapsed_time(e1) / 50 * 1e3:6.2f} us per pass, {nnz} non-zero entries of T")
