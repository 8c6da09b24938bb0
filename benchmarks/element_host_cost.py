#!/usr/bin/env python3
"""Wall time of one `track()` of each element type on a small beam (1e4 particles, fp32): what the host layer costs per element
when an element is tracked on its own (inside Segment.track runs of linear / drift-kick-drift elements are one C call)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402

dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)  # noqa: E731
beam = ca.ParticleBeam.from_parameters(num_particles=10_000, energy=t(1e8), **kw)
pbeam = ca.ParameterBeam.from_parameters(energy=t(1e8), **kw)
cases = {
    "Drift linear": ca.Drift(t(0.3), **kw),
    "Quadrupole linear": ca.Quadrupole(t(0.2), k1=t(3.0), **kw),
    "Quadrupole second_order": ca.Quadrupole(t(0.2), k1=t(3.0), tracking_method="second_order", **kw),
    "Quadrupole drift_kick_drift": ca.Quadrupole(t(0.2), k1=t(3.0), tracking_method="drift_kick_drift", **kw),
    "Dipole linear": ca.Dipole(t(0.5), angle=t(0.02), **kw),
    "Dipole drift_kick_drift": ca.Dipole(t(0.5), angle=t(0.02), tracking_method="drift_kick_drift", **kw),
    "Cavity active": ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(-10.0), frequency=t(1.3e9), **kw),
    "HorizontalCorrector": ca.HorizontalCorrector(t(0.1), angle=t(1e-4), **kw),
    "Solenoid": ca.Solenoid(t(0.2), k=t(0.5), **kw),
    "Screen active": ca.Screen(is_active=True, **kw),
    "BPM active": ca.BPM(is_active=True, **kw),
    "Marker": ca.Marker(),
    "Aperture active": ca.Aperture(x_max=t(1e-3), y_max=t(1e-3), is_active=True, **kw),
    "SpaceChargeKick 32^3": ca.SpaceChargeKick(t(0.2), grid_shape=(32, 32, 32), **kw),
}
for name, el in cases.items():
    for b, tag in ((beam, "ParticleBeam"), (pbeam, "ParameterBeam")):
        try:
            for _ in range(20):
                el.track(b)
        except Exception as exc:  # noqa: BLE001
            print(f"{name:30s} {tag:14s} -- ({type(exc).__name__})")
            continue
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            el.track(b)
        torch.cuda.synchronize()
        print(f"{name:30s} {tag:14s} {(time.perf_counter() - t0) / 300 * 1e6:7.1f} us")
