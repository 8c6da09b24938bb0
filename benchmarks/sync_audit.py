#!/usr/bin/env python3
"""Which steady-state tracking calls make torch synchronise with the device? Each workload runs twice to warm its caches, then
once more under torch.cuda.set_sync_debug_mode("warn"): every synchronising torch call (.item(), host reads, pageable
host-to-device copies) raises a warning that is printed with its origin. (libchx calls are not seen by this switch: they only
launch.)"""
import os
import sys
import traceback
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402
from benchmarks import run_configs as rc  # noqa: E402

dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)  # noqa: E731


def audit(label, fn):
    fn(); fn()
    torch.cuda.synchronize()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode("warn")
        try:
            fn()
        finally:
            torch.cuda.set_sync_debug_mode("default")
    hits = [w for w in rec if "synchroniz" in str(w.message).lower()]
    print(f"{label:58s} {len(hits)} synchronising call(s)")
    for w in hits[:6]:
        print("      ", w.filename.replace(os.getcwd() + "/", ""), w.lineno, "-", str(w.message)[:90])


beam = ca.ParticleBeam.from_parameters(num_particles=100_000, **kw)
seg = rc.ares_subcell(dt, rc.t(8.2, dt))
seg.AREABSCR1.is_active = True
audit("C1 lattice: Segment.track", lambda: seg.track(beam))
audit("C1 lattice: track + Screen.reading", lambda: (seg.track(beam), seg.AREABSCR1.reading))
vals = [t(10.0), t(-9.0)]
state = [0]


def rl():
    state[0] ^= 1
    seg.AREAMQZM1.k1 = vals[state[0]]
    seg.track(beam)
    return seg.AREABSCR1.reading


audit("control step (assign, track, read)", rl)
pb = ca.ParameterBeam.from_parameters(**kw)
audit("ParameterBeam: Segment.track + reading", lambda: (seg.track(pb), seg.AREABSCR1.reading))
for name, el in (("Marker", ca.Marker(**kw)), ("BPM active", ca.BPM(is_active=True, **kw)),
                 ("Aperture active", ca.Aperture(x_max=t(1e-3), y_max=t(1e-3), is_active=True, **kw)),
                 ("Screen active", ca.Screen(is_active=True, **kw)), ("Drift", ca.Drift(t(1.0), **kw)),
                 ("Quadrupole", ca.Quadrupole(t(0.2), k1=t(3.0), **kw)),
                 ("Cavity active", ca.Cavity(t(1.0), voltage=t(1e7), phase=t(10.0), frequency=t(1.3e9), **kw)),
                 ("Dipole dkd", ca.Dipole(t(0.5), angle=t(0.1), tracking_method="drift_kick_drift", **kw)),
                 ("Quadrupole second order", ca.Quadrupole(t(0.2), k1=t(3.0), tracking_method="second_order", **kw)),
                 ("SpaceChargeKick 32^3", ca.SpaceChargeKick(t(0.1), grid_shape=(32, 32, 32), **kw))):
    audit(f"{name}.track", lambda el=el: el.track(beam))
audit("beam.sigma_x, mu_y, emittance_x", lambda: (beam.sigma_x, beam.mu_y, beam.emittance_x))
audit("beam.clone()", lambda: beam.clone())
seg4 = ca.Segment([ca.Drift(t(0.1), **kw), ca.SpaceChargeKick(t(0.2), grid_shape=(32, 32, 32), **kw), ca.Drift(t(0.1), **kw),
                   ca.Quadrupole(t(0.1), k1=t(4.2), **kw)])
audit("[Drift, SpaceChargeKick, Drift, Quad].track", lambda: seg4.track(beam))
k1 = torch.nn.Parameter(t(3.0))
seg5 = ca.Segment([ca.Drift(t(1.0), **kw), ca.Quadrupole(t(0.2), k1=k1, **kw), ca.Drift(t(1.0), **kw)])


def c5():
    k1.grad = None
    seg5.track(beam).sigma_x.backward()


audit("forward + backward through [Drift, Quad(k1), Drift]", c5)
vb = ca.Segment([ca.Drift(t(1.0), **kw), ca.Quadrupole(t(0.2), k1=torch.linspace(-3, 3, 8, **kw), **kw), ca.Drift(t(1.0), **kw)])
audit("vectorised k1 (8 settings): Segment.track", lambda: vb.track(beam))
audit("Segment.track_moments", lambda: vb.track_moments(beam))
# calibration: what the switch does and does not see
audit("calibration: torch.tensor(1.0, device='cuda')", lambda: torch.tensor(1.0, device="cuda"))
audit("calibration: float(beam.sigma_x)", lambda: float(beam.sigma_x))
