#!/usr/bin/env python3
"""cProfile of the host side of the small-beam paths: python benchmarks/host_profile.py [rl|c1|linac|c2|dkd|c5]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import run_configs as rc  # noqa: E402
import cheetah_amd as ca  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "rl"
dt = torch.float32
if which in ("rl", "c1", "control"):
    seg = rc.ares_subcell(dt, rc.t(8.2, dt))
    seg.AREABSCR1.is_active = True
    beam = ca.ParticleBeam.from_twiss(beta_x=rc.t(3.14, dt), beta_y=rc.t(42.0, dt), num_particles=10_000, dtype=dt, device="cuda")
    actions = torch.randn(300, 5, device="cuda", dtype=dt)
    counter = [0]

    scale = torch.tensor([10.0, 10.0, 1e-4, 10.0, 1e-4], device="cuda", dtype=dt)
    settings = [seg.AREAMQZM1.k1, seg.AREAMQZM2.k1, seg.AREAMCVM1.angle, seg.AREAMQZM3.k1, seg.AREAMCHM1.angle]

    def step():
        if which == "control":
            scaled = actions[counter[0] % 300] * scale
            counter[0] += 1
            for i, target in enumerate(settings):
                target.copy_(scaled[i])
        if which == "rl":
            a = actions[counter[0] % 300]
            counter[0] += 1
            seg.AREAMQZM1.k1, seg.AREAMQZM2.k1, seg.AREAMCVM1.angle = a[0] * 10, a[1] * 10, a[2] * 1e-4
            seg.AREAMQZM3.k1, seg.AREAMCHM1.angle = a[3] * 10, a[4] * 1e-4
        seg.track(beam)
        return seg.AREABSCR1.reading
elif which == "linac":
    kw = {"dtype": dt, "device": "cuda"}
    els = []
    for i in range(16):
        els += [ca.Drift(rc.t(0.3, dt), **kw), ca.Quadrupole(rc.t(0.2, dt), k1=rc.t(3.0 if i % 2 else -3.0, dt), **kw),
                ca.Cavity(rc.t(1.0377, dt), voltage=rc.t(18e6, dt), phase=rc.t(-10.0, dt), frequency=rc.t(1.3e9, dt), **kw)]
    seg = ca.Segment(els)
    beam = ca.ParticleBeam.from_parameters(num_particles=10_000, energy=rc.t(1e8, dt), **kw)

    def step():
        with torch.no_grad():
            return seg.track(beam).particles
elif which == "c2":
    kw = {"dtype": dt, "device": "cuda"}
    els = []
    for i in range(50):
        els += [ca.Drift(rc.t(0.2, dt), **kw), ca.Quadrupole(rc.t(0.1, dt), k1=rc.t(4.2 if i % 2 == 0 else -4.2, dt), **kw)]
    seg = ca.Segment(els)
    beam = ca.ParticleBeam.from_parameters(num_particles=1_000_000, dtype=dt, device="cuda")

    def step():
        return seg.track(beam)
elif which == "dkd":
    kw = {"dtype": dt, "device": "cuda"}
    els = []
    for _ in range(25):
        els += [ca.Quadrupole(rc.t(0.2, dt), k1=rc.t(4.2, dt), tracking_method="drift_kick_drift", **kw),
                ca.Drift(rc.t(0.8, dt), tracking_method="drift_kick_drift", **kw),
                ca.Quadrupole(rc.t(0.2, dt), k1=rc.t(-4.2, dt), tracking_method="drift_kick_drift", **kw),
                ca.Drift(rc.t(0.8, dt), tracking_method="drift_kick_drift", **kw)]
    for e in els:
        e.dkd_precision = "storage"
    beam = ca.ParticleBeam.from_parameters(num_particles=10_000, dtype=dt, device="cuda")

    def step():
        b = beam
        for e in els[:10]:
            b = e.track(b)
        return b
else:
    kw = {"dtype": dt, "device": "cuda"}
    k1 = torch.nn.Parameter(rc.t(3.142, dt))
    seg = ca.Segment([ca.Drift(rc.t(1.0, dt)), ca.Quadrupole(rc.t(0.2, dt), k1=k1, **kw), ca.Drift(rc.t(1.0, dt)),
                      ca.Screen(is_active=True, name="scr", **kw)])
    beam = ca.ParticleBeam.from_parameters(num_particles=1_000_000, dtype=dt, device="cuda")

    def step():
        k1.grad = None
        seg.track(beam)
        seg.scr.get_read_beam().sigma_x.backward()

for _ in range(50):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(500):
    step()
torch.cuda.synchronize()
print(which, "us per step:", round((time.perf_counter() - t0) / 500 * 1e6, 1))
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats(os.environ.get("SORT", "tottime")).print_stats(int(os.environ.get("TOP", "28")))
