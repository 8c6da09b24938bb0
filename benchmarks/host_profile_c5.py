#!/usr/bin/env python3
"""cProfile of the host side of config C5 (forward + backward) and of the control loop: where the Python time goes."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import run_configs as rc  # noqa: E402
import cheetah_amd as ca  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c5"
dt = torch.float32
if which == "c5":
    kw = {"dtype": dt, "device": "cuda"}
    k1 = torch.nn.Parameter(rc.t(3.142, dt))
    seg = ca.Segment([ca.Drift(rc.t(1.0, dt)), ca.Quadrupole(rc.t(0.2, dt), k1=k1, **kw), ca.Drift(rc.t(1.0, dt)),
                      ca.Screen(is_active=True, name="scr", **kw)])
    beam = ca.ParticleBeam.from_parameters(num_particles=1_000_000, dtype=dt, device="cuda")

    def step():
        k1.grad = None
        with torch.no_grad():
            k1.add_(0.0)
        seg.track(beam)
        loss = seg.scr.get_read_beam().sigma_x
        loss.backward()
else:
    seg = rc.ares_subcell(dt, rc.t(8.2, dt))
    seg.AREABSCR1.is_active = True
    beam = ca.ParticleBeam.from_twiss(beta_x=rc.t(3.14, dt), beta_y=rc.t(42.0, dt), num_particles=10_000, dtype=dt, device="cuda")
    actions = torch.randn(300, 5, device="cuda", dtype=dt)
    counter = [0]

    def step():
        a = actions[counter[0] % 300]
        counter[0] += 1
        seg.AREAMQZM1.k1, seg.AREAMQZM2.k1, seg.AREAMCVM1.angle = a[0] * 10, a[1] * 10, a[2] * 1e-4
        seg.AREAMQZM3.k1, seg.AREAMCHM1.angle = a[3] * 10, a[4] * 1e-4
        seg.track(beam)
        return seg.AREABSCR1.reading

for _ in range(20):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats(os.environ.get("SORT", "cumulative")).print_stats(45)
