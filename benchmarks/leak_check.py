#!/usr/bin/env python3
"""Do the persistent plans, workspaces and caches grow? Device memory (torch allocator) and host RSS before and after thousands
of steps of the control loop, hundreds of space-charge tracks and optimisation steps."""
import gc
import os
import resource
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402
from benchmarks import run_configs as rc  # noqa: E402

dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)  # noqa: E731


def rss_mb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


def measure(label, fn, n, warm=50):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); gc.collect()
    a0, r0 = torch.cuda.memory_allocated(), rss_mb()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); gc.collect()
    a1, r1 = torch.cuda.memory_allocated(), rss_mb()
    print(f"{label:44s} {n:6d} steps: device allocated {a0 / 2**20:9.2f} -> {a1 / 2**20:9.2f} MiB   host max RSS {r0:8.1f} -> {r1:8.1f} MiB")


seg = rc.ares_subcell(dt, rc.t(8.2, dt))
seg.AREABSCR1.is_active = True
beam = ca.ParticleBeam.from_parameters(num_particles=10_000, **kw)
i = [0]


def rl():
    i[0] += 1
    seg.AREAMQZM1.k1 = t(8.0 + 1e-3 * (i[0] % 7))
    seg.AREAMCVM1.angle = t(1e-4 * (i[0] % 5))
    seg.track(beam)
    return seg.AREABSCR1.reading


measure("control loop (new setting tensors every step)", rl, 5000)
els = []
for k in range(4):
    els += [ca.Drift(t(0.1), **kw), ca.SpaceChargeKick(t(0.2), grid_shape=(64, 64, 64), **kw), ca.Drift(t(0.1), **kw),
            ca.Quadrupole(t(0.1), k1=t(4.2 if k % 2 == 0 else -4.2), **kw)]
seg4 = ca.Segment(els)
b4 = ca.ParticleBeam.from_parameters(num_particles=200_000, **kw)
measure("4 space-charge kicks at 64^3, 2e5 particles", lambda: seg4.track(b4), 300, warm=10)
k1 = torch.nn.Parameter(t(3.0))
seg5 = ca.Segment([ca.Drift(t(1.0), **kw), ca.Quadrupole(t(0.2), k1=k1, **kw), ca.Drift(t(1.0), **kw), ca.Screen(is_active=True, name="scr", **kw)])
b5 = ca.ParticleBeam.from_parameters(num_particles=100_000, **kw)


def opt():
    k1.grad = None
    seg5.track(b5)
    seg5.scr.get_read_beam().sigma_x.backward()
    with torch.no_grad():
        k1.add_(1e-4 * k1.grad.sign())


measure("optimisation step (forward + backward)", opt, 1000)
vec = ca.Segment([ca.Drift(t(1.0), **kw), ca.Quadrupole(t(0.2), k1=torch.linspace(-3, 3, 64, **kw), **kw), ca.Drift(t(1.0), **kw)])
measure("vectorised track (64 settings)", lambda: vec.track(beam), 1000)
pb = ca.ParameterBeam.from_parameters(**kw)
measure("ParameterBeam control step", lambda: (seg.track(pb), seg.AREABSCR1.reading), 3000)
