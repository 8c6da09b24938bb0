#!/usr/bin/env python3
"""The control loop on the ARES subcell (1e4 particles, fp32), us per step: five settings ASSIGNED from an action tensor (moves the
epoch), the same five written IN PLACE, track alone; and a 100-element FODO with one strength assigned per step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import run_configs as rc
import cheetah_amd as ca
dt = torch.float32


def timeit(fn, reps=300, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


seg = rc.ares_subcell(dt, rc.t(8.2, dt))
seg.AREABSCR1.is_active = True
beam = ca.ParticleBeam.from_twiss(beta_x=rc.t(3.14, dt), beta_y=rc.t(42.0, dt), num_particles=10_000, dtype=dt, device="cuda")
pbeam = ca.ParameterBeam.from_twiss(beta_x=rc.t(3.14, dt), beta_y=rc.t(42.0, dt), dtype=dt, device="cuda")
actions = torch.randn(300, 5, device="cuda", dtype=dt)
counter = [0]


def assign(b):
    a = actions[counter[0] % 300]
    counter[0] += 1
    seg.AREAMQZM1.k1, seg.AREAMQZM2.k1, seg.AREAMCVM1.angle = a[0] * 10, a[1] * 10, a[2] * 1e-4
    seg.AREAMQZM3.k1, seg.AREAMCHM1.angle = a[3] * 10, a[4] * 1e-4
    seg.track(b)
    return seg.AREABSCR1.reading


def in_place(b):
    a = actions[counter[0] % 300]
    counter[0] += 1
    seg.AREAMQZM1.k1.copy_(a[0] * 10); seg.AREAMQZM2.k1.copy_(a[1] * 10); seg.AREAMCVM1.angle.copy_(a[2] * 1e-4)
    seg.AREAMQZM3.k1.copy_(a[3] * 10); seg.AREAMCHM1.angle.copy_(a[4] * 1e-4)
    seg.track(b)
    return seg.AREABSCR1.reading


with torch.no_grad():
    res = {"assign_particles": timeit(lambda: assign(beam)), "assign_parameter_beam": timeit(lambda: assign(pbeam)),
           "in_place_particles": timeit(lambda: in_place(beam)), "track_only_particles": timeit(lambda: seg.track(beam)),
           "track_only_parameter_beam": timeit(lambda: seg.track(pbeam))}
    kw = {"dtype": dt, "device": "cuda"}
    els = []
    for i in range(50):
        els += [ca.Quadrupole(rc.t(0.2, dt), k1=rc.t(4.2 if i % 2 == 0 else -4.2, dt), **kw), ca.Drift(rc.t(0.8, dt), **kw)]
    fodo = ca.Segment(els)

    def fodo_step():
        a = actions[counter[0] % 300]
        counter[0] += 1
        els[14].k1 = a[0] * 10
        fodo.track(beam)

    res["fodo100_assign_one"] = timeit(fodo_step)
print("  ".join(f"{k} {v:6.1f}" for k, v in res.items()))
