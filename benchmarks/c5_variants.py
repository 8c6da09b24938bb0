#!/usr/bin/env python3
"""Config C5's eager step with and without the optimiser-like in-place touch of k1, us per step (what each piece of the step costs)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import run_configs as rc  # noqa: E402
import cheetah_amd as ca  # noqa: E402

dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
k1 = torch.nn.Parameter(rc.t(3.142, dt))
seg = ca.Segment([ca.Drift(rc.t(1.0, dt)), ca.Quadrupole(rc.t(0.2, dt), k1=k1, **kw), ca.Drift(rc.t(1.0, dt)),
                  ca.Screen(is_active=True, name="scr", **kw)])
beam = ca.ParticleBeam.from_parameters(num_particles=int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, dtype=dt, device="cuda")


def timed(fn, n=300, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


def plain():
    k1.grad = None
    seg.track(beam)
    seg.scr.get_read_beam().sigma_x.backward()


def touched():
    k1.grad = None
    with torch.no_grad():
        k1.add_(0.0)
    seg.track(beam)
    seg.scr.get_read_beam().sigma_x.backward()


def forward_only():
    seg.track(beam)
    return seg.scr.get_read_beam().sigma_x


def track_only():
    seg.track(beam)


def touch_only():
    with torch.no_grad():
        k1.add_(0.0)


def engine_floor():
    k1.grad = None
    (k1 * 2.0).backward()


def engine_floor_sqrt():
    k1.grad = None
    (k1 * 2.0).sqrt().backward()


print("cpu threads", torch.get_num_threads(), "engine floor (one mul node)", round(timed(engine_floor), 1), "two nodes", round(timed(engine_floor_sqrt), 1))
print({"plain": round(timed(plain), 1), "touched": round(timed(touched), 1), "forward_only": round(timed(forward_only), 1),
       "track_only": round(timed(track_only), 1), "touch_only": round(timed(touch_only), 1), "plain_again": round(timed(plain), 1)})
