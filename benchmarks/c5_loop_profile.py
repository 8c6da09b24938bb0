#!/usr/bin/env python3
"""Host profile of the bench's C5 loop (benchmarks/run_configs.py c5: grad reset, in-place version bump of k1, track, sigma_x of the
read beam, backward) against the bare step of benchmarks/c5_stages.py: which part of the loop costs what."""
import cProfile
import os
import pstats
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
beam = ca.ParticleBeam.from_parameters(num_particles=1_000_000, dtype=dt, device="cuda")
res = {}


def bare():
    k1.grad = None
    seg.track(beam)
    seg.scr.get_read_beam().sigma_x.backward()


def bump():
    k1.grad = None
    with torch.no_grad():
        k1.add_(0.0)
    seg.track(beam)
    seg.scr.get_read_beam().sigma_x.backward()


def full():
    k1.grad = None
    with torch.no_grad():
        k1.add_(0.0)
    seg.track(beam)
    loss = seg.scr.get_read_beam().sigma_x
    loss.backward()
    res["sigma_x"], res["dk1"] = loss.detach(), k1.grad


def timeit(fn, reps=300, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for name, fn in (("bare step", bare), ("+ version bump of k1", bump), ("the bench's loop", full), ("bare step", bare)):
    print(f"{name:28s} {timeit(fn):7.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    bump()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
