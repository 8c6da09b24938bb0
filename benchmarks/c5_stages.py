#!/usr/bin/env python3
"""Config C5 (forward + backward through [Drift, Quad(k1), Drift, Screen], 1e6 particles) by stage: host time of track /
read beam / sigma / backward, the whole step, and the torch profiler's per-op table (CPU and GPU time)."""
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
beam = ca.ParticleBeam.from_parameters(num_particles=1_000_000, dtype=dt, device="cuda")


def stage_times(n=300):
    T = [0.0] * 4
    for _ in range(n):
        k1.grad = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seg.track(beam)
        t1 = time.perf_counter()
        rb = seg.scr.get_read_beam()
        t2 = time.perf_counter()
        loss = rb.sigma_x
        t3 = time.perf_counter()
        loss.backward()
        t4 = time.perf_counter()
        T[0] += t1 - t0
        T[1] += t2 - t1
        T[2] += t3 - t2
        T[3] += t4 - t3
    return [1e6 * v / n for v in T]


def full(n=300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        k1.grad = None
        seg.track(beam)
        seg.scr.get_read_beam().sigma_x.backward()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


stage_times(30)
print("host us per stage (track, get_read_beam, sigma_x, backward):", [round(v, 1) for v in stage_times()])
print("whole step us:", round(full(), 1))
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(20):
        k1.grad = None
        seg.track(beam)
        seg.scr.get_read_beam().sigma_x.backward()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=55))
