#!/usr/bin/env python3
"""Config C5 (forward + backward through [Drift, Quad(k1), Drift, Screen], 1e6 particles) captured ONCE into a device graph
(torch.cuda.CUDAGraph = hipGraph on ROCm) and replayed: the host side of a step is one graph launch instead of ~270 us of Python and
autograd bookkeeping. libchx's launches go to the capturing stream like any other kernel (plain launches on the current stream, no
synchronisation, workspaces from torch's allocator); the settings are read through their pointers, so an optimiser that updates
k1 in place between replays is seen by the replayed kernels."""
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


def step():
    seg.track(beam)
    loss = seg.scr.get_read_beam().sigma_x
    loss.backward()
    return loss


def eager(n=200):
    for _ in range(20):
        k1.grad = None
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        k1.grad = None
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


# capture (the recipe of torch's CUDA-graphs notes: warm up on a side stream, then capture forward + backward). The capture comes
# BEFORE any backward pass on the default stream: a Parameter whose AccumulateGrad node was created there breaks the capture.
def clean_step():
    k1.grad = None
    return step()


captured = ca.graph.capture(clean_step, constant_beam=True)      # an optimisation over k1: the incoming beam stays
graph, static_loss = captured.graph, captured.outputs
static_grad = k1.grad
graph.replay()
torch.cuda.synchronize()
print("replayed loss, grad:", float(static_loss), float(static_grad))
# a new setting, in place: the replay follows it
with torch.no_grad():
    k1.fill_(2.5)
graph.replay()
torch.cuda.synchronize()
g_replay, l_replay = float(static_grad), float(static_loss)
for _ in range(20):
    graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    graph.replay()
torch.cuda.synchronize()
replay_us = (time.perf_counter() - t0) / 500 * 1e6
print("graph replay us per step", round(replay_us, 1))

# the same step eagerly (after the capture: see above), same numbers
k1.grad = None
l_eager = float(step())
print("k1 = 2.5: replay", l_replay, g_replay, " eager", l_eager, float(k1.grad))
assert abs(l_replay - l_eager) <= 1e-6 * abs(l_eager) and abs(g_replay - float(k1.grad)) <= 1e-5 * abs(float(k1.grad))
eager_us = eager()
print("eager us per step", round(eager_us, 1))
import json  # noqa: E402

print(json.dumps({"c5_graph": {"graph_replay_us": replay_us, "eager_us": eager_us, "loss": l_replay, "grad": g_replay,
                               "loss_eager": l_eager, "grad_eager": float(k1.grad)}}))
