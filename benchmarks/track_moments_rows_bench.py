#!/usr/bin/env python3
"""chx_track_moments, rows path (one beam, one map per setting): 4096 settings x 1e5 particles, fp32 and fp64; the fp32 result
against the fp64 one."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402,F401
from cheetah_amd import _ops  # noqa: E402

B, N = 4096, 100_000
g = torch.Generator("cuda").manual_seed(3)
beam = ca.ParticleBeam.from_parameters(num_particles=N, dtype=torch.float64, device="cuda")
x64 = beam.particles
R64 = torch.eye(7, dtype=torch.float64, device="cuda") + 0.2 * torch.randn(B, 7, 7, dtype=torch.float64, device="cuda", generator=g)
R64[:, 6] = 0
R64[:, 6, 6] = 1
w64 = torch.rand(N, dtype=torch.float64, device="cuda", generator=g)
res = {}
for dt in (torch.float32, torch.float64):
    x, R, w = x64.to(dt), R64.to(dt), w64.to(dt)
    for _ in range(2):
        out = _ops.track_moments(x, w, R)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = _ops.track_moments(x, w, R)
    torch.cuda.synchronize()
    res[dt] = out
    print(dt, "ms per call", round((time.perf_counter() - t0) / 5 * 1e3, 3))
ref = _ops.track_moments(x64.float().double(), w64.float().double(), R64.float().double())
err = ((res[torch.float32] - ref).abs() / (ref.abs() + 1e-300))
print("fp32 vs fp64 arithmetic on the same fp32 inputs: max rel", float(err[:, :8].max()), float(err[:, 8:].max()))
