#!/usr/bin/env python3
"""The 100-launch element-by-element step at the size one rank of an 8-GPU strong run holds (1.25e5 particles): wall time per
step, and (under rocprofv3 --kernel-trace) where it goes — kernel durations against the gaps between consecutive kernels."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cheetah_amd as ca  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
seg = bench.build_fodo(ca, torch, "cuda", torch.float32)
torch.manual_seed(4321)
beam = ca.ParticleBeam.from_parameters(num_particles=n, dtype=torch.float32, device="cuda")
for fused in (False, True):
    for _ in range(5):
        seg.track_elementwise(beam, fused=fused)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        seg.track_elementwise(beam, fused=fused)
    torch.cuda.synchronize()
    print("fused" if fused else "100 launches", n, "particles:", round((time.perf_counter() - t0) / 30 * 1e3, 4), "ms per step")
