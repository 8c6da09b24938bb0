#!/usr/bin/env python3
"""Time chx_moments (one-pass, 29 sums) on 1e6 fp32 particles; CHX_TUNE_MOMENTS_WGS selects the workgroup count."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402
from cheetah_amd import _ops  # noqa: E402

beam = ca.ParticleBeam.from_parameters(num_particles=1_000_000, dtype=torch.float32, device="cuda")
x, w = beam.particles, beam.survival_probabilities
for _ in range(20):
    _ops.moments(x, w)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(200):
    _ops.moments(x, w)
e1.record()
torch.cuda.synchronize()
print(f"CHX_TUNE_MOMENTS_WGS={os.environ.get('CHX_TUNE_MOMENTS_WGS', 'default')}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per chx_moments "
      f"(sigma_x {float(_ops.moments(x, w)[8].sqrt()):.6e})")
