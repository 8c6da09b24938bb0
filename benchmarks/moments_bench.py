#!/usr/bin/env python3
"""Time chx_moments on 1e6 fp32 particles: CHX_TUNE_MOMENTS_WGS
selects the workgroup count. Run under `rocprofv3 --kernel-trace --stats` for the kernel durations."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402
from cheetah_amd import _ops  # noqa: E402

for n in [int(v) for v in os.environ.get('MOMENTS_N', '1000000,100000,10000').split(',')]:
    beam = ca.ParticleBeam.from_parameters(num_particles=n, dtype=torch.float32, device="cuda")
    x, w = beam.particles, beam.survival_probabilities
    for fused in (False,):
        for _ in range(20):
            _ops.moments(x, w)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(200):
            _ops.moments(x, w)
        e1.record()
        torch.cuda.synchronize()
        print(f"N={n} fused={fused} WGS={os.environ.get('CHX_TUNE_MOMENTS_WGS', 'default')}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per "
              f"chx_moments (sigma_x {float(_ops.moments(x, w)[8].sqrt()):.9e})")
