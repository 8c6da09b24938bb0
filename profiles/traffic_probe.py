#!/usr/bin/env python3
"""Launches the dominant kernel (apply_tile_kernel<float>) a fixed number of times at two sizes so
that rocprofv3 --pmc passes can attribute HBM traffic per launch:
  N = 1e6   the benchmark size (28 MB in + 28 MB out: resident in the 256 MiB Infinity Cache)
  N = 1.6e7 448 MB in + 448 MB out per launch: cannot be cache resident -> calibrates the counters
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cheetah_amd import _ops  # noqa: E402

R = torch.eye(7, device="cuda").reshape(1, 7, 7).contiguous()
R[0, 0, 1] = 0.8
for n in (1_000_000, 16_000_000):
    x = torch.randn(1, n, 7, device="cuda")
    y = torch.empty_like(x)
    for _ in range(20):
        _ops._apply_raw(x, R, 1, 1, 1, n)
    torch.cuda.synchronize()
print("probe done")
