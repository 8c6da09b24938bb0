#!/usr/bin/env python3
"""Direct (global atomics) vs sorted (LDS-privatised) cloud-in-cell deposit at the C3 / C4 sizes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cheetah_amd import _ops  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


torch.manual_seed(0)
N = 1_000_000
x = torch.zeros(N, 7, device="cuda")
x[:, 0], x[:, 2], x[:, 4] = (torch.randn(N, device="cuda") for _ in range(3))
q = torch.full((N,), 1e-15, device="cuda")
for name, cols, bins, ext in (("3-D 128^3, +-3 sigma", (0, 2, 4), (128, 128, 128), [[-3.0, 3.0]] * 3),
                              ("3-D 32^3, +-3 sigma", (0, 2, 4), (32, 32, 32), [[-3.0, 3.0]] * 3),
                              ("2-D 2448x2040, sigma = 50x70 px", (0, 2), (2448, 2040), [[-24.48, 24.48], [-14.57, 14.57]]),
                              ("2-D 1024x1024, sigma = 170 px", (0, 2), (1024, 1024), [[-3.0, 3.0], [-3.0, 3.0]])):
    e = torch.tensor(ext, device="cuda")
    for mode in ("direct", "sorted"):
        us = timeit(lambda: _ops.cic_deposit(x, cols, bins, e, charge=q, mode=mode))
        print(f"{name:36s} {mode:7s} {us:9.1f} us")
