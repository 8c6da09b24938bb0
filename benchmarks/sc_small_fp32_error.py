#!/usr/bin/env python3
"""Measured error of the float32 SpaceChargeKick on the SMALL grids of tests/golden/space_charge.npz against the float64 oracle on
the same float32 inputs — the number behind the bound of tests/test_gpu_parity.py::test_space_charge_kick_vs_reference[f32]."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cheetah_amd as ca  # noqa: E402
from oracle import chx_oracle as oracle  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "space_charge.npz"))
dt = torch.float32
for gi in (0, 1):
    k = f"g{gi}_f32"
    grid = tuple(int(v) for v in g[f"{k}_grid"])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    beam = ca.ParticleBeam(dev(g[f"{k}_in"]), torch.tensor(float(g["energy"]), dtype=dt, device="cuda"),
                           particle_charges=dev(g[f"{k}_charges"]), survival_probabilities=dev(g[f"{k}_survival"]),
                           species=ca.Species("electron", dtype=dt, device="cuda"))
    sc = ca.SpaceChargeKick(effect_length=torch.tensor(float(g["effect_length"]), dtype=dt, device="cuda"), grid_shape=grid, dtype=dt,
                            device="cuda")
    got = sc.track(beam).particles.cpu().numpy().astype(np.float64)
    inp = g[f"{k}_in"].astype(np.float64)
    truth = oracle.space_charge_kick(inp[None], float(g["energy"]), g[f"{k}_charges"].astype(np.float64),
                                     g[f"{k}_survival"].astype(np.float64), float(g["effect_length"]), grid_shape=grid)[0]
    kick = np.max(np.abs(truth - inp), axis=0)
    err = np.max(np.abs(got - truth), axis=0)
    eps_term = 2 * np.finfo(np.float32).eps * np.max(np.abs(truth), axis=0)
    ref32 = g[f"{k}_out"].astype(np.float64)
    print(f"grid {grid} N={inp.shape[0]}: max err / kick (px, py, delta) =", [f"{err[c] / kick[c]:.2e}" for c in (1, 3, 5)],
          " rounding floor 2 eps max|coord| / kick =", [f"{eps_term[c] / kick[c]:.2e}" for c in (1, 3, 5)],
          " reference's own fp32 run vs ours / kick =", [f"{np.max(np.abs(got[:, c] - ref32[:, c])) / kick[c]:.2e}" for c in (1, 3, 5)])
