#!/usr/bin/env python3
"""Golden vectors for the GENERAL second-order tensor (k2 together with k1 and / or hx): the reference's
`cheetah.track_methods.base_ttensor` (track_methods.py:80-281) on drawn settings and on the special points of its divided
differences (a == b, a == 0, b == 0), with the gradient of a fixed random contraction of T with respect to every input.
Run in the build container: PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/generate_golden_ttensor_general.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import cheetah  # noqa: E402
from cheetah.track_methods import base_ttensor  # noqa: E402

torch.set_default_dtype(torch.float64)
rng = np.random.default_rng(20260929)
rows = []
for _ in range(40):
    rows.append([rng.uniform(0.05, 1.5), rng.uniform(-8, 8), rng.uniform(-40, 40), rng.uniform(-1.5, 1.5), 10 ** rng.uniform(6.8, 10)])
# special points, exactly representable: k1 = 0 (b = 0), hx^2 = -k1 (a = 0), hx^2 = -2 k1 (a == b), only k2, low energy.
# (A kx2 that is merely rounding-small is NOT among them: there the reference's closed forms cancel catastrophically and its
# own digits mean nothing, while the builder kernel switches to series.)
rows += [[0.4, 0.0, 12.0, 0.7, 1e8], [0.4, -0.25, 12.0, 0.5, 1e8], [0.4, -0.125, -9.0, 0.5, 1e8], [0.6, 0.0, 5.0, 0.0, 6e6],
         [0.3, 2.0, 0.0, 0.0, 1e8], [0.3, 0.0, 0.0, 0.5, 1e8], [0.25, -3.0, 7.0, 0.0, 2e7], [1.2, 1e-9, 3.0, 1e-5, 1e9]]
settings = np.asarray(rows)
coef = rng.standard_normal((7, 7, 7))
species = cheetah.Species("electron")
T_all, G_all = [], []
for L, k1, k2, hx, E in settings:
    args = [torch.tensor(v, requires_grad=True) for v in (L, k1, k2, hx, E)]
    T = base_ttensor(args[0], args[1], args[2], args[3], species, args[4])
    loss = (T * torch.from_numpy(coef)).sum()
    grads = torch.autograd.grad(loss, args, allow_unused=True)
    T_all.append(T.detach().numpy())
    G_all.append([0.0 if g is None else float(g) for g in grads])
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ttensor_general.npz")
np.savez_compressed(out, settings=settings, coef=coef, T=np.asarray(T_all), grads=np.asarray(G_all),
                    mass_eV=float(species.mass_eV), torch_version=torch.__version__)
print("wrote", out, np.asarray(T_all).shape)
