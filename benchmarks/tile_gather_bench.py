#!/usr/bin/env python3
"""Gather + kick at 1e6 particles / 128^3: the untiled kernel (chx_sc_gather_kick_phi) on rows in the caller's order and on
tile-ordered rows, the tile kernels (chx_sc_tile_gather_kick), and the tile deposit against the generic sorted deposit."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402
from cheetah_amd import _lib, _ops  # noqa: E402

dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)  # noqa: E731
N, bins = 1_000_000, (128, 128, 128)
lib = _lib.lib()
beam = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=N, total_charge=t(1e-9), energy=t(2.5e8), radius_x=t(1e-3), radius_y=t(1e-3),
                                            radius_tau=t(1e-3), sigma_px=t(1e-6), sigma_py=t(1e-6), sigma_p=t(1e-6), **kw)
x = beam.particles.contiguous()
if os.environ.get("GAUSS"):          # C4's shape: a Gaussian bunch (the ellipsoid is uniform)
    x = ca.ParticleBeam.from_parameters(num_particles=N, sigma_x=t(1e-3), sigma_y=t(1e-3), sigma_tau=t(1e-3), energy=t(2.5e8), **kw).particles.contiguous()
    beam = ca.ParticleBeam(x, t(2.5e8), particle_charges=beam.particle_charges, **kw)
energy = torch.tensor([2.5e8], **kw)
gamma = energy / 510998.95069
beta = (1 - 1 / gamma**2).sqrt()
sig = torch.stack([beam.sigma_x, beam.sigma_y, beam.sigma_tau]).reshape(1, 3)
half = (3.0 * sig).contiguous()
cell = (2 * half / 128.0).contiguous()
extent = torch.stack([-half[0], half[0]], dim=-1).reshape(1, 3, 2).contiguous()
scale = torch.stack([torch.ones_like(beta), torch.ones_like(beta), -beta], dim=-1).contiguous()
if os.environ.get("PRESORT"):
    # experiment: rows handed over in CELL order (the tile sort keeps the order of its input inside a tile, coarsely): what would a
    # cell-ordered beam buy the deposit (equal-cell neighbours) and the gather (fewer distinct lines per load)?
    pos = torch.stack([x[:, 0], x[:, 2], -beta * x[:, 4]], dim=1)
    idx = ((pos + half) / cell).floor().clamp(0, 127).long()
    key = (idx[:, 0] * 128 + idx[:, 1]) * 128 + idx[:, 2]
    order = key.argsort()
    if os.environ["PRESORT"] == "shuffle":
        order = torch.randperm(N, device=x.device)
    x = x[order].contiguous()
dtt = torch.tensor([0.2 / 299792458.0], **kw)
phi = torch.randn(1, 132, 132, 132, **kw) * 1e3
q = beam.particle_charges.contiguous()
w = beam.survival_probabilities.contiguous()
b3, dtc = _ops._bins3(bins), _ops.dtype_code(dt)
state = _ops.sc_tile_state(N, bins, dt, x.device)
rho = torch.empty(bins, **kw)


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def sort():
    _ops.check(lib.chx_sc_tile_sort(x.data_ptr(), q.data_ptr(), w.data_ptr(), extent.data_ptr(), scale.data_ptr(), N, b3, dtc,
                                    state.data_ptr(), state.numel(), _ops.stream_ptr()), "sort")


def deposit():
    _ops.check(lib.chx_sc_tile_deposit(None, extent.data_ptr(), scale.data_ptr(), N, b3, dtc, state.data_ptr(), state.numel(),
                                       rho.data_ptr(), 0, _ops.stream_ptr()), "deposit")


out = torch.empty_like(x)


def tile_gather(unperm):
    _ops.check(lib.chx_sc_tile_gather_kick(None, phi.data_ptr(), half.data_ptr(), cell.data_ptr(), gamma.data_ptr(), energy.data_ptr(),
                                           dtt.data_ptr(), 510998.95069, N, b3, dtc, None, state.data_ptr(), state.numel(), unperm,
                                           out.data_ptr(), _ops.stream_ptr()), "gather")


sort()
deposit()
tile_gather(0)
rows_sorted = out.clone()          # kicked rows in tile order (positions unchanged by the kick)
print("sort us", round(timeit(sort), 1))
print("tile deposit (3 kernels) us", round(timeit(deposit), 1))
print("generic sorted deposit us", round(timeit(lambda: _ops.sc_deposit_overwrite(x.reshape(1, N, 7), q.reshape(1, N), w.reshape(1, N), extent, scale, 1, N, bins)), 1))
print("tile gather, tile order out us", round(timeit(lambda: tile_gather(0)), 1))
print("tile gather, caller order out us", round(timeit(lambda: tile_gather(1)), 1))
for name, rows in (("caller order", x), ("tile order", rows_sorted)):
    r = rows.reshape(1, N, 7)
    print(f"untiled gather on rows in {name} us",
          round(timeit(lambda: _ops.sc_gather_kick_phi(r, phi, half, cell, gamma, energy, dtt, 510998.95069, 1, N, bins)), 1))
