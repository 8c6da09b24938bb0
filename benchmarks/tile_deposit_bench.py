#!/usr/bin/env python3
"""Tile deposit (chx_sc_tile_deposit) at 1e6 particles / 128^3 against the share of misfiled particles: the tile-ordered rows are
shifted by a fraction of a cell before the deposit."""
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
N, bins = int(os.environ.get("NPART", "1000000")), (128, 128, 128)
lib = _lib.lib()
beam = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=N, total_charge=t(1e-9), energy=t(2.5e8), radius_x=t(1e-3), radius_y=t(1e-3),
                                            radius_tau=t(1e-3), sigma_px=t(1e-6), sigma_py=t(1e-6), sigma_p=t(1e-6), **kw)
x = beam.particles.contiguous()
energy = torch.tensor([2.5e8], **kw)
gamma = energy / 510998.95069
beta = (1 - 1 / gamma**2).sqrt()
sig = torch.stack([beam.sigma_x, beam.sigma_y, beam.sigma_tau]).reshape(1, 3)
half = (3.0 * sig).contiguous()
cell = (2 * half / 128.0).contiguous()
extent = torch.stack([-half[0], half[0]], dim=-1).reshape(1, 3, 2).contiguous()
scale = torch.stack([torch.ones_like(beta), torch.ones_like(beta), -beta], dim=-1).contiguous()
dtt = torch.tensor([0.2 / 299792458.0], **kw)
phi = torch.zeros(1, 132, 132, 132, **kw)
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



def deposit_rows(rows):
    _ops.check(lib.chx_sc_tile_deposit(rows.data_ptr(), extent.data_ptr(), scale.data_ptr(), N, b3, dtc, state.data_ptr(), state.numel(),
                                       rho.data_ptr(), 0, _ops.stream_ptr()), "deposit")


sort()
deposit()
tile_gather(0)
rows_sorted = out.clone()          # rows in tile order (zero potential: positions unchanged)
ref = _ops.sc_deposit_overwrite(x.reshape(1, N, 7), q.reshape(1, N), w.reshape(1, N), extent, scale, 1, N, bins).reshape(bins)
print("state rows: us", round(timeit(deposit), 1), " max |rho - generic| / max rho", float((rho - ref).abs().max() / ref.abs().max()))
hdr = state[:32].view(torch.int32)
for sx, sy, sz in ((0, 0, 0), (0.08, 0, 0), (0.25, 0, 0), (0.5, 0, 0), (1.0, 0, 0), (2.0, 0, 0), (0.5, 0.5, 0.5), (1.0, 1.0, 1.0), (8.0, 8.0, 8.0)):
    rows = rows_sorted.clone()
    rows[:, 0] += sx * cell[0, 0]
    rows[:, 2] += sy * cell[0, 1]
    rows[:, 4] += sz * cell[0, 2] / beta[0]
    hdr[2] = 0
    deposit_rows(rows)
    torch.cuda.synchronize()
    mis = int(hdr[4])
    print(f"shift ({sx}, {sy}, {sz}) cells: misfiled {mis / N * 100:5.1f} %  deposit us {timeit(lambda: deposit_rows(rows)):6.1f}")
