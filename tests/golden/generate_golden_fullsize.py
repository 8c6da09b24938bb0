#!/usr/bin/env python3
"""Full-size reference numbers for config C4 (tests/golden/fullsize_c4.npz): the REAL reference tracks the 50-element
space-charge linac with 1e6 particles on a 128^3 grid in fp64; stored are scalars and a 3985-particle sample (every 251st
particle) after the first kick and at the end of the lattice — no 28 MB particle arrays: the inputs are regenerated
bit-identically from tests/fullsize_inputs.py on both sides.

Run in the build container only (imports /root/reference read-only; ~1 min, ~6 GB):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_golden_fullsize.py
"""
import os
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import cheetah  # noqa: E402  (the reference)
from tests import fullsize_inputs as fi  # noqa: E402

torch.set_num_threads(8)
f64 = torch.float64
t = lambda v: torch.tensor(v, dtype=f64)  # noqa: E731


def stats(beam):
    p = beam.particles[:, :6]
    return np.concatenate([p.mean(dim=0).numpy(), p.std(dim=0).numpy()])


def main():
    x = fi.c4_particles()
    beam = cheetah.ParticleBeam(torch.from_numpy(x), t(fi.C4_ENERGY), particle_charges=torch.from_numpy(fi.c4_charges()),
                                dtype=f64)
    arrays = {"in_stats": stats(beam)}
    sl = slice(None, None, fi.C4_SAMPLE_STRIDE)
    t0 = time.time()
    first = True
    for cell in range(fi.C4_CELLS):
        elements = [cheetah.Drift(t(0.1), dtype=f64),
                    cheetah.SpaceChargeKick(t(0.2), grid_shape=fi.C4_GRID, dtype=f64),
                    cheetah.Drift(t(0.1), dtype=f64),
                    cheetah.Quadrupole(t(0.1), k1=t(fi.c4_quad_k1(cell)), dtype=f64),
                    cheetah.Drift(t(0.1), dtype=f64)]
        for e in elements:
            before = beam
            beam = e.track(beam)
            if first and isinstance(e, cheetah.SpaceChargeKick):
                first = False
                arrays["kick1_in_sample"] = before.particles[sl].numpy()
                arrays["kick1_out_sample"] = beam.particles[sl].numpy()
                arrays["kick1_stats"] = stats(beam)
                # intermediates of the first kick (same calls as SpaceChargeKick.track, space_charge_kick.py:531-556)
                flat = cheetah.ParticleBeam(before.particles.unsqueeze(0), before.energy.unsqueeze(0),
                                            particle_charges=before.particle_charges.unsqueeze(0), dtype=f64)
                half = torch.stack([e.grid_extent_x * flat.sigma_x, e.grid_extent_y * flat.sigma_y,
                                    e.grid_extent_tau * flat.sigma_tau], dim=-1)
                cellsz = 2 * half / torch.tensor(fi.C4_GRID, dtype=f64)
                xp = flat.to_xyz_pxpypz()
                rho = e._array_rho(flat, xp, cellsz, half)
                phi = e._solve_poisson_equation(flat, xp, cellsz, half)
                g = fi.C4_GRID
                arrays["kick1_half"] = half.numpy()
                arrays["kick1_rho_sum"] = np.asarray(float(rho.sum()))
                arrays["kick1_rho_max"] = np.asarray(float(rho.abs().max()))
                arrays["kick1_rho_line"] = rho[0, : g[0], g[1] // 2, g[2] // 2].numpy()
                arrays["kick1_phi_absmax"] = np.asarray(float(phi.abs().max()))
                arrays["kick1_phi_line"] = phi[0, :, g[1] // 2, g[2] // 2].numpy()
                arrays["kick1_phi_diag"] = phi[0].diagonal(dim1=0, dim2=1).diagonal(dim1=0, dim2=1).numpy()[0] \
                    if False else np.asarray([float(phi[0, i, i, i]) for i in range(0, g[0], 4)])
                print(f"first kick done after {time.time() - t0:.1f} s")
    arrays["out_stats"] = stats(beam)
    arrays["out_sample"] = beam.particles[sl].numpy()
    arrays["energy"] = np.asarray(fi.C4_ENERGY)
    path = os.path.join(HERE, "fullsize_c4.npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote fullsize_c4.npz: {os.path.getsize(path) / 1024:.1f} KiB in {time.time() - t0:.1f} s")
    print("sigma growth x:", arrays["out_stats"][6] / arrays["in_stats"][6])


if __name__ == "__main__":
    main()
