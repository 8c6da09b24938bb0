#!/usr/bin/env python3
"""SpaceChargeKick of the reference on drawn configurations -> tests/golden/space_charge_random.npz: grids with three different
edge lengths (powers of two and not), grid extents between 2 and 4.5 sigma per axis, effect lengths from 1 cm to 1 m, beams
from gamma = 3 to 4000 with flat / long / round shapes, 300 particles each with non-uniform charges and some dead particles;
plus five VECTORISED set-ups (vector dimensions on the particles, the energy, the effect length, the survival probabilities).
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_random_sc.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import cheetah  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
f64 = {"dtype": torch.float64}
t = lambda v: torch.tensor(v, **f64)  # noqa: E731
rng = np.random.default_rng(777)
grids = [(16, 32, 64), (32, 16, 16), (24, 20, 28), (16, 16, 128), (40, 16, 32), (64, 64, 16)]
arrays = {"n_cases": np.asarray(len(grids))}
for i, grid in enumerate(grids):
    energy = float(np.exp(rng.uniform(np.log(1.6e6), np.log(2e9))))
    n = 300
    torch.manual_seed(50 + i)
    sig = [float(np.exp(rng.uniform(np.log(2e-5), np.log(2e-3)))) for _ in range(3)]
    beam = cheetah.ParticleBeam.from_parameters(num_particles=n, energy=t(energy), sigma_x=t(sig[0]), sigma_y=t(sig[1]), sigma_tau=t(sig[2]),
                                                sigma_px=t(1e-5), sigma_py=t(1e-5), sigma_p=t(1e-3), total_charge=t(float(rng.uniform(0.1e-9, 2e-9))),
                                                **f64)
    charges = beam.particle_charges * t(rng.uniform(0.5, 1.5, size=n))
    survival = t((rng.random(n) > 0.1).astype(np.float64))
    beam = cheetah.ParticleBeam(beam.particles, beam.energy, particle_charges=charges, survival_probabilities=survival, species=beam.species)
    ext = [float(rng.uniform(2.0, 4.5)) for _ in range(3)]
    L = float(np.exp(rng.uniform(np.log(1e-2), np.log(1.0))))
    sc = cheetah.SpaceChargeKick(effect_length=t(L), grid_shape=grid, grid_extent_x=t(ext[0]), grid_extent_y=t(ext[1]),
                                 grid_extent_tau=t(ext[2]), **f64)
    out = sc.track(beam)
    assert torch.isfinite(out.particles).all()
    arrays[f"grid_{i}"] = np.asarray(grid)
    arrays[f"energy_{i}"] = np.asarray(energy)
    arrays[f"length_{i}"] = np.asarray(L)
    arrays[f"extent_{i}"] = np.asarray(ext)
    arrays[f"in_{i}"] = beam.particles.numpy()
    arrays[f"charges_{i}"] = charges.numpy()
    arrays[f"survival_{i}"] = survival.numpy()
    arrays[f"out_{i}"] = out.particles.numpy()
    kick = (out.particles - beam.particles).abs().amax(dim=0)
    print(i, grid, f"gamma {energy / 510998.95:.1f}", "sigmas", [f"{s:.1e}" for s in sig], "max kick", [f"{float(k):.2e}" for k in kick[[1, 3, 5]]])
# ---- vectorised set-ups (drawn after the cases above, which keep their draws): vector dimensions on the particles, on the
# energy, on the effect length and on the survival probabilities, alone and combined (reference tests/test_space_charge_kick.py:74-160)
VEC = [  # (particles batch, energy shape, effect-length shape, survival batch)
    ((3,), (), (), ()),
    ((), (2,), (), ()),
    ((3,), (), (3,), ()),
    ((2,), (2,), (2,), (2,)),
    ((4,), (), (), (4,)),
]
arrays["n_vectorised"] = np.asarray(len(VEC))
for i, (pb, es, ls, sb) in enumerate(VEC):
    n = 200
    grid = [(16, 16, 16), (8, 16, 32), (16, 8, 8), (12, 10, 14), (16, 16, 8)][i]
    energy = np.exp(rng.uniform(np.log(5e6), np.log(1e9), es))
    sig = [float(np.exp(rng.uniform(np.log(5e-5), np.log(1e-3)))) for _ in range(3)]
    x = rng.standard_normal((*pb, n, 7)) * np.array([sig[0], 1e-5, sig[1], 1e-5, sig[2], 1e-3, 0.0])
    x[..., 6] = 1.0
    charges = rng.uniform(0.5, 1.5, n) * 1e-9 / n
    survival = (rng.random((*sb, n)) > 0.1).astype(np.float64)
    L = np.exp(rng.uniform(np.log(2e-2), np.log(0.5), ls))
    beam = cheetah.ParticleBeam(t(x), t(energy), particle_charges=t(charges), survival_probabilities=t(survival), **f64)
    sc = cheetah.SpaceChargeKick(effect_length=t(L), grid_shape=grid, **f64)
    out = sc.track(beam)
    assert torch.isfinite(out.particles).all()
    arrays[f"v_grid_{i}"], arrays[f"v_energy_{i}"], arrays[f"v_length_{i}"] = np.asarray(grid), np.asarray(energy), np.asarray(L)
    arrays[f"v_in_{i}"], arrays[f"v_charges_{i}"], arrays[f"v_survival_{i}"] = x, charges, survival
    arrays[f"v_out_{i}"] = out.particles.numpy()
    arrays[f"v_energy_out_{i}"] = out.energy.numpy()
    print("vectorised", i, pb, es, ls, sb, "->", tuple(out.particles.shape), "max kick", float((out.particles - beam.particles).abs().max()))
path = os.path.join(OUT, "space_charge_random.npz")
np.savez_compressed(path, **arrays)
print("->", path, f"{os.path.getsize(path) / 1024:.1f} KiB")
