#!/usr/bin/env python3
"""Every derived beam property of the reference on drawn beams -> tests/golden/beam_properties_random.npz: eight ParticleBeams
(correlated 6-D distributions from a random linear mixing, unequal and partly negative charges, survival probabilities with
exact zeros, electrons / positrons / protons / a custom ion, energies from gamma = 1.3 to 10^4, vector shapes (), (3,) and
(2, 2) on the particles or only on the energy) and the ParameterBeams with the same moments: all properties of
particles/beam.py (relativistic factors, p0c, projected / geometric / normalised emittances, beta, alpha, the four
dispersions), of particle_beam.py (total_charge, particle counts, 6 means, 6 sigmas, 15 covariances, energies, momenta) and of
parameter_beam.py, in float64.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_random_beam_properties.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import cheetah  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
f64 = {"dtype": torch.float64}
rng = np.random.default_rng(777001)

BASE = ["relativistic_gamma", "relativistic_beta", "p0c", "projected_emittance_x", "emittance_x", "normalized_emittance_x", "beta_x",
        "alpha_x", "projected_emittance_y", "emittance_y", "normalized_emittance_y", "beta_y", "alpha_y", "dispersion_x",
        "dispersion_px", "dispersion_y", "dispersion_py"]
COORDS = ["x", "px", "y", "py", "tau", "p"]
MOMENTS = [f"mu_{c}" for c in COORDS] + [f"sigma_{c}" for c in COORDS] + [
    "cov_xpx", "cov_ypy", "cov_taup", "cov_xp", "cov_pxp", "cov_yp", "cov_pyp", "cov_xy", "cov_xpy", "cov_xtau", "cov_pxy",
    "cov_pxpy", "cov_pxtau", "cov_ytau", "cov_pytau"]
PARTICLE_ONLY = ["total_charge", "num_particles_survived", "energies", "momenta"]

SHAPES = [((), ()), ((), ()), ((3,), ()), ((), (3,)), ((2, 2), ()), ((), ()), ((3,), (3,)), ((), ())]
SPECIES = ["electron", "proton", "electron", "positron", "electron", "ion", "proton", "electron"]


def species(name):
    if name == "ion":
        return cheetah.Species("ion", num_elementary_charges=torch.tensor(6.0, **f64), mass_eV=torch.tensor(1.1178e10, **f64))
    return cheetah.Species(name, **f64)


if __name__ == "__main__":
    arrays = {"n_beams": np.asarray(len(SHAPES)), "base": np.asarray(BASE), "moments": np.asarray(MOMENTS),
              "particle_only": np.asarray(PARTICLE_ONLY)}
    for i, ((pshape, eshape), sp_name) in enumerate(zip(SHAPES, SPECIES)):
        n = int(rng.integers(200, 700))
        scales = np.array([rng.uniform(5e-5, 2e-3), rng.uniform(1e-5, 5e-4), rng.uniform(5e-5, 2e-3), rng.uniform(1e-5, 5e-4),
                           rng.uniform(1e-5, 2e-3), rng.uniform(1e-4, 1e-2)])
        mix = np.eye(6) + 0.4 * rng.standard_normal((6, 6))          # correlates every pair of coordinates
        raw = rng.standard_normal((*pshape, n, 6)) @ mix.T * scales + rng.uniform(-1, 1, 6) * scales
        particles = np.concatenate([raw, np.ones((*pshape, n, 1))], axis=-1)
        charges = rng.uniform(0.2, 1.0, n) * 1e-15 * (-1.0 if i % 3 == 0 else 1.0)
        if i == 5:
            charges[::7] *= -1.0                                      # mixed signs
        survival = rng.uniform(0.0, 1.0, (*pshape, n))
        survival[..., rng.integers(0, n, n // 5)] = 0.0               # dead particles
        if i == 1:
            survival[:] = 1.0
        sp = species(sp_name)
        mass = float(sp.mass_eV)
        gamma = np.exp(rng.uniform(np.log(1.3), np.log(1e4 if mass < 1e8 else 50.0), eshape))
        energy = gamma * mass
        beam = cheetah.ParticleBeam(torch.tensor(particles), torch.tensor(energy, **f64), particle_charges=torch.tensor(charges),
                                    survival_probabilities=torch.tensor(survival), species=sp, **f64)
        arrays[f"particles_{i}"], arrays[f"energy_{i}"] = particles, np.asarray(energy)
        arrays[f"charges_{i}"], arrays[f"survival_{i}"] = charges, survival
        arrays[f"species_{i}"] = np.asarray(sp_name)
        for name in BASE + MOMENTS + PARTICLE_ONLY:
            try:
                arrays[f"pb_{name}_{i}"] = np.asarray(getattr(beam, name).numpy())
            except RuntimeError as err:    # e.g. `energies` of unvectorised particles with a vectorised energy does not broadcast
                print("   reference raises for", name, "-", str(err)[:80])
        arrays[f"pb_num_particles_{i}"] = np.asarray(beam.num_particles)
        conv = beam.as_parameter_beam()
        par = cheetah.ParameterBeam(conv.mu, conv.cov, beam.energy, total_charge=beam.total_charge, species=sp, **f64)
        arrays[f"mu_{i}"], arrays[f"cov_{i}"] = conv.mu.numpy(), conv.cov.numpy()
        for name in BASE + MOMENTS:
            try:
                arrays[f"par_{name}_{i}"] = np.asarray(getattr(par, name).numpy())
            except RuntimeError as err:
                print("   reference raises for ParameterBeam", name, "-", str(err)[:80])
        print(i, sp_name, pshape, eshape, n, "emittance_x", arrays[f"pb_emittance_x_{i}"].reshape(-1)[:2])
    np.savez_compressed(os.path.join(OUT, "beam_properties_random.npz"), **arrays)
    print("wrote", len(arrays), "arrays")
