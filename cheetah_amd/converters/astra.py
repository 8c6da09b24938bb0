"""Astra particle distributions -> Cheetah coordinates (behavioural mirror of cheetah/converters/astra.py:8-62, which
follows Ocelot's astra2ocelot adaptor).

An Astra file has one row per macro-particle: x, y, z [m], px, py, pz [eV/c], clock [ns], charge [nC], species index,
status flag. Row 0 is the reference particle; all other rows give z, pz (and the clock) RELATIVE to it. Rows with a
status flag <= 0 are lost particles."""

from __future__ import annotations

import numpy as np

ELECTRON_MASS_EV = 0.51099895069e6  # scipy.constants "electron mass energy equivalent in MeV" * 1e6 (CODATA 2022)


def from_astrabeam(path: str) -> tuple[np.ndarray, float, np.ndarray]:
    """(particles (N, 6) in Cheetah order x, px, y, py, tau, delta; reference energy [eV]; charges (N,) [C])."""
    table = np.loadtxt(path)
    table = table[table[:, 9] > 0]                 # keep live particles only
    p_ref = table[0, 5]                            # reference longitudinal momentum [eV/c]
    pos = table[:, 0:3].copy()
    mom = table[:, 3:6].copy()
    pos[0, 2] = 0.0                                # the reference row carries absolute z / pz: make it relative
    mom[0, 2] = 0.0
    mom[:, 2] += p_ref                             # absolute momenta

    gamma_ref = np.sqrt((p_ref / ELECTRON_MASS_EV) ** 2 + 1.0)
    beta_ref = np.sqrt(1.0 - gamma_ref**-2)
    energy = gamma_ref * ELECTRON_MASS_EV

    gamma = np.sqrt(1.0 + (mom * mom).sum(axis=1) / ELECTRON_MASS_EV**2)
    beta = np.sqrt(1.0 - gamma**-2)
    direction = mom / np.linalg.norm(mom, axis=1, keepdims=True)
    # drift every particle to the plane z = 0 of the reference particle: c dt = -z / (beta n_z)
    cdt = -pos[:, 2] / (beta * direction[:, 2])

    particles = np.zeros((table.shape[0], 6))
    particles[:, 0] = pos[:, 0] + beta * direction[:, 0] * cdt
    particles[:, 1] = mom[:, 0] / p_ref
    particles[:, 2] = pos[:, 1] + beta * direction[:, 1] * cdt
    particles[:, 3] = mom[:, 1] / p_ref
    particles[:, 4] = cdt
    particles[:, 5] = (gamma / gamma_ref - 1.0) / beta_ref
    charges = np.abs(table[:, 7]) * 1e-9           # nC -> C
    return particles, energy, charges
