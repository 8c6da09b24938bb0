"""Mirror of cheetah/utils/physics.py."""
from __future__ import annotations

import torch


def compute_relativistic_factors(energy: torch.Tensor, particle_mass_eV: torch.Tensor):
    """(gamma, 1 / gamma^2, beta) for a total energy in eV (utils/physics.py:4-19)."""
    gamma = energy / particle_mass_eV
    igamma2 = gamma.square().reciprocal()
    return gamma, igamma2, (1.0 - igamma2).sqrt()
