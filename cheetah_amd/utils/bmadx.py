"""Coordinate conversions between this package's (x, px, y, py, tau, delta, 1) vectors and Bmad's (x, px, y, py, z, pz)
(mirror of cheetah/utils/bmadx.py:7-111). They are format glue around test data and converters; the Bmad-X tracking
routines of that reference module (offset_particle_set, track_a_drift, low_energy_z_correction, ...) have no Python
counterpart here: they run inside `chx_dkd_track` (csrc/chx_nonlinear.hip)."""
from __future__ import annotations

import torch


def _as_tensor(mc2, like: torch.Tensor) -> torch.Tensor:
    return torch.as_tensor(mc2, dtype=like.dtype, device=like.device)


def cheetah_to_bmad_z_pz(tau: torch.Tensor, delta: torch.Tensor, ref_energy: torch.Tensor, mc2):
    """(z, pz, p0c) from (tau, delta) at reference energy `ref_energy` [eV] (utils/bmadx.py:7-29)."""
    mc2 = _as_tensor(mc2, ref_energy)
    p0c = (ref_energy.square() - mc2.square()).sqrt().unsqueeze(-1)
    energy = ref_energy.unsqueeze(-1) + delta * p0c
    momentum = (energy.square() - mc2.square()).sqrt()
    return -(momentum / energy) * tau, (momentum - p0c) / p0c, p0c.squeeze(-1)


def bmad_to_cheetah_z_pz(z: torch.Tensor, pz: torch.Tensor, p0c: torch.Tensor, mc2):
    """(tau, delta, ref_energy) from Bmad (z, pz) at reference momentum `p0c` [eV/c] (utils/bmadx.py:32-54)."""
    mc2 = _as_tensor(mc2, p0c)
    ref_energy = (p0c.square() + mc2.square()).sqrt()
    momentum = (1 + pz) * p0c.unsqueeze(-1)
    energy = (momentum.square() + mc2.square()).sqrt()
    return -z * energy / momentum, (energy - ref_energy.unsqueeze(-1)) / p0c.unsqueeze(-1), ref_energy


def cheetah_to_bmad_coords(cheetah_coords: torch.Tensor, ref_energy: torch.Tensor, mc2):
    """(…, N, 7) particle vectors -> ((…, N, 6) Bmad coordinates, p0c) (utils/bmadx.py:57-82)."""
    z, pz, p0c = cheetah_to_bmad_z_pz(cheetah_coords[..., 4], cheetah_coords[..., 5], ref_energy, mc2)
    return torch.cat([cheetah_coords[..., :4], z.unsqueeze(-1), pz.unsqueeze(-1)], dim=-1), p0c


def bmad_to_cheetah_coords(bmad_coords: torch.Tensor, p0c: torch.Tensor, mc2):
    """(…, N, 6) Bmad coordinates -> ((…, N, 7) particle vectors, ref_energy) (utils/bmadx.py:85-111)."""
    tau, delta, ref_energy = bmad_to_cheetah_z_pz(bmad_coords[..., 4], bmad_coords[..., 5], p0c, mc2)
    ones = torch.ones_like(tau).unsqueeze(-1)
    return torch.cat([bmad_coords[..., :4], tau.unsqueeze(-1), delta.unsqueeze(-1), ones], dim=-1), ref_energy
