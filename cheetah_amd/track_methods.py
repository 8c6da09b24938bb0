"""First- and second-order map helpers under their reference names (mirror of cheetah/track_methods.py), evaluated by the
libchx builder kernels. Custom elements written against the reference call these directly."""

from __future__ import annotations

import torch

from . import _ops
from .particles.species import Species


def _flat(values, dtype):
    shape = torch.broadcast_shapes(*[v.shape for v in values])
    B = max(_ops.numel(shape), 1)
    cols = [v.to(dtype).expand(shape).reshape(B) for v in values]
    return torch.stack(cols, dim=-1).contiguous(), shape, B


def drift_matrix(length: torch.Tensor, energy: torch.Tensor, species: Species) -> torch.Tensor:
    """track_methods.py:284-299."""
    params, shape, B = _flat([length, energy], length.dtype)
    R = _ops.build_rmatrix(_ops.KIND["drift"], params[:, :1].contiguous(), params[:, 1].contiguous(),
                           species.mass_eV_float, species.num_elementary_charges_float, B)
    return R.reshape(*shape, 7, 7)


def base_rmatrix(length, k1, hx, species: Species, energy: torch.Tensor | None = None) -> torch.Tensor:
    """Universal first-order map of a combined-function magnet (track_methods.py:17-77): the dipole builder kind
    [length, angle = hx * length, k1, e1, e2, tilt, fint, fint_exit, gap] without faces or tilt."""
    energy = energy if energy is not None else length.new_zeros(())
    zero = length.new_zeros(())
    params, shape, B = _flat([length, hx * length, k1, zero, zero, zero, zero, zero, zero, energy], length.dtype)
    R = _ops.build_rmatrix(_ops.KIND["dipole"], params[:, :9].contiguous(), params[:, 9].contiguous(),
                           species.mass_eV_float, species.num_elementary_charges_float, B)
    return R.reshape(*shape, 7, 7)


def base_ttensor(length, k1, k2, hx, species: Species, energy: torch.Tensor | None = None) -> torch.Tensor:
    """Second-order tensor without the first-order block (track_methods.py:80-281) for any combination of quadrupole,
    sextupole strength and curvature: `chx_build_ttensor` kind CHX_T_GENERAL [length, k1, k2, hx] (differentiable in all
    four and in the energy through chx_build_ttensor_vjp)."""
    energy = energy if energy is not None else length.new_zeros(())
    params, pshape = _ops.stack_params([length, k1, k2, hx], length.dtype, length.device)
    return _ops.build_ttensor(_ops.T_KIND["general"], params, pshape, energy.to(length.dtype), species.mass_eV_float)


def rotation_matrix(angle: torch.Tensor) -> torch.Tensor:
    """x-y rotation of the coordinate system (track_methods.py:302-323)."""
    cs, sn = angle.cos(), angle.sin()
    tm = torch.eye(7, dtype=angle.dtype, device=angle.device).repeat(*angle.shape, 1, 1)
    for i, j, v in ((0, 0, cs), (1, 1, cs), (2, 2, cs), (3, 3, cs), (0, 2, sn), (1, 3, sn), (2, 0, -sn), (3, 1, -sn)):
        tm[..., i, j] = v
    return tm


def misalignment_matrix(misalignment: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """(entry, exit) shift maps of a transversely misaligned element (track_methods.py:326-342)."""
    eye = torch.eye(7, dtype=misalignment.dtype, device=misalignment.device).repeat(*misalignment.shape[:-1], 1, 1)
    R_entry, R_exit = eye.clone(), eye
    R_exit[..., 0, 6], R_exit[..., 2, 6] = misalignment[..., 0], misalignment[..., 1]
    R_entry[..., 0, 6], R_entry[..., 2, 6] = -misalignment[..., 0], -misalignment[..., 1]
    return R_entry, R_exit


def combined_rotation_misalignment_matrix(angle: torch.Tensor, misalignment: torch.Tensor):
    """Misalign, then rotate: (entry, exit) maps (track_methods.py:345-382)."""
    shape = torch.broadcast_shapes(angle.shape, misalignment.shape[:-1])
    tm_entry = rotation_matrix(angle.expand(shape))
    tm_exit = tm_entry.clone().mT
    cs, sn = angle.cos(), angle.sin()
    tm_entry[..., 0, 6] = -misalignment[..., 0] * cs - misalignment[..., 1] * sn
    tm_entry[..., 2, 6] = misalignment[..., 0] * sn - misalignment[..., 1] * cs
    tm_exit[..., 0, 6] = misalignment[..., 0]
    tm_exit[..., 2, 6] = misalignment[..., 1]
    return tm_entry, tm_exit
