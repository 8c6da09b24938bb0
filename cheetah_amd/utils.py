"""Helper functions user code imports from `cheetah.utils` (mirror of cheetah/utils/__init__.py). The statistics helpers
are thin tensor expressions for arbitrary inputs; beams themselves take their moments from `chx_moments`."""

from __future__ import annotations

import torch

from . import _ops
from .accelerator.element import merge_element_names  # noqa: F401
from .warnings import (  # noqa: F401
    DefaultParameterWarning,
    DirtyNameWarning,
    NoBeamPropertiesInLatticeWarning,
    NotUnderstoodPropertyWarning,
    PhysicsWarning,
    UnknownElementWarning,
    VisualizationWarning,
)


def compute_relativistic_factors(energy: torch.Tensor, particle_mass_eV: torch.Tensor):
    """(gamma, 1 / gamma^2, beta) for a total energy in eV (utils/physics.py:4-19)."""
    gamma = energy / particle_mass_eV
    igamma2 = gamma.square().reciprocal()
    return gamma, igamma2, (1.0 - igamma2).sqrt()


def _weighted_moments(weights: torch.Tensor, dim):
    total = weights.sum(dim=dim)
    return total, total - weights.square().sum(dim=dim) / total   # sum of weights, reliability-weights correction


def unbiased_weighted_covariance(inputs1, inputs2, weights, dim: int = None) -> torch.Tensor:
    """Weighted covariance with reliability weights (utils/statistics.py:4-27)."""
    total, correction = _weighted_moments(weights, dim)
    mean1 = (inputs1 * weights).sum(dim=dim) / total
    mean2 = (inputs2 * weights).sum(dim=dim) / total
    keep = (lambda m: m.unsqueeze(dim)) if dim is not None else (lambda m: m)
    return (weights * (inputs1 - keep(mean1)) * (inputs2 - keep(mean2))).sum(dim=dim) / correction


def unbiased_weighted_variance(inputs, weights, dim: int = None) -> torch.Tensor:
    """utils/statistics.py:30-48."""
    return unbiased_weighted_covariance(inputs, inputs, weights, dim=dim)


def unbiased_weighted_std(inputs, weights, dim: int = None) -> torch.Tensor:
    """utils/statistics.py:51-62."""
    return unbiased_weighted_variance(inputs, weights, dim=dim).sqrt()


def unbiased_weighted_covariance_matrix(inputs: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """(…, N, D) samples with (…, N) weights -> (…, D, D) (utils/statistics.py:65-88)."""
    total = weights.sum(dim=-1)
    correction = total - weights.square().sum(dim=-1) / total
    centred = inputs - ((inputs * weights.unsqueeze(-1)).sum(dim=-2) / total.unsqueeze(-1)).unsqueeze(-2)
    return (centred * weights.unsqueeze(-1)).mT @ centred / correction.unsqueeze(-1).unsqueeze(-1)


def match_distribution_moments(samples, target_mu, target_cov, weights=None) -> torch.Tensor:
    """Affine-transform (…, N, D) samples so that their mean / covariance become the targets (utils/statistics.py:91-130):
    whiten with the Cholesky factor of the sample covariance, colour with the target's."""
    if weights is None:
        weights = torch.ones_like(samples[..., 0])
    total = weights.sum(dim=-1, keepdim=True)
    mean = (samples * weights.unsqueeze(-1)).sum(dim=-2) / total
    chol_have = torch.linalg.cholesky(unbiased_weighted_covariance_matrix(samples, weights))
    chol_want = torch.linalg.cholesky(target_cov)
    white = torch.linalg.solve_triangular(chol_have, (samples - mean.unsqueeze(-2)).mT, upper=False)
    return (chol_want @ white).mT + target_mu.unsqueeze(-2)


def elementwise_linspace(start: torch.Tensor, end: torch.Tensor, steps: int) -> torch.Tensor:
    """`steps` evenly spaced values between every pair of entries, shape start.shape + (steps,)
    (utils/elementwise_linspace.py; the reference builds each ramp with torch.linspace in the default dtype)."""
    ramp = torch.linspace(0.0, 1.0, steps, device=start.device)
    out = start.unsqueeze(-1) + (end - start).unsqueeze(-1) * ramp.to(start.dtype if start.is_floating_point() else ramp.dtype)
    if steps > 0:
        out[..., -1] = end        # torch.linspace ends exactly on `end`
    return out


def cloud_in_cell_charge_deposition(positions, bins, extent, charges) -> torch.Tensor:
    """Cloud-in-cell deposit of (…, N, d) positions with (…, N) charges onto a grid of `bins` cells over `extent`
    (…, d, 2) (utils/cloud_in_cell.py:8-41) through `chx_cic_deposit`."""
    d = positions.shape[-1]
    padded = positions.new_zeros((*positions.shape[:-1], 7))
    padded[..., :d] = positions
    return _ops.cic_deposit(padded, tuple(range(d)), tuple(int(b) for b in bins), extent, charge=charges)


def squash_index_for_unavailable_dims(index: tuple, shape: tuple) -> tuple:
    """Index a tensor of `shape` with the trailing part of a longer vector index; size-1 dims take index 0
    (utils/vector.py)."""
    if len(shape) == 0:
        return ()
    tail = index[-len(shape):]
    return tuple(0 if size == 1 else i for i, size in zip(tail, shape))


class UniqueNameGenerator:
    """Callable producing `prefix_0`, `prefix_1`, ... (utils/names.py:4-14)."""

    def __init__(self, prefix: str):
        self._prefix, self._counter = prefix, 0

    def __call__(self) -> str:
        name = f"{self._prefix}_{self._counter}"
        self._counter += 1
        return name
