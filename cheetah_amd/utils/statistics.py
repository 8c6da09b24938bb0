"""Weighted statistics on arbitrary tensors (mirror of cheetah/utils/statistics.py). Thin tensor expressions; beams
themselves take their moments from `chx_moments`."""
from __future__ import annotations

import torch


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
