"""Kernel-density histograms with the reference's call signatures (mirror of cheetah/utils/kde.py:116-204) on top of
`chx_kde_values` (Gaussian kernel values on device) and a rocBLAS GEMM over the sample axis."""
from __future__ import annotations

import torch

from .. import _ops


def _as_particles(x1: torch.Tensor, x2: torch.Tensor | None = None) -> torch.Tensor:
    p = x1.new_zeros((*x1.shape, 7))
    p[..., 0] = x1
    if x2 is not None:
        p[..., 2] = x2
    return p


def kde_histogram_1d(x: torch.Tensor, bins: torch.Tensor, bandwidth: torch.Tensor, weights: torch.Tensor | None = None,
                     epsilon: float = 1e-10) -> torch.Tensor:
    """(…, N) samples -> (…, len(bins)) normalised density (utils/kde.py:116-152)."""
    return _ops.kde_histogram_1d(_as_particles(x), bins, bandwidth, survival=weights, epsilon=float(epsilon))


def kde_histogram_2d(x1: torch.Tensor, x2: torch.Tensor, bins1: torch.Tensor, bins2: torch.Tensor, bandwidth: torch.Tensor,
                     weights: torch.Tensor | None = None, epsilon: float = 1e-10) -> torch.Tensor:
    """(…, N) sample pairs -> (…, len(bins1), len(bins2)) normalised joint density (utils/kde.py:155-204)."""
    return _ops.kde_histogram_2d(_as_particles(x1, x2), bins1, bins2, bandwidth, survival=weights,
                                 epsilon=float(epsilon)).mT
