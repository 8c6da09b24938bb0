"""Mirror of cheetah/utils/cloud_in_cell.py on top of `chx_cic_deposit`."""
from __future__ import annotations

import torch

from .. import _ops


def cloud_in_cell_charge_deposition(positions, bins, extent, charges) -> torch.Tensor:
    """Cloud-in-cell deposit of (…, N, d) positions with (…, N) charges onto a grid of `bins` cells over `extent`
    (…, d, 2) (utils/cloud_in_cell.py:8-41) through `chx_cic_deposit`."""
    d = positions.shape[-1]
    padded = positions.new_zeros((*positions.shape[:-1], 7))
    padded[..., :d] = positions
    return _ops.cic_deposit(padded, tuple(range(d)), tuple(int(b) for b in bins), extent, charge=charges)
