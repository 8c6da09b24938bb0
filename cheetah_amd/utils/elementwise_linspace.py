"""Mirror of cheetah/utils/elementwise_linspace.py."""
from __future__ import annotations

import torch


def elementwise_linspace(start: torch.Tensor, end: torch.Tensor, steps: int) -> torch.Tensor:
    """`steps` evenly spaced values between every pair of entries, shape start.shape + (steps,)
    (utils/elementwise_linspace.py; the reference builds each ramp with torch.linspace in the default dtype)."""
    ramp = torch.linspace(0.0, 1.0, steps, device=start.device)
    out = start.unsqueeze(-1) + (end - start).unsqueeze(-1) * ramp.to(start.dtype if start.is_floating_point() else ramp.dtype)
    if steps > 0:
        out[..., -1] = end        # torch.linspace ends exactly on `end`
    return out
