"""The reference's singularity-free special functions as differentiable element-wise ops (mirror of
cheetah/utils/autograd.py:4-74). Values and partial derivatives come from one `chx_special` launch each; reverse mode,
forward mode and vmap are all expressed through those stored partials. Inputs must live on a ROCm device."""
from __future__ import annotations

import torch

from .. import _ops


class _Special1(torch.autograd.Function):
    @staticmethod
    def forward(x, kind):
        y, dy = _ops.special(kind, x)
        return y, dy

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.mark_non_differentiable(output[1])
        ctx.save_for_backward(output[1])
        ctx.save_for_forward(output[1])

    @staticmethod
    def backward(ctx, grad, _):
        (dy,) = ctx.saved_tensors
        return grad * dy, None

    @staticmethod
    def jvp(ctx, tangent, _):
        (dy,) = ctx.saved_tensors
        return tangent * dy, None

    @staticmethod
    def vmap(info, in_dims, x, kind):   # element-wise: the vmapped dimension is just another dimension
        y, dy = _Special1.apply(x, kind)
        return (y, dy), (in_dims[0], in_dims[0])


class _Special2(torch.autograd.Function):
    @staticmethod
    def forward(a, b, kind):
        a, b = torch.broadcast_tensors(a, b)
        return _ops.special(kind, a, b)

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.mark_non_differentiable(output[1], output[2])
        ctx.save_for_backward(output[1], output[2])
        ctx.save_for_forward(output[1], output[2])
        ctx.shapes = (inputs[0].shape, inputs[1].shape)

    @staticmethod
    def backward(ctx, grad, _a, _b):
        da, db = ctx.saved_tensors
        return (grad * da).sum_to_size(ctx.shapes[0]), (grad * db).sum_to_size(ctx.shapes[1]), None

    @staticmethod
    def jvp(ctx, ta, tb, _):
        da, db = ctx.saved_tensors
        return ta * da + tb * db, None, None

    @staticmethod
    def vmap(info, in_dims, a, b, kind):
        # vmapped dimension to the front, remaining dimensions lined up from the right for broadcasting
        n = max(a.dim() - (in_dims[0] is not None), b.dim() - (in_dims[1] is not None))

        def front(t, d):
            if d is None:
                return t
            t = t.movedim(d, 0)
            return t.reshape(t.shape[0], *([1] * (n - (t.dim() - 1))), *t.shape[1:])

        return _Special2.apply(front(a, in_dims[0]), front(b, in_dims[1]), kind), (0, 0, 0)


def log1pdiv(x: torch.Tensor) -> torch.Tensor:
    """`log(1 + x) / x`, 1 at x = 0 (utils/autograd.py:4-6, 77-105)."""
    return _Special1.apply(x, "log1pdiv")[0]


def si1mdiv(x: torch.Tensor) -> torch.Tensor:
    """`(1 - si(sqrt(x))) / x`, 1/6 at x = 0 (utils/autograd.py:9-11, 108-146)."""
    return _Special1.apply(x, "si1mdiv")[0]


def sicos1mdiv(x: torch.Tensor) -> torch.Tensor:
    """`(1 - si(sqrt(x)) cos(sqrt(x))) / x`, 2/3 at x = 0 (utils/autograd.py:14-19, 149-206)."""
    return _Special1.apply(x, "sicos1mdiv")[0]


def sipsicos3mdiv(x: torch.Tensor) -> torch.Tensor:
    """`(3 - 4 si(sqrt(x)) + si(sqrt(x)) cos(sqrt(x))) / (2x)`, 0 at x = 0 (utils/autograd.py:22-27, 209-278)."""
    return _Special1.apply(x, "sipsicos3mdiv")[0]


def sicoskuddelmuddel15mdiv(x: torch.Tensor) -> torch.Tensor:
    """`(15 - 22.5 si + 9 si cos - 1.5 si cos^2 + x si^3) / x^3` of sqrt(x), 1/56 substituted at x = 0
    (utils/autograd.py:30-39, 281-358; unused by the reference's maps as well)."""
    return _Special1.apply(x, "sicoskuddelmuddel15mdiv")[0]


def cossqrtmcosdivdiff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """`(cos(sqrt(b)) - cos(sqrt(a))) / (a - b)`, si(sqrt(a)) / 2 at a = b (utils/autograd.py:42-47, 361-430)."""
    return _Special2.apply(a, b, "cossqrtmcosdivdiff")[0]


def simsidivdiff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """`(si(sqrt(a)) - si(sqrt(b))) / (b - a)` with its a = b limit (utils/autograd.py:50-55, 433-543)."""
    return _Special2.apply(a, b, "simsidivdiff")[0]


def si2msi2divdiff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """`(si^2(sqrt(b)) - si^2(sqrt(a))) / (a - b)` with its a = b limit (utils/autograd.py:58-66, 546-649)."""
    return _Special2.apply(a, b, "si2msi2divdiff")[0]


def sqrta2minusbdiva(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """`(sqrt(a^2 + b) - a) / b`, 1 / (2a) at b = 0 (utils/autograd.py:69-74, 652-700)."""
    return _Special2.apply(a, b, "sqrta2minusbdiva")[0]
