"""Multi-GPU sharding of the tracking path (one process per GPU, torch.distributed over RCCL/xGMI).

The path shards on two independent axes and needs NO data-path collective for tracking itself:
  * vector (batch) axis: independent lattice settings  -> `shard_range(B, rank, world)` per rank;
  * particle axis: particles are independent under linear maps and cavities -> each rank tracks its
    own slice of the beam.
Only *global beam moments* need communication: every rank reduces its own particles with the one-pass
`chx_moments` kernel, ONE all-gather moves 29 doubles per rank and batch row, and the per-rank moments are
merged exactly (Chan et al.: M = sum_r [M_r + W_r (mu_r - mu)(mu_r - mu)^T]) — the reference's weighted
statistics (cheetah/utils/statistics.py:30-48) distributed over ranks. (`allreduce_moments` is the older
two-all-reduce formulation of the same result.) Space charge additionally all-reduces the g^3 charge grid
(one exchange per kick).

The collective logic is written against `torch.distributed` only, so it runs over RCCL on GPUs and
over gloo in the CPU tests (tests/test_sharding_gloo.py, world_size 2, partials from the oracle).
"""

from __future__ import annotations

import contextlib
import os
from typing import Callable

import torch
import torch.distributed as dist

_ACTIVE_GROUP: list = []   # stack of (process group, force_collectives) inside `particle_sharded(...)`
_FORCE: list = [os.environ.get("CHX_FORCE_COLLECTIVES", "0") == "1"]   # process-wide default of `force_collectives`


def force_collectives(on: bool = True) -> bool:
    """Process-wide switch (also the environment variable CHX_FORCE_COLLECTIVES=1): take the collective branch of every
    exchange below even in a group of ONE rank. A one-rank all-gather / all-reduce returns its input, so results are
    bit-identical to the un-sharded ones — the point is that the RCCL calls, their stream ordering against libchx's own
    streams and `chx_merge_moments` execute on a single-GPU box (tests, `bench.py --force-collectives`). Returns the
    previous setting."""
    old, _FORCE[0] = _FORCE[0], bool(on)
    return old


def collectives_on(group=None) -> bool:
    """Does an exchange over `group` go through torch.distributed? More than one rank, or forced (see above)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    if dist.get_world_size(group) > 1:
        return True
    return _ACTIVE_GROUP[-1][1] if _ACTIVE_GROUP else _FORCE[0]


@contextlib.contextmanager
def particle_sharded(group=None, force_collectives=None):
    """Inside this context every rank of `group` holds a SLICE of the particles of one beam. Elements whose physics couples
    the particles exchange what they need: a `SpaceChargeKick` takes its grid from the global beam moments (one
    all-gather of 29 doubles per rank) and sums the deposited charge over the ranks (one all-reduce of the g^3 grid), a
    `Screen` sums its image, an active `BPM` reads the mean of all shards (the same 29-double all-gather). Linear maps, cavities
    and apertures need nothing. `force_collectives` (default: the
    process-wide switch): run the exchanges even when the group has a single rank."""
    force = _FORCE[0] if force_collectives is None else bool(force_collectives)
    _ACTIVE_GROUP.append((group if group is not None else (dist.group.WORLD if dist.is_initialized() else None), force))
    try:
        yield
    finally:
        _ACTIVE_GROUP.pop()


def active_group():
    """The process group of the innermost `particle_sharded` context, or None outside of one / when there is nothing to
    exchange (a single rank and collectives not forced)."""
    if not _ACTIVE_GROUP or not (dist.is_available() and dist.is_initialized()):
        return None
    group, force = _ACTIVE_GROUP[-1]
    return group if (dist.get_world_size(group) > 1 or force) else None


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [start, stop) of `n_total` items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_total), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def allreduce_moments(local_sums: torch.Tensor, centred_fn: Callable[[torch.Tensor], torch.Tensor],
                      finalize_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], group=None) -> torch.Tensor:
    """Global weighted moments from per-rank partials.

    local_sums  (B,8) fp64 on this rank: [sum w, sum w^2, sum w x_0..5] over the LOCAL particles
    centred_fn  global_sums -> (B,21) fp64 local centred sums about the global mean
    finalize_fn (global_sums, global_m2) -> (B,29) [W, W2, mu(6), unbiased cov upper triangle (21)]
    """
    sums = local_sums.clone()
    on = collectives_on(group)
    if on:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    m2 = centred_fn(sums).clone()
    if on:
        dist.all_reduce(m2, op=dist.ReduceOp.SUM, group=group)
    return finalize_fn(sums, m2)


_TRI = [(i, j) for i in range(6) for j in range(i, 6)]


def merge_moments(per_rank: torch.Tensor) -> torch.Tensor:
    """(R, B, 29) per-rank [W, W2, mu(6), unbiased cov upper triangle(21)] -> (B, 29) of the union of the shards.
    Ranks without weight (W = 0) contribute nothing."""
    W_r, W2_r = per_rank[..., 0], per_rank[..., 1]
    has = W_r > 0
    safe_W = torch.where(has, W_r, torch.ones_like(W_r))
    mu_r = torch.where(has.unsqueeze(-1), per_rank[..., 2:8], torch.zeros_like(per_rank[..., 2:8]))
    cf_r = W_r - W2_r / safe_W
    M_r = torch.where(has.unsqueeze(-1), per_rank[..., 8:29] * cf_r.unsqueeze(-1), torch.zeros_like(per_rank[..., 8:29]))
    W, W2 = W_r.sum(0), W2_r.sum(0)
    mu = (W_r.unsqueeze(-1) * mu_r).sum(0) / W.unsqueeze(-1)
    d = mu_r - mu.unsqueeze(0)
    dd = torch.stack([d[..., i] * d[..., j] for i, j in _TRI], dim=-1)
    M = (M_r + W_r.unsqueeze(-1) * dd).sum(0)
    cov = M / (W - W2 / W).unsqueeze(-1)
    return torch.cat([W.unsqueeze(-1), W2.unsqueeze(-1), mu, cov], dim=-1)


def gather_merge_moments(local: torch.Tensor, group=None) -> torch.Tensor:
    """Global moments from this rank's local (B,29) moments: one all-gather + exact merge."""
    if not collectives_on(group):
        return local
    staged = local.contiguous()
    if staged.is_cuda and dist.get_backend(group) == "gloo":
        staged = staged.cpu()   # gloo has no device all-gather (29 doubles per row: the detour costs nothing)
    world = dist.get_world_size(group)
    if staged.is_cuda:   # RCCL: one all-gather into a (R, B, 29) buffer, merged by one kernel
        from . import _lib, _ops

        gathered = torch.empty((world, *staged.shape), dtype=staged.dtype, device=staged.device)
        dist.all_gather_into_tensor(gathered, staged, group=group)
        out = torch.empty_like(staged)
        _ops.check(_lib.lib().chx_merge_moments(gathered.data_ptr(), world, staged.shape[0], out.data_ptr(), _ops.stream_ptr()),
                   "chx_merge_moments")
        return out
    parts = [torch.empty_like(staged) for _ in range(world)]
    dist.all_gather(parts, staged, group=group)
    merged = merge_moments(torch.stack(parts, dim=0))
    if local.is_cuda:    # came through the gloo detour: merge on the device with the kernel as well
        return merged.to(local.device)
    return merged


class _SumOverRanks(torch.autograd.Function):
    """y = the sum of x over the ranks of a group, delivered to every rank, as a differentiable node. Every rank evaluates the
    SAME loss of y, so the cotangent that reaches rank r's x is the loss's own: backward is the identity and needs no exchange.
    (torch.distributed.nn.all_reduce sums the cotangents instead: that is the rule for ranks with DIFFERENT losses.)"""

    @staticmethod
    def forward(ctx, x, group):
        y = x.detach().clone().contiguous()
        if collectives_on(group):
            staged = y.cpu() if (y.is_cuda and dist.get_backend(group) == "gloo") else y
            dist.all_reduce(staged, op=dist.ReduceOp.SUM, group=group)
            if staged is not y:
                y.copy_(staged)
        return y

    @staticmethod
    def backward(ctx, g):
        return g, None


def sum_over_ranks(x: torch.Tensor, group=None) -> torch.Tensor:
    """Sum of `x` over the particle shards on every rank (total charge, surviving particles, a screen image); keeps the
    autograd graph of this rank's contribution."""
    if torch.is_grad_enabled() and x.requires_grad:
        return _SumOverRanks.apply(x, group)
    return allreduce_grid(x.detach().clone().contiguous(), group)


def all_reduce_gradients(tensors, group=None) -> None:
    """Sum the `.grad` of replicated settings over the ranks (in place). Inside `particle_sharded` every rank evaluates the same
    loss of the GLOBAL beam statistics, and its backward pass leaves on every replicated setting (a quadrupole strength, a
    cavity phase) the share of the gradient that comes through ITS particles; the gradient of the loss is the sum of the shares
    — one all-reduce per setting, after `backward()`. A setting a rank has no gradient for counts as zero."""
    if group is None and _ACTIVE_GROUP:
        group = _ACTIVE_GROUP[-1][0]
    if not collectives_on(group):
        return
    for t in tensors:
        if t.grad is None:
            t.grad = torch.zeros_like(t)
        g = t.grad
        staged = g.cpu() if (g.is_cuda and dist.get_backend(group) == "gloo") else g
        dist.all_reduce(staged, op=dist.ReduceOp.SUM, group=group)
        if staged is not g:
            g.copy_(staged)


def gather_moments_rows(local: torch.Tensor, group=None):
    """This rank's (1,29) moments -> (rows, R): over RCCL the (R,29) moments of all shards exactly as all-gathered (R = world
    size; the consumer merges them itself: `chx_sc_kick_sorted_begin`'s geometry kernel), otherwise the merged (1,29) row and
    R = 0. One all-gather either way."""
    if collectives_on(group) and local.is_cuda and dist.get_backend(group) != "gloo":
        world = dist.get_world_size(group)
        staged = local.contiguous()
        gathered = torch.empty((world, staged.shape[-1]), dtype=staged.dtype, device=staged.device)
        dist.all_gather_into_tensor(gathered, staged.reshape(1, -1), group=group)
        return gathered, world
    return gather_merge_moments(local, group).contiguous(), 0


def global_moments(beam, group=None) -> torch.Tensor:
    """(…,29) global moments of a particle-sharded beam: local one-pass HIP reduction (chx_moments) + one
    all-gather over RCCL + exact merge."""
    from . import _ops

    local = _ops.moments(beam.particles, beam.survival_probabilities)
    return gather_merge_moments(local.reshape(-1, _ops.MOM_NOUT), group).reshape(local.shape)


def allreduce_grid(grid: torch.Tensor, group=None) -> torch.Tensor:
    """Sum a locally deposited charge grid / screen image over the particle shards (in place)."""
    if collectives_on(group):
        dist.all_reduce(grid, op=dist.ReduceOp.SUM, group=group)
    return grid
