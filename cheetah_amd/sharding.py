"""Multi-GPU sharding of the tracking path (one process per GPU, torch.distributed over RCCL/xGMI).

The path shards on two independent axes and needs NO data-path collective for tracking itself:
  * vector (batch) axis: independent lattice settings  -> `shard_range(B, rank, world)` per rank;
  * particle axis: particles are independent under linear maps and cavities -> each rank tracks its
    own slice of the beam.
Only *global beam moments* need communication: two all-reduces of tiny fp64 buffers per query
(8 sums, then 21 centred sums about the global mean), i.e. the reference's two-pass statistics
(cheetah/utils/statistics.py:30-48) distributed over ranks. Space charge additionally all-reduces the
g^3 charge grid (one exchange per kick).

The collective logic is written against `torch.distributed` only, so it runs over RCCL on GPUs and
over gloo in the CPU tests (tests/test_sharding_gloo.py, world_size 2, partials from the oracle).
"""

from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist


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
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    m2 = centred_fn(sums).clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(m2, op=dist.ReduceOp.SUM, group=group)
    return finalize_fn(sums, m2)


def global_moments(beam, group=None) -> torch.Tensor:
    """(…,29) global moments of a particle-sharded beam (HIP partial reductions + RCCL all-reduce)."""
    from . import _ops

    p, w = beam.particles, beam.survival_probabilities
    batch_shape = torch.broadcast_shapes(p.shape[:-2], w.shape[:-1])
    B = _ops.numel(batch_shape)
    x, _ = _ops.flat_bcast(p, batch_shape, 2)
    x = _ops.aligned(x)
    wf, _ = _ops.flat_bcast(w.to(p.dtype), batch_shape, 1)
    wf = wf.contiguous()
    out = allreduce_moments(_ops.moment_sums(x, wf, B), lambda s: _ops.moment_centred(x, wf, s, B),
                            _ops.moment_finalize, group)
    return out.reshape(*batch_shape, _ops.MOM_NOUT)


def allreduce_grid(grid: torch.Tensor, group=None) -> torch.Tensor:
    """Sum a locally deposited charge grid / screen image over the particle shards (in place)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(grid, op=dist.ReduceOp.SUM, group=group)
    return grid
