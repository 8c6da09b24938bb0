"""world_size-2 gloo test of the multi-GPU collective logic (cheetah_amd/sharding.py) on CPU.

The per-rank partial reductions come from the CPU oracle here (on the GPU box they come from the
chx_moment_* kernels); what is under test is the sharding arithmetic and the two all-reduces."""
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _partials(x, w):
    """numpy restatement of chx_moment_sums / chx_moment_centred for one shard."""
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    sums = np.concatenate([[w64.sum(), (w64 * w64).sum()], (w64[:, None] * x64[:, :6]).sum(0)])[None]

    def centred(gs):
        mu = gs[0, 2:8] / gs[0, 0]
        d = x64[:, :6] - mu
        out = []
        for i in range(6):
            for j in range(i, 6):
                out.append((w64 * d[:, i] * d[:, j]).sum())
        return np.asarray(out)[None]

    return sums, centred


def _finalize(sums, m2):
    W, W2 = sums[:, 0], sums[:, 1]
    out = torch.zeros(sums.shape[0], 29, dtype=torch.float64)
    out[:, 0], out[:, 1] = W, W2
    out[:, 2:8] = sums[:, 2:8] / W[:, None]
    out[:, 8:] = m2 / (W - W2 / W)[:, None]
    return out


def _worker(rank, world, port, x, w, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cheetah_amd.sharding import allreduce_grid, allreduce_moments, shard_range

    lo, hi = shard_range(x.shape[0], rank, world)
    sums, centred = _partials(x[lo:hi], w[lo:hi])
    out = allreduce_moments(torch.from_numpy(sums), lambda gs: torch.from_numpy(centred(gs.numpy())), _finalize)
    grid = torch.full((4, 4), float(rank + 1), dtype=torch.float64)
    allreduce_grid(grid)
    if rank == 0:
        q.put((out.numpy(), grid.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from cheetah_amd.sharding import shard_range

    for n, world in ((10, 3), (4096, 8), (7, 8), (1_000_003, 8)):
        cuts = [shard_range(n, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1


def test_global_moments_two_ranks_gloo(oracle):
    rng = np.random.default_rng(5)
    N = 20001  # odd on purpose: unequal shards
    x = (rng.standard_normal((N, 7)) * [1e-3, 1e-5, 2e-3, 1e-5, 1e-4, 1e-3, 0] + [5e-3, 0, -1e-3, 0, 0, 0, 1]).astype(np.float32)
    w = rng.random(N).astype(np.float32)
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, x, w, q)) for r in range(2)]
    for p in procs:
        p.start()
    out, grid = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = oracle.moments(x[None], w[None])["raw"]
    assert np.allclose(out[:, :8], ref[:, :8], rtol=1e-12, atol=1e-18)
    assert np.allclose(out[:, 8:], ref[:, 8:], rtol=1e-9, atol=1e-24)
    assert np.array_equal(grid, np.full((4, 4), 3.0))


def _worker_merge(rank, world, port, x, w, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cheetah_amd.sharding import gather_merge_moments, shard_range
    from oracle import chx_oracle

    lo, hi = shard_range(x.shape[1], rank, world)
    if rank == world - 1:
        w = w.copy()
        w[1, lo:hi] = 0.0  # batch row 1: this rank's shard carries no weight at all
    local = chx_oracle.moments(x[:, lo:hi], w[:, lo:hi])["raw"]  # on the GPU box: chx_moments of the local shard
    out = gather_merge_moments(torch.from_numpy(np.nan_to_num(local, nan=0.0)))
    if rank == 0:
        q.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gather_merge_moments_three_ranks_gloo(oracle):
    """The one-collective formulation used by sharding.global_moments: per-rank moments, all-gather, exact merge."""
    rng = np.random.default_rng(9)
    B, N, world = 2, 9001, 3
    x = (rng.standard_normal((B, N, 7)) * [1e-3, 1e-5, 2e-3, 1e-5, 1e-4, 1e-3, 0] + [5e-3, 0, -1e-3, 0, 0, 2e-2, 1]).astype(np.float64)
    x[:, 6000:, 0] += 3e-3  # shards with different means: the merge term matters
    w = rng.random((B, N))
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_merge, args=(r, world, port, x, w, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    w_ref = w.copy()
    w_ref[1, 6001:] = 0.0  # shard_range(9001, 2, 3) = [6001, 9001)
    ref = oracle.moments(x, w_ref)["raw"]
    assert np.allclose(out[:, :2], ref[:, :2], rtol=1e-13)
    sig = np.sqrt(ref[:, [8, 14, 19, 23, 26, 28]])
    assert np.all(np.abs(out[:, 2:8] - ref[:, 2:8]) <= 1e-12 * (sig + np.abs(ref[:, 2:8])))
    k = 8
    for i in range(6):
        for j in range(i, 6):
            assert np.all(np.abs(out[:, k] - ref[:, k]) <= 1e-11 * sig[:, i] * sig[:, j]), (i, j)
            k += 1


def test_force_collectives_on_one_rank_gloo():
    """`force_collectives` / CHX_FORCE_COLLECTIVES: a group of ONE rank still goes through all_gather / all_reduce (what the
    `-m gpu` test tests/test_gpu_rccl_single_rank.py runs on RCCL); without the switch the exchanges are skipped."""
    from cheetah_amd import sharding

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    calls = {"gather": 0, "reduce": 0}
    real_gather, real_reduce = dist.all_gather, dist.all_reduce
    dist.all_gather = lambda *a, **k: (calls.__setitem__("gather", calls["gather"] + 1), real_gather(*a, **k))[1]
    dist.all_reduce = lambda *a, **k: (calls.__setitem__("reduce", calls["reduce"] + 1), real_reduce(*a, **k))[1]
    try:
        rng = np.random.default_rng(3)
        local = torch.from_numpy(np.abs(rng.standard_normal((2, 29))) + 1.0)
        local[:, 1] = local[:, 0] ** 2 / 50.0                       # W2 < W^2: a valid weight pair
        grid = torch.full((3, 3), 2.0, dtype=torch.float64)
        assert not sharding.collectives_on() and sharding.active_group() is None
        with sharding.particle_sharded():
            assert sharding.active_group() is None                   # one rank, not forced: nothing to exchange
            assert sharding.gather_merge_moments(local) is local
        assert calls == {"gather": 0, "reduce": 0}
        with sharding.particle_sharded(force_collectives=True):
            assert sharding.active_group() is not None and sharding.collectives_on()
            merged = sharding.gather_merge_moments(local)
            sharding.allreduce_grid(grid)
        assert calls == {"gather": 1, "reduce": 1}
        assert torch.allclose(merged, local, rtol=1e-14, atol=0) and torch.equal(grid, torch.full((3, 3), 2.0, dtype=torch.float64))
        old = sharding.force_collectives(True)                       # the process-wide switch (bench.py --force-collectives)
        try:
            assert sharding.collectives_on()
            with sharding.particle_sharded():
                assert sharding.active_group() is not None
            with sharding.particle_sharded(force_collectives=False):
                assert sharding.active_group() is None
        finally:
            sharding.force_collectives(old)
    finally:
        dist.all_gather, dist.all_reduce = real_gather, real_reduce
        dist.destroy_process_group()


def _worker_chain_rows(rank, world, port, x, w, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cheetah_amd.sharding import gather_moments_rows, shard_range
    from oracle import chx_oracle

    lo, hi = shard_range(x.shape[0], rank, world)
    xs, ws = x[lo:hi].astype(np.float64), w[lo:hi].astype(np.float64)
    if rank == 0:
        # a rank on the staged kick-by-kick path: the full moments of its shard (chx_moments)
        local = chx_oracle.moments(xs[None], ws[None])["raw"]
    else:
        # a rank on the tile-ordered chain: what chx_sc_partials_moments makes of the sums the gather pass left behind — W, W2,
        # the means and variances of x, y, tau from sums about the ORIGIN; zeros elsewhere
        W, W2 = ws.sum(), (ws * ws).sum()
        local = np.zeros((1, 29))
        local[0, 0], local[0, 1] = W, W2
        for col, diag in ((0, 0), (2, 11), (4, 18)):
            m = (ws * xs[:, col]).sum() / W
            local[0, 2 + col] = m
            local[0, 8 + diag] = ((ws * xs[:, col] ** 2).sum() - W * m * m) / (W - W2 / W)
    rows, n = gather_moments_rows(torch.from_numpy(local))
    q.put((rank, rows.numpy(), n))
    dist.barrier()
    dist.destroy_process_group()


def test_chain_and_staged_ranks_exchange_compatible_moment_rows(oracle):
    """Inside `sharding.particle_sharded` a SpaceChargeKick needs sigma_x, sigma_y, sigma_tau of ALL particles
    (space_charge_kick.py:531-538). A rank on the tile-ordered chain contributes a row with only those three entries filled
    (from the sums its gather pass accumulated), a rank on the staged path its full chx_moments row: both go through the same
    29-double all-gather, and the merged x / y / tau entries must be the whole beam's — on every rank."""
    rng = np.random.default_rng(21)
    N = 30_001
    x = rng.standard_normal((N, 7)) * [3e-4, 1e-5, 2e-4, 1e-5, 5e-5, 1e-3, 0] + [2e-4, 0, -1e-4, 0, 3e-5, 0, 1]
    x[:12_000, 0] += 4e-4                      # the two shards have different means
    w = rng.random(N)
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_chain_rows, args=(r, 2, port, x, w, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        rank, rows, n = q.get()
        got[rank] = (rows, n)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = oracle.moments(x[None], w[None])["raw"][0]
    for rank in (0, 1):
        rows, n = got[rank]
        assert n == 0 and rows.shape == (1, 29)           # gloo / CPU: merged on the host (RCCL hands the rows to the kernel)
        for col, diag in ((0, 0), (2, 11), (4, 18)):
            assert abs(rows[0, 2 + col] - ref[2 + col]) <= 1e-12 * np.sqrt(ref[8 + diag])
            assert rows[0, 8 + diag] == __import__("pytest").approx(ref[8 + diag], rel=1e-11)
    assert np.array_equal(got[0][0], got[1][0])            # bit-identical on both ranks: the same grid geometry everywhere


def _worker_beam_properties(rank, world, port, x, w, qc, q):
    """A particle-sharded ParticleBeam's properties through the PRODUCT's classes. The local reduction is libchx's on the GPU box;
    here (no GPU) `_ops.moments` is replaced by the oracle's, everything behind it — the cache, the exchange, the merge, the
    property algebra — is the product's."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cheetah_amd as ca
    from cheetah_amd import _ops, sharding
    from oracle import chx_oracle

    calls = {"n": 0}

    def oracle_moments(particles, survival):
        calls["n"] += 1
        raw = chx_oracle.moments(particles.numpy()[None].astype(np.float64), survival.numpy()[None].astype(np.float64))["raw"]
        return torch.from_numpy(raw).reshape(29)

    _ops.moments = oracle_moments
    lo, hi = sharding.shard_range(x.shape[0], rank, world)
    beam = ca.ParticleBeam(torch.from_numpy(x[lo:hi]), torch.tensor(1e8, dtype=torch.float64), particle_charges=torch.from_numpy(qc[lo:hi]),
                           survival_probabilities=torch.from_numpy(w[lo:hi]), dtype=torch.float64)
    gathers = {"n": 0}
    real = dist.all_gather
    dist.all_gather = lambda *a, **k: (gathers.__setitem__("n", gathers["n"] + 1), real(*a, **k))[1]
    with torch.no_grad(), sharding.particle_sharded():
        names = ("sigma_x", "sigma_px", "sigma_y", "sigma_p", "mu_x", "mu_y", "mu_tau", "cov_xpx", "cov_taup", "emittance_x", "beta_y", "alpha_x",
                 "total_charge", "num_particles_survived")
        got = {n: float(getattr(beam, n)) for n in names}
        assert calls["n"] == 1 and gathers["n"] == 1, (calls, gathers)          # ONE local reduction and ONE exchange for all of them
        beam.particles[:, 0] *= 2.0                                               # a new version of the beam: reduced and exchanged again
        got["sigma_x_doubled"] = float(beam.sigma_x)
        assert calls["n"] == 2 and gathers["n"] == 2
    local_sigma = float(beam.sigma_x)                                            # outside the context: this shard's own statistics
    assert calls["n"] == 3 and gathers["n"] == 2
    dist.all_gather = real
    q.put((rank, got, local_sigma))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, target, args):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, *args, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, deadline = [], time.monotonic() + 180
    while len(out) < world:                       # (a worker that died must fail the test, not park it on the queue)
        if not q.empty():
            out.append(q.get())
            continue
        dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
        if dead or time.monotonic() > deadline:
            for p in procs:
                if p.is_alive():
                    p.kill()
            raise AssertionError(f"worker exit codes {dead}" if dead else "workers timed out")
        time.sleep(0.05)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return sorted(out, key=lambda t: t[0])


def test_sharded_beam_properties_are_global_two_ranks_gloo(oracle):
    """particle_beam.py:1699-1943 on a particle-sharded beam: inside `sharding.particle_sharded` every rank reads the statistics of
    the UNION of the shards (the reference's weighted statistics, utils/statistics.py:4-62, over all particles) from one cached
    exchange; outside the context a shard's own."""
    rng = np.random.default_rng(21)
    N = 15_001
    x = rng.standard_normal((N, 7)) * [3e-4, 2e-5, 2e-4, 1e-5, 1e-4, 2e-3, 0] + [1e-4, 0, -2e-4, 0, 1e-5, 0, 1]
    x[:, 1] += 0.05 * x[:, 0]                                        # x-px and x-p correlations: the optics functions are not trivial
    x[:, 0] += 0.02 * x[:, 5]
    x[:9000, 0] += 2e-4                                              # the shards have different means: the merge term matters
    w = rng.random(N)
    qc = rng.random(N) * 1e-15
    res = _run(2, _worker_beam_properties, (x, w, qc))
    mom = oracle.moments(x[None], w[None])
    import cheetah_amd as ca

    union = ca.ParameterBeam._from_moment_vector(torch.from_numpy(mom["raw"]).reshape(29), torch.float64, torch.tensor(1e8, dtype=torch.float64))
    x2 = x.copy()
    x2[:, 0] *= 2.0
    sig2 = float(np.sqrt(oracle.moments(x2[None], w[None])["cov"][0, 0, 0]))
    for rank, got, local_sigma in res:
        for name, v in got.items():
            if name == "total_charge":
                want = float((qc * w).sum())
            elif name == "num_particles_survived":
                want = float(w.sum())
            elif name == "sigma_x_doubled":
                want = sig2
            else:
                want = float(getattr(union, name))
            assert v == pytest.approx(want, rel=1e-9, abs=1e-30), (rank, name, v, want)
        lo, hi = (0, 7501) if rank == 0 else (7501, N)
        own = x[lo:hi].copy()
        own[:, 0] *= 2.0
        assert local_sigma == pytest.approx(float(np.sqrt(oracle.moments(own[None], w[None, lo:hi])["cov"][0, 0, 0])), rel=1e-10)


def _worker_uneven(rank, world, port, x, w, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cheetah_amd import sharding
    from oracle import chx_oracle

    out = []
    for n_total in (x.shape[0], 5, 8, 11):                # 5 < world: three EMPTY shards; 11: shards of 2 and 1
        lo, hi = sharding.shard_range(n_total, rank, world)
        if hi > lo:
            local = np.nan_to_num(chx_oracle.moments(x[None, lo:hi], w[None, lo:hi])["raw"], nan=0.0, posinf=0.0, neginf=0.0)
        else:
            local = np.zeros((1, 29))                     # a rank without particles: no weight, contributes nothing
        merged = sharding.gather_merge_moments(torch.from_numpy(local))
        grid = torch.full((2, 3), float(hi - lo), dtype=torch.float64)
        sharding.allreduce_grid(grid)
        out.append((merged.numpy(), float(grid[0, 0]), (lo, hi)))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_uneven_and_empty_shards_gloo(oracle):
    """World 8 (the driver's scaling run): `shard_range` + all-gather + exact merge with shards of unequal size, shards of ONE
    particle and EMPTY shards (fewer particles than ranks) — every rank ends with the moments of the union."""
    rng = np.random.default_rng(8)
    N, world = 1003, 8
    x = rng.standard_normal((N, 7)) * [1e-3, 1e-5, 2e-3, 1e-5, 1e-4, 1e-3, 0] + [5e-3, 0, -1e-3, 0, 0, 2e-2, 1]
    w = 0.25 + rng.random(N)
    res = _run(world, _worker_uneven, (x, w))
    for k, n_total in enumerate((N, 5, 8, 11)):
        ref = oracle.moments(x[None, :n_total], w[None, :n_total])["raw"]
        cuts = [r[1][k][2] for r in res]
        assert cuts[0][0] == 0 and cuts[-1][1] == n_total and all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        if n_total == 5:
            assert sum(1 for lo, hi in cuts if hi == lo) == 3
        sig = np.sqrt(ref[0, [8, 14, 19, 23, 26, 28]])
        for rank, per_case in res:
            merged, count, _ = per_case[k]
            assert count == float(n_total), (rank, n_total, count)
            assert np.allclose(merged[0, :2], ref[0, :2], rtol=1e-13)
            assert np.all(np.abs(merged[0, 2:8] - ref[0, 2:8]) <= 1e-12 * (sig + np.abs(ref[0, 2:8]))), (rank, n_total)
            j = 8
            for a in range(6):
                for b in range(a, 6):
                    assert abs(merged[0, j] - ref[0, j]) <= 1e-10 * sig[a] * sig[b], (rank, n_total, a, b)
                    j += 1


# ---------------------------------------------------------------------------------------------- gradients across the shards
def _cpu_moments_bwd(x, w, out, d_out, B, need_x, need_w):
    """chx_moments_bwd_w restated in torch (csrc/chx_moments.hip moments_bwd_kernel): every term from the moment vector `out`
    and the row itself — the property that makes the backward pass of a sharded beam local."""
    o, g = out.reshape(B, 29), d_out.reshape(B, 29)
    dX = torch.zeros(B, x.shape[1], 7, dtype=torch.float64) if need_x else None
    dW = torch.zeros(B, x.shape[1], dtype=torch.float64) if need_w else None
    for b in range(B):
        xb = x[0 if x.shape[0] == 1 else b].double()
        wb = torch.ones(xb.shape[0], dtype=torch.float64) if w is None else w[0 if w.shape[0] == 1 else b].double()
        W, W2 = o[b, 0], o[b, 1]
        icf = 1.0 / (W - W2 / W)
        G = torch.zeros(6, 6, dtype=torch.float64)
        k = 8
        for i in range(6):
            for j in range(i, 6):
                if i == j:
                    G[i, i] = 2.0 * g[b, k]
                else:
                    G[i, j] = G[j, i] = g[b, k]
                k += 1
        d = xb[:, :6] - o[b, 2:8]
        s = d @ G.T
        if need_x:
            dX[b, :, :6] = wb[:, None] * (g[b, 2:8] / W + icf * s)
        if need_w:
            S = (g[b, 8:] * o[b, 8:]).sum()
            kcf = 1.0 + W2 / (W * W)
            dW[b] = g[b, 0] + 2.0 * wb * g[b, 1] + (d * g[b, 2:8]).sum(1) / W + icf * (0.5 * (d * s).sum(1) - S * (kcf - 2.0 * wb / W))
    return (None if dX is None else dX.to(x.dtype)), (None if dW is None else dW.to(x.dtype))


def _worker_sharded_gradients(rank, world, port, x, w, qc, q):
    """d(loss of the GLOBAL statistics) / d(a replicated setting) through the product's classes on a particle-sharded beam. The two
    libchx calls of the node — the local one-pass moments and chx_moments_bwd_w — are replaced by CPU restatements (no GPU here);
    the node itself, the exchange, the merge, the property algebra, `total_charge` and `all_reduce_gradients` are the product's."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cheetah_amd as ca
    from cheetah_amd import _ops, sharding
    from oracle import chx_oracle

    def cpu_moments_raw(xx, ww, B, N, entry=None):
        xs = xx.detach().numpy().astype(np.float64)
        ws = np.ones((1, N)) if ww is None else ww.detach().numpy().astype(np.float64)
        xs, ws = np.broadcast_to(xs, (B, N, 7)), np.broadcast_to(ws, (B, N))
        return torch.from_numpy(np.nan_to_num(chx_oracle.moments(xs, ws)["raw"], nan=0.0)).reshape(B, 29)

    _ops._moments_raw = cpu_moments_raw
    _ops._moments_bwd_raw = _cpu_moments_bwd
    _ops.require_device = lambda *a: None
    lo, hi = sharding.shard_range(x.shape[0], rank, world)
    f64 = torch.float64
    k = torch.tensor(0.7, dtype=f64, requires_grad=True)          # replicated "settings": a focusing strength ...
    a = torch.tensor(1.3, dtype=f64, requires_grad=True)          # ... and an aperture-like weight scale
    x0, w0, q0 = torch.from_numpy(x[lo:hi]), torch.from_numpy(w[lo:hi]), torch.from_numpy(qc[lo:hi])

    def beam_of(xx, ww, qq):
        R = torch.eye(7, dtype=f64)
        R = R + k * torch.tensor([[0, 0.5, 0, 0, 0, 0, 0], [-0.8, 0, 0, 0, 0, 0.1, 0]] + [[0] * 7] * 5, dtype=f64)
        weights = torch.sigmoid(a * (1.0 - (xx[:, 0] / 4e-4) ** 2)) * ww
        return ca.ParticleBeam(xx @ R.T, torch.tensor(1e8, dtype=f64), particle_charges=qq, survival_probabilities=weights, dtype=f64)

    def loss_of(beam):
        return beam.sigma_x * 3e3 + beam.mu_px * 1e4 + beam.cov_xpx * 1e8 + beam.emittance_x * 1e9 + beam.total_charge * 1e12 \
            + beam.num_particles_survived * 1e-4

    with sharding.particle_sharded():
        loss = loss_of(beam_of(x0, w0, q0))
        loss.backward()
        shares = (float(k.grad), float(a.grad))
        sharding.all_reduce_gradients([k, a])
        # an active BPM inside a differentiable track of a sharded beam: the reading is formed under no_grad, nothing raises
        bpm = ca.BPM(is_active=True, dtype=f64)
        bpm.track(beam_of(x0, w0, q0))
        reading = bpm.reading.clone()
    q.put((rank, float(loss.detach()), float(k.grad), float(a.grad), shares, reading.tolist(), bool(bpm.reading.requires_grad)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_beam_properties_are_differentiable_two_ranks_gloo(oracle):
    """utils/statistics.py:4-62 and particle_beam.py:1699-1717 are differentiable whatever the layout of the particles: a loss
    of the global beam statistics, back-propagated on every rank of a particle-sharded beam, gives shares that add up to the
    single-process gradient (computed here with plain torch autograd on the union of the shards)."""
    rng = np.random.default_rng(33)
    N = 9_001
    x = rng.standard_normal((N, 7)) * [3e-4, 2e-5, 2e-4, 1e-5, 1e-4, 2e-3, 0] + [1e-4, 0, -2e-4, 0, 1e-5, 0, 1]
    x[:, 1] += 0.05 * x[:, 0]
    x[:5000, 0] += 2e-4
    w = 0.2 + 0.8 * rng.random(N)
    qc = rng.random(N) * 1e-15
    res = _run(2, _worker_sharded_gradients, (x, w, qc))
    # the single-process truth: the same expressions on all particles, statistics written out in torch (utils/statistics.py:4-62)
    f64 = torch.float64
    k = torch.tensor(0.7, dtype=f64, requires_grad=True)
    a = torch.tensor(1.3, dtype=f64, requires_grad=True)
    xx, ww, qq = torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(qc)
    R = torch.eye(7, dtype=f64) + k * torch.tensor([[0, 0.5, 0, 0, 0, 0, 0], [-0.8, 0, 0, 0, 0, 0.1, 0]] + [[0] * 7] * 5, dtype=f64)
    y = xx @ R.T
    wt = torch.sigmoid(a * (1.0 - (xx[:, 0] / 4e-4) ** 2)) * ww
    W = wt.sum()
    mean = lambda v: (wt * v).sum() / W  # noqa: E731
    cov = lambda u, v: (wt * (u - mean(u)) * (v - mean(v))).sum() / (W - (wt * wt).sum() / W)  # noqa: E731
    sxx, spp, sxp = cov(y[:, 0], y[:, 0]), cov(y[:, 1], y[:, 1]), cov(y[:, 0], y[:, 1])
    sdd, sxd, spd = cov(y[:, 5], y[:, 5]), cov(y[:, 0], y[:, 5]), cov(y[:, 1], y[:, 5])
    emit = ((sxx - sxd * sxd / sdd) * (spp - spd * spd / sdd) - (sxp - sxd * spd / sdd) ** 2).sqrt()     # beam.py:442-470 (dispersion corrected)
    loss = sxx.sqrt() * 3e3 + mean(y[:, 1]) * 1e4 + sxp * 1e8 + emit * 1e9 + (qq * wt).sum() * 1e12 + W * 1e-4
    loss.backward()
    mu_xy = (float(mean(y[:, 0]).detach()), float(mean(y[:, 2]).detach()))
    for rank, got_loss, gk, ga, shares, reading, reading_grad in res:
        assert got_loss == pytest.approx(float(loss.detach()), rel=1e-10)
        assert gk == pytest.approx(float(k.grad), rel=1e-8) and ga == pytest.approx(float(a.grad), rel=1e-8), (rank, gk, ga)
        assert reading == pytest.approx(mu_xy, rel=1e-9) and not reading_grad
    # the shares differ between the ranks (they come through different particles) and add up
    assert res[0][4] != res[1][4]
    assert res[0][4][0] + res[1][4][0] == pytest.approx(float(k.grad), rel=1e-8)
    assert res[0][4][1] + res[1][4][1] == pytest.approx(float(a.grad), rel=1e-8)
