"""A beam whose particles are spread over two processes (sharding.particle_sharded): the SpaceChargeKick takes its grid
from the global moments and sums the charge grid over the ranks, so the union of the two outgoing slices equals the
single-process result. Both ranks share the one GPU of the test box, hence the gloo backend (RCCL needs one device per
rank); the collectives are the ones RCCL runs on a multi-GPU node."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make(ca, x, q, dtype):
    kw = {"dtype": dtype, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    seg = ca.Segment([ca.Drift(t(0.1), **kw), ca.SpaceChargeKick(t(0.3), grid_shape=(32, 32, 32), **kw), ca.Drift(t(0.2), **kw),
                      ca.Quadrupole(t(0.1), k1=t(3.0), **kw), ca.SpaceChargeKick(t(0.3), grid_shape=(32, 32, 32), **kw)])
    beam = ca.ParticleBeam(x.to("cuda"), t(2e7), particle_charges=q.to("cuda"), species=ca.Species("electron", **kw))
    return seg, beam


def _worker(rank, world, port, x, q, queue, cuts=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cheetah_amd as ca
    from cheetah_amd import _ops, sharding

    lo, hi = sharding.shard_range(x.shape[0], rank, world) if cuts is None else (cuts[rank], cuts[rank + 1])
    seg, beam = _make(ca, x[lo:hi], q[lo:hi], x.dtype)
    links, orig = [], _ops.sc_kick_sorted
    _ops.sc_kick_sorted = lambda *a, **k: (links.append(k.get("group") is not None), orig(*a, **k))[1]
    kw = {"dtype": x.dtype, "device": "cuda"}
    screen = ca.Screen(resolution=(64, 48), pixel_size=torch.tensor([4e-5, 5e-5], **kw), is_active=True, **kw)
    with sharding.particle_sharded():
        out = seg.track(beam)
        sigma = sharding.global_moments(out)[8].sqrt()
        screen.track(out)
        image_sum = float(screen.reading.sum())       # summed over the ranks inside the context
    queue.put((rank, out.particles.cpu().numpy(), float(sigma), image_sum, links))   # by value: the process may be gone when it is read
    dist.barrier()
    dist.destroy_process_group()


# (dtype, N, cuts): float64 / 60 000 — both ranks below the tile sort's minimum: the staged kick with the direct deposit;
# float32 / 150 001 — both ranks on the tile-ordered CHAIN (chx_sc_kick_sorted_begin / _finish around the two exchanges), an odd
# split; float64 / 180 000 cut at 120 000 — rank 0 on the chain, rank 1 (60 000 rows) kick by kick, and float32 / 140 000 cut at 50 000 the other way
# round: the exchanges carry the same contents on both paths, so the ranks need not agree on the path
@pytest.mark.parametrize("dtype,N,cuts,chained", [(torch.float64, 60_000, None, (False, False)), (torch.float32, 150_001, None, (True, True)),
                                                  (torch.float64, 180_000, (0, 120_000, 180_000), (True, False)),
                                                  (torch.float32, 140_000, (0, 50_000, 140_000), (False, True))])
def test_particle_sharded_space_charge_equals_single_process(dtype, N, cuts, chained):
    import cheetah_amd as ca

    torch.manual_seed(12)
    x = torch.randn(N, 7, dtype=dtype) * torch.tensor([3e-4, 2e-5, 2e-4, 3e-5, 2e-5, 1e-3, 0.0], dtype=dtype)
    x[:, 6] = 1.0
    q = torch.full((N,), 2e-9 / N, dtype=dtype)
    seg, beam = _make(ca, x, q, dtype)
    whole = seg.track(beam)
    ref, sigma_ref = whole.particles.cpu(), float(whole.sigma_x)

    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, x, q, queue, cuts)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict()
    for _ in range(2):
        rank, parts, sigma, image_sum, links = queue.get(timeout=300)
        results[rank] = (parts, sigma, image_sum, links)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):      # two kicks: two links of a chain with the group handed down, or none (staged kick by kick)
        assert results[r][3] == ([True, True] if chained[r] else []), (r, results[r][3])
    joined = torch.cat([torch.from_numpy(results[0][0]), torch.from_numpy(results[1][0])], dim=0)
    kick = (ref - x).abs().max(dim=0).values                   # size of the effect per coordinate
    tol = 1e-9 if dtype == torch.float64 else 2e-4             # fp32: the charge sum order differs between the runs (measured 3e-5)
    assert torch.all((joined - ref).abs() <= tol * kick + 1e-30), ((joined - ref).abs().max(dim=0).values / kick)
    kw = {"dtype": dtype, "device": "cuda"}
    screen = ca.Screen(resolution=(64, 48), pixel_size=torch.tensor([4e-5, 5e-5], **kw), is_active=True, **kw)
    screen.track(whole)
    total = float(screen.reading.sum())
    for r in range(2):
        assert results[r][1] == pytest.approx(sigma_ref, rel=1e-6 if dtype == torch.float64 else 1e-4)
        assert results[r][2] == pytest.approx(total, rel=1e-5)


def test_merge_moments_kernel_equals_tensor_formula():
    """chx_merge_moments (what the RCCL path runs after its all-gather) against sharding.merge_moments."""
    import cheetah_amd as ca  # noqa: F401
    from cheetah_amd import _lib, _ops, sharding

    torch.manual_seed(1)
    R, B, N = 3, 5, 4000
    shards = []
    for r in range(R):
        x = torch.randn(B, N + 100 * r, 7, dtype=torch.float64, device="cuda") * (1 + r) + 0.1 * r
        w = torch.rand(B, N + 100 * r, dtype=torch.float64, device="cuda")
        if r == 1:
            w[2] = 0.0                                           # a shard without weight in one batch row
        shards.append(_ops.moments(x, w))
    per_rank = torch.stack(shards, dim=0).contiguous()
    per_rank[1, 2] = 0.0                                         # (its moments are NaN / undefined: they must be ignored)
    ref = sharding.merge_moments(per_rank)
    out = torch.empty((B, 29), dtype=torch.float64, device="cuda")
    _ops.check(_lib.lib().chx_merge_moments(per_rank.data_ptr(), R, B, out.data_ptr(), _ops.stream_ptr()), "chx_merge_moments")
    assert torch.allclose(out, ref, rtol=1e-12, atol=1e-300)


def _bpm_worker(rank, world, port, x, w, queue):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cheetah_amd as ca
    from cheetah_amd import sharding

    lo, hi = sharding.shard_range(x.shape[0], rank, world)
    seg, bpms = _bpm_lattice(ca)
    beam = ca.ParticleBeam(x[lo:hi].to("cuda"), torch.tensor(1e8, device="cuda"), survival_probabilities=w[lo:hi].to("cuda"), device="cuda",
                           dtype=torch.float32)
    with sharding.particle_sharded(), torch.no_grad():
        out = seg.track(beam)
    queue.put((rank, out.particles.cpu().numpy(), torch.stack([b.reading for b in bpms]).cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _bpm_lattice(ca):
    kw = {"dtype": torch.float32, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    els, bpms = [], []
    for i in range(5):
        bpm = ca.BPM(is_active=True, misalignment=t([1e-5 * i, -1e-5]), **kw)
        bpms.append(bpm)
        els += [ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw), ca.Drift(t(0.5), **kw),
                ca.Aperture(x_max=t(1.5e-3), y_max=t(1.5e-3), **kw), bpm]
    return ca.Segment(els), bpms


def test_particle_sharded_bpms_read_the_means_of_all_shards():
    """Two ranks hold the halves of one beam: inside sharding.particle_sharded an active BPM reads the GLOBAL weighted means (this
    rank's one-pass moments, the 29-double all-gather, the exact merge) — the single-process readings, to rounding — while the
    particles of each half are tracked as before."""
    import cheetah_amd as ca

    torch.manual_seed(21)
    N = 40_001
    x = torch.randn(N, 7) * torch.tensor([5e-4, 2e-5, 4e-4, 3e-5, 2e-5, 1e-3, 0.0])
    x[:, 0] += 2e-4
    x[:, 6] = 1.0
    w = torch.rand(N)
    seg, bpms = _bpm_lattice(ca)
    with torch.no_grad():
        whole = seg.track(ca.ParticleBeam(x.to("cuda"), torch.tensor(1e8, device="cuda"), survival_probabilities=w.to("cuda"), device="cuda",
                                          dtype=torch.float32))
    want = torch.stack([b.reading for b in bpms]).cpu().numpy()
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bpm_worker, args=(r, 2, port, x, w, queue)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict()
    for _ in range(2):
        rank, parts, readings = queue.get(timeout=180)
        results[rank] = (parts, readings)
    for p in procs:
        p.join(timeout=60)
    import numpy as np

    union = np.concatenate([results[0][0], results[1][0]])
    assert np.array_equal(union, whole.particles.cpu().numpy())
    for r in (0, 1):
        assert np.allclose(results[r][1], want, rtol=2e-6, atol=2e-10), np.abs(results[r][1] - want).max()
    assert np.array_equal(results[0][1], results[1][1])                 # every rank holds the same global readings
    assert np.abs(want[:, 0]).min() > 1e-5
