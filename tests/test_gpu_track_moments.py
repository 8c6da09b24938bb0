"""fused track + moments (`chx_track_moments`, `Segment.track_moments`).

The kernel must return what the two-step path returns — moments (a10) of the linearly tracked beam (a1) — so it
is checked against the oracle's `moments(apply(x, R, mode=1))` (same fma chain, fp64 two-pass moments) and against
the product's own two-step path. Differences are summation order (and the one-pass shifted second moments) only.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _beam(N, dt, seed, B=None):
    rng = np.random.default_rng(seed)
    shape = (N, 7) if B is None else (B, N, 7)
    x = rng.standard_normal(shape) * np.array([2e-4, 4e-6, 2e-4, 4e-6, 8e-6, 2e-3, 0.0])
    x[..., :6] += np.array([3e-4, -1e-6, -2e-4, 2e-6, 1e-5, 1e-3])  # off-centre beam: the shift matters
    x[..., 6] = 1.0
    w = rng.uniform(0.0, 1.0, shape[:-1])
    return x.astype(dt), w.astype(dt)


def _maps(B, dt, seed):
    """FODO-like cell maps with a scanned quadrupole strength, built by the oracle (fp64) and cast."""
    from oracle import chx_oracle as O

    k1 = np.linspace(-30.0, 30.0, B)
    E = np.full(B, 1e8)
    q = O.build_rmatrix("quadrupole", np.stack([np.full(B, 0.122), k1, np.full(B, 0.01), np.full(B, 1e-4), np.full(B, -2e-4)], -1), E)
    d = O.build_rmatrix("drift", np.full((B, 1), 0.428), E)
    h = O.build_rmatrix("hcor", np.stack([np.full(B, 0.02), np.full(B, 9e-5)], -1), E)
    return O.compose([q, d, h]).astype(dt)


def _check(raw, ora, tag, packed=False):
    """raw (B,29) from the device vs oracle moments dict of the tracked particles. `packed`: the float32 rows kernel forms its
    products in packed float32 and sums 32 particles in float32 before every float64 add (csrc/chx_moments.hip
    track_moments_rows_f32_kernel): ~1e-7 relative instead of float64 rounding — still two decades below what the reference's
    own float32 reductions give."""
    ref = ora["raw"]
    t_w, t_mu, t_cov = (1e-6, 3e-7, 2e-6) if packed else (1e-13, 1e-12, 1e-11)
    assert np.allclose(raw[:, :2], ref[:, :2], rtol=t_w, atol=0)
    sig6 = np.sqrt(np.stack([ora["cov"][:, j, j] for j in range(6)], axis=-1))
    assert np.all(np.abs(raw[:, 2:8] - ref[:, 2:8]) <= t_mu * (sig6 + np.abs(ref[:, 2:8])))
    k = 8
    for i in range(6):
        for j in range(i, 6):
            assert np.all(np.abs(raw[:, k] - ref[:, k]) <= t_cov * sig6[:, i] * sig6[:, j]), (tag, i, j)
            k += 1


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("B,N,weighted", [(256, 20_000, True), (100, 3_001, False), (64, 1_500, True)])
def test_rows_path_shared_beam(oracle, tag, B, N, weighted):
    from cheetah_amd import _ops

    dt = np.float32 if tag == "f32" else np.float64
    x, w = _beam(N, dt, 1)
    R = _maps(B, dt, 2)
    tx, tR = torch.tensor(x, device="cuda"), torch.tensor(R, device="cuda")
    tw = torch.tensor(w, device="cuda") if weighted else None
    raw = _ops.track_moments(tx, tw, tR).cpu().numpy()
    y = oracle.apply(x[None], R, mode=1)
    ora = oracle.moments(y, w if weighted else None)
    packed = tag == "f32"
    _check(raw, ora, tag, packed=packed)
    # and against the product's own two-step path (bit-identical tracked particles)
    two = _ops.moments(_ops.apply_map(tx, tR), tw).cpu().numpy()
    sig = np.sqrt(np.abs(two[:, [8, 14, 19, 23, 26, 28]]))
    assert np.all(np.abs(raw[:, 2:8] - two[:, 2:8]) <= (3e-7 if packed else 1e-12) * (sig + np.abs(two[:, 2:8])))
    assert np.allclose(raw[:, [8, 14, 19, 23, 26, 28]], two[:, [8, 14, 19, 23, 26, 28]], rtol=2e-6 if packed else 1e-10, atol=0)


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_particle_path_per_row_beams_and_single_row(oracle, tag):
    from cheetah_amd import _ops

    dt = np.float32 if tag == "f32" else np.float64
    for B, N, shared_map in ((3, 40_000, False), (1, 100_003, True), (5, 777, True)):
        x, w = _beam(N, dt, 3, B=B)
        R = _maps(1 if shared_map else B, dt, 4)
        raw = _ops.track_moments(torch.tensor(x, device="cuda"), torch.tensor(w, device="cuda"),
                                 torch.tensor(R[0] if shared_map else R, device="cuda")).cpu().numpy()
        y = oracle.apply(x, R, mode=1)
        _check(raw.reshape(B, 29), oracle.moments(y, w), tag)


def test_segment_track_moments_matches_track_then_moments():
    import cheetah_amd as ca

    f32 = torch.float32
    kw = {"dtype": f32, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(0)
    beam = ca.ParticleBeam.from_parameters(num_particles=50_000, mu_x=t(1e-4), sigma_px=t(2e-6), energy=t(1e8), **kw)
    B = 128
    cell = [ca.Drift(t(0.175), **kw), ca.Quadrupole(t(0.122), k1=torch.linspace(-30, 30, B, **kw), **kw), ca.Drift(t(0.428), **kw),
            ca.Quadrupole(t(0.122), k1=t(-14.3), **kw), ca.Drift(t(0.204), **kw), ca.VerticalCorrector(t(0.02), angle=t(9e-5), **kw),
            ca.Drift(t(0.45), **kw)]
    seg = ca.Segment(cell)
    fused = seg.track_moments(beam)
    ref = seg.track(beam)
    assert fused.mu.shape == (B, 7) and fused.cov.shape == (B, 7, 7)
    for n in ("x", "px", "y", "py", "tau", "p"):
        a, b = getattr(fused, f"sigma_{n}"), getattr(ref, f"sigma_{n}")
        assert torch.allclose(a, b, rtol=1e-6, atol=0), n
        assert torch.allclose(getattr(fused, f"mu_{n}"), getattr(ref, f"mu_{n}"), rtol=1e-6, atol=1e-7 * float(b.max())), n
    assert torch.allclose(fused.cov_xpx, ref.cov_xpx, rtol=1e-5, atol=1e-20)
    assert torch.equal(fused.s, ref.s) and torch.equal(fused.energy, ref.energy)
    assert torch.allclose(fused.total_charge, ref.total_charge)
    # a non-skippable element in the middle: particles are tracked up to it, only the tail is fused
    seg2 = ca.Segment(cell[:3] + [ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(30.0), frequency=t(1.3e9), **kw)] + cell[3:])
    f2, r2 = seg2.track_moments(beam), seg2.track(beam)
    assert torch.allclose(f2.sigma_x, r2.sigma_x, rtol=1e-6) and torch.allclose(f2.sigma_p, r2.sigma_p, rtol=1e-6)
    assert torch.allclose(f2.energy, r2.energy)
    # nothing linear at the end: plain moments of the tracked beam
    seg3 = ca.Segment(cell[:2] + [ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(30.0), frequency=t(1.3e9), **kw)])
    f3, r3 = seg3.track_moments(beam), seg3.track(beam)
    assert torch.allclose(f3.sigma_y, r3.sigma_y, rtol=1e-6) and torch.equal(f3.s, r3.s)


def test_track_moments_rejects_bad_arguments():
    from cheetah_amd import _lib

    lib = _lib.lib()
    x = torch.zeros(8, 7, device="cuda")
    R = torch.eye(7, device="cuda").repeat(2, 1, 1)
    out = torch.zeros(2, 29, dtype=torch.float64, device="cuda")
    ws = torch.zeros(lib.chx_track_moments_workspace_bytes(2, 8), dtype=torch.uint8, device="cuda")
    ok = lib.chx_track_moments(x.data_ptr(), None, R.data_ptr(), None, 2, 1, 2, 1, 8, 0, out.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert ok == 0
    assert lib.chx_track_moments(x.data_ptr(), None, R.data_ptr(), None, 2, 1, 2, 1, 8, 0, out.data_ptr(), ws.data_ptr(), 8, None) == -5
    assert lib.chx_track_moments(x.data_ptr(), None, None, None, 2, 1, 2, 1, 8, 0, out.data_ptr(), ws.data_ptr(), ws.numel(), None) == -1
    assert lib.chx_track_moments(x.data_ptr(), None, R.data_ptr(), None, 2, 1, 3, 1, 8, 0, out.data_ptr(), ws.data_ptr(), ws.numel(), None) == -1
    assert lib.chx_track_moments(x.data_ptr(), None, R.data_ptr(), None, 2, 1, 2, 1, 8, 9, out.data_ptr(), ws.data_ptr(), ws.numel(), None) == -2
