"""Pruned / symmetry-aware Poisson solve (csrc/chx_fft.hip: chx_sc_igf_table, chx_sc_green_spectrum, chx_sc_convolve)
against the dense hipFFT formulation of the same convolution (chx_sc_igf + chx_sc_fft_exec + chx_sc_spectral_mul) and,
for the Green spectrum, against numpy's FFT of the oracle's dense Green function."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dense(rho, cell, gamma, scale, g):
    from cheetah_amd import _ops

    B = rho.shape[0]
    G2 = tuple(2 * v for v in g)
    plan = _ops.ScFftPlan(B, g, rho.dtype)
    green = _ops.sc_igf(cell, gamma, g, padded=True)
    plan.forward(green, which=1)
    pad = torch.zeros((B, G2[0], G2[1], G2[2] + 2), dtype=rho.dtype, device=rho.device)
    pad[:, : g[0], : g[1], : g[2]] = rho
    plan.forward(pad, which=0)
    _ops.sc_spectral_mul(pad, green, scale)
    plan.inverse(pad)
    return pad[:, : g[0], : g[1], : g[2]].clone()


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("g", [(32, 32, 32), (16, 32, 64), (64, 64, 64), (128, 128, 128), (256, 16, 32), (16, 32, 256)])
def test_pruned_convolution_matches_dense_fft(tag, g):
    from cheetah_amd import _ops

    dt = torch.float32 if tag == "f32" else torch.float64
    # line lengths 32 ... 256 run the register-resident kernels, 512 the LDS radix-2 kernel (along x: (256, 16, 32); along z,
    # where it replaces the fused forward-multiply-inverse pass by three kernels: (16, 32, 256))
    if not _ops.sc_pruned_supported(g, dt):
        assert dt == torch.float64 and max(g) == 256
        pytest.skip("fp64 lines of 512 points do not fit the LDS tile")
    B = 2
    torch.manual_seed(0)
    rho = torch.rand((B, *g), dtype=dt, device="cuda") ** 4
    cell = torch.tensor([[1.1e-4, 0.9e-4, 2.0e-6], [2.0e-4, 1.0e-4, 1.0e-6]], dtype=dt, device="cuda")
    gamma = torch.tensor([489.0, 20.0], dtype=dt, device="cuda")
    scale = torch.tensor([1.0 / (8 * g[0] * g[1] * g[2]), 3.0 / (8 * g[0] * g[1] * g[2])], dtype=torch.float64, device="cuda")
    ref = _dense(rho, cell, gamma, scale, g)
    Ghat = _ops.sc_green_spectrum(cell, gamma, g)
    phi = _ops.sc_convolve(rho, Ghat, scale, g)
    err = (phi - ref).abs().amax(dim=(1, 2, 3)) / ref.abs().amax(dim=(1, 2, 3))
    assert torch.all(err < (2e-5 if tag == "f32" else 1e-12)), err


def test_green_spectrum_is_the_fft_of_the_dense_green_function(oracle):
    from cheetah_amd import _ops

    g = (16, 32, 16)
    cell = np.array([[1.1e-4, 0.9e-4, 2.0e-6]])
    gamma = np.array([100.0])
    dense = oracle.igf(cell * np.array([1.0, 1.0, gamma[0]]), g)[0]                  # (2gx, 2gy, 2gz)
    spec = np.fft.fftn(dense)
    assert np.abs(spec.imag).max() < 1e-9 * np.abs(spec.real).max()                   # real and even
    want = spec.real[: g[0] + 1, : g[1] + 1, : g[2] + 1]
    got = _ops.sc_green_spectrum(torch.tensor(cell, device="cuda"), torch.tensor(gamma, device="cuda"), g)[0].cpu().numpy()
    assert np.max(np.abs(got - want)) < 1e-11 * np.abs(want).max()


def test_unsupported_grids_are_reported():
    from cheetah_amd import _ops

    assert not _ops.sc_pruned_supported((48, 32, 32), torch.float32)
    assert not _ops.sc_pruned_supported((32, 32, 8), torch.float32)
    assert _ops.sc_pruned_supported((512, 32, 32), torch.float32) and not _ops.sc_pruned_supported((512, 32, 32), torch.float64)


# float32 particle step against the float64 one, as a fraction of the largest kick of the coordinate: measured <= 5.1e-6 on MI355X
# on this test's rough potential (uniform noise to the fourth power: neighbouring nodes differ by the full amplitude, so the eight
# corner terms cancel; on C4's smooth potential the two steps agree to 1e-7 of the kick, benchmarks/sc_fp32_error.py), bound 4 x that
FP32_STEP_BOUND = 2.0e-5


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("g", [(16, 16, 16), (32, 16, 64), (128, 128, 128)])
def test_gather_from_the_potential_is_bit_identical_to_gradient_then_gather(tag, g):
    """chx_sc_convolve_halo + chx_sc_gather_kick_phi (what chx_sc_kick runs) against chx_sc_convolve + chx_sc_gradient +
    chx_sc_gather_kick(_mapped): the same potential bits inside the halo, the same kicked particles bit for bit — including
    particles in the boundary cells, in the half cell outside the first / last node and far outside the grid."""
    from cheetah_amd import _ops

    dt = torch.float32 if tag == "f32" else torch.float64
    B, N = 2, 40_000
    torch.manual_seed(3)
    rho = torch.rand((B, *g), dtype=dt, device="cuda") ** 4 * 1e-3
    cell = torch.tensor([[1.1e-4, 0.9e-4, 2.0e-6], [2.0e-4, 1.0e-4, 1.0e-6]], dtype=dt, device="cuda")
    gamma = torch.tensor([489.0, 20.0], dtype=dt, device="cuda")
    energy = gamma * 510998.95069
    scale = torch.tensor([1.0e9, 3.0e9], dtype=torch.float64, device="cuda")
    Ghat = _ops.sc_green_spectrum(cell, gamma, g)
    phi = _ops.sc_convolve(rho, Ghat, scale, g)
    halo = torch.full((B, g[0] + 4, g[1] + 4, g[2] + 4), float("nan"), dtype=dt, device="cuda")   # poison: must never matter
    lib = _ops._lib.lib()
    b3, code = _ops._bins3(g), _ops.dtype_code(dt)
    ws_bytes = lib.chx_sc_convolve_workspace_bytes(B, b3, code)
    ws = _ops.workspace(ws_bytes, rho.device)
    _ops.check(lib.chx_sc_convolve_halo(rho.data_ptr(), Ghat.data_ptr(), scale.data_ptr(), B, b3, code, halo.data_ptr(), ws.data_ptr(),
                                        ws_bytes, _ops.stream_ptr()), "chx_sc_convolve_halo")
    assert torch.equal(halo[:, 2:-2, 2:-2, 2:-2], phi)
    inner = torch.zeros_like(halo, dtype=torch.bool)
    inner[:, 2:-2, 2:-2, 2:-2] = True
    assert torch.isnan(halo[~inner]).all()                      # the halo is not written
    assert torch.equal(_ops.sc_convolve_halo(rho, Ghat, scale, g)[:, 2:-2, 2:-2, 2:-2], phi)

    half = cell * torch.tensor(g, dtype=dt, device="cuda") / 2
    # positions in SI metres: uniform over 1.3 x the grid (so ~40 % of the particles miss it in some axis), plus particles
    # pinned to the first / last cells, exactly on nodes, and far away
    u = (torch.rand(B, N, 3, dtype=dt, device="cuda") * 2 - 1) * 1.3
    u[:, :200] = (torch.rand(B, 200, 3, dtype=dt, device="cuda") * 2 - 1) * 1.0
    u[:, 200:400, 0] = -1 + torch.rand(B, 200, dtype=dt, device="cuda") * 2 / g[0]
    u[:, 400:600, 1] = 1 - torch.rand(B, 200, dtype=dt, device="cuda") * 2 / g[1]
    u[:, 600:800, 2] = 1 - torch.rand(B, 200, dtype=dt, device="cuda") * 3 / g[2]
    u[:, 800:810] = torch.tensor([-1.0, 1.0, -1.0], dtype=dt, device="cuda")
    u[:, 810:820] = 1e6
    pos = u * half[:, None, :]
    xyz = torch.zeros(B, N, 7, dtype=dt, device="cuda")
    xyz[..., 0], xyz[..., 2], xyz[..., 4] = pos[..., 0], pos[..., 1], pos[..., 2]
    xyz[..., 1], xyz[..., 3], xyz[..., 5] = 1e-23, -2e-23, 2.6e-19
    xyz[..., 6] = 1
    x = torch.empty_like(xyz)
    _ops.check(lib.chx_from_xyz_pxpypz(xyz.data_ptr(), energy.data_ptr(), 510998.95069, B, B, B, N, code, x.data_ptr(), _ops.stream_ptr()),
               "from_xyz")
    dtk = torch.tensor([1e-9, 3e-9], dtype=dt, device="cuda")

    def same(a, b, before):
        # float64 rows: one arithmetic on both routes, bit for bit. float32 rows (round 6): chx_sc_gather_kick_phi evaluates the
        # particle step in float32 (csrc/chx_spacecharge.hip, sc_kick_row32), the force-grid route in float64 — the same real-number
        # map, so they agree to a few 1e-7 of the kick plus the last bit of the coordinate. Measured on MI355X: see FP32_STEP_BOUND.
        if tag == "f64":
            assert torch.equal(a, b)
            return
        kick = (b - before).abs().amax(dim=1, keepdim=True)
        excess = ((a - b).abs() - 2 * torch.finfo(dt).eps * b.abs()).clamp_min(0) / kick.clamp_min(1e-30)
        assert (excess <= FP32_STEP_BOUND).all(), excess.amax(dim=1)

    F = _ops.sc_gradient(phi, cell, gamma, g)
    want = _ops.sc_gather_kick(x, F, half, cell, energy, dtk, 510998.95069, B, N, g)
    got = _ops.sc_gather_kick_phi(x, halo, half, cell, gamma, energy, dtk, 510998.95069, B, N, g)
    assert not torch.isnan(got).any()
    same(got, want, x)
    assert not torch.equal(want, x)
    # with the linear run folded in, per batch row
    R = (torch.eye(7, dtype=dt, device="cuda") + 0.05 * torch.randn(B, 7, 7, dtype=dt, device="cuda")).contiguous()
    R[:, 6] = 0
    R[:, 6, 6] = 1
    want_m = torch.empty_like(x)
    _ops.check(lib.chx_sc_gather_kick_mapped(x.data_ptr(), F.data_ptr(), half.data_ptr(), cell.data_ptr(), energy.data_ptr(),
                                             dtk.data_ptr(), 510998.95069, B, B, B, N, b3, code, R.data_ptr(), B, want_m.data_ptr(),
                                             _ops.stream_ptr()), "gather mapped")
    got_m = _ops.sc_gather_kick_phi(x, halo, half, cell, gamma, energy, dtk, 510998.95069, B, N, g, post_map=R)
    if tag == "f64":
        assert torch.equal(got_m, want_m)
    else:
        # the kicked row goes through R: a difference of FP32_STEP_BOUND kicks in px, py, delta reaches every coordinate
        kick = (want - x).abs().amax(dim=1, keepdim=True)                                   # (B, 1, 7)
        reach = (R.abs() @ kick.transpose(1, 2)).transpose(1, 2)                            # (B, 1, 7)
        lim = FP32_STEP_BOUND * reach + 4 * torch.finfo(dt).eps * want_m.abs().amax(dim=1, keepdim=True)
        assert ((got_m - want_m).abs() <= lim).all(), ((got_m - want_m).abs() / lim).amax(dim=1)
