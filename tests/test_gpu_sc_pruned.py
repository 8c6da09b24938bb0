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
