"""Screen(method="kde") (chx_kde_values + GEMM) against images made by the reference (tests/golden/kde.npz)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def screen(ca, dtype):
    kw = {"dtype": dtype, "device": "cuda"}
    return ca.Screen(resolution=(64, 48), pixel_size=torch.tensor([4e-5, 5e-5], **kw), method="kde",
                     kde_bandwidth=torch.tensor(6e-5, **kw), misalignment=torch.tensor([1e-4, -5e-5], **kw), is_active=True,
                     **kw)


@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("name", ["scalar", "vector"])
def test_kde_images_match_reference(golden, tag, name):
    import cheetah_amd as ca

    g = golden("kde.npz")
    dtype = torch.float64 if tag == "f64" else torch.float32
    kw = {"dtype": dtype, "device": "cuda"}
    x = torch.tensor(g["x"] if name == "vector" else g["x"][0], **kw)
    beam = ca.ParticleBeam(x, torch.tensor(1e8, **kw), particle_charges=torch.tensor(g["q"], **kw),
                           survival_probabilities=torch.tensor(g["surv"], **kw), species=ca.Species("electron", **kw))
    scr = screen(ca, dtype)
    scr.track(beam)
    img = scr.reading.cpu().numpy()
    ref = g[f"{name}_{tag}"]
    assert img.shape == ref.shape and img.dtype == ref.dtype
    rt = 1e-11 if tag == "f64" else 2e-4
    assert np.allclose(img, ref, rtol=rt, atol=rt * ref.max())
    assert np.allclose(img.sum(axis=(-2, -1)), 1.0, rtol=1e-5)        # a probability density over the pixels


def test_kde_gradient_matches_reference(golden):
    import cheetah_amd as ca

    g = golden("kde.npz")
    kw = {"dtype": torch.float64, "device": "cuda"}
    x = torch.tensor(g["x"][0], **kw).requires_grad_(True)
    beam = ca.ParticleBeam(x, torch.tensor(1e8, **kw), particle_charges=torch.tensor(g["q"], **kw),
                           survival_probabilities=torch.tensor(g["surv"], **kw), species=ca.Species("electron", **kw))
    scr = screen(ca, torch.float64)
    scr.track(beam)
    img = scr.reading
    assert img.grad_fn is not None
    assert np.allclose(img.detach().cpu().numpy(), g["scalar_f64"], rtol=1e-11, atol=1e-11 * g["scalar_f64"].max())
    (img * torch.tensor(g["W"], **kw)).sum().backward()
    ref = g["scalar_f64_dx"]
    assert np.allclose(x.grad.cpu().numpy(), ref, rtol=1e-8, atol=1e-10 * np.abs(ref).max())


def test_kde_chunked_accumulation(monkeypatch):
    """More particles than one GEMM slab: the image equals the single-slab one."""
    import cheetah_amd as ca
    from cheetah_amd import _ops

    kw = {"dtype": torch.float32, "device": "cuda"}
    torch.manual_seed(4)
    x = torch.randn(5000, 7, **kw) * torch.tensor([4e-4, 1e-5, 3e-4, 1e-5, 1e-5, 1e-3, 0.0], **kw)
    beam = ca.ParticleBeam(x, torch.tensor(1e8, **kw), species=ca.Species("electron", **kw))
    scr = screen(ca, torch.float32)
    scr.track(beam)
    whole = scr.reading.clone()
    monkeypatch.setattr(_ops, "KDE_CHUNK", 1024)
    scr.track(beam)
    assert torch.allclose(scr.reading, whole, rtol=1e-4, atol=1e-6 * float(whole.max()))
    with pytest.raises(AssertionError, match="Invalid method"):      # the reference asserts (screen.py:87-91)
        ca.Screen(method="nearest")
