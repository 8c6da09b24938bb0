"""Gradients through Screen.reading (cloud-in-cell): HIP forward + chx_cic_deposit_bwd vs the reference's autograd."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_screen_reading_gradients(golden, tag):
    import cheetah_amd as ca

    g = golden("screen_grad.npz")
    dt = torch.float64 if tag == "f64" else torch.float32
    t = lambda v: torch.tensor(v, dtype=dt, device="cuda")  # noqa: E731
    kw = {"dtype": dt, "device": "cuda"}
    k1 = torch.nn.Parameter(t(2.5))
    seg = ca.Segment([ca.Quadrupole(t(0.3), k1=k1, **kw), ca.Drift(t(0.5), **kw),
                      ca.Screen(resolution=(40, 30), pixel_size=t([5e-5, 6e-5]), misalignment=t([1e-4, -5e-5]),
                                is_active=True, method="cloud-in-cell", name="scr", **kw)])
    parts = torch.tensor(g[f"in_{tag}"], device="cuda").requires_grad_(True)
    q = torch.tensor(g[f"q_{tag}"], device="cuda").requires_grad_(True)
    mass, nq = g[f"species_{tag}"]
    sp = ca.Species("custom_electron", num_elementary_charges=torch.tensor(nq, dtype=dt, device="cuda"),
                    mass_eV=torch.tensor(mass, dtype=torch.float64, device="cuda"))
    beam = ca.ParticleBeam(parts, torch.tensor(g[f"energy_{tag}"], device="cuda"), particle_charges=q, species=sp)
    seg.track(beam)
    img = seg.scr.reading
    W = torch.tensor(g[f"W_{tag}"], device="cuda")
    loss = (img * W).sum() * 1e15
    loss.backward()
    rt = 1e-9 if tag == "f64" else 5e-3
    assert np.allclose(img.detach().cpu().numpy(), g[f"img_{tag}"], rtol=10 * rt, atol=rt * g[f"img_{tag}"].max())
    assert float(loss.detach()) == pytest.approx(float(g[f"loss_{tag}"]), rel=rt)
    assert float(k1.grad) == pytest.approx(float(g[f"dk1_{tag}"]), rel=10 * rt)
    for name, got in (("dparticles", parts.grad), ("dq", q.grad)):
        ref = g[f"{name}_{tag}"]
        got = got.cpu().numpy()
        scale = np.abs(ref).max(axis=0)
        scale = np.where(scale == 0, 1.0, scale)
        assert np.max(np.abs(got - ref) / scale) < 10 * rt, name
