"""Gradients through Cavity.track on a ParticleBeam: CavityTrack (chx_apply_affine7_bwd + chx_cavity_track_bwd for the
particle-sized work, autograd over the (B,)-sized coefficient expressions) against the reference's autograd result
(tests/golden/cavity_grad.npz, fp64)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["sw_acc", "tw_acc", "sw_dec"])
def test_cavity_gradients_match_reference(golden, name):
    import cheetah_amd as ca

    g = golden("cavity_grad.npz")
    kw = {"dtype": torch.float64, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    L, V, ph, f, E = g[f"{name}_params"]
    length, voltage = torch.nn.Parameter(t(L)), torch.nn.Parameter(t(V))
    phase, freq = torch.nn.Parameter(t(ph)), torch.nn.Parameter(t(f))
    energy = t(E).requires_grad_(True)
    xin = t(g["x"]).requires_grad_(True)
    W = t(g["W"])
    cav = ca.Cavity(length=length, voltage=voltage, phase=phase, frequency=freq, cavity_type=str(g[f"{name}_type"]), **kw)
    beam = ca.ParticleBeam(xin, energy, species=ca.Species("electron", **kw))
    out = cav.track(beam)
    assert out.particles.grad_fn is not None and out.energy.grad_fn is not None
    assert np.allclose(out.particles.detach().cpu().numpy(), g[f"{name}_out"], rtol=1e-11, atol=1e-16)
    loss = (out.particles * W).sum() + 1e-9 * out.energy
    assert float(loss) == pytest.approx(float(g[f"{name}_loss"]), rel=1e-10)
    loss.backward()
    got = np.array([float(length.grad), float(voltage.grad), float(phase.grad), float(freq.grad), float(energy.grad)])
    ref = g[f"{name}_grads"]
    assert np.allclose(got, ref, rtol=1e-8, atol=1e-12 * np.abs(ref).max()), (got, ref)
    dx, dx_ref = xin.grad.cpu().numpy(), g[f"{name}_dx"]
    assert np.allclose(dx[:, :6], dx_ref[:, :6], rtol=1e-9, atol=1e-12 * np.abs(dx_ref).max())


def test_cavity_gradients_vectorised_and_shared_beam():
    """k-scan style: (3,) voltages on one shared beam; d/dx sums over the batch, d/dV stays per entry."""
    import cheetah_amd as ca

    kw = {"dtype": torch.float64, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(0)
    x = (torch.randn(500, 7, **kw) * t([2e-4, 4e-6, 2e-4, 4e-6, 8e-6, 2e-3, 0.0]))
    x[:, 6] = 1.0
    x.requires_grad_(True)
    V = torch.nn.Parameter(t([1.0e7, 1.5e7, 2.0e7]))
    cav = ca.Cavity(length=t(1.0), voltage=V, phase=t(20.0), frequency=t(1.3e9), **kw)
    out = cav.track(ca.ParticleBeam(x, t(6e6), species=ca.Species("electron", **kw)))
    assert out.particles.shape == (3, 500, 7) and out.energy.shape == (3,)
    out.particles[..., 4].square().sum().backward()
    assert V.grad.shape == (3,) and torch.all(V.grad != 0) and x.grad.shape == (500, 7)
    # finite-difference check of d/dV on entry 1
    eps = 1.0
    vals = []
    for dv in (-eps, eps):
        with torch.no_grad():
            c2 = ca.Cavity(length=t(1.0), voltage=t([1.0e7, 1.5e7 + dv, 2.0e7]), phase=t(20.0), frequency=t(1.3e9), **kw)
            o = c2.track(ca.ParticleBeam(x.detach(), t(6e6), species=ca.Species("electron", **kw)))
            vals.append(float(o.particles[..., 4].square().sum()))
    fd = (vals[1] - vals[0]) / (2 * eps)
    assert float(V.grad[1]) == pytest.approx(fd, rel=1e-5)


def test_parameter_beam_gradients_flow():
    """tests/test_differentiable.py:58-75: mu / cov of an incoming ParameterBeam as Parameters."""
    import cheetah_amd as ca

    kw = {"dtype": torch.float64, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    seg = ca.Segment([ca.Drift(t(0.5), **kw), ca.Quadrupole(t(0.2), k1=t(4.2), **kw), ca.Drift(t(1.0), **kw),
                      ca.HorizontalCorrector(t(0.02), angle=t(1e-4), **kw)])
    beam = ca.ParameterBeam.from_parameters(sigma_x=t(2e-4), sigma_px=t(3e-6), energy=t(1e8), **kw)
    ref = seg.track(beam)
    beam.mu = torch.nn.Parameter(beam.mu.clone())
    beam.cov = torch.nn.Parameter(beam.cov.clone())
    out = seg.track(beam)
    assert out.mu.grad_fn is not None and out.cov.grad_fn is not None
    assert torch.allclose(out.mu, ref.mu, rtol=1e-13, atol=1e-20) and torch.allclose(out.cov, ref.cov, rtol=1e-12, atol=1e-30)
    out.sigma_x.backward()
    assert beam.cov.grad is not None and float(beam.cov.grad[0, 0]) > 0
