"""Gradients through SpaceChargeKick.track: Moments + CicDeposit + ScPoisson + ScGradient + ScGatherKick (the
backward kernels of csrc/chx_spacecharge_bwd.hip) against the reference's autograd results (tests/golden/sc_grad.npz,
fp64) and against the physics check of the reference's own tests (tests/test_space_charge_kick.py:202-261)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KW = {"dtype": torch.float64, "device": "cuda"}


def t(v):
    return torch.tensor(np.asarray(v), **KW)


@pytest.mark.parametrize("name", ["e50MeV", "e8MeV_anis"])
def test_space_charge_gradients_match_reference(golden, name):
    import cheetah_amd as ca

    g = golden("sc_grad.npz")
    L, E, ext, gx, gy, gz = g[f"{name}_meta"]
    length = torch.nn.Parameter(t(L))
    energy = t(E).requires_grad_(True)
    xin = t(g["x"]).requires_grad_(True)
    q = t(g["q"]).requires_grad_(True)
    W = t(g["W"])
    sc = ca.SpaceChargeKick(effect_length=length, grid_shape=(int(gx), int(gy), int(gz)), grid_extent_x=t(ext),
                            grid_extent_y=t(ext), grid_extent_tau=t(ext), **KW)
    beam = ca.ParticleBeam(xin, energy, particle_charges=q, species=ca.Species("electron", **KW))
    out = sc.track(beam)
    assert out.particles.grad_fn is not None
    ref_out = g[f"{name}_out"]
    assert np.allclose(out.particles.detach().cpu().numpy(), ref_out, rtol=1e-8, atol=1e-10 * np.abs(ref_out).max(axis=0))
    # the differentiable pipeline and the fused forward-only pipeline agree
    with torch.no_grad():
        fused = sc.track(ca.ParticleBeam(xin.detach(), energy.detach(), particle_charges=q.detach(),
                                         species=ca.Species("electron", **KW)))
    assert torch.allclose(fused.particles, out.particles.detach(), rtol=1e-9, atol=1e-14)
    loss = (out.particles * W).sum()
    assert float(loss.detach()) == pytest.approx(float(g[f"{name}_loss"]), rel=1e-9)
    loss.backward()
    got = np.array([float(length.grad), float(energy.grad)])
    ref = g[f"{name}_grads"]
    assert np.allclose(got, ref, rtol=1e-6), (got, ref)
    dx, dx_ref = xin.grad.cpu().numpy(), g[f"{name}_dx"]
    scale = np.abs(dx_ref).max(axis=0)
    assert np.allclose(dx, dx_ref, rtol=1e-6, atol=1e-8 * scale), np.abs(dx - dx_ref).max(axis=0) / scale
    dq, dq_ref = q.grad.cpu().numpy(), g[f"{name}_dq"]
    assert np.allclose(dq, dq_ref, rtol=1e-6, atol=1e-8 * np.abs(dq_ref).max())


def test_gradient_value_backward_ad():
    """tests/test_space_charge_kick.py:202-261: d(beam radius)/d(section length) of a cold uniform sphere against the
    analytic expansion rate, through three kicks and four drifts (default dtype fp32)."""
    import cheetah_amd as ca
    from scipy import constants
    from scipy.constants import physical_constants

    kw = {"dtype": torch.float32, "device": "cuda"}
    tt = lambda v: torch.tensor(v, **kw)  # noqa: E731
    R0 = tt(0.001)
    energy = tt(2.5e8)
    species = ca.Species("electron", **kw)
    gamma = energy / species.mass_eV
    beta = (1 - 1 / gamma**2).sqrt()
    torch.manual_seed(0)
    incoming = ca.ParticleBeam.uniform_3d_ellipsoid(
        num_particles=100_000, total_charge=tt(1e-8), energy=energy, radius_x=R0, radius_y=R0,
        radius_tau=R0 / gamma / beta, sigma_px=tt(1e-15), sigma_py=tt(1e-15), sigma_p=tt(1e-15), species=species, **kw)
    electron_radius = tt(physical_constants["classical electron radius"][0])
    kappa = 1 + (tt(2.0).sqrt() / 4) * (3 + 2 * tt(2.0).sqrt()).log()
    Nb = incoming.total_charge / constants.elementary_charge
    segment_length = torch.nn.Parameter((beta * gamma * kappa * (R0.pow(3) / (Nb * electron_radius)).sqrt()).detach())
    segment = ca.Segment(elements=[
        ca.Drift(segment_length / 6, **kw), ca.SpaceChargeKick(segment_length / 3, **kw),
        ca.Drift(segment_length / 3, **kw), ca.SpaceChargeKick(segment_length / 3, **kw),
        ca.Drift(segment_length / 3, **kw), ca.SpaceChargeKick(segment_length / 3, **kw),
        ca.Drift(segment_length / 6, **kw)])
    outgoing = segment.track(incoming)
    # the beam doubles in size over this length (forward check of the same test file, :19-71)
    assert float(outgoing.sigma_x.detach()) == pytest.approx(2 * float(incoming.sigma_x), rel=0.02)
    outgoing.sigma_x.backward()
    dradius_dlength = 5**0.5 * segment_length.grad
    expected = (Nb * electron_radius / R0).sqrt() / gamma
    assert float(dradius_dlength) == pytest.approx(float(expected), rel=0.1)


def test_gradients_on_a_grid_that_is_not_a_power_of_two():
    """Grids outside the pruned transforms take the dense hipFFT plans under autograd as well (ScPoissonDense): the forward
    values equal the forward-only path, and d loss / d (length, particles) agree with central differences of that path.
    (Against the reference's autograd: tests/test_gpu_grad_flags.py, `SpaceChargeKick@odd_grid`.)"""
    import cheetah_amd as ca

    torch.manual_seed(11)
    x = torch.randn(600, 7, **KW) * t([2e-4, 3e-5, 2e-4, 2e-5, 1e-4, 1e-3, 0.0])
    x[:, 6] = 1.0
    length = torch.nn.Parameter(t(0.1))
    sc = ca.SpaceChargeKick(length, grid_shape=(24, 20, 12), **KW)
    xin = x.clone().requires_grad_(True)
    out = sc.track(ca.ParticleBeam(xin, t(1e8), particle_charges=torch.full((600,), 2e-13, **KW), species=ca.Species("electron", **KW)))
    with torch.no_grad():
        plain = sc.track(ca.ParticleBeam(x, t(1e8), particle_charges=torch.full((600,), 2e-13, **KW), species=ca.Species("electron", **KW)))
    assert torch.allclose(out.particles.detach(), plain.particles, rtol=1e-10, atol=1e-16)
    w = torch.randn(600, 3, **KW)
    loss = ((out.particles[:, [1, 3, 5]] - xin[:, [1, 3, 5]]) * w).sum() * 1e9
    gl, gx = torch.autograd.grad(loss, (length, xin), retain_graph=True)

    def f(L, xv):
        with torch.no_grad():
            o = ca.SpaceChargeKick(t(L), grid_shape=(24, 20, 12), **KW).track(
                ca.ParticleBeam(xv, t(1e8), particle_charges=torch.full((600,), 2e-13, **KW), species=ca.Species("electron", **KW)))
            return float(((o.particles[:, [1, 3]] - xv[:, [1, 3]]) * w[:, :2]).sum() * 1e9)

    # momentum columns only for the differences (delta goes through p / p0 - 1: its round trip is noise at this step size)
    loss13 = ((out.particles[:, [1, 3]] - xin[:, [1, 3]]) * w[:, :2]).sum() * 1e9
    gl13, gx13 = torch.autograd.grad(loss13, (length, xin))
    h = 1e-6
    assert float(gl13) == pytest.approx((f(0.1 + h, x) - f(0.1 - h, x)) / (2 * h), rel=1e-6)
    for i, j in ((3, 0), (77, 2), (410, 4)):
        xp, xm = x.clone(), x.clone()
        xp[i, j] += 1e-9
        xm[i, j] -= 1e-9
        assert float(gx13[i, j]) == pytest.approx((f(0.1, xp) - f(0.1, xm)) / 2e-9, rel=2e-4, abs=1e-3 * float(gx13.abs().max()))
    assert torch.isfinite(gx).all() and torch.isfinite(gl)


def test_vectorised_space_charge_gradients_equal_separate_runs():
    """Two effect lengths and two beams as one vectorised call: every gradient equals the one of the separate runs."""
    import cheetah_amd as ca

    torch.manual_seed(2)
    N = 4000
    xs = torch.randn(2, N, 7, **KW) * t([3e-4, 2e-5, 2e-4, 3e-5, 1e-5, 1e-3, 0.0])
    xs[..., 6] = 1.0
    W = torch.randn(2, N, 7, **KW)
    lengths = [0.2, 0.35]
    q = torch.full((N,), 1e-12, **KW)

    def run(x, L):
        x = x.clone().requires_grad_(True)
        length = torch.nn.Parameter(t(L))
        energy = t(3e7).requires_grad_(True) if not isinstance(L, list) else t([3e7, 3e7]).requires_grad_(True)
        sc = ca.SpaceChargeKick(effect_length=length, grid_shape=(16, 16, 16), **KW)
        out = sc.track(ca.ParticleBeam(x, energy, particle_charges=q, species=ca.Species("electron", **KW)))
        return x, length, energy, out

    x2, l2, e2, out2 = run(xs, lengths)
    assert out2.particles.shape == (2, N, 7)
    (out2.particles * W).sum().backward()
    for b in range(2):
        x1, l1, e1, out1 = run(xs[b], lengths[b])
        assert torch.allclose(out1.particles, out2.particles[b].detach(), rtol=1e-10, atol=1e-16)
        (out1.particles * W[b]).sum().backward()
        assert float(l2.grad[b]) == pytest.approx(float(l1.grad), rel=1e-8)
        assert float(e2.grad[b]) == pytest.approx(float(e1.grad), rel=1e-7)
        scale = x1.grad.abs().max(dim=0).values
        assert torch.all((x2.grad[b] - x1.grad).abs() <= 1e-7 * scale + 1e-30)
