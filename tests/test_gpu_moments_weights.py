"""Gradient of the beam moments with respect to the survival weights (and the particles) from ONE kernel pass,
`chx_moments_bwd_w`, against autograd through the tensor expressions of the reference's weighted statistics
(/root/reference/cheetah/utils/statistics.py:4-62, /root/reference/cheetah/particles/particle_beam.py:1699-1717)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _torch_moments(x, w):
    """[W, W2, mu(6), cov upper triangle (21)] in float64 as tensor expressions."""
    xd, wd = x.to(torch.float64)[..., :6], w.to(torch.float64)
    W, W2 = wd.sum(dim=-1), (wd * wd).sum(dim=-1)
    mu = (wd.unsqueeze(-1) * xd).sum(dim=-2) / W.unsqueeze(-1)
    c = xd - mu.unsqueeze(-2)
    cov = torch.einsum("bn,bni,bnj->bij", wd, c, c) / (W - W2 / W).reshape(-1, 1, 1)
    iu = torch.triu_indices(6, 6, device=x.device)
    return torch.cat([W.unsqueeze(-1), W2.unsqueeze(-1), mu, cov[:, iu[0], iu[1]]], dim=-1)


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
@pytest.mark.parametrize("shapes", [((1, 5000), (1, 5000)), ((3, 4000), (3, 4000)), ((1, 4000), (3, 4000)), ((3, 4000), (1, 4000))])
def test_weight_and_particle_gradients(dt, shapes):
    import cheetah_amd  # noqa: F401
    from cheetah_amd import _ops

    (Bx, N), (Bw, _) = shapes
    torch.manual_seed(3)
    kw = {"dtype": dt, "device": "cuda"}
    x = (torch.randn(Bx, N, 7, **kw) * torch.tensor([1e-3, 1e-5, 2e-3, 1e-5, 1e-4, 1e-3, 0.0], **kw)
         + torch.tensor([2e-4, 0, -1e-3, 0, 0, 1e-3, 1.0], **kw)).requires_grad_(True)
    w = torch.rand(Bw, N, **kw).requires_grad_(True)
    B = max(Bx, Bw)
    coef = torch.randn(B, 29, dtype=torch.float64, device="cuda") * torch.tensor(
        [1e-3, 1e-3] + [1e3] * 6 + [1e6] * 21, dtype=torch.float64, device="cuda")
    out = _ops.moments(x if Bx > 1 else x[0], w if Bw > 1 else w[0])
    out = out.reshape(B, 29)
    ref = _torch_moments(x.expand(B, N, 7), w.expand(B, N))
    assert torch.allclose(out, ref, rtol=1e-9 if dt == torch.float64 else 1e-5, atol=1e-30)
    gx, gw = torch.autograd.grad((out * coef).sum(), (x, w))
    rx, rw = torch.autograd.grad((ref * coef).sum(), (x, w))
    tol = 1e-9 if dt == torch.float64 else 2e-4
    assert float((gw - rw).abs().max() / rw.abs().max()) < tol
    for j in range(6):
        assert float((gx[..., j] - rx[..., j]).abs().max() / rx[..., j].abs().max()) < tol, j


def test_aperture_upstream_of_a_moment_is_differentiable_through_the_kernel():
    """A survival array that carries a graph reaches the moments through Moments (no eager fallback)."""
    import cheetah_amd as ca
    from cheetah_amd import _ops

    kw = {"dtype": torch.float64, "device": "cuda"}
    torch.manual_seed(1)
    beam = ca.ParticleBeam.from_parameters(num_particles=20_000, **kw)
    logits = torch.randn(20_000, **kw, requires_grad=True)
    surv = torch.sigmoid(logits)
    b2 = ca.ParticleBeam(beam.particles, beam.energy, particle_charges=beam.particle_charges, survival_probabilities=surv,
                         species=beam.species)
    calls = []
    orig = _ops.Moments.apply
    _ops.Moments.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        s = b2.sigma_x
    finally:
        _ops.Moments.apply = orig
    assert calls
    s.backward()
    ref = _torch_moments(beam.particles.reshape(1, -1, 7), torch.sigmoid(logits.detach().clone().requires_grad_(True)).reshape(1, -1))
    assert logits.grad is not None and torch.isfinite(logits.grad).all() and float(logits.grad.abs().sum()) > 0
    assert float(s) == pytest.approx(float(ref[0, 8].sqrt()), rel=1e-10)
