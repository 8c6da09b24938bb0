"""Gradients of the ParameterBeam transport (mu', cov') = (R mu, R cov R^T) through the HIP node `_ops.ParameterTrack`
(chx_parameter_track_bwd) against torch's own autograd of the reference expressions
(/root/reference/cheetah/accelerator/element.py:167-179, tests/test_differentiable.py:58-75), broadcast inputs included."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(dt, mu_shape, cov_shape, tm_shape, seed=0):
    g = torch.Generator("cuda").manual_seed(seed)
    mu = torch.randn(*mu_shape, 7, dtype=dt, device="cuda", generator=g)
    a = torch.randn(*cov_shape, 7, 7, dtype=dt, device="cuda", generator=g)
    cov = a @ a.mT                                             # symmetric like a covariance (the formulas do not need it)
    tm = torch.eye(7, dtype=dt, device="cuda") + 0.3 * torch.randn(*tm_shape, 7, 7, dtype=dt, device="cuda", generator=g)
    return [t.requires_grad_(True) for t in (mu, cov, tm)]


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
@pytest.mark.parametrize("shapes", [((), (), ()), ((), (), (5,)), ((5,), (5,), ()), ((3, 1), (1, 4), (3, 4)), ((4,), (), (3, 1))])
def test_parameter_track_backward_equals_autograd_of_the_matmuls(dt, shapes):
    from cheetah_amd import _ops

    mu, cov, tm = _inputs(dt, *shapes)
    w_mu = torch.randn(7, dtype=dt, device="cuda")
    w_cov = torch.randn(7, 7, dtype=dt, device="cuda")         # NOT symmetric: d_R needs both G R cov^T and G^T R cov

    def loss(m, c):
        return (m * w_mu).sum() + (c * w_cov).sum()

    mo, co = _ops.parameter_track(mu, cov, tm)
    got = torch.autograd.grad(loss(mo, co), (mu, cov, tm))
    mr, cr = (tm @ mu.unsqueeze(-1)).squeeze(-1), tm @ cov @ tm.mT
    # same forward values on the broadcast batch shape
    assert torch.allclose(mo, mr.expand(mo.shape), rtol=1e-12 if dt == torch.float64 else 1e-5, atol=0)
    assert torch.allclose(co, cr.expand(co.shape), rtol=1e-11 if dt == torch.float64 else 1e-4, atol=1e-12 if dt == torch.float64 else 1e-5)
    ref = torch.autograd.grad(loss(mr.expand(mo.shape), cr.expand(co.shape)), (mu, cov, tm))
    for a, b, name in zip(got, ref, ("mu", "cov", "tm")):
        assert a.shape == b.shape, name
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= (1e-12 if dt == torch.float64 else 2e-5) * scale, name


def test_only_the_wanted_gradients_are_formed():
    from cheetah_amd import _ops

    mu, cov, tm = _inputs(torch.float64, (), (), (6,))
    mu.requires_grad_(False)
    cov.requires_grad_(False)
    mo, co = _ops.parameter_track(mu, cov, tm)
    (g,) = torch.autograd.grad(co[..., 0, 0].sqrt().sum(), (tm,))
    ref_c = tm @ cov @ tm.mT
    (r,) = torch.autograd.grad(ref_c[..., 0, 0].sqrt().sum(), (tm,))
    assert torch.allclose(g, r, rtol=1e-12, atol=1e-14)


def test_graph_flags_follow_the_two_expressions():
    """mu' = R mu hangs on (mu, R), cov' = R cov R^T on (cov, R): an input that only feeds one of them must not put the other
    output into the graph (the reference's flags, tests/golden/grad_flags.json)."""
    from cheetah_amd import _ops

    mu, cov, tm = _inputs(torch.float64, (), (), ())
    for req in ((True, False, False), (False, True, False), (False, False, True)):
        for t, r in zip((mu, cov, tm), req):
            t.requires_grad_(r)
        mo, co = _ops.parameter_track(mu, cov, tm)
        assert mo.requires_grad == (req[0] or req[2]) and co.requires_grad == (req[1] or req[2]), req
