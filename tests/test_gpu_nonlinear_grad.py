"""Gradients through drift_kick_drift and second_order tracking: DkdTrack (chx_dkd_track_bwd), BuildTTensor
(chx_build_ttensor_vjp) and ApplySecondOrder (chx_apply_second_order_bwd) against the reference's autograd results
(tests/golden/nonlinear_grad.npz, fp64). Where the reference itself returns NaN (second order with kx2 == 0, see the
generator's docstring) the entry is checked against central finite differences of our own forward pass."""
import ast

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KW = {"dtype": torch.float64, "device": "cuda"}
NAMES = ["dkd_drift", "dkd_drift_lowE", "dkd_quad", "dkd_quad_defocus", "dkd_dipole", "dkd_dipole_entrance", "dkd_tdc",
         "so_drift", "so_quad", "so_quad_defocus", "so_dipole", "so_sextupole"]


def t(v):
    return torch.tensor(np.asarray(v), **KW)


def build(ca, g, name, values):
    cls = getattr(ca, str(g[f"{name}__class"]))
    opts = ast.literal_eval(str(g[f"{name}__opts"]))
    return cls(**values, **opts, tracking_method=str(g[f"{name}__method"]), **KW)


def loss_of(ca, g, name, values, energy, x, W):
    out = build(ca, g, name, values).track(ca.ParticleBeam(x, energy, species=ca.Species("electron", **KW)))
    return (out.particles * W).sum() + 1e-9 * out.energy.sum(), out


@pytest.mark.parametrize("name", NAMES)
def test_gradients_match_reference(golden, name):
    import cheetah_amd as ca

    g = golden("nonlinear_grad.npz")
    pnames = [str(p) for p in g[f"{name}__pnames"]]
    values = {k: torch.nn.Parameter(t(g[f"{name}__p__{k}"])) for k in pnames}
    energy = t(g[f"{name}__energy"]).requires_grad_(True)
    xin = t(g["x"]).requires_grad_(True)
    W = t(g["W"])
    loss, out = loss_of(ca, g, name, values, energy, xin, W)
    assert out.particles.grad_fn is not None
    assert np.allclose(out.particles.detach().cpu().numpy(), g[f"{name}__out"], rtol=1e-9, atol=1e-15)
    assert float(loss.detach()) == pytest.approx(float(g[f"{name}__loss"]), rel=1e-10)
    loss.backward()

    def fd(key, index=None):
        """central difference of our own forward with respect to one parameter entry"""
        base = {k: v.detach().clone() for k, v in values.items()}
        h = 1e-6 * max(1.0, abs(float(base[key].reshape(-1)[index or 0])))
        vals = []
        for sgn in (-1.0, 1.0):
            trial = {k: v.clone() for k, v in base.items()}
            trial[key].reshape(-1)[index or 0] += sgn * h
            with torch.no_grad():
                vals.append(float(loss_of(ca, g, name, trial, energy.detach(), xin.detach(), W)[0]))
        return (vals[1] - vals[0]) / (2 * h)

    for k in pnames:
        got, ref = values[k].grad.cpu().numpy(), g[f"{name}__g__{k}"]
        assert got.shape == ref.shape and np.all(np.isfinite(got)), (k, got)
        for idx in np.ndindex(ref.shape or (1,)):
            r = ref[idx] if ref.shape else float(ref)
            v = got[idx] if ref.shape else float(got)
            if np.isfinite(r):
                assert v == pytest.approx(r, rel=2e-7, abs=1e-13), (name, k, idx, v, r)
            else:  # the reference's own gradient is NaN here
                assert name.startswith("so_")
                assert v == pytest.approx(fd(k, idx[0] if ref.shape else None), rel=1e-5, abs=1e-9), (name, k)
    ge, ge_ref = float(energy.grad), float(g[f"{name}__g_energy"])
    if np.isfinite(ge_ref):
        assert ge == pytest.approx(ge_ref, rel=1e-6, abs=1e-16), (name, ge, ge_ref)
    dx, dx_ref = xin.grad.cpu().numpy(), g[f"{name}__dx"]
    assert np.all(np.isfinite(dx))
    if np.all(np.isfinite(dx_ref)):
        assert np.allclose(dx[:, :6], dx_ref[:, :6], rtol=1e-7, atol=1e-11 * np.abs(dx_ref).max()), name


def test_gradients_vectorised_shared_beam_fp32():
    """A k1 scan (4,) over one shared fp32 beam: d/dk1 stays per entry, d/dx sums over the scan; both methods give
    finite gradients that agree with each other to first order (thin, weak quadrupole)."""
    import cheetah_amd as ca

    kw = {"dtype": torch.float32, "device": "cuda"}
    torch.manual_seed(3)
    x = torch.randn(4000, 7, **kw) * torch.tensor([3e-4, 5e-5, 3e-4, 5e-5, 2e-5, 1e-4, 0.0], **kw)
    x[:, 6] = 1.0
    grads = {}
    for method in ("drift_kick_drift", "second_order", "linear"):
        xin = x.clone().requires_grad_(True)
        k1 = torch.nn.Parameter(torch.tensor([1.0, -2.0, 0.5, 3.0], **kw))
        quad = ca.Quadrupole(length=torch.tensor(0.1, **kw), k1=k1, tracking_method=method, **kw)
        out = quad.track(ca.ParticleBeam(xin, torch.tensor(1e8, **kw), species=ca.Species("electron", **kw)))
        assert out.particles.shape == (4, 4000, 7)
        out.particles[..., 1].square().sum().backward()
        assert k1.grad.shape == (4,) and xin.grad.shape == (4000, 7)
        assert torch.isfinite(k1.grad).all() and torch.isfinite(xin.grad).all()
        grads[method] = (k1.grad.cpu().numpy(), xin.grad.cpu().numpy())
    for method in ("drift_kick_drift", "second_order"):
        assert np.allclose(grads[method][0], grads["linear"][0], rtol=2e-2)
        assert np.allclose(grads[method][1][:, :4], grads["linear"][1][:, :4], rtol=5e-2,
                           atol=1e-3 * np.abs(grads["linear"][1]).max())  # chromatic terms only exist beyond first order


def test_segment_optimisation_step_through_nonlinear_elements():
    """A gradient step on a quadrupole strength through a mixed-method lattice reduces the loss."""
    import cheetah_amd as ca

    torch.manual_seed(5)
    x = torch.randn(3000, 7, **KW) * t([3e-4, 5e-5, 3e-4, 5e-5, 2e-5, 1e-3, 0.0])
    x[:, 6] = 1.0
    k1 = torch.nn.Parameter(t(2.0))
    seg = ca.Segment([
        ca.Drift(length=t(0.5), tracking_method="drift_kick_drift", **KW),
        ca.Quadrupole(length=t(0.2), k1=k1, tracking_method="drift_kick_drift", num_steps=3, **KW),
        ca.Drift(length=t(0.5), tracking_method="second_order", **KW),
        ca.Sextupole(length=t(0.1), k2=t(5.0), **KW),
        ca.Drift(length=t(1.0), **KW),
    ])
    beam = ca.ParticleBeam(x, t(1e8), species=ca.Species("electron", **KW))

    def loss_fn():
        return seg.track(beam).sigma_x

    l0 = loss_fn()
    l0.backward()
    assert torch.isfinite(k1.grad) and k1.grad != 0
    with torch.no_grad():
        k1 -= 0.5 * torch.sign(k1.grad)
    assert float(loss_fn()) < float(l0)


@pytest.mark.parametrize("method", ["drift_kick_drift", "second_order"])
def test_vectorised_dipole_gradients_equal_separate_runs(method):
    """Angles (3,) x per-row beams and energies: the vectorised backward equals three separate ones."""
    import cheetah_amd as ca

    torch.manual_seed(8)
    N = 2000
    xs = torch.randn(3, N, 7, **KW) * t([3e-4, 5e-5, 3e-4, 5e-5, 2e-5, 1e-3, 0.0])
    xs[..., 6] = 1.0
    W = torch.randn(3, N, 7, **KW)
    angles, energies = [0.1, 0.2, -0.15], [5e7, 6e7, 8e7]

    def run(x, angle, energy):
        x = x.clone().requires_grad_(True)
        a = torch.nn.Parameter(t(angle))
        e = t(energy).requires_grad_(True)
        length = torch.nn.Parameter(t(0.5))
        dip = ca.Dipole(length=length, angle=a, dipole_e1=t(0.05), dipole_e2=t(0.02), tracking_method=method, **KW)
        out = dip.track(ca.ParticleBeam(x, e, species=ca.Species("electron", **KW)))
        return x, a, e, length, out

    xv, av, ev, lv, outv = run(xs, angles, energies)
    assert outv.particles.shape == (3, N, 7)
    (outv.particles * W).sum().backward()
    length_total = 0.0
    for b in range(3):
        x1, a1, e1, l1, out1 = run(xs[b], angles[b], energies[b])
        assert torch.allclose(out1.particles, outv.particles[b].detach(), rtol=1e-12, atol=1e-18)
        (out1.particles * W[b]).sum().backward()
        assert float(av.grad[b]) == pytest.approx(float(a1.grad), rel=1e-9)
        assert float(ev.grad[b]) == pytest.approx(float(e1.grad), rel=1e-8, abs=1e-20)
        assert torch.allclose(xv.grad[b], x1.grad, rtol=1e-9, atol=1e-12 * float(x1.grad.abs().max()))
        length_total += float(l1.grad)
    assert float(lv.grad) == pytest.approx(length_total, rel=1e-9)      # a shared parameter collects all rows


@pytest.mark.parametrize("kind,params,extra", [
    ("drift", (1.0,), ()), ("quadrupole", (0.2, 4.2, 0.1, 1e-4, -1e-4), (10,)),
    ("dipole", (0.5, 0.35, 0.17, 0.17, 0.1, 0.5, 0.5, 0.05, 0.05), (1, 3)), ("tdc", (1.0, 1e7, 0.2, 1e9, 0.0, 0.0, 0.0), ())])
def test_storage_precision_dkd_stays_within_float32_rounding_of_the_float64_map(kind, params, extra):
    """`Element.dkd_precision = "storage"` (chx_dkd_track_p): float32 beams evaluated in float32 like the reference's Bmad-X
    tensor code (/root/reference/cheetah/utils/bmadx.py runs in the beam dtype). The float64 kernel on the same input is the
    yardstick: errors of a few float32 ulps of each coordinate's scale; the default (float64 arithmetic) is at the rounding of the
    store."""
    import cheetah_amd  # noqa: F401
    from cheetah_amd import _ops

    torch.manual_seed(0)
    N = 50_000
    x = torch.randn(N, 7, device="cuda") * torch.tensor([2e-4, 4e-6, 2e-4, 4e-6, 8e-6, 2e-3, 0.0], device="cuda")
    x[:, 6] = 1.0
    E = torch.tensor(1e8, device="cuda")
    k = _ops.DKD_KIND[kind]
    p32 = torch.tensor([list(params)], device="cuda")
    ref = _ops.dkd_track(k, x.double(), p32.double(), torch.Size(()), E.double(), 510998.95069, -1.0, *extra)[0]
    scale = ref.abs().max(dim=0).values[:6]
    errs = {}
    for flag in (False, True):
        got, e_out = _ops.dkd_track(k, x, p32, torch.Size(()), E, 510998.95069, -1.0, *extra, storage_precision=flag)
        assert got.dtype == torch.float32 and torch.all(got[:, 6] == 1)
        errs[flag] = (got.double() - ref).abs().max(dim=0).values[:6] / scale
    assert float(errs[False].max()) < 2e-7           # one rounding of the stored float32
    # float32 arithmetic through the whole map: a few ulps on the transverse coordinates; tau and delta go through the
    # Cheetah <-> Bmad conversion (bmadx.py:7-56), whose energy differences cancel five digits of a float32 — the price of the
    # reference's own arithmetic width (measured: delta 2.4e-5, tau up to 2.6e-4 of the coordinate's scale)
    # the dipole body (dipole.py:246-336: asin, atan2 and differences of nearly equal lengths) loses more: 1.3e-4 on px
    bound = torch.tensor([2e-4, 5e-4, 2e-4, 5e-4, 3e-3, 2e-4] if kind == "dipole" else [5e-6, 5e-6, 5e-6, 5e-6, 1e-3, 1e-4],
                         dtype=torch.float64, device="cuda")
    assert torch.all(errs[True] < bound), errs[True]


def test_dkd_precision_attribute_reaches_the_kernel():
    import cheetah_amd as ca

    kw = {"dtype": torch.float32, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    beam = ca.ParticleBeam.from_parameters(num_particles=20_000, **kw)
    quad = ca.Quadrupole(t(0.2), k1=t(4.2), tracking_method="drift_kick_drift", **kw)
    a = quad.track(beam).particles
    quad.dkd_precision = "storage"
    b = quad.track(beam).particles
    scale = a.abs().max(dim=0).values[:6]
    assert not torch.equal(a, b) and torch.all((a - b).abs().max(dim=0).values[:6] <= 2e-3 * scale)
    quad.dkd_precision = "half"
    with pytest.raises(ValueError):
        quad.track(beam)


@pytest.mark.parametrize("kind,params,extra", [
    ("drift", (1.0,), ()), ("quadrupole", (0.2, 4.2, 0.1, 1e-4, -1e-4), (10,)), ("quadrupole", (0.2, -4.2, 0.0, 0.0, 0.0), (1,)),
    ("quadrupole", (1.0, 10.0, 0.5, 0.01, -0.02), (10,)), ("quadrupole", (0.3, 0.0, 0.0, 0.0, 0.0), (3,))])
def test_mixed_precision_dkd_is_float64_grade_in_tau_and_delta(kind, params, extra):
    """`Element.dkd_precision = "mixed"`, the default of float32 beams (dkd_mixed_kernel): the longitudinal pair in float64, the
    transverse map in float32. Yardstick: the float64 kernel on the same float32 input (itself pinned to Bmad-X at 1e-14,
    tests/test_nonlinear.py). tau and delta must come out at the rounding of the float32 store like the float64 arithmetic
    does; x, px, y, py within a few float32 ulps of the coordinate's scale."""
    import cheetah_amd  # noqa: F401
    from cheetah_amd import _ops

    torch.manual_seed(0)
    N = 50_000
    x = torch.randn(N, 7, device="cuda") * torch.tensor([2e-4, 4e-6, 2e-4, 4e-6, 8e-6, 2e-3, 0.0], device="cuda")
    x[:, 6] = 1.0
    E = torch.tensor(1e8, device="cuda")
    k = _ops.DKD_KIND[kind]
    p32 = torch.tensor([list(params)], device="cuda")
    ref = _ops.dkd_track(k, x.double(), p32.double(), torch.Size(()), E.double(), 510998.95069, -1.0, *extra)[0]
    scale = ref.abs().max(dim=0).values[:6]
    got, e_mixed = _ops.dkd_track(k, x, p32, torch.Size(()), E, 510998.95069, -1.0, *extra, storage_precision=2)
    dbl, e_double = _ops.dkd_track(k, x, p32, torch.Size(()), E, 510998.95069, -1.0, *extra, storage_precision=0)
    assert got.dtype == torch.float32 and torch.all(got[:, 6] == 1) and torch.equal(e_mixed, e_double)
    err = (got.double() - ref).abs().max(dim=0).values[:6] / scale
    print(f"\nmixed {kind} {params}: " + " ".join(f"{float(v):.1e}" for v in err))
    # MEASURED (units of the coordinate's scale; the float64 arithmetic gives 6e-8 = the rounding of the float32 store):
    #   drift                                   x, y 3.5e-8, tau 5.1e-8, delta 4.4e-14
    #   quadrupole, one step, on the axis       x ... py 1.8e-7, tau 5.0e-8, delta 4.4e-14
    #   quadrupole, three steps, k1 = 0         x, y 1.1e-7, tau 5.0e-8
    #   a quadrupole with a misalignment (here half a beam size, and the 100 beam sizes of the reference's Bmad-X test element)
    #   is evaluated in float64 altogether: the shifted coordinate's float32 steps and path-length terms would cost 1e-6
    # Bounds 4x measured.
    shifted = kind == "quadrupole" and (params[3] != 0.0 or params[4] != 0.0)
    assert float(err[5]) < 2e-13, err
    assert float(err[4]) < 2.4e-7, err
    assert float(err[:4].max()) < (2.4e-7 if shifted else 7.2e-7), err
    if shifted:
        assert torch.equal(got, dbl)
