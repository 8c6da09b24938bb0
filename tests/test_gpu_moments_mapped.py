"""Gradient of a beam moment with respect to lattice settings when the particles carry no graph: the backward pass is 7x7
algebra on the INCOMING beam's moments (`chx_moments_mapped_bwd`: mu' = A mu + b, cov' = A C A^T) instead of two passes over
the tracked particles (`chx_moments_bwd` + the dR reduction of `chx_apply_affine7_bwd`). Both must give the same dR — the
particle path is autograd's definition (reference: particles @ tm.mT then the weighted statistics,
/root/reference/cheetah/accelerator/element.py:180-191, utils/statistics.py:4-62, tests/test_differentiable.py:10-32) —
and the particle path's dR reduction at 10^6 particles is pinned against a float64 torch reduction of the same dY, X."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ca():
    assert torch.cuda.is_available()
    import cheetah_amd

    cheetah_amd._lib.lib()
    return cheetah_amd


def _beam(ca, n, dt, weights=False, seed=3):
    kw = {"dtype": dt, "device": "cuda"}
    torch.manual_seed(seed)
    beam = ca.ParticleBeam.from_parameters(num_particles=n, mu_x=torch.tensor(2e-4, **kw), mu_py=torch.tensor(-3e-6, **kw),
                                           cov_xpx=torch.tensor(3e-10, **kw), **kw)
    if weights:
        beam.survival_probabilities = torch.rand(n, **kw)
    return beam


def _grads(ca, beam, dt, algebraic, names=("sigma_x", "mu_y", "sigma_p", "cov_xpx"), vector=False):
    from cheetah_amd import _ops

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    k1 = torch.nn.Parameter(t([3.142, -2.0, 0.5]) if vector else t(3.142))
    L = torch.nn.Parameter(t(0.2))
    ang = torch.nn.Parameter(t(2e-4))
    seg = ca.Segment([ca.Drift(t(1.0), **kw), ca.Quadrupole(L, k1=k1, **kw), ca.HorizontalCorrector(t(0.05), angle=ang, **kw),
                      ca.Drift(t(1.0), **kw), ca.Screen(is_active=True, name="scr", **kw)])
    out = seg.track(beam)
    read = seg.scr.get_read_beam()
    assert hasattr(out.particles, "_chx_lin") and hasattr(read.particles, "_chx_lin")
    if not algebraic:
        del read.particles._chx_lin                     # the particle-sized backward (Moments + Apply)
    res = {}
    for name in names:
        val = getattr(read, name)
        g = torch.autograd.grad(val.sum(), (k1, L, ang), retain_graph=True)
        res[name] = (val.detach(), [v.detach() for v in g])
    return res, _ops


@pytest.mark.parametrize("dt,weights,vector", [(torch.float64, False, False), (torch.float64, True, False),
                                               (torch.float64, True, True), (torch.float32, False, False)])
def test_algebraic_backward_equals_particle_backward(ca, dt, weights, vector):
    beam = _beam(ca, 50_000, dt, weights)
    alg, _ = _grads(ca, beam, dt, True, vector=vector)
    par, _ = _grads(ca, beam, dt, False, vector=vector)
    rtol = 1e-9 if dt == torch.float64 else 2e-4     # fp32: the particle path differentiates through rounded y = R x
    for name in alg:
        # the same statistic forward: the algebraic route reads the one-pass sums the particle pass of the [run | Screen] stretch left
        # (round 6, chx_lattice_screen.mom_partials), the particle route runs chx_moments over the rows — two summation orders
        ftol = 1e-12 if dt == torch.float64 else 2e-7
        assert torch.allclose(alg[name][0], par[name][0], rtol=ftol, atol=ftol * float(alg["sigma_x"][0].abs().max()) ** (2 if name.startswith("cov") else 1)), name
        # a setting the moment does not depend on has gradient 0 algebraically and rounding noise on the particle path:
        # errors are measured in units of the moment's largest sensitivity, every setting weighted with its own size
        sizes = [3.142, 0.2, 2e-4]
        scale = max(float(p.abs().max()) * sz for p, sz in zip(par[name][1], sizes))
        for a, p, sz in zip(alg[name][1], par[name][1], sizes):
            assert float((a - p).abs().max()) * sz <= rtol * scale, (name, a, p)


def test_incoming_moments_are_reduced_once_per_beam(ca):
    """The incoming beam's moments are memoised on its particle tensor: later steps of an optimisation loop launch no
    particle-sized kernel in the backward pass; an in-place edit of the beam invalidates the memo."""
    from cheetah_amd import _ops

    dt = torch.float32
    beam = _beam(ca, 20_000, dt)
    calls = {"n": 0}
    real = _ops._moments_raw

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    _ops._moments_raw = counting
    try:
        first, _ = _grads(ca, beam, dt, True, names=("sigma_x",))
        n_first = calls["n"]                      # forward reduction of y + the incoming moments
        again, _ = _grads(ca, beam, dt, True, names=("sigma_x",))
        # forward reduction only — and that one inside the C++ node (cheetah_amd._chxtorch MomentEntryMappedNode) when one beam
        # goes through one map: the incoming beam is not reduced again either way
        assert calls["n"] - n_first in (0, 1)
        assert torch.equal(first["sigma_x"][1][0], again["sigma_x"][1][0])
        with torch.no_grad():
            beam.particles[:, 0] *= 2.0           # version moves
        n0 = calls["n"]
        changed, _ = _grads(ca, beam, dt, True, names=("sigma_x",))
        assert calls["n"] - n0 in (1, 2)          # the incoming beam again (+ the forward reduction when Python launches it)
        assert float(changed["sigma_x"][0]) > 1.5 * float(first["sigma_x"][0])
    finally:
        _ops._moments_raw = real


def test_particle_gradients_keep_the_particle_path(ca):
    dt = torch.float64
    beam = _beam(ca, 5000, dt)
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    x = beam.particles.clone().requires_grad_(True)
    k1 = torch.nn.Parameter(t(1.5))
    seg = ca.Segment([ca.Drift(t(0.5), **kw), ca.Quadrupole(t(0.2), k1=k1, **kw), ca.Drift(t(0.5), **kw)])
    out = seg.track(ca.ParticleBeam(x, beam.energy, species=beam.species))
    assert not hasattr(out.particles, "_chx_lin")
    out.sigma_x.backward()
    assert x.grad is not None and k1.grad is not None and float(x.grad.abs().sum()) > 0


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_full_size_dR_reduction_against_float64_torch(ca, dt):
    """10^6 particles: dR[b] = sum_n dY_n x_n^T of chx_apply_affine7_bwd (fp64 accumulation) against the same reduction by
    torch in float64, and the algebraic dR of a sigma_x loss against both."""
    from cheetah_amd import _ops

    N = 1_000_000
    beam = _beam(ca, N, dt, weights=True, seed=11)
    x = _ops.aligned(beam.particles).reshape(1, N, 7)
    R = (torch.eye(7, dtype=dt, device="cuda") + 0.1 * torch.randn(7, 7, dtype=dt, device="cuda",
                                                                   generator=torch.Generator("cuda").manual_seed(2)))
    R[6] = 0
    R[6, 6] = 1
    R = R.reshape(1, 7, 7).requires_grad_(True)
    y = _ops.Apply.apply(x, R, 1)
    mom = _ops.Moments.apply(y, beam.survival_probabilities.reshape(1, N).contiguous(), 1)
    loss = mom[0, 8].sqrt() + 3.0 * mom[0, 4] + mom[0, 8 + 7]          # sigma_x, mu_y, cov_pxpx-ish mix
    dY, = torch.autograd.grad(loss, y, retain_graph=True)
    dR_kernel, = torch.autograd.grad(loss, R, retain_graph=True)
    dR_torch = torch.einsum("ni,nj->ij", dY[0].double(), x[0].double())
    scale = dR_torch.abs().max()
    assert float((dR_kernel[0].double() - dR_torch).abs().max() / scale) < (1e-6 if dt == torch.float32 else 1e-12)
    # algebraic: the same loss through MomentsMapped
    lin = _ops._LinearSource(beam.particles, x, R, (), 0)
    mom2 = _ops.MomentsMapped.apply(R, y.detach(), beam.survival_probabilities.reshape(1, N).contiguous(),
                                    (lin, beam.survival_probabilities), 1)
    loss2 = mom2[0, 8].sqrt() + 3.0 * mom2[0, 4] + mom2[0, 8 + 7]
    dR_alg, = torch.autograd.grad(loss2, R)
    assert float((dR_alg[0].double() - dR_torch).abs().max() / scale) < (3e-4 if dt == torch.float32 else 1e-10)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("E", [3, 16, 23])
def test_run_vjp_entry_equals_its_two_launches(dt, E):
    """`chx_run_vjp_entry` (ABI 9) through the C-ABI: the builders' VJP that forms dL/d(composed map) from the gradient of ONE entry
    of the tracked beam's moments itself, against the two calls it replaces — `chx_moment_entry_mapped_bwd` writing that cotangent
    as a tensor, `chx_run_vjp_masked` reading it: the same bits, for runs the fused kernel takes (E <= 16: one launch) and for a
    longer one (its fallback: the two launches behind the one entry point); every entry index, with and without the square root;
    a mask that leaves settings out."""
    import ctypes

    from cheetah_amd import _lib, _ops

    lib = _lib.lib()
    dev = "cuda"
    torch.manual_seed(E)
    kinds_py, params = [], []
    for e in range(E):
        if e % 3 == 1:
            kinds_py.append(_ops.KIND["quadrupole"])
            params.append([0.1 + 0.01 * e, 3.0 * (-1) ** e, 0.05 * (e % 2), 1e-4 * (e % 3), -5e-5])
        elif e % 3 == 2:
            kinds_py.append(_ops.KIND["hcor"] if "hcor" in _ops.KIND else _ops.KIND["horizontal_corrector"])
            params.append([0.05, 1e-4 * e])
        else:
            kinds_py.append(_ops.KIND["drift"])
            params.append([0.2 + 0.01 * e])
    tensors = [[torch.tensor(v, dtype=dt, device=dev) for v in p] for p in params]
    ptrs = (ctypes.c_void_p * (E * _ops.MAX_PARAMS))()
    for e, ts in enumerate(tensors):
        for k, t in enumerate(ts):
            ptrs[e * _ops.MAX_PARAMS + k] = t.data_ptr()
    kinds = (ctypes.c_int32 * E)(*kinds_py)
    energy = torch.tensor([1.2e8], dtype=dt, device=dev)
    mass, nq, code = 510998.95069, -1.0, _ops.dtype_code(dt)
    maps = torch.empty((E, 7, 7), dtype=dt, device=dev)
    C = torch.empty((7, 7), dtype=dt, device=dev)
    _ops.check(lib.chx_run_build_compose(kinds, ptrs, E, energy.data_ptr(), mass, nq, code, maps.data_ptr(), C.data_ptr(), _ops.stream_ptr()),
               "chx_run_build_compose")
    x = torch.randn(1, 4000, 7, dtype=dt, device=dev) * torch.tensor([2e-4, 3e-5, 1.5e-4, 2e-5, 1e-4, 1e-3, 0.0], dtype=dt, device=dev)
    x[..., 6] = 1.0
    w = (0.3 + 0.7 * torch.rand(1, 4000, dtype=dt, device=dev))
    mom_x = _ops._moments_raw(x, w, 1, 4000)
    y = x @ C.T
    mom_y = _ops._moments_raw(y.contiguous(), w, 1, 4000)
    need = (ctypes.c_uint16 * E)(*[(0xFFFF if e % 4 else 0x0002) for e in range(E)])       # (every fourth element: its slot 1 only)
    g = torch.tensor([0.7], dtype=dt, device=dev)
    ws_bytes = lib.chx_run_vjp_entry_workspace_bytes(E)
    assert ws_bytes == lib.chx_run_vjp_workspace_bytes(E) + 49 * 8
    for index, take_sqrt in ((8, 1), (8, 0), (2, 0), (3, 0), (14, 1), (9, 0), (28, 1), (19, 0)):
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        got = torch.full((E, _ops.MAX_PARAMS + 1), float("nan"), dtype=dt, device=dev)
        _ops.check(lib.chx_run_vjp_entry(kinds, ptrs, E, energy.data_ptr(), mass, nq, code, maps.data_ptr(), need, g.data_ptr(), mom_y.data_ptr(),
                                         index, take_sqrt, C.data_ptr(), mom_x.data_ptr(), got.data_ptr(), ws.data_ptr(), ws_bytes,
                                         _ops.stream_ptr()), "chx_run_vjp_entry")
        dC = torch.empty((7, 7), dtype=dt, device=dev)
        _ops.check(lib.chx_moment_entry_mapped_bwd(g.data_ptr(), mom_y.data_ptr(), index, take_sqrt, C.data_ptr(), mom_x.data_ptr(), 1, 1, 1, code,
                                                   dC.data_ptr(), 0, _ops.stream_ptr()), "chx_moment_entry_mapped_bwd")
        want = torch.full_like(got, float("nan"))
        ws2 = torch.empty(lib.chx_run_vjp_workspace_bytes(E), dtype=torch.uint8, device=dev)
        _ops.check(lib.chx_run_vjp_masked(kinds, ptrs, E, energy.data_ptr(), mass, nq, code, maps.data_ptr(), dC.data_ptr(), need, want.data_ptr(),
                                          ws2.data_ptr(), ws2.numel(), _ops.stream_ptr()), "chx_run_vjp_masked")
        assert torch.equal(got, want), (index, take_sqrt, (got - want).abs().max())
        assert torch.isfinite(got).all() and (float(got.abs().max()) > 0.0 or index == 28)     # (sigma_p does not depend on a linear map's settings)
    # argument checks: W / W2 have no gradient to a map; a missing workspace
    assert lib.chx_run_vjp_entry(kinds, ptrs, E, energy.data_ptr(), mass, nq, code, maps.data_ptr(), need, g.data_ptr(), mom_y.data_ptr(), 1, 0,
                                 C.data_ptr(), mom_x.data_ptr(), got.data_ptr(), ws.data_ptr(), ws_bytes, _ops.stream_ptr()) == -1
    assert lib.chx_run_vjp_entry(kinds, ptrs, E, energy.data_ptr(), mass, nq, code, maps.data_ptr(), need, g.data_ptr(), mom_y.data_ptr(), 8, 1,
                                 C.data_ptr(), mom_x.data_ptr(), got.data_ptr(), None, 0, _ops.stream_ptr()) == -5
