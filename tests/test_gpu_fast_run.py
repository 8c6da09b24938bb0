"""The persistent device plan of a run of scalar-parameter elements (`chx_run_track`, `Segment._run_apply_fast`): one C
call per merged `Segment.track`, the device decides whether the stored map is still valid. Results must be bit-identical
to the general path (per-element maps -> chx_compose_maps -> chx_apply_affine7) and follow every kind of change of the
settings (in-place edit, re-assignment, dtype move, vectorised or trainable settings falling back to the general path).
Reference behaviour: /root/reference/cheetah/accelerator/segment.py:534-574, utils/cache.py:29-52."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ca():
    assert torch.cuda.is_available()
    import cheetah_amd

    cheetah_amd._lib.lib()
    return cheetah_amd


def lattice(ca, dt):
    t = lambda v: torch.tensor(v, dtype=dt, device="cuda")  # noqa: E731
    kw = {"dtype": dt, "device": "cuda"}
    return ca.Segment([
        ca.Marker(**kw), ca.Drift(t(0.175), **kw), ca.Quadrupole(t(0.122), k1=t(8.2), name="q1", **kw), ca.Drift(t(0.428), **kw),
        ca.Quadrupole(t(0.122), k1=t(-14.3), tilt=t(0.01), misalignment=t([1e-4, -2e-4]), name="q2", **kw),
        ca.VerticalCorrector(t(0.02), angle=t(9e-5), name="cv", **kw), ca.Dipole(t(0.3), angle=t(0.05), dipole_e1=t(0.01), name="d", **kw),
        ca.HorizontalCorrector(t(0.02), angle=t(-1e-4), name="ch", **kw), ca.Solenoid(t(0.1), k=t(0.3), **kw),
        ca.BPM(**kw), ca.Drift(t(0.45), **kw), ca.Screen(name="scr", **kw),
    ])


def general_path(ca, seg, beam):
    """Per-element maps composed and applied without the persistent plan."""
    from cheetah_amd import _ops

    # identity elements (Marker, inactive BPM / Screen) are left out of the product, as Segment does: the association of the
    # fp64 products — and with it the last bit of an fp64 map — depends on the number of factors
    maps = [e.first_order_transfer_map(beam.energy, beam.species) for e in seg.elements
            if e._chx_kind != _ops.KIND["identity"]]
    tm = _ops.compose_maps(maps, (), maps[0].dtype, maps[0].device)
    return _ops.apply_map(beam.particles, tm)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_fast_run_is_bit_identical_and_follows_setting_changes(ca, dt):
    seg = lattice(ca, dt)
    beam = ca.ParticleBeam.from_parameters(num_particles=10_000, dtype=dt, device="cuda")
    out = seg.track(beam)
    run = seg._plan()[0][1]
    assert run.fast is not None and run.fast.ok, "the all-scalar lattice must take the persistent plan"
    assert torch.equal(out.particles, general_path(ca, seg, beam))
    assert torch.equal(seg.track(beam).particles, out.particles)            # unchanged settings: stored map reused
    # in-place edit of a setting: no host-side counter is consulted, the device notices the new value
    seg.q1.k1.add_(1.5)
    out2 = seg.track(beam)
    assert not torch.equal(out2.particles, out.particles)
    assert torch.equal(out2.particles, general_path(ca, seg, beam))
    seg.q2.misalignment[0] = 3e-4
    assert torch.equal(seg.track(beam).particles, general_path(ca, seg, beam))
    # re-assignment (new tensor, new address): the epoch moves, the plan re-reads that element and patches its pointers
    epoch = run.fast.epoch
    seg.cv.angle = torch.tensor(-2e-5, dtype=dt, device="cuda")
    out3 = seg.track(beam)
    assert seg._plan()[0][1].fast.epoch != epoch and seg._plan()[0][1].fast.ok
    assert torch.equal(out3.particles, general_path(ca, seg, beam))
    # a control loop: new tensors for several settings every step, old ones freed in between
    for step in range(20):
        seg.q1.k1 = torch.tensor(1.0 + step, dtype=dt, device="cuda")
        seg.ch.angle = torch.tensor(1e-5 * step, dtype=dt, device="cuda")
        seg.q2.misalignment = torch.tensor([1e-5 * step, -2e-5], dtype=dt, device="cuda")
        assert torch.equal(seg.track(beam).particles, general_path(ca, seg, beam))
    # another beam energy / species: part of the device-side comparison
    beam2 = ca.ParticleBeam(beam.particles, torch.tensor(2.3e9, dtype=dt, device="cuda"), species=ca.Species("proton", dtype=dt, device="cuda"))
    assert torch.equal(seg.track(beam2).particles, general_path(ca, seg, beam2))
    assert torch.equal(seg.track(beam).particles, general_path(ca, seg, beam))
    # s and the other beam attributes as in the general path
    assert float(out3.s) == pytest.approx(float(seg.length), rel=1e-6)


def test_fast_run_falls_back_where_it_must(ca):
    dt = torch.float32
    seg = lattice(ca, dt)
    beam = ca.ParticleBeam.from_parameters(num_particles=4096, dtype=dt, device="cuda")
    seg.track(beam)
    # vectorised setting -> general path, vectorised result
    seg.q1.k1 = torch.linspace(-5, 5, 3, dtype=dt, device="cuda")
    out = seg.track(beam)
    assert out.particles.shape == (3, 4096, 7)
    assert not seg._plan()[0][1].fast.ok
    # trainable setting -> general path with gradients
    seg.q1.k1 = torch.nn.Parameter(torch.tensor(4.2, dtype=dt, device="cuda"))
    out = seg.track(beam)
    out.particles[:, 0].square().mean().backward()
    assert seg.q1.k1.grad is not None and float(seg.q1.k1.grad.abs()) > 0
    # a buffer switched to requires_grad in place moves no counter: still found
    seg = lattice(ca, dt)     # (a registered Parameter cannot be re-assigned a plain tensor: fresh lattice)
    seg.track(beam)
    assert seg._plan()[0][1].fast.ok
    seg.q2.k1.requires_grad_(True)
    out = seg.track(beam)
    assert out.particles.requires_grad
    seg.q2.k1.requires_grad_(False)
    with torch.no_grad():
        assert not seg.track(beam).particles.requires_grad
    # particles that require grad
    b2 = ca.ParticleBeam(beam.particles.clone().requires_grad_(True), beam.energy, species=beam.species)
    assert seg.track(b2).particles.requires_grad
    # the lattice moved to another dtype after tracking
    seg64 = seg.double()
    beam64 = ca.ParticleBeam(beam.particles.double(), beam.energy.double(), species=ca.Species("electron", dtype=torch.float64, device="cuda"))
    o64 = seg64.track(beam64)
    assert o64.particles.dtype == torch.float64
    assert torch.equal(o64.particles, general_path(ca, seg64, beam64))
    # changing the element list
    seg64.elements.append(ca.Drift(torch.tensor(1.0, dtype=torch.float64, device="cuda"), dtype=torch.float64, device="cuda"))
    o2 = seg64.track(beam64)
    assert torch.equal(o2.particles, general_path(ca, seg64, beam64)) and not torch.equal(o2.particles, o64.particles)
    # an active element in the middle splits the lattice into two runs, each with its own plan
    seg64.scr.is_active = True
    o3 = seg64.track(beam64)
    plan = seg64._plan()
    assert [k for k, _ in plan] == ["run", "element", "run"] and all(item.fast.ok for k, item in plan if k == "run")
    # two passes (each rounding the coordinates once) instead of one merged pass: equal to rounding, not bit for bit
    assert torch.allclose(o3.particles, o2.particles, rtol=1e-12, atol=1e-18)


def test_c_abi_run_map_against_build_and_compose(ca):
    """chx_run_map through the C-ABI: NaN-filled state -> map; same call again -> unchanged; a changed value -> new map
    equal to chx_build_rmatrix_scalars + chx_compose_maps."""
    import ctypes

    from cheetah_amd import _lib, _ops

    lib = _lib.lib()
    dt = torch.float64
    vals = [torch.tensor(v, dtype=dt, device="cuda") for v in (0.3, 0.2, 4.2, 0.0, 0.0, 0.0, 0.8)]
    kinds = [_ops.KIND["drift"], _ops.KIND["quadrupole"], _ops.KIND["drift"]]
    rows = [[vals[0]], [vals[1], vals[2], vals[3], vals[4], vals[5]], [vals[6]]]
    ptrs = []
    for r in rows:
        ptrs += [t.data_ptr() for t in r] + [None] * (_ops.MAX_PARAMS - len(r))
    E = 3
    energy = torch.tensor(1e8, dtype=dt, device="cuda")
    nbytes = lib.chx_run_state_bytes(E)
    assert nbytes > 0 and lib.chx_run_state_bytes(0) == 0 and lib.chx_run_state_bytes(193) == 0
    state = torch.full((nbytes // 8,), float("nan"), dtype=torch.float64, device="cuda")
    R_addr = ctypes.c_void_p()
    karr, parr = (ctypes.c_int32 * E)(*kinds), (ctypes.c_void_p * (E * _ops.MAX_PARAMS))(*ptrs)

    def run():
        _ops.check(lib.chx_run_map(karr, parr, E, energy.data_ptr(), 510998.95069, -1.0, 1, state.data_ptr(), nbytes,
                                   ctypes.byref(R_addr), None, None, _ops.stream_ptr()), "chx_run_map")
        off = (R_addr.value - state.data_ptr()) // 8
        return state[off:off + 49].reshape(7, 7).clone()

    def two_call():
        maps = torch.empty((E, 7, 7), dtype=dt, device="cuda")
        _ops.check(lib.chx_build_rmatrix_scalars(karr, parr, E, energy.data_ptr(), 510998.95069, -1.0, 1, maps.data_ptr(),
                                                 _ops.stream_ptr()), "build")
        return _ops.compose_maps([maps[0], maps[1], maps[2]], (), dt, maps.device)

    R1 = run()
    assert torch.equal(R1, two_call())
    assert torch.equal(run(), R1)
    vals[2].fill_(-3.3)
    R2 = run()
    assert not torch.equal(R2, R1) and torch.equal(R2, two_call())
    assert lib.chx_run_map(karr, parr, E, energy.data_ptr(), 510998.95069, -1.0, 1, state.data_ptr(), 16, None, None, None, None) == -5
    assert lib.chx_run_map(karr, parr, 0, energy.data_ptr(), 510998.95069, -1.0, 1, state.data_ptr(), nbytes, None, None, None, None) == -1


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_kick_then_run_in_one_pass_is_bit_identical(ca, dt):
    """[SpaceChargeKick, run of linear elements] inside a Segment: the run's map is applied in the kick's particle kernel
    (`chx_sc_kick` with post_map). Must equal tracking the kick and the elements one after the other, bit for bit."""
    t = lambda v: torch.tensor(v, dtype=dt, device="cuda")  # noqa: E731
    kw = {"dtype": dt, "device": "cuda"}
    torch.manual_seed(3)
    beam = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=70_000, total_charge=t(1e-9), energy=t(2.5e8), radius_x=t(1e-3),
                                                radius_y=t(1e-3), radius_tau=t(1e-4), sigma_px=t(1e-5), sigma_py=t(1e-5),
                                                sigma_p=t(1e-5), **kw)
    els = [ca.Drift(t(0.1), **kw), ca.SpaceChargeKick(t(0.2), grid_shape=(32, 32, 32), **kw), ca.Drift(t(0.1), **kw),
           ca.Quadrupole(t(0.1), k1=t(4.2), **kw), ca.Drift(t(0.1), **kw), ca.SpaceChargeKick(t(0.2), grid_shape=(32, 32, 32), **kw),
           ca.Marker(**kw), ca.Drift(t(0.3), **kw)]
    seg = ca.Segment(els)
    out = seg.track(beam)
    plan = seg._plan()
    assert [k for k, _ in plan] == ["run", "element", "run", "element", "run"]
    assert all(item.fast is not None and item.fast.ok for k, item in plan[2::2] if k == "run")
    # reference order: one element after the other through their public track methods, maps merged per run like Segment does
    from cheetah_amd import _ops

    b = beam
    for kind, item in plan:
        if kind == "run":
            maps = [e.first_order_transfer_map(b.energy, b.species) for e in item.elements if e._chx_kind != _ops.KIND["identity"]]
            tm = _ops.compose_maps(maps, (), dt, maps[0].device) if len(maps) > 1 else maps[0]
            b = ca.ParticleBeam(_ops.apply_map(b.particles, tm), b.energy, particle_charges=b.particle_charges,
                                survival_probabilities=b.survival_probabilities, species=b.species)
        else:
            b = item.track(b)
    # two evaluations of a kick differ in the last bits (the order of the LDS atomics of the deposit is not fixed), so the
    # end-to-end comparison is to rounding; the bit-exactness of the fused map is checked on a fixed field below
    scale = b.particles.abs().amax(dim=0).clamp_min(1e-30)
    assert float(((out.particles - b.particles).abs() / scale).max()) < (2e-6 if dt == torch.float32 else 1e-10)
    assert float(out.s) == pytest.approx(float(seg.length), rel=1e-6)
    # chx_sc_gather_kick_mapped == chx_sc_gather_kick followed by chx_apply_affine7, bit for bit (deterministic inputs)
    from cheetah_amd import _lib

    lib = _lib.lib()
    g3 = (16, 16, 16)
    N = 5000
    x = (torch.randn(1, N, 7, dtype=dt, device="cuda") * 3e-4).contiguous()
    x[..., 6] = 1
    F = torch.randn(1, 16 * 16 * 16, 4, dtype=dt, device="cuda") * 1e3
    half = torch.tensor([[1e-3, 1e-3, 1e-3]], dtype=dt, device="cuda")
    cell = half * 2 / 16
    energy = torch.tensor([2.5e8], dtype=dt, device="cuda")
    dtk = torch.tensor([1e-9], dtype=dt, device="cuda")
    R = (torch.eye(7, dtype=dt, device="cuda") + 0.05 * torch.randn(7, 7, dtype=dt, device="cuda")).reshape(1, 7, 7).contiguous()
    R[0, 6] = 0
    R[0, 6, 6] = 1
    b3 = _ops._bins3(g3)
    code = _ops.dtype_code(dt)
    kicked, both, fused = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    _ops.check(lib.chx_sc_gather_kick(x.data_ptr(), F.data_ptr(), half.data_ptr(), cell.data_ptr(), energy.data_ptr(), dtk.data_ptr(),
                                      510998.95069, 1, 1, 1, N, b3, code, kicked.data_ptr(), _ops.stream_ptr()), "gather")
    _ops.check(lib.chx_apply_affine7(kicked.data_ptr(), R.data_ptr(), both.data_ptr(), 1, 1, 1, N, code, _ops.stream_ptr()), "apply")
    _ops.check(lib.chx_sc_gather_kick_mapped(x.data_ptr(), F.data_ptr(), half.data_ptr(), cell.data_ptr(), energy.data_ptr(),
                                             dtk.data_ptr(), 510998.95069, 1, 1, 1, N, b3, code, R.data_ptr(), 1, fused.data_ptr(),
                                             _ops.stream_ptr()), "gather mapped")
    assert torch.equal(fused, both) and not torch.equal(kicked, x)
    # a setting of the fused run changed in place: picked up by the device-side validation inside the fused path as well
    els[3].k1.fill_(-2.0)
    out2 = seg.track(beam)
    assert not torch.equal(out2.particles, out.particles)
    b = beam
    for e in els:
        b = e.track(b)
    # element by element (three rounded passes) vs merged (one): equal to rounding, relative to each coordinate's spread
    scale = b.particles.abs().amax(dim=0).clamp_min(1e-30)
    assert float(((out2.particles - b.particles).abs() / scale).max()) < (1e-4 if dt == torch.float32 else 1e-10)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_run_of_scalar_settings_with_gradients_is_one_autograd_node(dt):
    """A run whose scalar settings (and the beam energy) carry gradients goes through `_ops.RunMapPlanned` (forward:
    chx_run_build_compose on the persistent plan; backward: chx_run_vjp_masked) or, without a plan, `_ops.RunMapScalars`. Gradients against the element-by-element path (a BuildMap node per
    element, torch matmuls in between) and, in fp64, against central differences."""
    import cheetah_amd as ca
    from cheetah_amd import _ops

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731

    def lattice(vals):
        k1, ang, tilt, mis, bend, L = vals
        return [ca.Drift(L, **kw), ca.Quadrupole(t(0.2), k1=k1, tilt=tilt, misalignment=mis, **kw), ca.Marker(),
                ca.HorizontalCorrector(t(0.05), angle=ang, **kw), ca.Dipole(t(0.5), angle=bend, dipole_e1=t(0.02), **kw),
                ca.Quadrupole(t(0.2), k1=k1, **kw), ca.Drift(t(0.3), **kw)]          # k1 is shared by two quadrupoles

    def fresh():
        return [torch.nn.Parameter(t(v)) for v in (4.2, 1e-3, 0.1)] + [torch.nn.Parameter(t([1e-4, -2e-4])),
                                                                       torch.nn.Parameter(t(0.05)), torch.nn.Parameter(t(0.7))]

    torch.manual_seed(5)
    beam0 = ca.ParticleBeam.from_parameters(num_particles=20_000, energy=t(1.2e8), sigma_x=t(2e-4), sigma_y=t(1e-4), **kw)

    def loss_of(out):
        p = out.particles
        return (p[:, 0].square().mean() * 3e6 + p[:, 2].mean() * 2e3 + p[:, 3].square().mean() * 1e7 + p[:, 4].var() * 1e9
                + p[:, 1].mean() * 1e3)

    def run(merged: bool, vals, energy):
        beam = ca.ParticleBeam(beam0.particles, energy, particle_charges=beam0.particle_charges, species=beam0.species)
        els = lattice(vals)
        if merged:
            return loss_of(ca.Segment(els).track(beam))
        for e in els:
            beam = e.track(beam)
        return loss_of(beam)

    calls = []
    orig = _ops.RunMapPlanned.apply
    try:
        _ops.RunMapPlanned.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
        va, ea = fresh(), torch.nn.Parameter(t(1.2e8))
        la = run(True, va, ea)
        la.backward()
    finally:
        _ops.RunMapPlanned.apply = orig
    assert calls, "the merged run did not take the one-node path (persistent differentiable plan)"
    # the plan-less form of the same node (RunMapScalars: runs beyond the plan's size) gives the same numbers
    from cheetah_amd.accelerator.segment import Segment

    planned = Segment._run_map_grad
    try:
        Segment._run_map_grad = staticmethod(lambda *a: None)
        vc, ec = fresh(), torch.nn.Parameter(t(1.2e8))
        lc = run(True, vc, ec)
        lc.backward()
    finally:
        Segment._run_map_grad = planned
    assert float(lc.detach()) == float(la.detach())
    for a, c in zip(va + [ea], vc + [ec]):
        assert torch.allclose(a.grad, c.grad, rtol=1e-6 if dt == torch.float32 else 1e-13, atol=0), (a.grad, c.grad)
    vb, eb = fresh(), torch.nn.Parameter(t(1.2e8))
    lb = run(False, vb, eb)
    lb.backward()
    rtol = 2e-3 if dt == torch.float32 else 1e-8
    assert float(la.detach()) == pytest.approx(float(lb.detach()), rel=1e-4 if dt == torch.float32 else 1e-10)
    for a, b in zip(va + [ea], vb + [eb]):
        assert a.grad is not None and torch.isfinite(a.grad).all()
        scale = float(b.grad.abs().max())
        assert float((a.grad - b.grad).abs().max()) <= rtol * scale + 1e-30, (a.grad, b.grad)
    if dt == torch.float64:
        with torch.no_grad():
            for i, h in enumerate((1e-5, 1e-8, 1e-6)):
                up, dn = fresh(), fresh()
                up[i].add_(h)
                dn[i].sub_(h)
                fd = (float(run(True, up, t(1.2e8))) - float(run(True, dn, t(1.2e8)))) / (2 * h)
                assert float(va[i].grad) == pytest.approx(fd, rel=2e-5, abs=1e-12)


def test_parameter_beam_takes_the_persistent_plan_too():
    """ParameterBeam through a run of scalar settings: the map comes from the persistent device plan (chx_run_map); the
    result equals the general path (maps built and composed per call) bit for bit, and follows in-place edits and
    re-assignments of the settings."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator.segment import Segment

    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    q1 = ca.Quadrupole(t(0.2), k1=t(4.2), name="q1", **kw)
    seg = ca.Segment([ca.Drift(t(0.4), **kw), q1, ca.Marker(), ca.HorizontalCorrector(t(0.03), angle=t(2e-4), **kw),
                      ca.Drift(t(0.6), **kw), ca.Quadrupole(t(0.2), k1=t(-3.0), **kw)])
    beam = ca.ParameterBeam.from_twiss(beta_x=t(3.0), beta_y=t(8.0), energy=t(1.5e8), **kw)

    def general():
        run = seg._plan()[0][1]
        maps = [e.first_order_transfer_map(beam.energy, beam.species) for e in run.elements]
        from cheetah_amd import _ops

        tm = _ops.compose_maps([m for m, e in zip(maps, run.elements) if e._chx_kind != _ops.KIND["identity"]], (), dt, maps[0].device)
        return _ops.parameter_track(beam.mu, beam.cov, tm)

    calls = []
    orig, descriptor = Segment._run_map_fast, Segment.__dict__["_run_map_fast"]
    Segment._run_map_fast = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        out = seg.track(beam)
    finally:
        Segment._run_map_fast = descriptor
    assert calls and seg._plan()[0][1].fast is not None and seg._plan()[0][1].fast.ok
    mu, cov = general()
    assert torch.equal(out.mu, mu.reshape(out.mu.shape)) and torch.equal(out.cov, cov.reshape(out.cov.shape))
    q1.k1.fill_(-1.5)                                   # in place: found by the device
    out2 = seg.track(beam)
    mu, cov = general()
    assert torch.equal(out2.mu, mu.reshape(out2.mu.shape)) and not torch.equal(out2.cov, out.cov)
    q1.k1 = t(7.0)                                      # re-assigned: pointer patched on the host
    out3 = seg.track(beam)
    mu, cov = general()
    assert torch.equal(out3.cov, cov.reshape(out3.cov.shape)) and not torch.equal(out3.cov, out2.cov)
    assert torch.equal(out.cov, out.cov.clone())        # earlier results are separate tensors, not views of the plan's state
    assert float(out3.s) == pytest.approx(1.43, rel=1e-6)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("cavity_type", ["standing_wave", "traveling_wave"])
def test_scalar_cavity_in_one_call_is_bit_identical(dt, cavity_type):
    """Cavity.track for one beam and scalar settings goes through chx_cavity_track_scalars (map, coefficients, outgoing energy
    by one thread + the particle pass: one C call). Same bits as the general path (chx_build_rmatrix, chx_cavity_coeffs,
    chx_cavity_track from packed tensors) for accelerating, decelerating and switched-off cavities."""
    import cheetah_amd as ca

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(2)
    beam = ca.ParticleBeam.from_parameters(num_particles=10_007, energy=t(6e7), sigma_tau=t(2e-4), sigma_p=t(3e-3), **kw)
    for voltage, phase in ((18e6, -12.0), (-7e6, 30.0), (0.0, 0.0), (5e6, 95.0)):
        cav = ca.Cavity(t(1.0377), voltage=t(voltage), phase=t(phase), frequency=t(1.3e9), cavity_type=cavity_type, **kw)
        calls = []
        orig = ca.Cavity._track_scalars
        ca.Cavity._track_scalars = lambda self, b: (calls.append(1), orig(self, b))[1]
        try:
            fast = cav.track(beam)
        finally:
            ca.Cavity._track_scalars = orig
        assert calls and fast.particles.shape == beam.particles.shape
        ca.Cavity._track_scalars = lambda self, b: None          # the general path
        try:
            slow = cav.track(beam)
        finally:
            ca.Cavity._track_scalars = orig
        assert torch.equal(fast.particles, slow.particles), (voltage, phase)
        assert torch.equal(fast.energy, slow.energy) and fast.energy.shape == slow.energy.shape == ()
        assert torch.equal(fast.s, slow.s)
        assert not torch.equal(fast.particles, beam.particles)
    # a vectorised voltage or a trainable one takes the general path
    assert ca.Cavity(t(1.0), voltage=t([1e6, 2e6]), phase=t(0.0), frequency=t(1.3e9), **kw)._track_scalars(beam) is None
    vp = torch.nn.Parameter(t(1e6))
    assert ca.Cavity(t(1.0), voltage=vp, phase=t(0.0), frequency=t(1.3e9), **kw)._track_scalars(beam) is None


@pytest.mark.parametrize("trainable_beam", [False, True])
def test_gradients_through_random_lattices_vs_reference(trainable_beam):
    """Eight drawn beamlines (tests/golden/generate_golden_random_grads.py) with up to four trainable scalar settings each:
    loss and d loss / d setting against the reference's autograd in float64 — through the one-node run path (RunMapScalars /
    chx_run_vjp) when only settings are trainable, and through the general per-element path when the incoming particles
    and the beam energy are trainable as well (then those gradients are compared too)."""
    import json
    import os

    import numpy as np

    import cheetah_amd as ca

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lattices_random_grads.npz"))
    kw = {"dtype": torch.float64, "device": "cuda"}
    for i in range(int(g["n_lattices"])):
        spec = json.loads(str(g[f"spec_{i}"]))
        trainable = json.loads(str(g[f"trainable_{i}"]))
        elements, params = [], []
        for e, (kind, args) in enumerate(spec):
            targs = {}
            for k, v in args.items():
                if isinstance(v, float):
                    t = torch.tensor(v, **kw)
                    if [e, k] in trainable:
                        t = torch.nn.Parameter(t)
                        params.append(t)
                    targs[k] = t
                else:
                    targs[k] = v
            elements.append(getattr(ca, kind)(**targs, **kw))
        x = torch.tensor(g[f"in_{i}"], **kw)
        en = torch.tensor(float(g[f"energy_{i}"]), **kw)
        if trainable_beam:
            x, en = torch.nn.Parameter(x), torch.nn.Parameter(en)
        out = ca.Segment(elements).track(ca.ParticleBeam(x, en, species=ca.Species("electron", **kw)))
        w1, w2 = torch.tensor(g[f"w1_{i}"], **kw), torch.tensor(g[f"w2_{i}"], **kw)
        loss = (out.particles[:, :6] * w1).sum(dim=1).mean() + (out.particles[:, :6].square() * w2).sum(dim=1).mean()
        loss.backward()
        names = [k for k, _ in spec]
        assert float(loss.detach()) == pytest.approx(float(g[f"loss_{i}"]), rel=1e-10), (i, names)
        ref = g[f"grads_{i}"]
        got = np.array([float(p.grad) for p in params])
        assert np.allclose(got, ref, rtol=1e-7, atol=1e-9 * np.abs(ref).max()), (i, names, trainable, got, ref)
        if trainable_beam:
            assert float(en.grad) == pytest.approx(float(g[f"grad_energy_{i}"]), rel=1e-6, abs=1e-12 * abs(float(g[f"loss_{i}"])))
            gp = g[f"grad_particles_{i}"]
            assert np.allclose(x.grad.cpu().numpy(), gp, rtol=1e-8, atol=1e-10 * np.abs(gp).max()), (i, names)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_path_length_follows_every_edit_of_a_length(dt):
    """`s` behind a run is `incoming.s + (L_0 + L_1 + ...)` (segment.py:54-58). The persistent plan computes it in the launch
    that validates the settings, so a length that was re-assigned or edited IN PLACE shows up at once — for a ParticleBeam, a
    ParameterBeam, behind a SpaceChargeKick and on the general path (regression: the fast path used to hand out a cached sum)."""
    import cheetah_amd as ca

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    d = ca.Drift(t(1.0), **kw)
    q = ca.Quadrupole(t(0.2), k1=t(1.0), **kw)
    seg = ca.Segment([d, ca.Marker(), q])
    beam = ca.ParticleBeam.from_parameters(num_particles=2000, **kw)
    pbeam = ca.ParameterBeam.from_parameters(**kw)

    def expect(total, s0=0.0):
        want = float(t(s0) + (t(total[0]) + t(total[1])))          # the reference's association, in dtype
        for b in (beam, pbeam):
            if s0:
                b = b.clone()
                b.s = t(s0)
            out = seg.track(b)
            assert float(out.s) == want, (type(b).__name__, float(out.s), want)
            assert out.s is not b.s

    expect((1.0, 0.2))
    expect((1.0, 0.2))                      # steady state (nothing changed)
    d.length = t(2.0)
    expect((2.0, 0.2))
    d.length.add_(1.0)                      # in place: no host counter moves
    expect((3.0, 0.2))
    q.length.mul_(2.0)
    expect((3.0, 0.4), s0=5.0)
    # behind a kick, the run's map and its length are folded into the kick's pass
    sc_seg = ca.Segment([ca.Drift(t(0.1), **kw), ca.SpaceChargeKick(t(0.25), grid_shape=(16, 16, 16), **kw), d, q])
    big = ca.ParticleBeam.from_parameters(num_particles=20_000, total_charge=t(1e-10), **kw)
    s1 = float(sc_seg.track(big).s)
    d.length.add_(0.5)
    s2 = float(sc_seg.track(big).s)
    assert s1 == pytest.approx(0.1 + 3.4, rel=1e-6) and s2 == pytest.approx(s1 + 0.5, rel=1e-6)   # the kick itself is thin
    # the general path (a vectorised setting rules the plan out) validates the lengths as well
    q.k1 = t([1.0, 2.0])
    assert float(seg.track(beam).s) == pytest.approx(3.9, rel=1e-6)
    d.length.sub_(1.0)
    assert float(seg.track(beam).s) == pytest.approx(2.9, rel=1e-6)


def test_deepcopy_and_pickle_after_a_gpu_track(ca):
    """the persistent plan (ctypes pointer arrays in `Segment.__dict__`) must not leak into copies — a copy plans
    again and follows ITS OWN settings (the reference's Segment is a plain nn.Module and deep-copies, segment.py:45-71)."""
    import copy
    import pickle

    dt = torch.float32
    seg = lattice(ca, dt)
    torch.manual_seed(4)
    beam = ca.ParticleBeam.from_parameters(num_particles=5000, dtype=dt, device="cuda")
    out = seg.track(beam).particles.clone()
    assert seg._plan()[0][1].fast is not None and seg._plan()[0][1].fast.ok
    dup = copy.deepcopy(seg)
    back = pickle.loads(pickle.dumps(seg))
    assert torch.equal(dup.track(beam).particles, out) and torch.equal(back.track(beam).particles, out)
    dup.q1.k1 = torch.tensor(-2.5, dtype=dt, device="cuda")         # assignment on the copy
    back.q1.k1.fill_(-2.5)                                            # in-place edit on the unpickled one
    fresh = lattice(ca, dt)
    fresh.q1.k1 = torch.tensor(-2.5, dtype=dt, device="cuda")
    want = fresh.track(beam).particles
    assert torch.equal(dup.track(beam).particles, want) and torch.equal(back.track(beam).particles, want)
    assert torch.equal(seg.track(beam).particles, out)               # the original never noticed


def test_fused_screen_reading_keeps_gradients_of_lattice_settings(ca):
    """`track_screen_reading` must not hand back a detached image when a lattice setting / the charges / the
    screen geometry carry a graph — it then takes `track` + `reading` (CicDeposit has a backward, the fused deposit none)."""
    dt = torch.float64
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(8)
    beam = ca.ParticleBeam.from_parameters(num_particles=20_000, sigma_x=t(2e-4), sigma_y=t(2e-4), **kw)

    def build(k1):
        return ca.Segment([ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=k1, **kw), ca.Drift(t(0.8), **kw),
                           ca.Screen(resolution=(32, 24), pixel_size=t([8e-5, 9e-5]), is_active=True, name="scr", **kw)])

    weights = torch.linspace(0.0, 1.0, 32 * 24, **kw).reshape(24, 32)
    k1a = torch.nn.Parameter(t(2.0))
    img = build(k1a).track_screen_reading(beam)
    assert img.requires_grad
    (img * weights).sum().backward()
    k1b = torch.nn.Parameter(t(2.0))
    seg = build(k1b)
    seg.track(beam)
    (seg.scr.reading * weights).sum().backward()
    assert k1a.grad is not None and torch.allclose(k1a.grad, k1b.grad, rtol=1e-12, atol=0)
    # without any graph the fused deposit is taken and gives the same image
    with torch.no_grad():
        plain = build(t(2.0)).track_screen_reading(beam)
    assert torch.allclose(plain, img.detach(), rtol=1e-10, atol=1e-30)
    # charges that require grad: same rule
    q = beam.particle_charges.clone().requires_grad_(True)
    beam_q = ca.ParticleBeam(beam.particles, beam.energy, particle_charges=q, species=beam.species)
    img_q = build(t(2.0)).track_screen_reading(beam_q)
    img_q.sum().backward()
    assert q.grad is not None and float(q.grad.abs().sum()) > 0


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_run_longer_than_one_device_plan_goes_through_in_pieces(ca, dt):
    """600 mergeable elements between the ends of a beamline: one persistent device plan holds 192 elements / 400 setting
    tensors, so the run's map is the product of its PIECES' maps (`Segment._run_map_parts`: a `chx_run_map` call per piece, one
    `chx_compose_maps`) instead of a host walk over every element. Against the product of all element maps (fp64: rounding of
    the association; fp32: the piece maps are rounded where the element maps were — measured 1.5e-7 of a coordinate's scale)."""
    t = lambda v: torch.tensor(v, dtype=dt, device="cuda")  # noqa: E731
    kw = {"dtype": dt, "device": "cuda"}
    els = []
    for i in range(150):
        els += [ca.Quadrupole(t(0.2), k1=t(1.1 if i % 2 == 0 else -1.1), misalignment=t([1e-5 * (i % 7), -2e-5]), tilt=t(1e-3 * (i % 3)), **kw),
                ca.Drift(t(0.8), **kw), ca.HorizontalCorrector(t(0.02), angle=t(1e-6 * (i % 5)), **kw),
                ca.Dipole(t(0.1), angle=t(1e-3), dipole_e1=t(5e-4), **kw) if i % 10 == 0 else ca.Drift(t(0.1), **kw)]
    seg = ca.Segment(els)
    beam = ca.ParticleBeam.from_parameters(num_particles=20_000, dtype=dt, device="cuda")
    with torch.no_grad():
        out = seg.track(beam)
        run = seg._plan()[0][1]
        assert run.parts and len(run.parts[0]) >= 4 and all(c and p.fast is not None and p.fast.ok for p, c in run.parts[0])
        ref = general_path(ca, seg, beam)
        scale = ref.abs().amax(dim=0)
        err = ((out.particles - ref).abs() / scale).max()
        assert err < (6e-7 if dt == torch.float32 else 1e-12), float(err)
        assert float(out.s) == pytest.approx(float(sum(float(e.length) for e in els)), rel=1e-6 if dt == torch.float32 else 1e-12)
        # an in-place edit inside the third piece is followed (the device compares the values), and so is a re-assignment
        els[300].k1.add_(0.5)
        out2 = seg.track(beam)
        ref2 = general_path(ca, seg, beam)
        assert not torch.equal(out2.particles, out.particles)
        assert ((out2.particles - ref2).abs() / scale).max() < (6e-7 if dt == torch.float32 else 1e-12)
        els[10].k1 = t(-0.7)
        ref3 = general_path(ca, seg, beam)
        assert ((seg.track(beam).particles - ref3).abs() / scale).max() < (6e-7 if dt == torch.float32 else 1e-12)
        # ParameterBeam: the same pieces
        pb = ca.ParameterBeam.from_parameters(dtype=dt, device="cuda")
        got = seg.track(pb)
        tm = ca._ops.compose_maps([e.first_order_transfer_map(pb.energy, pb.species) for e in els], (), dt, torch.device("cuda"))
        want_mu = tm @ pb.mu
        assert torch.allclose(got.mu, want_mu, rtol=0, atol=float(want_mu.abs().max()) * (2e-6 if dt == torch.float32 else 1e-12))
    # a trainable setting anywhere in the run: the differentiable path, as before
    els[4].k1 = torch.nn.Parameter(t(0.3))
    seg.track(beam).sigma_x.backward()
    assert els[4].k1.grad is not None and torch.isfinite(els[4].k1.grad) and float(els[4].k1.grad) != 0.0


def test_nested_segments_are_planned_through(ca):
    """A lattice of cells of cells (what lattice files give): the plain nested Segments' elements join the parent's runs — ONE run
    plan for 25 cells of 4 elements, the same numbers as the flattened lattice bit for bit (same elements in the same order);
    a nested segment holding an active Screen still records its beam; a SUBCLASS of Segment stays an element of its own."""
    dt = torch.float32
    t = lambda v: torch.tensor(v, dtype=dt, device="cuda")  # noqa: E731
    kw = {"dtype": dt, "device": "cuda"}

    def cell(i):
        return [ca.Quadrupole(t(0.2), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.8), **kw)]

    cells = [ca.Segment(cell(2 * i) + [ca.Segment(cell(2 * i + 1), name=f"inner{i}")], name=f"cell{i}") for i in range(25)]
    nested = ca.Segment(cells)
    flat = nested.flattened()
    beam = ca.ParticleBeam.from_parameters(num_particles=10_000, **kw)
    with torch.no_grad():
        out, ref = nested.track(beam), flat.track(beam)
    plan = nested._plan()
    assert len(plan) == 1 and plan[0][0] == "run" and len(plan[0][1].elements) == 100 and plan[0][1].fast.ok
    assert torch.equal(out.particles, ref.particles) and torch.equal(out.s, ref.s)
    # O(1) re-validation: nothing in the plan depends on tensor values any more
    assert nested.__dict__["_plan_cache"][3] is None
    # an edit deep inside is followed
    cells[3].elements[2].elements[0].k1 = t(1.5)
    with torch.no_grad():
        assert torch.equal(nested.track(beam).particles, flat.track(beam).particles)
    # an active screen inside a nested segment records the beam at its place
    scr = ca.Screen(resolution=(32, 32), is_active=True, name="scr", **kw)
    cells[10].elements.append(scr)
    flat2 = nested.flattened()
    with torch.no_grad():
        out2 = nested.track(beam)
        read = scr.get_read_beam().particles.clone()
        ref2 = flat2.track(beam)
    assert torch.equal(out2.particles, ref2.particles) and torch.equal(read, scr.get_read_beam().particles)
    assert [k for k, _ in nested._plan()] == ["run", "element", "run"]

    class MySegment(ca.Segment):
        def track(self, incoming):
            return super().track(incoming)

    own = ca.Segment([MySegment(cell(0)), ca.Drift(t(0.1), **kw)])
    assert any(isinstance(e, MySegment) for _, item in own._plan() for e in (item.elements if _ == "run" else [item]))
    with torch.no_grad():       # (its map enters the run as ONE factor: the same product in another association)
        a, b = own.track(beam).particles, own.flattened().track(beam).particles
    assert ((a - b).abs().amax(dim=0) / b.abs().amax(dim=0)).max() < 1e-6


def test_run_with_elements_no_device_plan_takes_goes_through_in_pieces(ca):
    """100 elements around a CustomTransferMap, a quadrupole with 8 strengths and (then) a trainable corrector: the stretches a
    persistent plan takes come from `chx_run_map`, the three odd elements from the general path, one `chx_compose_maps` puts
    them together — the product of all element maps (vectorised over the 8 strengths), with a gradient where one is asked for."""
    dt = torch.float32
    t = lambda v: torch.tensor(v, dtype=dt, device="cuda")  # noqa: E731
    kw = {"dtype": dt, "device": "cuda"}
    els = []
    for i in range(50):
        els += [ca.Quadrupole(t(0.2), k1=t(2.2 if i % 2 == 0 else -2.2), **kw), ca.Drift(t(0.8), **kw)]
    R = torch.eye(7, **kw)
    R[0, 1], R[2, 3], R[0, 6] = 0.3, 0.3, 1e-5
    els[31] = ca.CustomTransferMap(R, length=t(0.3), **kw)
    els[60] = ca.Quadrupole(t(0.2), k1=torch.linspace(-3.0, 3.0, 8, **kw), **kw)
    seg = ca.Segment(els)
    beam = ca.ParticleBeam.from_parameters(num_particles=5_000, **kw)
    with torch.no_grad():
        out = seg.track(beam)
        run = seg._plan()[0][1]
        assert run.parts and [c for _, c in run.parts[0]] == [True, False, True, False, True]
        maps = [e.first_order_transfer_map(beam.energy, beam.species) for e in els]
        tm = ca._ops.compose_maps(maps, (8,), dt, torch.device("cuda"))
        ref = ca._ops.apply_map(beam.particles, tm)
    assert out.particles.shape == (8, 5_000, 7)
    scale = ref.abs().amax(dim=(0, 1))
    assert ((out.particles - ref).abs().amax(dim=(0, 1)) / scale).max() < 1e-6
    assert float(out.s) == pytest.approx(sum(float(e.length) for e in els), rel=1e-6)
    # a trainable setting in one of the plan's stretches: that stretch takes the differentiable path, the product keeps the graph
    els[5] = ca.HorizontalCorrector(t(0.1), angle=torch.nn.Parameter(t(1e-4)), **kw)
    seg = ca.Segment(els)
    loss = seg.track(beam).mu_x.sum()
    loss.backward()
    g = els[5].angle.grad
    assert g is not None and torch.isfinite(g) and float(g) != 0.0
    # against finite differences of the same lattice
    with torch.no_grad():
        h = 1e-5
        els[5].angle.add_(h); up = float(seg.track(beam).mu_x.sum())
        els[5].angle.sub_(2 * h); dn = float(seg.track(beam).mu_x.sum())
    assert float(g) == pytest.approx((up - dn) / (2 * h), rel=2e-2)


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_long_lattice_vs_reference(ca, dt):
    """700 mergeable elements around a CustomTransferMap — more than one persistent device plan takes: the run's map is composed
    from its pieces — against the REFERENCE's own float64 run (tests/golden/long_lattice.npz, generator
    tests/golden/generate_golden_long_lattice.py; segment.py:534-574): ParticleBeam, ParameterBeam, three beams in one ParticleBeam.
    Measured on the MI355X, relative to a coordinate's scale: float64 particles 8.8e-15, ParameterBeam 2.6e-14, three beams 7.3e-15;
    float32 1.2e-6 / 2.7e-6 / 1.2e-6 — bounds = 4 x those."""
    import json
    import os

    import numpy as np

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "long_lattice.npz"))
    fk = {"dtype": dt, "device": "cuda"}
    t = lambda a: torch.tensor(np.asarray(a), **fk)  # noqa: E731
    els = []
    for kind, kw in json.loads(str(g["spec"])):
        args = {k: (torch.tensor(v, **fk) if isinstance(v, (float, list)) else v) for k, v in kw.items()}
        els.append(getattr(ca, kind)(**args, **fk))
    seg = ca.Segment(els)
    beam = ca.ParticleBeam(t(g["in"]), t(g["energy"]), **fk)
    with torch.no_grad():
        out = seg.track(beam)
        run = seg._plan()[0][1]
        assert len(seg._plan()) == 1 and run.parts and len(run.parts[0]) >= 6       # pieces around the CustomTransferMap
        pout = seg.track(ca.ParameterBeam(t(g["pb_mu_in"]), t(g["pb_cov_in"]), t(g["energy"]), **fk))
        parts = t(g["in"]).unsqueeze(0) * torch.tensor([0.5, 1.0, 1.5], **fk).reshape(3, 1, 1)
        parts[..., 6] = 1.0
        many = seg.track(ca.ParticleBeam(parts.contiguous(), t(g["energy"]), **fk))
    tol = 3.6e-14 if dt == torch.float64 else 5e-6
    ref = g["out"]
    err = (np.abs(out.particles.double().cpu().numpy() - ref).max(axis=0) / np.abs(ref).max(axis=0)).max()
    assert err < tol, err
    assert float(out.s) == pytest.approx(float(g["s_out"]), rel=1e-12 if dt == torch.float64 else 2e-6)
    e_mu = np.abs(pout.mu.double().cpu().numpy() - g["pb_mu"]).max() / np.abs(g["pb_mu"][:6]).max()
    e_cov = np.abs(pout.cov.double().cpu().numpy() - g["pb_cov"]).max() / np.abs(g["pb_cov"]).max()
    assert max(e_mu, e_cov) < 3 * tol, (e_mu, e_cov)
    mref = g["many_out"]
    merr = (np.abs(many.particles.double().cpu().numpy() - mref).max(axis=(0, 1)) / np.abs(mref).max(axis=(0, 1))).max()
    assert merr < tol, merr
    print(f"long lattice vs reference ({dt}): particles {err:.2e}, parameter beam {max(e_mu, e_cov):.2e}, three beams {merr:.2e}")


def test_settings_that_are_views_of_one_tensor(ca):
    """Several magnets' settings as 0-dim views of ONE tensor (a control loop writes them all with one `copy_`): the device plan
    holds their addresses and follows the write."""
    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    settings = torch.tensor([3.0, -3.0, 1e-4], **kw)
    seg = ca.Segment([ca.Quadrupole(t(0.2), k1=settings[0], **kw), ca.Drift(t(0.5), **kw), ca.Quadrupole(t(0.2), k1=settings[1], **kw),
                      ca.HorizontalCorrector(t(0.05), angle=settings[2], **kw), ca.Drift(t(0.5), **kw)])
    beam = ca.ParticleBeam.from_parameters(num_particles=4_000, **kw)
    with torch.no_grad():
        a = seg.track(beam).particles.clone()
        assert seg._plan()[0][1].fast.ok
        assert torch.equal(a, general_path(ca, seg, beam))
        epoch = ca.Element._epoch
        settings.copy_(torch.tensor([1.5, -2.0, -3e-4], **kw))
        b = seg.track(beam).particles
        assert ca.Element._epoch == epoch and not torch.equal(a, b) and torch.equal(b, general_path(ca, seg, beam))


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_vectorised_settings_of_a_run_in_one_launch(ca, dt):
    """A run whose settings are vectorised over a batch of lattice settings (k1 scans, batched environments): the composed maps of
    all rows from ONE launch (chx_run_map_batched) — bit for bit the per-element builds + chx_compose_maps of the general path
    (which tests/golden/lattices_random_vectorized.npz pins against the reference); runs that do not qualify keep that path."""
    from cheetah_amd import _ops
    from cheetah_amd.accelerator.segment import Segment

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(5)
    for shape in ((7,), (3, 2)):
        els = [ca.Marker(**kw), ca.Drift(t(0.175), **kw), ca.Quadrupole(t(0.122), k1=torch.randn(shape, **kw) * 8, **kw), ca.Drift(t(0.428), **kw),
               ca.Quadrupole(t(0.122), k1=t(-14.3), tilt=torch.randn(shape, **kw) * 0.05, misalignment=t([1e-4, -2e-4]), **kw),
               ca.VerticalCorrector(t(0.02), angle=torch.randn(shape, **kw) * 1e-4, **kw),
               ca.Dipole(t(0.3), angle=torch.randn(shape, **kw) * 0.02, dipole_e1=t(0.01), **kw), ca.Solenoid(t(0.1), k=t(0.3), **kw),
               ca.HorizontalCorrector(t(0.02), angle=t(-1e-4), **kw), ca.Drift(t(0.45), **kw)]
        seg = ca.Segment(els)
        beam = ca.ParticleBeam.from_parameters(num_particles=3_000, **kw)
        calls = []
        orig = Segment._run_map_vector
        Segment._run_map_vector = staticmethod(lambda run, energy, species: (calls.append(1), orig(run, energy, species))[1])
        try:
            with torch.no_grad():
                out = seg.track(beam)
        finally:
            Segment._run_map_vector = orig
        assert calls and out.particles.shape == (*shape, 3_000, 7)
        maps = [e.first_order_transfer_map(beam.energy, beam.species) for e in els if e._chx_kind != _ops.KIND["identity"]]
        tm = _ops.compose_maps(maps, shape, dt, torch.device("cuda"))
        run = seg._plan()[0][1]
        got = Segment._run_map_vector(run, beam.energy, beam.species)
        assert got is not None and got.shape == (*shape, 7, 7) and torch.equal(got, tm)
        with torch.no_grad():
            assert torch.equal(out.particles, _ops.apply_map(beam.particles, tm))
        # a ParameterBeam takes the same maps
        pb = ca.ParameterBeam.from_parameters(**kw)
        with torch.no_grad():
            pout = seg.track(pb)
            mu, cov = _ops.parameter_track(pb.mu, pb.cov, tm)
        assert torch.equal(pout.mu, mu) and torch.equal(pout.cov, cov)
    # what does not qualify: a vectorised length, batch shapes that do not broadcast, a setting that requires grad
    assert Segment._run_map_vector(ca.Segment([ca.Drift(torch.tensor([0.1, 0.2], **kw), **kw), ca.Quadrupole(t(0.1), k1=t(1.0), **kw)])._plan()[0][1],
                                   beam.energy, beam.species) is None
    clash = ca.Segment([ca.Quadrupole(t(0.1), k1=torch.randn(3, **kw), **kw), ca.Quadrupole(t(0.1), k1=torch.randn(2, **kw), **kw)])
    assert Segment._run_map_vector(clash._plan()[0][1], beam.energy, beam.species) is None
    # two shapes that BROADCAST ((3,) against (3, 1): a grid scan): one launch over the (3, 3) grid through expanded copies of the
    # settings, bit for bit the per-element builders + chx_compose_maps; an in-place edit of a setting is followed
    kb = torch.randn(3, 1, **kw)
    mixed = ca.Segment([ca.Quadrupole(t(0.1), k1=torch.randn(3, **kw), **kw), ca.Quadrupole(t(0.1), k1=kb, **kw)])
    for _ in range(2):
        got = Segment._run_map_vector(mixed._plan()[0][1], beam.energy, beam.species)
        maps = [e.first_order_transfer_map(beam.energy, beam.species) for e in mixed.elements]
        want = _ops.compose_maps(maps, (3, 3), dt, maps[0].device)
        assert got.shape == (3, 3, 7, 7) and torch.equal(got, want)
        kb.mul_(-1.5)
    with torch.no_grad():
        assert mixed.track(beam).particles.shape == (3, 3, 3_000, 7)
    k = torch.randn(4, **kw).requires_grad_(True)
    gseg = ca.Segment([ca.Quadrupole(t(0.1), k1=k, **kw), ca.Drift(t(0.5), **kw)])
    gseg.track(beam).sigma_x.sum().backward()
    assert k.grad is not None and torch.isfinite(k.grad).all()


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("with_vector_settings", [False, True])
def test_energy_scan_maps_in_one_launch(dt, with_vector_settings):
    """A scan of BEAM ENERGIES — `energy` a (B,) tensor, lattice settings scalar or vectorised over the same shape: the composed maps
    of a run for all B energies from one launch (chx_run_map_batched with energy_rows), bit for bit the per-element builders +
    chx_compose_maps of the general path; a ParticleBeam and a ParameterBeam tracked through both agree bit for bit, a fresh energy
    tensor per track included (nothing to cache)."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator.segment import Segment

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(17)
    B = 37
    els = []
    for i in range(30):
        k1 = torch.randn(B, **kw) * 4 if (with_vector_settings and i % 7 == 3) else t(4.2 if i % 2 == 0 else -4.2)
        els += [ca.Quadrupole(t(0.2), k1=k1, misalignment=t([1e-4, -2e-4]) if i == 5 else None, **kw), ca.Drift(t(0.8), **kw)]
        if i % 10 == 4:
            els += [ca.Dipole(t(0.3), angle=t(0.02), dipole_e1=t(0.01), **kw), ca.HorizontalCorrector(t(0.05), angle=t(1e-4), **kw)]
    seg = ca.Segment(els)
    energy = torch.linspace(6e7, 1.4e8, B, **kw)
    beam = ca.ParticleBeam.from_parameters(num_particles=2_000, energy=energy, **kw)
    pbeam = ca.ParameterBeam.from_parameters(energy=energy, sigma_p=t(1e-3), **kw)
    run = seg._plan()[0][1]
    calls = []
    orig = Segment._run_map_vector
    try:
        Segment._run_map_vector = staticmethod(lambda run, energy, species: (lambda r: (calls.append(r is not None), r)[1])(orig(run, energy, species)))
        with torch.no_grad():
            out, pout = seg.track(beam), seg.track(pbeam)
            fresh = seg.track(ca.ParticleBeam(beam.particles, energy.clone(), particle_charges=beam.particle_charges, **kw))
        assert calls and all(calls), calls
        got = orig(run, energy, beam.species)
        Segment._run_map_vector = staticmethod(lambda run, energy, species: None)
        ca.Element._epoch += 1
        run.token = run.tm = None
        with torch.no_grad():
            ref, pref = seg.track(beam), seg.track(pbeam)
            want = Segment._run_map(run, energy.clone(), beam.species)
    finally:
        Segment._run_map_vector = orig
        ca.Element._epoch += 1
    assert got.shape == want.shape == (B, 7, 7) and torch.equal(got, want)
    assert out.particles.shape == (B, 2_000, 7) and torch.equal(out.particles, ref.particles) and torch.equal(fresh.particles, ref.particles)
    assert torch.equal(pout.mu, pref.mu) and torch.equal(pout.cov, pref.cov)
    assert (out.particles[0] - out.particles[-1]).abs().max() > 1e-6          # the rows see different energies
