"""Chains of SpaceChargeKicks inside one `Segment.track` on the tile-ordered beam (`chx_sc_kick_sorted`, csrc/chx_sc_tiles.h):
the rows are sorted by deposit tile once, every kick deposits from the ordered rows and gathers with its tile's potential block
in LDS, the last kick restores the caller's particle order. Must give what tracking the elements one by one gives (each kick
through `chx_sc_kick`; the reference's per-kick arithmetic, /root/reference/cheetah/accelerator/space_charge_kick.py:477-586,
/root/reference/tests/test_space_charge_kick.py:14-71) — the pieces bit for bit where they are deterministic:
* tile deposit == generic deposit up to the summation order (fp64 LDS sums rounded once, fp32 adds of the +1 layers);
* tile gather == `chx_sc_gather_kick_phi` exactly;
* misfiled particles, particles outside the extent, hot tiles and the device-side re-sort keep the result."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ca():
    assert torch.cuda.is_available()
    import cheetah_amd

    cheetah_amd._lib.lib()
    return cheetah_amd


def _beam(ca, dt, n, seed=5, gaussian=False):
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(seed)
    if gaussian:
        return ca.ParticleBeam.from_parameters(num_particles=n, sigma_x=t(3e-4), sigma_y=t(2e-4), sigma_tau=t(1e-4), sigma_px=t(2e-5),
                                               sigma_py=t(3e-5), sigma_p=t(1e-3), energy=t(5e7), total_charge=t(1e-9), **kw)
    return ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=n, total_charge=t(1e-9), energy=t(2.5e8), radius_x=t(1e-3),
                                                radius_y=t(1e-3), radius_tau=t(1e-3), sigma_px=t(1e-6), sigma_py=t(1e-6),
                                                sigma_p=t(1e-6), **kw)


def _linac(ca, dt, grid, n_kicks, k1=4.2, drift=0.1):
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    els = []
    for i in range(n_kicks):
        els += [ca.Drift(t(drift), **kw), ca.SpaceChargeKick(t(0.2), grid_shape=grid, **kw), ca.Drift(t(drift), **kw),
                ca.Quadrupole(t(0.1), k1=t(k1 if i % 2 == 0 else -k1), **kw), ca.Drift(t(drift), **kw)]
    return ca.Segment(els)


def _one_by_one(seg, beam):
    for e in seg.elements:
        beam = e.track(beam)
    return beam


def _unchained(seg, beam):
    """`Segment.track` with every kick through the one-call path `chx_sc_kick` (what a chain replaces)."""
    from cheetah_amd.accelerator.segment import Segment

    orig = Segment._chain_starts
    Segment._chain_starts = lambda self, plan, i, incoming: False
    try:
        return seg.track(beam)
    finally:
        Segment._chain_starts = orig


@pytest.mark.parametrize("dt,grid,n,gaussian", [(torch.float32, (64, 64, 64), 200_000, False), (torch.float64, (32, 32, 32), 100_000, True),
                                                (torch.float32, (128, 64, 32), 150_001, True),
                                                # dense tiles (the reference's default 32^3 grid under a large beam): the deposit
                                                # splits every tile's slots over 16 / 32 workgroups
                                                (torch.float32, (32, 32, 32), 400_000, False), (torch.float32, (16, 16, 64), 300_001, True)])
def test_chain_equals_kick_by_kick(ca, dt, grid, n, gaussian):
    seg, beam = _linac(ca, dt, grid, 4), _beam(ca, dt, n, gaussian=gaussian)
    calls = []
    from cheetah_amd import _ops

    orig = _ops.sc_kick_sorted
    _ops.sc_kick_sorted = lambda *a, **k: (calls.append((a[10], a[11])), orig(*a, **k))[1]
    try:
        out = seg.track(beam)
    finally:
        _ops.sc_kick_sorted = orig
    assert calls == [(True, False), (False, False), (False, False), (False, True)]      # (first, last) of the four links
    ref = _unchained(seg, beam)
    assert torch.equal(out.s, ref.s)
    effect = (ref.particles - _one_by_one(_no_charge(ca, seg), beam).particles).abs().max(dim=0).values     # size of the space-charge effect
    err = (out.particles - ref.particles).abs().max(dim=0).values
    # the two paths sum the charge in a different order: a relative error of the space-charge EFFECT plus a few units in the
    # last place of the coordinates themselves (the effect of this beam is ~1e-4 of a coordinate, i.e. ~1e3 ulp in fp32)
    tol = 2e-4 if dt == torch.float32 else 1e-9
    ulp = 16 * torch.finfo(dt).eps * ref.particles.abs().max(dim=0).values
    assert torch.all(err <= tol * effect + ulp), (err / (tol * effect + ulp))
    # the beam's own arrays were not permuted
    assert out.particle_charges is beam.particle_charges or torch.equal(out.particle_charges, beam.particle_charges)


def _no_charge(ca, seg):
    """The same lattice without the kicks (what the particles do without space charge)."""
    return ca.Segment([e for e in seg.elements if not isinstance(e, ca.SpaceChargeKick)])


def test_strong_mixing_takes_the_slow_paths_and_resorts(ca):
    """Quadrupoles strong enough to turn the beam inside out between kicks: most particles leave the tile of their slot
    (global atomics into the second grid / global potential loads), and the device re-sorts on the next kick."""
    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    els = []
    for i in range(5):
        els += [ca.SpaceChargeKick(t(0.05), grid_shape=(32, 32, 32), **kw), ca.Quadrupole(t(0.2), k1=t(14.0 if i % 2 == 0 else -12.0), **kw),
                ca.Drift(t(0.9), **kw)]
    seg = ca.Segment(els)
    beam = _beam(ca, dt, 120_000, gaussian=True)
    from cheetah_amd import _ops

    flags, orig = [], _ops.sc_kick_sorted

    def spying(*a, **k):
        res = orig(*a, **k)
        flags.append(int(a[9][:32].view(torch.int32)[5]))       # header.n_sorts after the kick (a host read: test only)
        return res

    _ops.sc_kick_sorted = spying
    try:
        out = seg.track(beam)
    finally:
        _ops.sc_kick_sorted = orig
    ref = _unchained(seg, beam)
    effect = (ref.particles - _one_by_one(_no_charge(ca, seg), beam).particles).abs().max(dim=0).values
    err = (out.particles - ref.particles).abs().max(dim=0).values
    assert torch.isfinite(ref.particles).all() and float(effect[:4].min()) > 0
    ulp = 4 * torch.finfo(dt).eps * ref.particles.abs().max(dim=0).values
    assert torch.all(err <= 1e-3 * effect + ulp), (err / effect)
    # the quadrupoles image the beam upside down: the slots then hold particles of the mirrored tiles — kicks after the first find
    # (nearly) all particles misfiled and have their gather write the rows in the new tile order
    assert flags[-1] >= 2, flags


def test_reshuffling_beam_turns_the_chain_off_and_a_laminar_one_keeps_it(ca):
    """The host's guard (Segment._chain_allowed): the chain's header — the mean share of particles each deposit found outside
    their slot's tile — comes back asynchronously after a track; a lattice whose beam is imaged upside down between kicks goes
    back to kick-by-kick tracking (same results), the laminar linac keeps its chain."""
    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    from cheetah_amd import _ops

    calls, orig = [], _ops.sc_kick_sorted
    _ops.sc_kick_sorted = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        els = []
        for i in range(5):
            els += [ca.SpaceChargeKick(t(0.05), grid_shape=(32, 32, 32), **kw),
                    ca.Quadrupole(t(0.2), k1=t(14.0 if i % 2 == 0 else -12.0), **kw), ca.Drift(t(0.9), **kw)]
        mixing, beam = ca.Segment(els), _beam(ca, dt, 120_000, gaussian=True)
        first = mixing.track(beam)
        assert len(calls) == 5                                   # chained
        torch.cuda.synchronize()                                  # (the guard itself never waits: it polls an event)
        second = mixing.track(beam)
        assert len(calls) == 5, "the second track still took the chain"
        guard = mixing._chain_guard(mixing._plan())
        assert guard["off"] and guard["pending"] is None
        ref = _unchained(mixing, beam)
        effect = (ref.particles - _one_by_one(_no_charge(ca, mixing), beam).particles).abs().max(dim=0).values
        ulp = 4 * torch.finfo(dt).eps * ref.particles.abs().max(dim=0).values
        # (`second` took the very path of `ref`; its hot tiles use float atomics, so not bit for bit)
        assert torch.all((second.particles - ref.particles).abs().max(dim=0).values <= 1e-5 * effect + ulp)
        assert torch.all((first.particles - ref.particles).abs().max(dim=0).values <= 1e-3 * effect + ulp)
        # a copy of the lattice starts without the verdict (derived state)
        import copy

        assert copy.deepcopy(mixing).__dict__["_chain_guard_state"] is None
        # the laminar linac: a few per cent misfiled per kick -> stays chained, and stops sampling after three tracks
        calls.clear()
        linac, beam = _linac(ca, dt, (64, 64, 64), 4), _beam(ca, dt, 200_000)
        for _ in range(5):
            linac.track(beam)
            torch.cuda.synchronize()
        assert len(calls) == 20
        guard = linac._chain_guard(linac._plan())
        assert not guard["off"] and guard["samples"] == 3
    finally:
        _ops.sc_kick_sorted = orig


def _tile_pieces(ca, x, q, w, extent, scale, bins):
    """chx_sc_tile_sort + chx_sc_tile_deposit on x; returns (rho, state)."""
    from cheetah_amd import _lib, _ops

    lib = _lib.lib()
    N = x.shape[0]
    b3 = _ops._bins3(bins)
    dtc = _ops.dtype_code(x.dtype)
    state = _ops.sc_tile_state(N, bins, x.dtype, x.device)
    _ops.check(lib.chx_sc_tile_sort(x.data_ptr(), q.data_ptr(), w.data_ptr(), extent.data_ptr(), scale.data_ptr(), N, b3, dtc,
                                    state.data_ptr(), state.numel(), _ops.stream_ptr()), "chx_sc_tile_sort")
    rho = torch.full(tuple(bins), float("nan"), dtype=x.dtype, device=x.device)     # every cell must be STORED
    _ops.check(lib.chx_sc_tile_deposit(None, extent.data_ptr(), scale.data_ptr(), N, b3, dtc, state.data_ptr(), state.numel(),
                                       rho.data_ptr(), 0, _ops.stream_ptr()), "chx_sc_tile_deposit")
    return rho, state


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("case", ["gaussian", "outside", "hot"])
def test_tile_deposit_equals_generic_deposit(ca, dt, case):
    from cheetah_amd import _ops

    torch.manual_seed(2)
    N, bins = 150_000, (32, 64, 16)
    kw = {"dtype": dt, "device": "cuda"}
    x = torch.randn(N, 7, **kw) * torch.tensor([1e-3, 1e-5, 2e-3, 1e-5, 5e-4, 1e-3, 0.0], **kw)
    x[:, 6] = 1.0
    half = torch.tensor([3e-3, 6e-3, 1.5e-3], **kw)
    if case == "outside":
        half = half / 2.5                                       # a good part of the beam lies outside the extent
    if case == "hot":
        x[: N // 2, [0, 2, 4]] *= 0.02                            # half of the beam inside one or two tiles (> 8192 per tile)
    extent = torch.stack([-half, half], dim=-1).reshape(1, 3, 2).contiguous()
    scale = torch.tensor([[1.0, 1.0, -0.999]], **kw)
    q = torch.rand(N, **kw) * 1e-15
    w = torch.rand(N, **kw)
    rho, state = _tile_pieces(ca, x, q, w, extent, scale, bins)
    ref = _ops.sc_deposit_overwrite(x.reshape(1, N, 7), q.reshape(1, N), w.reshape(1, N), extent, scale, 1, N, bins)[0]
    assert torch.isfinite(rho).all()
    scale_ = ref.abs().max()
    # a hot tile's overflow beyond 8192 particles is added with global float atomics in the grid dtype (the generic sorted
    # deposit keeps fp64 LDS sums for it): tens of thousands of fp32 addends into one cell round at the 1e-5 level
    tol = (5e-5 if case == "hot" else 2e-6) if dt == torch.float32 else 1e-13
    assert float((rho - ref).abs().max() / scale_) < tol
    assert float(rho.double().sum()) == pytest.approx(float(ref.double().sum()), rel=1e-6 if dt == torch.float32 else 1e-12)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_tile_gather_is_bit_identical_to_the_untiled_kernel(ca, dt):
    """Sort, then gather + kick from a random potential: rows restored to the caller's order must EQUAL chx_sc_gather_kick_phi's
    (same nodes, same arithmetic), with and without the linear map applied in the same pass."""
    from cheetah_amd import _lib, _ops

    lib = _lib.lib()
    torch.manual_seed(9)
    N, bins = 130_000, (32, 32, 64)
    kw = {"dtype": dt, "device": "cuda"}
    x = torch.randn(N, 7, **kw) * torch.tensor([1e-3, 1e-5, 1e-3, 1e-5, 1e-3, 1e-3, 0.0], **kw)
    x[:, 6] = 1.0
    x[:500, 0] *= 5.0                                             # some particles beyond the grid
    energy = torch.tensor([5e7], **kw)
    gamma = energy / 510998.95069
    beta = (1 - 1 / gamma**2).sqrt()
    half = torch.tensor([[3.2e-3, 3.1e-3, 3.3e-3]], **kw)
    cell = 2 * half / torch.tensor([list(map(float, bins))], **kw)
    extent = torch.stack([-half[0], half[0]], dim=-1).reshape(1, 3, 2).contiguous()
    scale = torch.stack([torch.ones_like(beta), torch.ones_like(beta), -beta], dim=-1).contiguous()
    dtt = torch.tensor([0.2 / 299792458.0], **kw)
    phi = torch.randn(1, bins[0] + 4, bins[1] + 4, bins[2] + 4, **kw) * 1e3
    q = torch.full((N,), 1e-15, **kw)
    w = torch.ones(N, **kw)
    _, state = _tile_pieces(ca, x, q, w, extent, scale, bins)
    R = (torch.eye(7, **kw) + 0.05 * torch.randn(7, 7, **kw)).contiguous()
    for post in (None, R):
        ref = _ops.sc_gather_kick_phi(x.reshape(1, N, 7), phi, half, cell, gamma, energy, dtt, 510998.95069, 1, N, bins,
                                      post_map=None if post is None else post.reshape(1, 7, 7))[0]
        out = torch.empty_like(x)
        _ops.check(lib.chx_sc_tile_gather_kick(None, phi.data_ptr(), half.data_ptr(), cell.data_ptr(), gamma.data_ptr(),
                                               energy.data_ptr(), dtt.data_ptr(), 510998.95069, N, _ops._bins3(bins), _ops.dtype_code(dt),
                                               None if post is None else post.data_ptr(), state.data_ptr(), state.numel(), 1,
                                               out.data_ptr(), _ops.stream_ptr()), "chx_sc_tile_gather_kick")
        assert torch.equal(out, ref)
        # in tile order: the same rows, permuted
        tiled = torch.empty_like(x)
        _ops.check(lib.chx_sc_tile_gather_kick(None, phi.data_ptr(), half.data_ptr(), cell.data_ptr(), gamma.data_ptr(),
                                               energy.data_ptr(), dtt.data_ptr(), 510998.95069, N, _ops._bins3(bins), _ops.dtype_code(dt),
                                               None if post is None else post.data_ptr(), state.data_ptr(), state.numel(), 0,
                                               tiled.data_ptr(), _ops.stream_ptr()), "chx_sc_tile_gather_kick")
        assert torch.equal(torch.sort(tiled[:, 0]).values, torch.sort(ref[:, 0]).values)


def test_isolated_kicks_and_small_beams_keep_the_one_call_path(ca):
    from cheetah_amd import _ops

    dt = torch.float32
    calls = []
    orig = _ops.sc_kick_sorted
    _ops.sc_kick_sorted = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        _linac(ca, dt, (32, 32, 32), 1).track(_beam(ca, dt, 100_000))            # one kick: nothing to amortise the sort over
        _linac(ca, dt, (32, 32, 32), 3).track(_beam(ca, dt, 20_000))             # below the sort threshold
        seg = _linac(ca, dt, (32, 32, 32), 2)
        seg.elements[6] = ca.SpaceChargeKick(torch.tensor(0.2, dtype=dt, device="cuda"), grid_shape=(16, 16, 16), dtype=dt, device="cuda")
        seg.track(_beam(ca, dt, 100_000))                                        # two kicks on different grids
    finally:
        _ops.sc_kick_sorted = orig
    assert calls == []


def _spy_links(fn):
    """(first, last) of every chain link `fn()` issues."""
    from cheetah_amd import _ops

    calls, orig = [], _ops.sc_kick_sorted
    _ops.sc_kick_sorted = lambda *a, **k: (calls.append((a[10], a[11])), orig(*a, **k))[1]
    try:
        out = fn()
    finally:
        _ops.sc_kick_sorted = orig
    return out, calls


@pytest.mark.parametrize("breaker", ["custom_map", "vector_quad", "parameter_quad", "long_run"])
def test_unfusable_run_ends_the_chain(ca, breaker):
    """A run between two kicks that cannot ride in the kick's particle pass (no device plan: a CustomTransferMap, vectorised
    settings, a trainable Parameter while gradients are enabled, more than 192 elements) must END the chain at the kick in front of it: the sums the gather pass
    leaves for the next kick's grid describe the rows BEFORE that run, so a chain that went on would build the next grid from
    stale beam sizes (silently wrong). Kicks behind the breaker may start a new chain."""
    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    grid = (32, 32, 32)
    sc = lambda: ca.SpaceChargeKick(t(0.2), grid_shape=grid, **kw)  # noqa: E731
    quad = lambda k: ca.Quadrupole(t(0.1), k1=t(k), **kw)  # noqa: E731
    if breaker == "custom_map":
        R = torch.eye(7, **kw)
        R[0, 1], R[2, 3], R[0, 0], R[2, 2] = 0.7, 0.4, 1.6, 0.5             # changes sigma_x / sigma_y a lot
        mid = [ca.CustomTransferMap(R, length=t(0.7), **kw)]
    elif breaker == "vector_quad":
        mid = [ca.Quadrupole(t(0.3), k1=torch.tensor([9.0], **kw), **kw), ca.Drift(t(0.8), **kw)]
    elif breaker == "parameter_quad":      # a trainable strength
        mid = [ca.Quadrupole(t(0.3), k1=torch.nn.Parameter(t(9.0)), **kw), ca.Drift(t(0.8), **kw)]
    else:
        mid = [ca.Drift(t(0.004), **kw) for _ in range(200)] + [quad(9.0), ca.Drift(t(0.6), **kw)]
    els = [sc(), ca.Drift(t(0.1), **kw), sc()] + mid + [sc(), quad(-4.0), ca.Drift(t(0.2), **kw), sc(), ca.Drift(t(0.1), **kw)]
    seg, beam = ca.Segment(els), _beam(ca, dt, 120_000, gaussian=True)
    if breaker == "vector_quad":
        # a vectorised run makes the beam behind it vectorised: no chain there (and none needed)
        out, calls = _spy_links(lambda: seg.track(beam))
        assert calls == [(True, False), (False, True)]
    elif breaker == "parameter_quad":
        # with gradients enabled the run with the trainable strength has no plan a track may use: the chain ends in front of it
        _, calls = _spy_links(lambda: seg.track(beam))
        assert calls[:2] == [(True, False), (False, True)], calls
        # under no_grad a trainable strength is a setting like any other (the plans read its value): ONE chain of four links
        with torch.no_grad():
            out, calls = _spy_links(lambda: seg.track(beam))
        assert calls == [(True, False), (False, False), (False, False), (False, True)], calls
    else:
        with torch.no_grad():
            out, calls = _spy_links(lambda: seg.track(beam))
        assert calls == [(True, False), (False, True), (True, False), (False, True)], calls
    with torch.no_grad():
        ref = _unchained(seg, beam)
    assert out.particles.shape == ref.particles.shape
    effect = (ref.particles - _one_by_one(_no_charge(ca, seg), beam).particles).abs().max(dim=-2).values.reshape(-1, 7).max(dim=0).values
    err = (out.particles - ref.particles).abs().max(dim=-2).values.reshape(-1, 7).max(dim=0).values
    ulp = 16 * torch.finfo(dt).eps * ref.particles.abs().max(dim=-2).values.reshape(-1, 7).max(dim=0).values
    assert torch.all(err <= 2e-4 * effect + ulp), (err / (2e-4 * effect + ulp))


def test_gradients_flow_through_a_lattice_of_kicks(ca):
    """[kick, Quad(k1 requires grad), kick, Drift, kick]: the run with the trainable strength is not fused, the particles behind
    it carry a graph, and every later kick must take its differentiable path — gradient equal to kick-by-kick tracking (measured:
    identical; the bound leaves room for the summation order of the deposits)."""
    dt = torch.float64
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    grid = (16, 16, 16)

    def lattice(k1):
        return ca.Segment([ca.SpaceChargeKick(t(0.2), grid_shape=grid, **kw), ca.Quadrupole(t(0.1), k1=k1, **kw), ca.Drift(t(0.3), **kw),
                           ca.SpaceChargeKick(t(0.2), grid_shape=grid, **kw), ca.Drift(t(0.2), **kw),
                           ca.SpaceChargeKick(t(0.2), grid_shape=grid, **kw), ca.Drift(t(0.1), **kw)])

    beam = _beam(ca, dt, 70_000, gaussian=True)
    k1 = torch.tensor(3.0, requires_grad=True, **kw)
    out, calls = _spy_links(lambda: lattice(k1).track(beam))
    assert calls == []                                  # no link: the first kick's run is not fusable
    assert out.particles.requires_grad
    out.sigma_x.backward()
    g_chain = float(k1.grad)
    k1b = torch.tensor(3.0, requires_grad=True, **kw)
    b = beam
    for e in lattice(k1b).elements:
        b = e.track(b)
    b.sigma_x.backward()
    assert g_chain != 0.0 and abs(g_chain - float(k1b.grad)) <= 1e-9 * abs(float(k1b.grad)), (g_chain, float(k1b.grad))
    # trainable strength BEHIND a fusable link: [kick, Drift, kick, Quad(k1), kick] — the first two kicks chain, the chain ends
    # in front of the trainable run and the third kick differentiates
    k1c = torch.tensor(3.0, requires_grad=True, **kw)
    seg = ca.Segment([ca.SpaceChargeKick(t(0.2), grid_shape=grid, **kw), ca.Drift(t(0.3), **kw),
                      ca.SpaceChargeKick(t(0.2), grid_shape=grid, **kw), ca.Quadrupole(t(0.1), k1=k1c, **kw), ca.Drift(t(0.2), **kw),
                      ca.SpaceChargeKick(t(0.2), grid_shape=grid, **kw), ca.Drift(t(0.1), **kw)])
    out, calls = _spy_links(lambda: seg.track(beam))
    assert calls == [(True, False), (False, True)]
    out.sigma_x.backward()
    k1d = torch.tensor(3.0, requires_grad=True, **kw)
    seg2 = ca.Segment([ca.SpaceChargeKick(t(0.2), grid_shape=grid, **kw), ca.Drift(t(0.3), **kw),
                       ca.SpaceChargeKick(t(0.2), grid_shape=grid, **kw), ca.Quadrupole(t(0.1), k1=k1d, **kw), ca.Drift(t(0.2), **kw),
                       ca.SpaceChargeKick(t(0.2), grid_shape=grid, **kw), ca.Drift(t(0.1), **kw)])
    b = beam
    for e in seg2.elements:
        b = e.track(b)
    b.sigma_x.backward()
    assert float(k1c.grad) != 0.0 and abs(float(k1c.grad) - float(k1d.grad)) <= 1e-6 * abs(float(k1d.grad)), (float(k1c.grad), float(k1d.grad))


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_dense_tile_chain_vs_reference(ca, dt):
    """The reference's default 32^3 grid under 400 000 particles, four kicks: the chain's deposit shares every tile among 16
    workgroups. Against the REFERENCE's own float64 run (tests/golden/sc_dense_tiles.npz, generator
    tests/golden/generate_golden_sc_dense_tiles.py; space_charge_kick.py:477-586): a 2000-particle sample of the outgoing beam,
    in units of the space-charge EFFECT on each coordinate (the difference to the same lattice without kicks), and the beam's
    first and second moments. Measured on the MI355X: float64 6.4e-11 of the effect at most, float32 6.5 ulp of a coordinate."""
    import os

    import numpy as np

    from tests import fullsize_inputs as fi

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sc_dense_tiles.npz"))
    N = int(g["n"])
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    x = torch.tensor(fi.c4_particles(N, seed=20260930), **kw)
    q = torch.tensor(fi.c4_charges(N) * (N / 1e6) * 2.5, **kw)
    beam = ca.ParticleBeam(x, t(fi.C4_ENERGY), particle_charges=q, **kw)
    els = []
    for cell in range(4):
        els += [ca.Drift(t(0.1), **kw), ca.SpaceChargeKick(t(0.2), grid_shape=(32, 32, 32), **kw), ca.Drift(t(0.1), **kw),
                ca.Quadrupole(t(0.1), k1=t(fi.c4_quad_k1(cell)), **kw), ca.Drift(t(0.1), **kw)]
    seg = ca.Segment(els)
    calls = []
    from cheetah_amd import _ops

    orig = _ops.sc_kick_sorted
    _ops.sc_kick_sorted = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            out = seg.track(beam)
    finally:
        _ops.sc_kick_sorted = orig
    assert len(calls) == 4                                                  # the chain, not kick by kick
    got = out.particles[::200].double().cpu().numpy()
    ref, free = g["out_sample"], g["free_sample"]
    effect = np.abs(ref - free)[:, :6].max(axis=0)
    assert effect[1] > 1e-7 and effect[3] > 1e-7                             # the kicks do act on px, py
    scale = np.abs(ref)[:, :6].max(axis=0)
    diff = np.abs(got - ref)[:, :6].max(axis=0)
    if dt == torch.float64:
        err = (diff / effect).max()                    # measured 6.4e-11 (tau; 2e-13 on the transverse coordinates)
        assert err < 2.6e-10, diff / effect
    else:
        # float32: the rounding of a coordinate's own storage through twenty elements dominates — measured 6.5 ulp of the
        # coordinate's scale at most, i.e. 1.4e-4 of the space-charge effect on x and y
        err = (diff / (float(torch.finfo(dt).eps) * scale)).max()
        assert err < 26, diff / (float(torch.finfo(dt).eps) * scale)
        assert (diff / effect)[:4].max() < 6e-4
    p = out.particles[:, :6].double()
    assert np.allclose(p.mean(dim=0).cpu().numpy(), g["mean"], rtol=0, atol=(1e-9 if dt == torch.float64 else 2e-4) * np.abs(g["std"]).max())
    assert np.allclose(p.std(dim=0).cpu().numpy(), g["std"], rtol=1e-9 if dt == torch.float64 else 2e-5, atol=0)
    print(f"dense-tile chain vs reference ({dt}): {err:.2e}")
