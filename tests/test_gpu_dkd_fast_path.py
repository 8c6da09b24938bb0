"""The short drift-kick-drift path (Element._track_drift_kick_drift for one plain beam and scalar settings: cached parameter
array + chx_dkd_track_p) against the general path (_ops.dkd_track), and the invalidation of the cached array
(reference behaviour: /root/reference/cheetah/accelerator/quadrupole.py:174-240, drift.py:106-154 re-read their settings on
every call)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _general(el, beam):
    """The same element through the general path (the cached array switched off)."""
    orig = type(el)._dkd_params_stacked
    type(el)._dkd_params_stacked = lambda self, dtype, device: None
    try:
        return el.track(beam)
    finally:
        type(el)._dkd_params_stacked = orig


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("precision", ["double", "storage"])
def test_short_path_equals_general_path_and_follows_the_settings(dt, precision):
    import cheetah_amd as ca

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(2)
    beam = ca.ParticleBeam.from_parameters(num_particles=20_000, sigma_x=t(2e-4), sigma_px=t(3e-5), sigma_p=t(1e-3), energy=t(8e7), **kw)
    quad = ca.Quadrupole(t(0.3), k1=t(2.5), misalignment=t([1e-4, -2e-4]), tilt=t(0.1), num_steps=3, tracking_method="drift_kick_drift", **kw)
    drift = ca.Drift(t(0.7), tracking_method="drift_kick_drift", **kw)
    for el in (quad, drift):
        el.dkd_precision = precision
        a, b = el.track(beam), _general(el, beam)
        assert el.__dict__["_dkd_cache"] is not None
        assert torch.equal(a.particles, b.particles) and torch.equal(a.energy, b.energy) and torch.equal(a.s, b.s)
        assert a.particles.shape == beam.particles.shape and a.energy.shape == beam.energy.shape
    first = quad.track(beam).particles.clone()
    # in-place edit (version counter), re-assignment (revision), both seen
    quad.k1.fill_(-1.0)
    edited = quad.track(beam)
    assert not torch.equal(edited.particles, first) and torch.equal(edited.particles, _general(quad, beam).particles)
    quad.k1 = t(2.5)
    assert torch.equal(quad.track(beam).particles, first)
    quad.misalignment[1] = 5e-4
    assert torch.equal(quad.track(beam).particles, _general(quad, beam).particles)
    # a setting that wants a gradient takes the differentiable path, and gets one
    quad.k1 = torch.nn.Parameter(t(2.5))
    out = quad.track(beam)
    out.particles[:, 1].square().sum().backward()
    assert quad.k1.grad is not None and float(quad.k1.grad.abs()) > 0
    with torch.no_grad():
        assert torch.isfinite(quad.track(beam).particles).all()
    # a copy starts without the cached array (derived state) and tracks the same
    clone = copy.deepcopy(drift)
    assert clone.__dict__["_dkd_cache"] is None
    assert torch.equal(clone.track(beam).particles, drift.track(beam).particles)
    # vector settings, a float64 element on a float32 beam: the general path as before
    wide = ca.Quadrupole(t(0.3), k1=t([1.0, 2.0]), tracking_method="drift_kick_drift", **kw)
    assert wide.track(beam).particles.shape == (2, 20_000, 7) and wide.__dict__.get("_dkd_cache") is None
    if dt == torch.float32:
        d64 = ca.Drift(torch.tensor(0.7, dtype=torch.float64, device="cuda"), tracking_method="drift_kick_drift",
                       dtype=torch.float64, device="cuda")
        assert d64.track(beam).particles.dtype == torch.float64


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_segment_tracks_a_drift_kick_drift_run_in_one_call(dt):
    """Segment.track hands consecutive drift-kick-drift elements to chx_dkd_chain: same particles, energy and s — bit for bit —
    as the elements tracked one after the other; elements that do not qualify (a vectorised setting, a Marker, a linear
    element, a setting that wants a gradient) end a run and are tracked as before."""
    import cheetah_amd as ca
    from cheetah_amd import _ops

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(4)
    beam = ca.ParticleBeam.from_parameters(num_particles=30_001, sigma_x=t(2e-4), sigma_px=t(3e-5), sigma_p=t(1e-3), energy=t(8e7), **kw)
    dkd = {"tracking_method": "drift_kick_drift"}
    els = [ca.Drift(t(0.31), **dkd, **kw), ca.Quadrupole(t(0.2), k1=t(3.1), num_steps=4, **dkd, **kw), ca.Drift(t(0.77), **dkd, **kw),
           ca.Quadrupole(t(0.2), k1=t(-2.9), tilt=t(0.2), misalignment=t([1e-4, 2e-4]), **dkd, **kw),
           ca.Dipole(t(0.5), angle=t(0.02), **dkd, **kw), ca.Drift(t(0.13), **dkd, **kw),
           ca.Marker(name="m"), ca.Drift(t(0.4), **kw),                      # a pass-through element and a linear one
           ca.Drift(t(0.21), **dkd, **kw), ca.Quadrupole(t(0.1), k1=t(1.0), **dkd, **kw), ca.Drift(t(0.5), **dkd, **kw)]
    els[1].dkd_precision = "storage"
    seg = ca.Segment(els)
    calls, orig = [], _ops.dkd_chain
    _ops.dkd_chain = lambda *a, **k: (calls.append(len(a[0])), orig(*a, **k))[1]
    try:
        out = seg.track(beam)
    finally:
        _ops.dkd_chain = orig
    # (a float32 run is cut where a stretch of one arithmetic mode begins: that stretch keeps its particles in registers, and
    # the run [Marker, linear Drift] rides in it)
    assert calls == ([2, 8] if dt == torch.float32 else [10]), calls
    ref = beam
    for e in els:
        ref = e.track(ref)
    assert torch.equal(out.particles, ref.particles) and torch.equal(out.energy, ref.energy) and torch.equal(out.s, ref.s)
    assert out.energy.shape == () and out.s.shape == ()
    # a vectorised setting in the middle: the run stops in front of it
    els[2].length = t([0.7, 0.8])
    calls.clear()
    _ops.dkd_chain = lambda *a, **k: (calls.append(len(a[0])), orig(*a, **k))[1]
    try:
        out = seg.track(beam)
    finally:
        _ops.dkd_chain = orig
    assert calls == [2], calls                                    # (behind it the beam is vectorised: no run)
    ref = beam
    for e in els:
        ref = e.track(ref)
    assert torch.equal(out.particles, ref.particles) and torch.equal(out.s, ref.s)
    els[2].length = t(0.77)
    # gradients: the differentiable path, same value
    els[1].k1 = torch.nn.Parameter(t(3.1))
    out = seg.track(beam)
    out.particles[:, 0].square().mean().backward()
    assert els[1].k1.grad is not None and float(els[1].k1.grad.abs()) > 0


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_segment_tracks_a_second_order_run_in_one_call(dt):
    """Consecutive elements tracked with their second-order maps go to the device in one chx_second_order_chain call: particles and
    s equal — bit for bit — the elements tracked one after the other (element.py:195-228); a sextupole and a dipole take part,
    a vectorised setting or a setting with a gradient ends the run."""
    import cheetah_amd as ca
    from cheetah_amd import _ops

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(6)
    beam = ca.ParticleBeam.from_parameters(num_particles=25_003, sigma_x=t(3e-4), sigma_px=t(4e-5), sigma_p=t(2e-3), energy=t(6e7), **kw)
    so = {"tracking_method": "second_order"}
    els = [ca.Drift(t(0.4), **so, **kw), ca.Quadrupole(t(0.2), k1=t(3.3), tilt=t(0.1), **so, **kw), ca.Drift(t(0.6), **so, **kw),
           ca.Sextupole(t(0.15), k2=t(25.0), **so, **kw), ca.Dipole(t(0.5), angle=t(0.03), **so, **kw), ca.Drift(t(0.2), **so, **kw),
           ca.Marker(name="m"), ca.Quadrupole(t(0.2), k1=t(-2.0), **so, **kw), ca.Drift(t(0.3), **so, **kw)]
    seg = ca.Segment(els)
    calls, orig = [], _ops.second_order_chain
    _ops.second_order_chain = lambda *a, **k: (calls.append(len(a[0])), orig(*a, **k))[1]
    try:
        out = seg.track(beam)
        assert calls == [8], calls                        # (the Marker between two of them does not end the stretch)
        ref = beam
        for e in els:
            ref = e.track(ref)
        assert torch.equal(out.particles, ref.particles) and torch.equal(out.s, ref.s) and torch.equal(out.energy, ref.energy)
        els[2].length = t([0.6, 0.7])                       # vectorised: ends the first run in front of it
        calls.clear()
        out = seg.track(beam)
        assert calls == [2], calls
        ref = beam
        for e in els:
            ref = e.track(ref)
        assert torch.equal(out.particles, ref.particles) and torch.equal(out.s, ref.s)
        els[2].length = t(0.6)
        els[1].k1 = torch.nn.Parameter(t(3.3))              # a gradient: the differentiable path
        calls.clear()
        out = seg.track(beam)
        out.particles[:, 0].square().mean().backward()
        assert els[1].k1.grad is not None and float(els[1].k1.grad.abs()) > 0
    finally:
        _ops.second_order_chain = orig


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_second_order_run_cache_follows_every_kind_of_change(dt):
    """A second-order run is looked up once and reused while nothing changed (Segment._second_order_run). What can change —
    a setting edited in place, an attribute assigned, a setting that starts to require a gradient, another energy tensor, another
    species — must be seen: every track equals the elements tracked one after the other (element.py:195-228), bit for bit."""
    import cheetah_amd as ca

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(8)
    beam = ca.ParticleBeam.from_parameters(num_particles=30_011, sigma_x=t(3e-4), sigma_px=t(4e-5), sigma_p=t(2e-3), energy=t(6e7), **kw)
    so = {"tracking_method": "second_order"}
    els = [ca.Drift(t(0.4), **so, **kw), ca.Quadrupole(t(0.2), k1=t(3.3), **so, **kw), ca.Drift(t(0.6), **so, **kw),
           ca.Dipole(t(0.5), angle=t(0.03), **so, **kw), ca.Quadrupole(t(0.2), k1=t(-2.0), **so, **kw), ca.Drift(t(0.3), **so, **kw)]
    seg = ca.Segment(els)

    def check(b):
        out = seg.track(b)
        ref = b
        for e in els:
            ref = e.track(ref)
        assert torch.equal(out.particles, ref.particles) and torch.equal(out.s, ref.s) and torch.equal(out.energy, ref.energy)
        return out

    first = check(beam)
    again = check(beam)                                   # from the cache
    assert torch.equal(first.particles, again.particles)
    assert seg.__dict__["_so_run_cache"][1], "the run was not kept"
    els[1].k1.mul_(1.5)                                   # in place
    edited = check(beam)
    assert not torch.equal(edited.particles, first.particles)
    els[2].length.add_(0.25)
    check(beam)
    els[4].k1 = t(-1.0)                                   # assigned
    check(beam)
    other = ca.ParticleBeam(beam.particles.clone(), t(7.5e7), particle_charges=beam.particle_charges, species=beam.species, **kw)
    check(other)                                          # another energy tensor
    check(beam)
    beam.energy.mul_(1.1)                                 # the same tensor, another value
    check(beam)
    protons = ca.ParticleBeam(beam.particles.clone(), t(2e9), particle_charges=beam.particle_charges, species=ca.Species("proton"), **kw)
    check(protons)
    check(beam)
    els[1].k1.requires_grad_(True)                        # no assignment, no version change: the differentiable path must take over
    out = seg.track(beam)
    out.particles[:, 0].square().mean().backward()
    assert els[1].k1.grad is not None and float(els[1].k1.grad.abs()) > 0
    with torch.no_grad():
        check(beam)


@pytest.mark.parametrize("precision,dt", [("mixed", torch.float32), ("double", torch.float32), ("storage", torch.float32),
                                          ("mixed", torch.float64)])
def test_drifts_and_quadrupoles_in_registers_equal_the_elements_one_by_one(precision, dt):
    """A run of Drifts, Quadrupoles and Dipoles (float32 beams: of one arithmetic mode) is two launches with the particles in registers
    (chx_dkd_chain -> dkd_chain_kernel): particles, energy and s equal the elements tracked one after the other (drift.py:106-154,
    quadrupole.py:174-240, dipole.py:183-370) bit for bit — low energy (the reference energy's float32 round trip matters), several steps, a tilted
    and a shifted quadrupole, a tile that is not full, a beam so small that the elements' constants need several passes."""
    import cheetah_amd as ca
    from cheetah_amd import _ops

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(12)
    dkd = {"tracking_method": "drift_kick_drift"}
    for energy, n in ((4.2e6, 20_011), (1.3e9, 777), (6e7, 70)):       # (70 particles: the run goes through in three passes)
        beam = ca.ParticleBeam.from_parameters(num_particles=n, sigma_x=t(2e-4), sigma_px=t(3e-5), sigma_y=t(1e-4), sigma_py=t(2e-5),
                                               sigma_tau=t(1e-4), sigma_p=t(2e-3), energy=t(energy), **kw)
        els = []
        for c in range(6):
            els += [ca.Quadrupole(t(0.2), k1=t(3.0 + 0.1 * c), num_steps=1 + c % 3, **dkd, **kw), ca.Drift(t(0.5 + 0.01 * c), **dkd, **kw),
                    ca.Quadrupole(t(0.15), k1=t(-2.5), tilt=t(0.05 * c), **dkd, **kw), ca.Drift(t(0.3), **dkd, **kw)]
        els[6] = ca.Quadrupole(t(0.2), k1=t(1.5), misalignment=t([2e-4, -1e-4]), **dkd, **kw)
        els[10] = ca.Quadrupole(t(0.2), k1=t(0.0), **dkd, **kw)
        els[14] = ca.Dipole(t(0.4), angle=t(0.05), dipole_e1=t(0.01), dipole_e2=t(0.02), fringe_integral=t(0.5), gap=t(0.02),
                            **dkd, **kw)
        els[18] = ca.Dipole(t(0.3), angle=t(-0.03), tilt=t(0.2), fringe_at="entrance", **dkd, **kw)
        els[19] = ca.RBend(t(0.3), angle=t(0.02), rbend_e1=t(0.004), **dkd, **kw)
        for e in els:
            e.dkd_precision = precision
        seg = ca.Segment(els)
        calls, orig = [], _ops.dkd_chain
        _ops.dkd_chain = lambda *a, **k: (calls.append(len(a[0])), orig(*a, **k))[1]
        try:
            out = seg.track(beam)
            again = seg.track(beam)                    # the run from Segment's cache
        finally:
            _ops.dkd_chain = orig
        assert calls == [24, 24], calls
        ref = beam
        for e in els:
            ref = e.track(ref)
        for o in (out, again):
            assert torch.equal(o.particles, ref.particles) and torch.equal(o.energy, ref.energy) and torch.equal(o.s, ref.s)
        assert torch.isfinite(out.particles).all()


def test_drift_kick_drift_run_cache_follows_every_kind_of_change():
    """Segment keeps a drift-kick-drift run as it found it (Segment._dkd_run) while nothing changed. A setting edited in place,
    an attribute assigned (a setting, the arithmetic mode, the step count), a setting that starts to require a gradient, another
    energy or species: every track still equals the elements tracked one after the other (the reference re-reads its settings on
    every call, quadrupole.py:174-240)."""
    import cheetah_amd as ca

    kw = {"dtype": torch.float32, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(13)
    dkd = {"tracking_method": "drift_kick_drift"}
    beam = ca.ParticleBeam.from_parameters(num_particles=10_007, sigma_x=t(2e-4), sigma_px=t(3e-5), sigma_p=t(1e-3), energy=t(9e7), **kw)
    els = [ca.Drift(t(0.3), **dkd, **kw), ca.Quadrupole(t(0.2), k1=t(3.1), **dkd, **kw), ca.Drift(t(0.7), **dkd, **kw),
           ca.Quadrupole(t(0.2), k1=t(-2.9), **dkd, **kw), ca.Drift(t(0.2), **dkd, **kw)]
    seg = ca.Segment(els)

    def check(b):
        out = seg.track(b)
        ref = b
        for e in els:
            ref = e.track(ref)
        assert torch.equal(out.particles, ref.particles) and torch.equal(out.s, ref.s) and torch.equal(out.energy, ref.energy)
        return out

    first = check(beam)
    assert torch.equal(check(beam).particles, first.particles)
    assert seg.__dict__["_dkd_run_cache"][1], "the run was not kept"
    els[1].k1.mul_(1.5)                                   # in place
    assert not torch.equal(check(beam).particles, first.particles)
    els[2].length.add_(0.25)
    check(beam)
    els[3].k1 = t(-1.0)                                   # assigned
    check(beam)
    els[3].num_steps = 3
    check(beam)
    els[1].dkd_precision = "double"                       # the run is cut differently now
    check(beam)
    els[1].dkd_precision = "mixed"
    check(beam)
    other = ca.ParticleBeam(beam.particles.clone(), t(7.5e7), particle_charges=beam.particle_charges, species=beam.species, **kw)
    check(other)
    protons = ca.ParticleBeam(beam.particles.clone(), t(2e9), particle_charges=beam.particle_charges, species=ca.Species("proton"), **kw)
    check(protons)
    check(beam)
    els[1].k1.requires_grad_(True)                        # no assignment, no version change: the differentiable path takes over
    out = seg.track(beam)
    out.particles[:, 0].square().mean().backward()
    assert els[1].k1.grad is not None and float(els[1].k1.grad.abs()) > 0
    with torch.no_grad():
        check(beam)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_long_second_order_run_on_a_small_beam_goes_through_in_several_passes(dt):
    """chx_second_order_chain keeps the folded coefficients of a run in the scratch rows of the beam (256 coefficients per element): 150
    particles hold four elements per pass, later passes run in place. Same bits as the elements one by one (element.py:195-228)."""
    import cheetah_amd as ca

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(21)
    so = {"tracking_method": "second_order"}
    for n in (150, 9):                                  # (9 particles: no room for two elements — element by element)
        beam = ca.ParticleBeam.from_parameters(num_particles=n, sigma_x=t(3e-4), sigma_px=t(4e-5), sigma_p=t(2e-3), energy=t(6e7), **kw)
        els = [ca.Drift(t(0.4), **so, **kw), ca.Quadrupole(t(0.2), k1=t(3.3), **so, **kw), ca.Drift(t(0.6), **so, **kw),
               ca.Dipole(t(0.5), angle=t(0.03), **so, **kw), ca.Quadrupole(t(0.2), k1=t(-2.0), tilt=t(0.3), **so, **kw),
               ca.Drift(t(0.3), **so, **kw), ca.Sextupole(t(0.1), k2=t(12.0), **so, **kw)]
        out = ca.Segment(els).track(beam)
        ref = beam
        for e in els:
            ref = e.track(ref)
        assert torch.equal(out.particles, ref.particles) and torch.equal(out.s, ref.s)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_linear_runs_between_second_order_elements_ride_in_the_same_pass(dt):
    """A lattice whose drifts are tracked linearly (the default) and whose magnets to second order: the merged runs of linear
    elements between the second-order elements go through `chx_second_order_chain_mixed` with them — one pass over the beam.
    Same particles and s as the walk piece by piece (segment.py:545-574: a run of skippable elements is tracked with its composed
    map; element.py:195-228 for the others), also after edits of a linear element's setting and of a magnet's."""
    import cheetah_amd as ca
    from cheetah_amd import _ops

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(31)
    so = {"tracking_method": "second_order"}
    beam = ca.ParticleBeam.from_parameters(num_particles=40_009, sigma_x=t(3e-4), sigma_px=t(4e-5), sigma_p=t(2e-3), energy=t(6e7), **kw)
    pieces = [[ca.Drift(t(0.4), **kw), ca.Quadrupole(t(0.1), k1=t(0.7), **kw), ca.Drift(t(0.2), **kw)],      # a linear run of three
              [ca.Quadrupole(t(0.2), k1=t(3.3), **so, **kw)],
              [ca.Drift(t(0.6), **kw)],
              [ca.Sextupole(t(0.15), k2=t(25.0), **so, **kw)],
              [ca.Dipole(t(0.5), angle=t(0.03), **so, **kw)],
              [ca.Drift(t(0.3), **kw), ca.HorizontalCorrector(t(0.1), angle=t(1e-4), **kw)],
              [ca.Quadrupole(t(0.2), k1=t(-2.0), tilt=t(0.2), **so, **kw)],
              [ca.Drift(t(0.25), **kw)]]
    els = [e for piece in pieces for e in piece]
    seg = ca.Segment(els)
    subs = [ca.Segment(piece) if len(piece) > 1 or piece[0].tracking_method == "linear" else piece[0] for piece in pieces]
    calls, orig = [], _ops.second_order_chain
    _ops.second_order_chain = lambda *a, **k: (calls.append(len(a[0])), orig(*a, **k))[1]

    def check():
        out = seg.track(beam)
        ref = beam
        for piece in subs:
            ref = piece.track(ref)
        assert torch.equal(out.particles, ref.particles) and torch.equal(out.s, ref.s) and torch.equal(out.energy, ref.energy)
        return out

    try:
        first = check()
        assert calls == [8], calls                       # every piece in ONE call
        check()                                          # the stretch from Segment's cache
        pieces[0][1].k1.mul_(2.0)                        # a linear element inside a run, in place
        assert not torch.equal(check().particles, first.particles)
        pieces[2][0].length = t(0.9)                     # assigned
        check()
        pieces[3][0].k2.mul_(0.5)                        # a second-order magnet, in place
        check()
        with torch.no_grad():
            check()
    finally:
        _ops.second_order_chain = orig


@pytest.mark.parametrize("dt,precision", [(torch.float32, "mixed"), (torch.float32, "storage"), (torch.float64, "mixed")])
def test_linear_runs_between_drift_kick_drift_elements_ride_in_the_same_pass(dt, precision):
    """A lattice whose drifts are tracked linearly (the default) and whose magnets with the Bmad-X maps: the merged runs of linear
    elements between them go through `chx_dkd_chain_mixed` with them — one pass over the beam; each run's map is built for the
    reference energy in front of it (every drift-kick-drift element hands on the float round trip of its own, bmadx.py:49).
    Same particles, energy and s as the walk piece by piece (segment.py:545-574), also after edits; a lone Marker between two
    magnets does not end the stretch."""
    import cheetah_amd as ca
    from cheetah_amd import _ops

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(33)
    dkd = {"tracking_method": "drift_kick_drift"}
    beam = ca.ParticleBeam.from_parameters(num_particles=30_011, sigma_x=t(3e-4), sigma_px=t(4e-5), sigma_p=t(2e-3), energy=t(4.7e6), **kw)
    pieces = [[ca.Drift(t(0.4), **kw), ca.Quadrupole(t(0.1), k1=t(0.7), **kw), ca.Drift(t(0.2), **kw)],      # a linear run of three
              [ca.Quadrupole(t(0.2), k1=t(3.3), num_steps=3, **dkd, **kw)],
              [ca.Drift(t(0.6), **kw)],
              [ca.Dipole(t(0.5), angle=t(0.03), dipole_e1=t(0.01), **dkd, **kw)],
              [ca.Marker(name="between")],
              [ca.Quadrupole(t(0.2), k1=t(-2.0), tilt=t(0.2), **dkd, **kw)],
              [ca.Drift(t(0.3), **kw), ca.VerticalCorrector(t(0.1), angle=t(1e-4), **kw)],
              [ca.Drift(t(0.25), **dkd, **kw)],
              [ca.Drift(t(0.15), **kw)]]
    els = [e for piece in pieces for e in piece]
    for e in els:
        e.dkd_precision = precision
    seg = ca.Segment(els)
    subs = [ca.Segment(piece) if len(piece) > 1 or piece[0].tracking_method == "linear" else piece[0] for piece in pieces]
    calls, orig = [], _ops.dkd_chain
    _ops.dkd_chain = lambda *a, **k: (calls.append(len(a[0])), orig(*a, **k))[1]

    def check():
        out = seg.track(beam)
        ref = beam
        for piece in subs:
            ref = piece.track(ref)
        assert torch.equal(out.particles, ref.particles) and torch.equal(out.s, ref.s) and torch.equal(out.energy, ref.energy)
        return out

    try:
        first = check()
        assert calls == [8], calls                       # every piece but the Marker, in ONE call
        check()                                          # the stretch from Segment's cache
        pieces[0][1].k1.mul_(2.0)                        # a linear element inside a run, in place
        assert not torch.equal(check().particles, first.particles)
        pieces[2][0].length = t(0.9)                     # assigned
        check()
        pieces[1][0].k1.mul_(0.5)                        # a magnet, in place
        check()
        beam.energy.mul_(1.5)                            # the linear maps depend on it
        check()
        with torch.no_grad():
            check()
    finally:
        _ops.dkd_chain = orig


@pytest.mark.parametrize("method", ["second_order", "drift_kick_drift"])
def test_mixed_stretches_on_tiny_beams_and_behind_each_other(method):
    """(a) A beam too small to hold the stretch's constants in its scratch rows takes the element passes inside the same C call
    (chx_apply_affine7 for the linear runs, the energy handed on by a copy): same numbers. (b) Two stretches with an active Screen
    between them: the second one's linear maps are built for the energy the first one leaves; tracked repeatedly (the stretches
    come from Segment's cache, the first hands the same energy tensor on while its input stands) and after the beam energy was
    edited in place, the result equals the walk piece by piece (segment.py:545-574)."""
    import cheetah_amd as ca

    kw = {"dtype": torch.float32, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(41)
    nl = {"tracking_method": method}

    def cell(k):
        return [[ca.Drift(t(0.3 + 0.01 * k), **kw)], [ca.Quadrupole(t(0.2), k1=t(2.0 + 0.1 * k), **nl, **kw)],
                [ca.Drift(t(0.4), **kw), ca.HorizontalCorrector(t(0.05), angle=t(1e-4), **kw)], [ca.Quadrupole(t(0.2), k1=t(-2.2), **nl, **kw)]]

    screen = ca.Screen(resolution=(64, 48), pixel_size=t([2e-5, 2e-5]), is_active=True, name="mid", **kw)
    pieces = cell(0) + cell(1) + [[screen]] + cell(2) + cell(3) + [[ca.Drift(t(0.2), **kw)]]
    seg = ca.Segment([e for piece in pieces for e in piece])
    subs = [ca.Segment(piece) if len(piece) > 1 or piece[0].tracking_method == "linear" and piece[0] is not screen else piece[0]
            for piece in pieces]
    for n in (9, 20_003):
        beam = ca.ParticleBeam.from_parameters(num_particles=n, sigma_x=t(2e-4), sigma_px=t(3e-5), sigma_p=t(1e-3), energy=t(5.1e6), **kw)
        for repeat in range(4):
            if repeat == 3:
                beam.energy.mul_(1.3)
            out = seg.track(beam)
            ref = beam
            for piece in subs:
                ref = piece.track(ref)
            assert torch.equal(out.particles, ref.particles) and torch.equal(out.s, ref.s) and torch.equal(out.energy, ref.energy), (n, repeat)
