"""The README-style control loop: every step ASSIGNS new tensors to a few settings (README.md:73-77 of the reference:
`segment.AREAMQZM1.k1 = torch.tensor(8.2)`), tracks and reads the screen. Round 6: such an assignment — a plain tensor in the place
of a registered buffer of the same dtype, device and shape — changes nothing but an address; the plans that hold the element's
addresses take the new one on the spot (Element._absorb -> _FastRun.absorb), the lattice's partition stands (Element._hard_epoch)
and the stretch's table is re-sent in the arguments of one small launch (chx_table_store). Anything else that is assigned — another
dtype, a (B,) tensor, a tensor with a graph — takes the full re-derivation as before. The checker for every step is a freshly built
lattice with the same settings (its plans derived from scratch)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _subcell(ca, dt, k1s=(8.2, -14.3, 3.142), angles=(9e-5, -1e-4), active_screen=True):
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    return ca.Segment([
        ca.Marker(name="AREASOLA1", **kw), ca.Drift(t(0.17504)), ca.Quadrupole(t(0.122), k1=t(k1s[0]), name="AREAMQZM1", **kw),
        ca.Drift(t(0.428)), ca.Quadrupole(t(0.122), k1=t(k1s[1]), name="AREAMQZM2", **kw), ca.Drift(t(0.204)),
        ca.VerticalCorrector(t(0.02), angle=t(angles[0]), name="AREAMCVM1", **kw), ca.Drift(t(0.204)),
        ca.Quadrupole(t(0.122), k1=t(k1s[2]), name="AREAMQZM3", **kw), ca.Drift(t(0.179)),
        ca.HorizontalCorrector(t(0.02), angle=t(angles[1]), name="AREAMCHM1", **kw), ca.Drift(t(0.45)),
        ca.Screen(resolution=(64, 48), pixel_size=t([6e-5, 6e-5]), is_active=active_screen, name="AREABSCR1", **kw)])


def _assign(seg, vals):
    seg.AREAMQZM1.k1, seg.AREAMQZM2.k1, seg.AREAMCVM1.angle, seg.AREAMQZM3.k1, seg.AREAMCHM1.angle = vals


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("kind", ["particles", "parameters"])
def test_control_loop_with_assigned_settings_equals_fresh_lattices(dt, kind):
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    kw = {"dtype": dt, "device": "cuda"}
    torch.manual_seed(0)
    beam = (ca.ParticleBeam.from_twiss(beta_x=torch.tensor(3.14, **kw), beta_y=torch.tensor(42.0, **kw), num_particles=5000, **kw)
            if kind == "particles" else ca.ParameterBeam.from_twiss(beta_x=torch.tensor(3.14, **kw), beta_y=torch.tensor(42.0, **kw), **kw))
    seg = _subcell(ca, dt)
    actions = torch.randn(12, 5, **kw)
    reads = {"n": 0}
    real_read = segment._FastRun._read

    def counting_read(self, e, i):
        reads["n"] += 1
        return real_read(self, e, i)

    segment._FastRun._read = counting_read
    try:
        with torch.no_grad():
            seg.track(beam)                                  # plans are derived here
            for step in range(12):
                a = actions[step]
                vals = [a[0] * 10, a[1] * 10, a[2] * 1e-4, a[3] * 10, a[4] * 1e-4]
                before = reads["n"]
                hard = ca.Element._hard_epoch
                _assign(seg, vals)
                out = seg.track(beam)
                img = seg.AREABSCR1.reading
                assert reads["n"] == before, "an assigned address was re-derived instead of patched"
                assert ca.Element._hard_epoch == hard
                segment._FastRun._read = real_read
                fresh = _subcell(ca, dt)
                _assign(fresh, [v.clone() for v in vals])
                want = fresh.track(beam)
                want_img = fresh.AREABSCR1.reading
                segment._FastRun._read = counting_read
                if kind == "particles":
                    assert torch.equal(out.particles, want.particles), step
                else:
                    assert torch.equal(out.mu, want.mu) and torch.equal(out.cov, want.cov), step
                assert torch.equal(img, want_img) or torch.allclose(img, want_img, rtol=1e-5, atol=0), step
                assert torch.equal(out.s, want.s)
            # an in-place edit of an assigned tensor is followed (the device reads the setting through its address)
            vals[0].mul_(0.5)
            out = seg.track(beam)
            fresh = _subcell(ca, dt)
            _assign(fresh, [v.clone() for v in vals])
            want = fresh.track(beam)
            assert torch.equal(out.particles, want.particles) if kind == "particles" else torch.equal(out.cov, want.cov)
    finally:
        segment._FastRun._read = real_read


def test_plans_verify_after_absorbed_assignments():
    """CHX_CHECK_PLANS's re-derivation agrees with the patched plans (every address of every run and of the stretch's table)."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    beam = ca.ParticleBeam.from_parameters(num_particles=2000, **kw)
    seg = _subcell(ca, dt)
    old = segment._CHECK_PLANS
    try:
        with torch.no_grad():
            seg.track(beam)
            for step in range(4):
                _assign(seg, [torch.tensor(v, **kw) for v in (1.0 + step, -2.0, 1e-5 * step, 0.5, -3e-5)])
                segment._CHECK_PLANS = True          # the next track re-derives every plan it uses and compares
                seg.track(beam)
                segment._CHECK_PLANS = False
            # a storage swapped behind the host's back is still found
            seg.AREAMQZM1.k1.data = torch.tensor(7.0, **kw)
            segment._CHECK_PLANS = True
            with pytest.raises(RuntimeError, match="storage"):
                seg.track(beam)
    finally:
        segment._CHECK_PLANS = old


def test_assignments_that_are_more_than_an_address_take_the_full_path():
    import cheetah_amd as ca

    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    beam = ca.ParticleBeam.from_parameters(num_particles=3000, **kw)

    def both(mutate, check):
        seg = _subcell(ca, dt)
        with torch.no_grad():
            seg.track(beam)
            _assign(seg, [torch.tensor(v, **kw) for v in (1.0, -2.0, 1e-5, 0.5, -3e-5)])   # a soft step first: plans patched
            seg.track(beam)
        hard = ca.Element._hard_epoch
        mutate(seg)
        assert ca.Element._hard_epoch > hard
        fresh = _subcell(ca, dt)
        _assign(fresh, [torch.tensor(v, **kw) for v in (1.0, -2.0, 1e-5, 0.5, -3e-5)])
        mutate(fresh)
        check(seg, fresh)

    # a vector of strengths: the outgoing beam is vectorised
    def vec(s):
        s.AREAMQZM2.k1 = torch.tensor([-2.0, -1.0, 0.5], **kw)

    def same_particles(a, b):
        with torch.no_grad():
            pa, pb = a.track(beam).particles, b.track(beam).particles
        assert pa.shape == pb.shape and torch.equal(pa, pb)

    both(vec, same_particles)

    # a tensor that carries a graph: gradients reach it
    def trainable(s):
        s.AREAMQZM3.k1 = torch.tensor(0.5, requires_grad=True, **kw)

    def same_grad(a, b):
        ga = torch.autograd.grad(a.track(beam).sigma_x, a.AREAMQZM3.k1)[0]
        gb = torch.autograd.grad(b.track(beam).sigma_x, b.AREAMQZM3.k1)[0]
        assert float(ga) == pytest.approx(float(gb), rel=1e-6) and float(ga) != 0.0

    both(trainable, same_grad)

    # a non-tensor attribute (an element switched off / on): structure
    def deactivate(s):
        s.AREABSCR1.is_active = False

    both(deactivate, same_particles)

    # an nn.Parameter in the place of a buffer
    def parameter(s):
        s.AREAMQZM1.k1 = torch.nn.Parameter(torch.tensor(2.5, **kw))

    both(parameter, same_particles)


def test_two_lattices_sharing_an_element():
    """One element object in two lattices: both lattices' plans registered with it, an assignment patches both."""
    import cheetah_amd as ca

    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    quad = ca.Quadrupole(t(0.2), k1=t(3.0), name="shared", **kw)
    a = ca.Segment([ca.Drift(t(0.3), **kw), quad, ca.Drift(t(0.5), **kw)])
    b = ca.Segment([ca.Drift(t(1.0), **kw), quad, ca.Drift(t(0.1), **kw), ca.Quadrupole(t(0.1), k1=t(-1.0), **kw)])
    beam = ca.ParticleBeam.from_parameters(num_particles=1000, **kw)
    with torch.no_grad():
        a.track(beam), b.track(beam)
        for v in (1.0, -4.0, 2.5):
            quad.k1 = t(v)
            fa = ca.Segment([ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=t(v), **kw), ca.Drift(t(0.5), **kw)])
            fb = ca.Segment([ca.Drift(t(1.0), **kw), ca.Quadrupole(t(0.2), k1=t(v), **kw), ca.Drift(t(0.1), **kw),
                             ca.Quadrupole(t(0.1), k1=t(-1.0), **kw)])
            assert torch.equal(a.track(beam).particles, fa.track(beam).particles)
            assert torch.equal(b.track(beam).particles, fb.track(beam).particles)
