"""`Segment.track` hands a stretch [run of skippable elements | active Cavity]+ with scalar settings to the device in ONE call
(chx_lattice_track: two launches for the whole stretch; Segment._lattice_stretch) — particles, energy and path length must
equal, BIT FOR BIT, the reference's walk element by element (/root/reference/cheetah/accelerator/segment.py:545-574,
cavity.py:100-251), i.e. this package's own per-element path; the reference values themselves: tests/golden/cavity.npz and the
linac of tests/golden (test_gpu_parity.py). Elements that do not qualify end a stretch and are tracked as before."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _spy():
    from cheetah_amd.accelerator import segment

    calls = []
    host = segment._lib.host()

    class Spy:
        def __getattr__(self, name):
            fn = getattr(host, name)
            if name != "lattice_track":
                return fn
            return lambda *a: (calls.append(a[2]), fn(*a))[1]

    return calls, Spy()


def _walk(seg, beam):
    """`Segment.track` item by item (merged runs through chx_run_track, cavities through chx_cavity_track_scalars): the path the
    stretch replaces."""
    from cheetah_amd.accelerator.segment import Segment

    orig = Segment._lattice_stretch
    Segment._lattice_stretch = lambda self, plan, i, incoming: None
    try:
        return seg.track(beam)
    finally:
        Segment._lattice_stretch = orig


def _linac(ca, dt, cells, cavity_type="standing_wave", phase=-10.0):
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    els = []
    for i in range(cells):
        els += [ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw),
                ca.HorizontalCorrector(t(0.05), angle=t(1e-4 * (i + 1)), **kw),
                ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(phase + 3 * i), frequency=t(1.3e9), cavity_type=cavity_type, **kw)]
    return els


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("cavity_type", ["standing_wave", "traveling_wave"])
def test_stretch_equals_element_by_element(dt, cavity_type):
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(3)
    beam = ca.ParticleBeam.from_parameters(num_particles=10_007, energy=t(6e6), sigma_p=t(1e-3), sigma_tau=t(1e-4), **kw)
    els = _linac(ca, dt, 6, cavity_type)
    els += [ca.Drift(t(0.4), **kw)]                              # a run behind the last cavity
    seg = ca.Segment(els)
    calls, spy = _spy()
    old = segment._HOST
    segment._HOST = spy
    try:
        with torch.no_grad():
            out = seg.track(beam)
        assert calls == [10_007], calls                            # ONE stretch call for the 25 elements
        with torch.no_grad():
            ref = _walk(seg, beam)
            one_by_one = beam
            for e in els:
                one_by_one = e.track(one_by_one)
        assert torch.equal(out.particles, ref.particles)
        # (element by element WITHOUT merging the runs differs by the rounding of the merged maps, like the reference's own
        # Segment.track does from its elements tracked one after the other)
        scale = one_by_one.particles.abs().amax(dim=0)
        assert ((out.particles - one_by_one.particles).abs().amax(dim=0) / scale).max() < (1e-5 if dt == torch.float32 else 1e-13)
        assert torch.equal(out.energy, ref.energy) and out.energy.shape == ()
        assert torch.equal(out.s, ref.s)
        assert float(out.energy) > 1e8                             # six cavities of 18 MV
        # deceleration (the other branch of cavity.py:157) and an in-place edit of a setting: the device reads the settings
        els[15].phase.fill_(170.0)                                  # (the fourth cavity: the beam has 60 MeV by then)
        els[7].voltage.mul_(0.5)
        calls.clear()
        with torch.no_grad():
            out2 = seg.track(beam)
            ref2 = _walk(seg, beam)
        assert calls == [10_007]
        assert torch.isfinite(out2.particles).all() and float(out2.energy) < float(out.energy) - 3e7
        assert torch.equal(out2.particles, ref2.particles) and torch.equal(out2.energy, ref2.energy)
        assert not torch.equal(out2.particles, out.particles)
        # a new tensor assigned to a setting (moves the epoch: the table is re-derived)
        els[1].k1 = t(5.5)
        calls.clear()
        with torch.no_grad():
            out3 = seg.track(beam)
            ref3 = _walk(seg, beam)
        assert calls == [10_007] and torch.equal(out3.particles, ref3.particles)
    finally:
        segment._HOST = old


def test_what_ends_a_stretch():
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(4)
    beam = ca.ParticleBeam.from_parameters(num_particles=5_000, energy=t(5e7), **kw)
    els = _linac(ca, dt, 2)
    els += [ca.Screen(is_active=True, name="scr", **kw)]           # records the beam and lets it pass: an item of the stretch (round 5)
    els += _linac(ca, dt, 2)
    els += [ca.Quadrupole(t(0.2), k1=torch.tensor([1.0, 2.0], **kw), **kw)]   # vectorised: general path
    seg = ca.Segment(els)
    calls, spy = _spy()
    old = segment._HOST
    segment._HOST = spy
    tcalls = []
    th = segment._lib.torch_host()

    class TorchSpy:
        def __getattr__(self, name):
            fn = getattr(th, name)
            return fn if name != "lattice_track_screens" else (lambda *a: (tcalls.append(a[1].shape[0]), fn(*a))[1])

    old_t = segment._TORCH_HOST
    segment._TORCH_HOST = TorchSpy()
    try:
        with torch.no_grad():
            out = seg.track(beam)
        assert calls == [] and tcalls == [5_000], (calls, tcalls)      # both linacs and the screen between them: ONE call
        read = seg.scr.get_read_beam().particles.clone()
        # with screens kept out of the stretches (the walk the screen tests compare with) the screen ends the first stretch
        segment.Segment._STRETCH_SCREENS = False
        try:
            with torch.no_grad():
                out_b = seg.track(beam)
            assert calls == [5_000, 5_000] and tcalls == [5_000], (calls, tcalls)
            assert torch.equal(out_b.particles, out.particles) and torch.equal(seg.scr.get_read_beam().particles, read)
        finally:
            segment.Segment._STRETCH_SCREENS = True
        with torch.no_grad():
            ref = _walk(seg, beam)
        assert out.particles.shape == (2, 5_000, 7) and torch.equal(out.particles, ref.particles)
        assert torch.equal(read, seg.scr.get_read_beam().particles)
        # a switched-off cavity is a skippable element (cavity.py:253-262): part of a run the device plan does not take -> the
        # element-by-element path, same numbers
        els[3].voltage = t(0.0)
        calls.clear()
        with torch.no_grad():
            out2 = seg.track(beam)
            ref2 = _walk(seg, beam)
        assert torch.equal(out2.particles, ref2.particles)
        # gradients: a trainable voltage keeps the differentiable path
        els[3].voltage = t(18e6)
        els[7].voltage = torch.nn.Parameter(t(18e6))
        calls.clear()
        out3 = seg.track(beam)
        out3.particles[..., 5].square().mean().backward()
        assert els[7].voltage.grad is not None and float(els[7].voltage.grad.abs()) > 0
        # a float64 cavity in a float32 lattice raises like the reference's matmul does
        bad = ca.Cavity(torch.tensor(1.0, dtype=torch.float64, device="cuda"), voltage=torch.tensor(1e6, dtype=torch.float64, device="cuda"),
                        frequency=torch.tensor(1.3e9, dtype=torch.float64, device="cuda"), dtype=torch.float64, device="cuda")
        with pytest.raises(RuntimeError):
            with torch.no_grad():
                ca.Segment(_linac(ca, dt, 1) + [bad]).track(beam)
    finally:
        segment._HOST = old
        segment._TORCH_HOST = old_t


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("with_cavities", [False, True])
def test_active_bpms_ride_in_the_stretch(dt, with_cavities):
    """A lattice with an active BPM in every cell: ONE stretch call (chx_lattice_track_bpm: prepare, particle pass with the
    weighted sums of x and y at every monitor, one finalize launch) — particles, energy and path length bit for bit as the walk
    item by item; every reading = (mu_x, mu_y) of the beam AT that monitor minus its misalignment (bpm.py:77-87), equal to the
    walk's reading to the rounding of the mean (fp64 sums in another order, then one rounding to the beam's dtype)."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(5)
    beam = ca.ParticleBeam.from_parameters(num_particles=20_011, energy=t(6e7), mu_x=t(2e-4), mu_y=t(-1e-4), sigma_p=t(1e-3), **kw)
    # some particles lost upstream: the means are weighted with the survival probabilities
    w = torch.ones(20_011, **kw)
    w[::7] = 0.25
    w[5] = 0.0
    beam = ca.ParticleBeam(beam.particles, beam.energy, particle_charges=beam.particle_charges, survival_probabilities=w, **kw)
    els, bpms = [], []
    for i in range(12):
        bpm = ca.BPM(is_active=True, misalignment=t([1e-5 * i, -2e-5 * i]), name=f"bpm{i}", **kw)
        bpms.append(bpm)
        els += [ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw), ca.Drift(t(0.5), **kw),
                ca.VerticalCorrector(t(0.05), angle=t(2e-5 * (i + 1)), **kw), bpm, ca.Drift(t(0.2), **kw)]
        if with_cavities and i % 4 == 1:
            els += [ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(-10.0), frequency=t(1.3e9), **kw)]
    els += [ca.BPM(is_active=True, name="last", **kw)]            # a monitor at the very end reads the outgoing beam
    bpms.append(els[-1])
    seg = ca.Segment(els)
    calls, spy = _spy()
    old = segment._HOST
    segment._HOST = spy
    try:
        with torch.no_grad():
            out = seg.track(beam)
            got = torch.stack([b.reading.clone() for b in bpms])
        assert calls == [20_011], calls                            # one call for the whole lattice
        with torch.no_grad():
            ref = _walk(seg, beam)
            want = torch.stack([b.reading.clone() for b in bpms])
        assert calls == [20_011] or not with_cavities              # (the walk's cavities take their own path)
        assert torch.equal(out.particles, ref.particles) and torch.equal(out.energy, ref.energy) and torch.equal(out.s, ref.s)
        assert got.shape == (13, 2) and torch.isfinite(got).all()
        eps = torch.finfo(dt).eps
        # one unit in the last place of the MEAN in the beam's dtype (the reading is the mean minus the misalignment) + the
        # rounding of fp64 sums over 20 011 coordinates of a few mm taken in another order (measured: 5.4e-20 m)
        mean_scale = (want + torch.stack([b.misalignment for b in bpms])).abs()
        order = 8 * torch.finfo(torch.float64).eps * 5e-3
        assert torch.all((got - want).abs() <= 2 * eps * mean_scale + order), ((got - want).abs() / (eps * mean_scale)).max()
        assert (got[1] - got[0]).abs().max() > 1e-6                # the monitors see different beams
        # the last monitor: the outgoing beam
        assert torch.allclose(got[-1, 0], out.mu_x, rtol=4 * eps, atol=order) and torch.allclose(got[-1, 1], out.mu_y, rtol=4 * eps, atol=order)
        # an in-place edit of a corrector and of a monitor's misalignment is followed; tracking does not move the epoch
        epoch = ca.Element._epoch
        els[2].angle.fill_(5e-4)
        bpms[3].misalignment[0] = 7e-4
        calls.clear()
        with torch.no_grad():
            out2 = seg.track(beam)
            got2 = torch.stack([b.reading.clone() for b in bpms])
            ref2 = _walk(seg, beam)
            want2 = torch.stack([b.reading.clone() for b in bpms])
        assert ca.Element._epoch == epoch
        assert torch.equal(out2.particles, ref2.particles)
        mean_scale2 = (want2 + torch.stack([b.misalignment for b in bpms])).abs()
        assert torch.all((got2 - want2).abs() <= 2 * eps * mean_scale2 + order)
        assert (got2[1] - got[1]).abs().max() > 1e-7             # (the kick shows at the NEXT monitor)
        # an inactive monitor is a pass-through element of its run; a monitor with a gradient-carrying misalignment ends the stretch
        bpms[5].is_active = False
        with torch.no_grad():
            out3, ref3 = seg.track(beam), _walk(seg, beam)
        assert torch.equal(out3.particles, ref3.particles)
    finally:
        segment._HOST = old


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("shape", ["rectangular", "elliptical"])
def test_active_apertures_ride_in_the_stretch(dt, shape):
    """Active apertures between the cells (aperture.py:90-135: survival *= inside(x, y)) and monitors behind them: one stretch
    call; outgoing particles AND survival probabilities bit for bit as the walk item by item (a particle on the boundary falls on
    the same side: the comparison is evaluated operation by operation in the beam's dtype), readings weighted with the thinned
    probabilities."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(6)
    beam = ca.ParticleBeam.from_parameters(num_particles=30_001, energy=t(6e7), sigma_x=t(4e-4), sigma_y=t(3e-4), mu_x=t(1e-4), **kw)
    w = torch.rand(30_001, **kw)
    beam = ca.ParticleBeam(beam.particles, beam.energy, particle_charges=beam.particle_charges, survival_probabilities=w, **kw)
    els, bpms, aps = [], [], []
    for i in range(8):
        ap = ca.Aperture(x_max=t(6e-4 + 1e-4 * i), y_max=t(5e-4), shape=shape, name=f"ap{i}", **kw)
        bpm = ca.BPM(is_active=True, name=f"bpm{i}", **kw)
        aps.append(ap)
        bpms.append(bpm)
        els += [ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw), ca.Drift(t(0.5), **kw), ap, bpm, ca.Drift(t(0.2), **kw)]
    seg = ca.Segment(els)
    calls, spy = _spy()
    old = segment._HOST
    segment._HOST = spy
    try:
        with torch.no_grad():
            out = seg.track(beam)
            got = torch.stack([b.reading.clone() for b in bpms])
        assert calls == [30_001], calls
        with torch.no_grad():
            ref = _walk(seg, beam)
            want = torch.stack([b.reading.clone() for b in bpms])
        assert torch.equal(out.particles, ref.particles) and torch.equal(out.s, ref.s)
        assert torch.equal(out.survival_probabilities, ref.survival_probabilities)
        lost = int((out.survival_probabilities == 0).sum())
        assert 1000 < lost < 29_000, lost                            # the apertures do cut
        assert out.survival_probabilities is not beam.survival_probabilities and torch.equal(beam.survival_probabilities, w)
        eps = torch.finfo(dt).eps
        order = 8 * torch.finfo(torch.float64).eps * 5e-3
        assert torch.all((got - want).abs() <= 2 * eps * want.abs() + order), (got - want).abs().max()
        # a particle exactly on the boundary of the first aperture: same side as the element-by-element kernel
        aps[0].x_max.copy_(ref.particles[0, 0].abs())
        with torch.no_grad():
            out2, ref2 = seg.track(beam), _walk(seg, beam)
        assert torch.equal(out2.survival_probabilities, ref2.survival_probabilities)
        # a negative half-width is refused like on the walk (aperture.py:72-73)
        aps[3].x_max.fill_(-1.0)
        with pytest.raises(AssertionError):
            seg.track(beam)
        aps[3].x_max.fill_(1e-3)
        # a ParameterBeam passes (with the reference's warning) as before
        with pytest.warns(Warning), torch.no_grad():
            seg.track(ca.ParameterBeam.from_parameters(**kw))
    finally:
        segment._HOST = old


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_several_beams_in_one_particle_beam_ride_in_the_stretch(dt):
    """A vectorised ParticleBeam — (3, 2) beams of 4001 particles under ONE lattice setting and energy — through cavities, active
    BPMs and an aperture: one stretch call (the beams are blockIdx.y of the particle pass, the maps are the same for all);
    every beam's particles, survival probabilities and readings as if it had been tracked on its own, bit for bit / to the
    rounding of the means."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(7)
    N = 4001
    base = ca.ParticleBeam.from_parameters(num_particles=N, energy=t(6e7), sigma_x=t(3e-4), sigma_p=t(1e-3), **kw)
    parts = base.particles.unsqueeze(0).unsqueeze(0).repeat(3, 2, 1, 1).contiguous()
    parts[..., :6] *= (1.0 + 0.1 * torch.arange(6, **kw).reshape(3, 2, 1, 1))         # six different beams
    w = torch.rand(3, 2, N, **kw)
    many = ca.ParticleBeam(parts, base.energy, particle_charges=base.particle_charges, survival_probabilities=w, **kw)
    els, bpms = [], []
    for i in range(6):
        bpm = ca.BPM(is_active=True, misalignment=t([1e-5 * i, 2e-5]), **kw)
        bpms.append(bpm)
        els += [ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw), ca.Drift(t(0.5), **kw), bpm]
        if i == 2:
            els += [ca.Aperture(x_max=t(4e-4), y_max=t(6e-4), **kw)]
        if i % 3 == 1:
            els += [ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(-10.0), frequency=t(1.3e9), **kw)]
    seg = ca.Segment(els)
    calls, spy = _spy()
    old = segment._HOST
    segment._HOST = spy
    try:
        with torch.no_grad():
            out = seg.track(many)
            got = torch.stack([b.reading.clone() for b in bpms])          # (6 monitors, 3, 2, 2)
        assert calls == [N], calls
        assert out.particles.shape == (3, 2, N, 7) and out.survival_probabilities.shape == (3, 2, N) and got.shape == (6, 3, 2, 2)
        eps = torch.finfo(dt).eps
        order = 8 * torch.finfo(torch.float64).eps * 5e-3
        for a in range(3):
            for b in range(2):
                single = ca.ParticleBeam(parts[a, b].clone(), base.energy, particle_charges=base.particle_charges,
                                         survival_probabilities=w[a, b].clone(), **kw)
                with torch.no_grad():
                    ref = _walk(seg, single)
                    want = torch.stack([m.reading.clone() for m in bpms])
                assert torch.equal(out.particles[a, b], ref.particles) and torch.equal(out.survival_probabilities[a, b], ref.survival_probabilities)
                assert torch.equal(out.energy, ref.energy) and torch.equal(out.s, ref.s)
                assert torch.all((got[:, a, b] - want).abs() <= 2 * eps * (want.abs() + 3e-5) + order)
        # a survival array shared by the beams (N,) is spread over them
        shared = ca.ParticleBeam(parts, base.energy, particle_charges=base.particle_charges, survival_probabilities=w[0, 0].clone(), **kw)
        with torch.no_grad():
            out2 = seg.track(shared)
            one = _walk(seg, ca.ParticleBeam(parts[2, 1].clone(), base.energy, particle_charges=base.particle_charges,
                                             survival_probabilities=w[0, 0].clone(), **kw))
        assert out2.survival_probabilities.shape == (3, 2, N) and torch.equal(out2.survival_probabilities[2, 1], one.survival_probabilities)
        assert torch.equal(out2.particles[2, 1], one.particles)
    finally:
        segment._HOST = old


def test_several_beams_under_scalar_settings_take_the_run_plan():
    import cheetah_amd as ca

    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    seg = ca.Segment([e for i in range(20) for e in (ca.Quadrupole(t(0.2), k1=t(2.0 if i % 2 else -2.0), **kw), ca.Drift(t(0.7), **kw))])
    base = ca.ParticleBeam.from_parameters(num_particles=3001, **kw)
    parts = (base.particles.unsqueeze(0) * torch.linspace(0.5, 1.5, 5, **kw).reshape(5, 1, 1)).contiguous()
    parts[..., 6] = 1.0
    many = ca.ParticleBeam(parts, base.energy, **kw)
    with torch.no_grad():
        out = seg.track(many)
        assert seg._plan()[0][1].fast.ok
        for b in range(5):
            ref = seg.track(ca.ParticleBeam(parts[b].clone(), base.energy, **kw))
            assert torch.equal(out.particles[b], ref.particles)
    assert out.particles.shape == (5, 3001, 7) and torch.equal(out.sigma_x.shape, torch.Size([5])) if False else out.sigma_x.shape == (5,)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_parameter_beam_takes_the_stretch(dt):
    """A ParameterBeam through cavities and active BPMs: one call (chx_parameter_lattice_track) — mu, cov, energy, path length and
    every monitor's reading bit for bit as the walk item by item (chx_parameter_track per run / cavity, bpm.py:77-87 per monitor);
    also for a vectorised beam (5 moment vectors under one lattice setting)."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator.segment import Segment

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    els, bpms = [], []
    for i in range(8):
        bpm = ca.BPM(is_active=True, misalignment=t([1e-5 * i, -3e-5]), **kw)
        bpms.append(bpm)
        els += [ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw),
                ca.HorizontalCorrector(t(0.05), angle=t(1e-4 * (i + 1)), **kw), bpm]
        if i % 2 == 0:
            els += [ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(-10.0 + 5 * i), frequency=t(1.3e9),
                              cavity_type="standing_wave" if i % 4 == 0 else "traveling_wave", **kw)]
    seg = ca.Segment(els)
    one = ca.ParameterBeam.from_parameters(energy=t(6e6), mu_x=t(1e-4), mu_py=t(2e-6), sigma_p=t(1e-3), sigma_tau=t(1e-4), **kw)
    many = ca.ParameterBeam(one.mu.unsqueeze(0) * torch.linspace(0.5, 1.5, 5, **kw).reshape(5, 1) + torch.tensor([0, 0, 0, 0, 0, 0, 1.0], **kw) * (1 - torch.linspace(0.5, 1.5, 5, **kw).reshape(5, 1)),
                            one.cov, one.energy, total_charge=one.total_charge, **kw)
    calls = []
    orig = Segment._lattice_stretch_parameter

    def spy(self, plan, i, incoming):
        out = orig(self, plan, i, incoming)
        calls.append(out is not None)
        return out

    for beam in (one, many):
        Segment._lattice_stretch_parameter = spy
        try:
            calls.clear()
            with torch.no_grad():
                out = seg.track(beam)
                got = torch.stack([b.reading.clone() for b in bpms])
            assert calls == [True], calls
        finally:
            Segment._lattice_stretch_parameter = lambda self, plan, i, incoming: None
        try:
            with torch.no_grad():
                ref = seg.track(beam)
                want = torch.stack([b.reading.clone() for b in bpms])
        finally:
            Segment._lattice_stretch_parameter = orig
        assert out.mu.shape == ref.mu.shape and out.cov.shape == ref.cov.shape
        assert torch.equal(out.mu, ref.mu) and torch.equal(out.cov, ref.cov)
        assert torch.equal(out.energy, ref.energy) and torch.equal(out.s, ref.s) and float(out.energy) > 7e7
        assert got.shape == want.shape and torch.equal(got, want)
        assert torch.equal(out.total_charge, ref.total_charge)
    # an in-place edit of a cavity phase is followed
    els[4].phase.fill_(25.0)
    with torch.no_grad():
        out2 = seg.track(one)
        Segment._lattice_stretch_parameter = lambda self, plan, i, incoming: None
        try:
            ref2 = seg.track(one)
        finally:
            Segment._lattice_stretch_parameter = orig
    assert torch.equal(out2.mu, ref2.mu) and torch.equal(out2.cov, ref2.cov) and not torch.equal(out2.mu, out.mu[0] if out.mu.dim() > 1 else out.mu)


@pytest.mark.parametrize("B,with_cavity", [(37, True), (600, False)])
@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_parameter_beam_stretch_with_vectorised_settings(dt, B, with_cavity):
    """An orbit-response measurement: every corrector's angle (and one quadrupole's strength) a (B,) tensor, a monitor in every cell,
    a cavity with scalar settings in between — ONE call for all B lattice settings (the preparation launch builds row b's maps from
    element b of the vectorised settings); mu, cov and the (B, 2) readings bit for bit as the walk item by item."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator.segment import Segment

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(9)
    els, bpms = [], []        # (37 rows and a cavity: a workgroup per (item, row) prepares the maps; 600 rows without: a wave does)
    for i in range(9):
        bpm = ca.BPM(is_active=True, misalignment=t([1e-5 * i, -1e-5]), **kw)
        bpms.append(bpm)
        k1 = torch.randn(B, **kw) * 3 if i == 4 else t(3.0 if i % 2 else -3.0)
        els += [ca.Quadrupole(t(0.2), k1=k1, **kw), ca.HorizontalCorrector(t(0.05), angle=1e-4 * torch.randn(B, **kw), **kw), ca.Drift(t(0.6), **kw),
                ca.VerticalCorrector(t(0.05), angle=t(2e-5 * i), **kw), bpm]
        if i == 5 and with_cavity:
            els += [ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(-10.0), frequency=t(1.3e9), **kw)]
    seg = ca.Segment(els)
    pb = ca.ParameterBeam.from_parameters(energy=t(6e7), mu_x=t(1e-4), sigma_p=t(1e-3), **kw)
    orig = Segment._lattice_stretch_parameter
    calls = []
    Segment._lattice_stretch_parameter = lambda self, plan, i, incoming: (lambda r: (calls.append(r is not None), r)[1])(orig(self, plan, i, incoming))
    try:
        with torch.no_grad():
            out = seg.track(pb)
            got = torch.stack([b.reading.clone() for b in bpms])
        assert calls == [True], calls
    finally:
        Segment._lattice_stretch_parameter = lambda self, plan, i, incoming: None
    try:
        with torch.no_grad():
            ref = seg.track(pb)
            want = torch.stack([b.reading.clone() for b in bpms])
    finally:
        Segment._lattice_stretch_parameter = orig
    assert out.mu.shape == (B, 7) and out.cov.shape == (B, 7, 7) and got.shape == (9, B, 2)
    assert torch.equal(out.mu, ref.mu) and torch.equal(out.cov, ref.cov) and torch.equal(out.energy, ref.energy) and torch.equal(out.s, ref.s)
    assert torch.equal(got, want)
    assert (got[8, 0] - got[8, 1]).abs().max() > 1e-7                     # the rows see different lattices
    # the settings written in place: followed; a ParticleBeam on the same lattice walks (and agrees with its own walk)
    els[1].angle.mul_(-2.0)
    with torch.no_grad():
        out2 = seg.track(pb)
        Segment._lattice_stretch_parameter = lambda self, plan, i, incoming: None
        try:
            ref2 = seg.track(pb)
        finally:
            Segment._lattice_stretch_parameter = orig
        assert torch.equal(out2.mu, ref2.mu) and not torch.equal(out2.mu, out.mu)
        beam = ca.ParticleBeam.from_parameters(num_particles=2_000, energy=t(6e7), **kw)
        a = seg.track(beam)
        assert a.particles.shape == (B, 2_000, 7)
        assert torch.equal(a.particles, _walk(seg, beam).particles)


@pytest.mark.parametrize("B,with_cavity,own_rows", [(5, True, False), (70, False, False), (6, True, True)])
@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_particle_beam_stretch_with_vectorised_settings(dt, B, with_cavity, own_rows):
    """A scan of lattice settings tracked with particles: corrector angles and one quadrupole strength (B,) tensors, monitors and
    apertures in front of and behind the first vectorised element, a cavity with scalar settings — ONE stretch call; row b of the
    (B, N, 7) result = the beam (one shared beam, or row b of a (B, N, 7) beam) through row b of the settings, bit for bit as the
    walk item by item; readings to the rounding of the mean; a monitor / aperture in front of the scan keeps the incoming shape."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(21)
    N = 3_001
    beam = ca.ParticleBeam.from_parameters(num_particles=N, energy=t(6e7), mu_x=t(2e-4), sigma_x=t(4e-4), sigma_y=t(4e-4), sigma_p=t(1e-3), **kw)
    if own_rows:
        parts = beam.particles.unsqueeze(0).repeat(B, 1, 1)
        parts[..., :6] *= 1 + 0.1 * torch.rand(B, 1, 1, **kw)
        beam = ca.ParticleBeam(parts, beam.energy, particle_charges=beam.particle_charges, **kw)
    front_bpm = ca.BPM(is_active=True, misalignment=t([3e-5, -1e-5]), **kw)
    front_ap = ca.Aperture(x_max=t(1.2e-3), y_max=t(1.5e-3), shape="elliptical", **kw)
    els = [ca.Drift(t(0.4), **kw), front_ap, ca.Quadrupole(t(0.2), k1=t(2.0), **kw), front_bpm]
    bpms = []
    for i in range(6):
        bpm = ca.BPM(is_active=True, misalignment=t([1e-5 * i, -1e-5]), **kw)
        bpms.append(bpm)
        k1 = torch.randn(B, **kw) * 3 if i == 3 else t(3.0 if i % 2 else -3.0)
        els += [ca.Quadrupole(t(0.2), k1=k1, **kw), ca.HorizontalCorrector(t(0.05), angle=2e-4 * torch.randn(B, **kw), **kw),
                ca.Drift(t(0.6), **kw), bpm]
        if i == 2:
            els += [ca.Aperture(x_max=t(1.5e-3), y_max=t(1.5e-3), shape="rectangular", **kw)]
        if i == 4 and with_cavity:
            els += [ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(-10.0), frequency=t(1.3e9), **kw)]
    seg = ca.Segment(els)
    calls, spy = _spy()
    old = segment._HOST
    segment._HOST = spy
    try:
        with torch.no_grad():
            out = seg.track(beam)
            got, got_front = torch.stack([b.reading.clone() for b in bpms]), front_bpm.reading.clone()
        assert calls == [N], calls
        with torch.no_grad():
            ref = _walk(seg, beam)
            want, want_front = torch.stack([b.reading.clone() for b in bpms]), front_bpm.reading.clone()
    finally:
        segment._HOST = old
    assert out.particles.shape == (B, N, 7) == ref.particles.shape
    assert torch.equal(out.particles, ref.particles) and torch.equal(out.energy, ref.energy) and torch.equal(out.s, ref.s)
    assert out.survival_probabilities.shape == ref.survival_probabilities.shape == (B, N)
    assert torch.equal(out.survival_probabilities, ref.survival_probabilities)
    lost = (out.survival_probabilities == 0).sum(dim=-1)
    assert lost.min() > 0 and lost.max() < N and (own_rows or len(set(lost.tolist())) > 1)     # the rows lose different particles
    assert got.shape == want.shape == (6, B, 2)
    assert got_front.shape == want_front.shape == ((B, 2) if own_rows else (2,))
    eps = torch.finfo(dt).eps
    order = 8 * torch.finfo(torch.float64).eps * 5e-3
    mis = torch.stack([b.misalignment for b in bpms]).unsqueeze(1)
    assert torch.all((got - want).abs() <= 2 * eps * (want + mis).abs() + order)
    assert torch.all((got_front - want_front).abs() <= 2 * eps * (want_front + front_bpm.misalignment).abs() + order)
    assert (got[5, 0] - got[5, 1]).abs().max() > 1e-6
    # settings written in place are followed by the next track
    els[5].angle.mul_(-1.5)
    with torch.no_grad():
        out2, ref2 = seg.track(beam), _walk(seg, beam)
    assert torch.equal(out2.particles, ref2.particles) and not torch.equal(out2.particles, out.particles)
    # only apertures in FRONT of the scan: the survival probabilities keep the incoming beam's shape, like the walk's
    els[16].is_active = False
    with torch.no_grad():
        out3, ref3 = seg.track(beam), _walk(seg, beam)
    assert out3.survival_probabilities.shape == ref3.survival_probabilities.shape
    assert torch.equal(out3.survival_probabilities, ref3.survival_probabilities) and torch.equal(out3.particles, ref3.particles)


@pytest.mark.parametrize("B,N", [(0, 4_300_003), (64, 70_001)])
@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_big_stretch_two_particles_per_lane(dt, B, N):
    """From 1e6 particle rows on the particle pass of a stretch takes two particles per lane (float32: the maps as packed FMAs,
    the per-particle order of operations unchanged): one plain beam of 4.3e6 particles, and a scan of 64 lattice settings over a
    shared beam of 70 001 — cavities, monitors and apertures inside; particles, survival probabilities, energy and s bit for bit
    as the walk item by item, readings to the rounding of the mean."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(33)
    beam = ca.ParticleBeam.from_parameters(num_particles=N, energy=t(6e7), mu_x=t(2e-4), sigma_x=t(4e-4), sigma_y=t(4e-4), sigma_p=t(1e-3), **kw)
    w = torch.rand(N, **kw)
    beam = ca.ParticleBeam(beam.particles, beam.energy, particle_charges=beam.particle_charges, survival_probabilities=w, **kw)
    els, bpms = [], []
    for i in range(5):
        bpm = ca.BPM(is_active=True, misalignment=t([1e-5 * i, -1e-5]), **kw)
        bpms.append(bpm)
        angle = 2e-4 * torch.randn(B, **kw) if B else t(1e-4 * (i + 1))
        els += [ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw), ca.HorizontalCorrector(t(0.05), angle=angle, **kw), ca.Drift(t(0.6), **kw), bpm]
        if i == 1:
            els += [ca.Aperture(x_max=t(1.2e-3), y_max=t(1.4e-3), shape="elliptical", **kw)]
        if i == 2:
            els += [ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(-10.0), frequency=t(1.3e9), **kw)]
    seg = ca.Segment(els)
    calls, spy = _spy()
    old = segment._HOST
    segment._HOST = spy
    try:
        with torch.no_grad():
            out = seg.track(beam)
            got = torch.stack([b.reading.clone() for b in bpms])
        assert calls == [N], calls
        with torch.no_grad():
            ref = _walk(seg, beam)
            want = torch.stack([b.reading.clone() for b in bpms])
    finally:
        segment._HOST = old
    lead = (B,) if B else ()
    assert out.particles.shape == (*lead, N, 7) and N * max(B, 1) >= 1_000_000
    assert torch.equal(out.particles, ref.particles) and torch.equal(out.energy, ref.energy) and torch.equal(out.s, ref.s)
    assert out.survival_probabilities.shape == ref.survival_probabilities.shape == (*lead, N)
    assert torch.equal(out.survival_probabilities, ref.survival_probabilities)
    assert 0 < int((out.survival_probabilities == 0).sum()) < out.survival_probabilities.numel() // 2
    eps = torch.finfo(dt).eps
    order = 64 * torch.finfo(torch.float64).eps * 5e-3           # (fp64 sums over up to 4.3e6 coordinates in another order)
    mis = torch.stack([b.misalignment for b in bpms])
    mis = mis.unsqueeze(1) if B else mis
    assert got.shape == want.shape == (5, *lead, 2)
    assert torch.all((got - want).abs() <= 2 * eps * (want + mis).abs() + order), ((got - want).abs()).max()


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("cavity_type", ["standing_wave", "traveling_wave"])
def test_switched_off_cavities_ride_in_the_runs(dt, cavity_type):
    """A linac with some cavities at voltage 0 (skippable, drift-like: cavity.py:253-262): they are elements of the runs' persistent
    plans, the whole linac stays ONE stretch call — the same bits as with the cavities kept out of the plans (the per-element
    builders + chx_compose_maps), for particles and for a ParameterBeam. Switching a cavity off and on again between tracks
    re-partitions the lattice; a partition seen before comes back with its plans."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment
    from cheetah_amd.accelerator.cavity import Cavity

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(8)
    beam = ca.ParticleBeam.from_parameters(num_particles=20_003, energy=t(5e7), sigma_p=t(1e-3), **kw)
    pbeam = ca.ParameterBeam.from_parameters(energy=t(5e7), sigma_p=t(1e-3), **kw)
    els = _linac(ca, dt, 8, cavity_type)
    cavities = [e for e in els if isinstance(e, Cavity)]
    for k in (2, 5):
        cavities[k].voltage = t(0.0)
    seg = ca.Segment(els)
    calls, spy = _spy()
    old = segment._HOST
    segment._HOST = spy
    try:
        with torch.no_grad():
            out = seg.track(beam)
            pout = seg.track(pbeam)
        assert calls == [20_003], calls
        orig = Cavity._plannable
        Cavity._plannable = lambda self: False
        try:
            ca.Element._epoch += 1                       # (the plans look at their elements again)
            with torch.no_grad():
                ref = seg.track(beam)
                pref = seg.track(pbeam)
        finally:
            Cavity._plannable = orig
            ca.Element._epoch += 1
        assert torch.equal(out.particles, ref.particles) and torch.equal(out.energy, ref.energy) and torch.equal(out.s, ref.s)
        assert torch.equal(pout.mu, pref.mu) and torch.equal(pout.cov, pref.cov) and torch.equal(pout.energy, pref.energy)
        # the voltage comes back (assignment), goes away again (in place): every partition tracks like a fresh lattice
        plan_off = seg._plan()
        cavities[2].voltage = t(18e6)
        with torch.no_grad():
            out_on = seg.track(beam)
        assert seg._plan() is not plan_off
        fresh = _linac(ca, dt, 8, cavity_type)
        [e for e in fresh if isinstance(e, Cavity)][5].voltage = t(0.0)
        with torch.no_grad():
            want_on = ca.Segment(fresh).track(beam)
        assert torch.equal(out_on.particles, want_on.particles) and torch.equal(out_on.energy, want_on.energy)
        cavities[2].voltage.zero_()
        with torch.no_grad():
            out_off = seg.track(beam)
        assert seg._plan() is plan_off                  # the partition seen before, with its plans
        assert torch.equal(out_off.particles, out.particles) and torch.equal(out_off.energy, out.energy)
    finally:
        segment._HOST = old


def test_trainable_settings_keep_their_plans_under_no_grad():
    """A lattice whose strengths, cavity voltage and a monitor's misalignment are nn.Parameters: evaluated under `torch.no_grad()`
    it is ONE stretch call with the numbers of the same lattice built from plain tensors; with gradients enabled the very same
    Segment takes the differentiable path (every parameter receives a gradient) and, after an optimiser step edited the parameters
    in place, the next no_grad evaluation follows the new values."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(12)
    beam = ca.ParticleBeam.from_parameters(num_particles=10_000, energy=t(5e7), sigma_p=t(1e-3), **kw)

    def lattice(wrap):
        els = []
        for i in range(6):
            els += [ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=wrap(t(3.0 if i % 2 else -3.0)), **kw),
                    ca.Cavity(t(1.0377), voltage=wrap(t(18e6)) if i == 2 else t(18e6), phase=t(-10.0), frequency=t(1.3e9), **kw),
                    ca.BPM(is_active=True, misalignment=wrap(t([1e-5, -2e-5])) if i == 3 else t([0.0, 0.0]), **kw)]
        return ca.Segment(els)

    trainable, plain = lattice(torch.nn.Parameter), lattice(lambda v: v)
    params = list(trainable.parameters())
    assert len(params) == 8
    calls, spy = _spy()
    old = segment._HOST
    segment._HOST = spy
    try:
        with torch.no_grad():
            out = trainable.track(beam)
            assert calls == [10_000], calls
            ref = plain.track(beam)
        assert torch.equal(out.particles, ref.particles) and torch.equal(out.energy, ref.energy)
        assert torch.equal(trainable.elements[15].reading, plain.elements[15].reading)
        # gradients: the same Segment, the differentiable path
        calls.clear()
        got = trainable.track(beam)
        assert calls == [] and got.particles.requires_grad
        assert torch.allclose(got.particles, out.particles, rtol=1e-4, atol=1e-9)
        loss = got.particles[:, 0].square().mean() + got.energy * 1e-20 + trainable.elements[15].reading.square().sum()
        loss.backward()
        assert all(p.grad is not None and bool(p.grad.abs().sum() > 0) for p in params), [p.grad for p in params]
        # an optimiser step edits the parameters in place: the plans read the new values
        with torch.no_grad():
            for p in params:
                p.mul_(1.01)
            calls.clear()
            out2 = trainable.track(beam)
            assert calls == [10_000]
            for e_t, e_p in zip(trainable.elements, plain.elements):
                for name in ("k1", "voltage", "misalignment"):
                    if hasattr(e_p, name) and isinstance(getattr(e_t, name), torch.nn.Parameter):
                        setattr(e_p, name, getattr(e_t, name).detach().clone())
            ref2 = plain.track(beam)
        assert torch.equal(out2.particles, ref2.particles) and not torch.equal(out2.particles, out.particles)
    finally:
        segment._HOST = old


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("with_vector_settings,with_cavities", [(False, True), (True, True), (False, False)])
def test_energy_scan_rides_in_the_stretch(dt, with_vector_settings, with_cavities):
    """A scan of BEAM ENERGIES — `energy` a (B,) tensor — through a cavity linac with monitors (one in front of the first element, where
    the beam is not yet spread over the energies): ONE stretch call for a ParticleBeam (one shared beam) and for a ParameterBeam; row b
    = the beam at energy b (and row b of vectorised settings): particles, moments, outgoing (B,) energies and s bit for bit as the walk
    item by item, every reading with the walk's shape."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment
    from cheetah_amd.accelerator.segment import Segment

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(41)
    B, N = 23, 3_000
    energy = torch.linspace(4e7, 9e7, B, **kw)
    front = ca.BPM(is_active=True, misalignment=t([2e-5, 1e-5]), **kw)
    els, bpms = [front], [front]
    for i in range(6):
        bpm = ca.BPM(is_active=True, misalignment=t([1e-5 * i, -1e-5]), **kw)
        bpms.append(bpm)
        angle = 2e-4 * torch.randn(B, **kw) if (with_vector_settings and i % 2 == 1) else t(1e-4 * (i + 1))
        els += [ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw), ca.HorizontalCorrector(t(0.05), angle=angle, **kw)]
        if with_cavities:
            els += [ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(-10.0 + 3 * i), frequency=t(1.3e9),
                              cavity_type="standing_wave" if i % 2 else "traveling_wave", **kw)]
        els += [bpm]
    seg = ca.Segment(els)
    beam = ca.ParticleBeam.from_parameters(num_particles=N, energy=energy, mu_x=t(1e-4), sigma_p=t(1e-3), **kw)
    pbeam = ca.ParameterBeam.from_parameters(energy=energy, mu_x=t(1e-4), sigma_p=t(1e-3), **kw)
    calls, spy = _spy()
    pcalls = []
    orig_p = Segment._lattice_stretch_parameter
    old = segment._HOST
    segment._HOST = spy
    try:
        Segment._lattice_stretch_parameter = lambda self, plan, i, incoming: (lambda r: (pcalls.append(r is not None), r)[1])(orig_p(self, plan, i, incoming))
        with torch.no_grad():
            out = seg.track(beam)
            got = [b.reading.clone() for b in bpms]
            pout = seg.track(pbeam)
            pgot = [b.reading.clone() for b in bpms]
        assert calls == [N] and pcalls == [True], (calls, pcalls)
        Segment._lattice_stretch_parameter = lambda self, plan, i, incoming: None
        with torch.no_grad():
            ref = _walk(seg, beam)
            want = [b.reading.clone() for b in bpms]
            pref = seg.track(pbeam)
            pwant = [b.reading.clone() for b in bpms]
    finally:
        segment._HOST = old
        Segment._lattice_stretch_parameter = orig_p
    assert out.particles.shape == ref.particles.shape == (B, N, 7) and torch.equal(out.particles, ref.particles)
    assert out.energy.shape == ref.energy.shape == (B,) and torch.equal(out.energy, ref.energy) and torch.equal(out.s, ref.s)
    assert torch.equal(pout.mu, pref.mu) and torch.equal(pout.cov, pref.cov) and torch.equal(pout.energy, pref.energy) and torch.equal(pout.s, pref.s)
    if with_cavities:
        assert float((out.energy - energy).min()) > 5e7
    eps = torch.finfo(dt).eps
    order = 8 * torch.finfo(torch.float64).eps * 5e-3
    for k, (g, w, pg, pw) in enumerate(zip(got, want, pgot, pwant)):
        assert g.shape == w.shape == ((2,) if k == 0 else (B, 2)), (k, g.shape, w.shape)
        assert pg.shape == pw.shape == ((2,) if k == 0 else (B, 2)), (k, pg.shape, pw.shape)
        assert torch.all((g - w).abs() <= 2 * eps * (w + bpms[k].misalignment).abs() + order), k
        assert torch.equal(pg, pw), k
    assert (out.particles[0] - out.particles[-1]).abs().max() > 1e-6


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("scanned", ["phase_of_one", "phase_and_voltage_of_all", "frequency_of_one", "decelerating_rows"])
def test_cavity_scans_ride_in_the_stretch(dt, scanned):
    """A PHASE / voltage / frequency scan of cavities — the setting a (B,) tensor — stays ONE stretch call: row b of the beam through
    row b of the cavity's settings, the outgoing energy one per row (cavity.py:113-122), `(delta_energy > 0).any()` of cavity.py:157
    taken over the rows of that cavity (rows that lose energy next to rows that gain it). Particles, moments, energies, s bit for bit
    as the walk item by item; readings with the walk's shapes ((2,) in front of the first scanned cavity)."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment
    from cheetah_amd.accelerator.segment import Segment

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(52)
    B, N = 19, 3_000
    els, bpms = [], []
    for i in range(6):
        phase, voltage, freq = t(-10.0 + 3 * i), t(18e6), t(1.3e9)
        if scanned == "phase_of_one" and i == 2:
            phase = torch.linspace(-40.0, 40.0, B, **kw)
        elif scanned == "phase_and_voltage_of_all":
            phase, voltage = torch.linspace(-30.0, 30.0, B, **kw) + i, 1.5e7 + 5e6 * torch.rand(B, **kw)
        elif scanned == "frequency_of_one" and i == 3:
            freq = torch.linspace(1.2e9, 1.4e9, B, **kw)
        elif scanned == "decelerating_rows" and i == 1:
            phase = torch.linspace(0.0, 180.0, B, **kw)          # half of the rows lose energy in this cavity
        bpm = ca.BPM(is_active=True, misalignment=t([1e-5 * i, -1e-5]), **kw)
        bpms.append(bpm)
        els += [ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw), ca.HorizontalCorrector(t(0.05), angle=t(1e-4), **kw), bpm,
                ca.Cavity(t(1.0377), voltage=voltage, phase=phase, frequency=freq, cavity_type="standing_wave" if i % 2 else "traveling_wave", **kw)]
    seg = ca.Segment(els)
    beam = ca.ParticleBeam.from_parameters(num_particles=N, energy=t(9e7), mu_x=t(1e-4), sigma_p=t(1e-3), **kw)
    pbeam = ca.ParameterBeam.from_parameters(energy=t(9e7), mu_x=t(1e-4), sigma_p=t(1e-3), **kw)
    calls, spy = _spy()
    pcalls = []
    orig_p = Segment._lattice_stretch_parameter
    old = segment._HOST
    segment._HOST = spy
    try:
        Segment._lattice_stretch_parameter = lambda self, plan, i, incoming: (lambda r: (pcalls.append(r is not None), r)[1])(orig_p(self, plan, i, incoming))
        with torch.no_grad():
            out = seg.track(beam)
            got = [b.reading.clone() for b in bpms]
            pout = seg.track(pbeam)
            pgot = [b.reading.clone() for b in bpms]
        assert calls == [N] and pcalls == [True], (calls, pcalls)
        Segment._lattice_stretch_parameter = lambda self, plan, i, incoming: None
        with torch.no_grad():
            ref = _walk(seg, beam)
            want = [b.reading.clone() for b in bpms]
            pref = seg.track(pbeam)
            pwant = [b.reading.clone() for b in bpms]
    finally:
        segment._HOST = old
        Segment._lattice_stretch_parameter = orig_p
    # (the row at exactly 90 degrees gains nothing: the reference's formulas give 0 / 0 there, NaN in the walk and in the stretch alike)
    same = lambda a, b: torch.equal(torch.nan_to_num(a, nan=1234.5), torch.nan_to_num(b, nan=1234.5))  # noqa: E731
    assert out.particles.shape == ref.particles.shape == (B, N, 7) and same(out.particles, ref.particles)
    assert int(torch.isnan(out.particles).any(dim=-1).any(dim=-1).sum()) <= (1 if scanned == "decelerating_rows" else 0)
    assert out.energy.shape == ref.energy.shape and torch.equal(out.energy, ref.energy) and torch.equal(out.s, ref.s)
    assert out.energy.shape == (() if scanned == "frequency_of_one" else (B,))
    assert pout.mu.shape == pref.mu.shape and same(pout.mu, pref.mu) and same(pout.cov, pref.cov)
    assert pout.energy.shape == pref.energy.shape and torch.equal(pout.energy, pref.energy) and torch.equal(pout.s, pref.s)
    eps = torch.finfo(dt).eps
    order = 8 * torch.finfo(torch.float64).eps * 5e-3
    for k, (g, w, pg, pw) in enumerate(zip(got, want, pgot, pwant)):
        assert g.shape == w.shape and pg.shape == pw.shape, (k, g.shape, w.shape, pg.shape, pw.shape)
        live = torch.isfinite(w)
        assert torch.equal(torch.isfinite(g), live) and torch.all(((g - w).abs() <= 2 * eps * (w + bpms[k].misalignment).abs() + order)[live]), k
        assert same(pg, pw), k
    assert got[0].shape == (2,) and got[-1].shape == (B, 2)
    assert (out.particles[0] - out.particles[-1]).abs().max() > 1e-7


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("with_cavity", [False, True])
def test_grid_scan_by_broadcasting_rides_in_the_stretch(dt, with_cavity):
    """A 2-D grid scan written by broadcasting — one strength of shape (5, 1), another (and a cavity phase) of shape (1, 4) — is ONE
    stretch call over the (5, 4) grid: settings whose own shape is not the grid's are addressed through expanded copies that follow
    in-place edits. Particles, moments, energies bit for bit as the walk; a monitor between the two axes reads a (5, 1, 2) beam, one in
    front of both (2,), like the walk's; the survival probabilities behind an aperture between the axes have the walk's shape."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment
    from cheetah_amd.accelerator.segment import Segment

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    torch.manual_seed(63)
    N = 2_500
    ka = torch.linspace(-4.0, 4.0, 5, **kw).reshape(5, 1).contiguous()
    kb = torch.linspace(-3.0, 3.0, 4, **kw).reshape(1, 4).contiguous()
    phase = torch.linspace(-30.0, 20.0, 4, **kw).reshape(1, 4).contiguous()
    bpms = [ca.BPM(is_active=True, misalignment=t([1e-5 * i, -1e-5]), **kw) for i in range(4)]
    els = [ca.Drift(t(0.3), **kw), bpms[0],
           ca.Quadrupole(t(0.2), k1=ka, **kw), ca.Drift(t(0.5), **kw), bpms[1], ca.Aperture(x_max=t(1.5e-3), y_max=t(1.5e-3), **kw),
           ca.Quadrupole(t(0.2), k1=t(2.0), **kw), ca.HorizontalCorrector(t(0.05), angle=t(1e-4), **kw), bpms[2],
           ca.Quadrupole(t(0.2), k1=kb, **kw), ca.Drift(t(0.4), **kw)]
    if with_cavity:
        els += [ca.Cavity(t(1.0377), voltage=t(18e6), phase=phase, frequency=t(1.3e9), **kw)]
    els += [ca.Drift(t(0.2), **kw), bpms[3]]
    seg = ca.Segment(els)
    beam = ca.ParticleBeam.from_parameters(num_particles=N, energy=t(6e7), mu_x=t(1e-4), sigma_x=t(5e-4), sigma_y=t(5e-4), sigma_p=t(1e-3), **kw)
    pbeam = ca.ParameterBeam.from_parameters(energy=t(6e7), mu_x=t(1e-4), sigma_p=t(1e-3), **kw)
    pels = [e for e in els if not isinstance(e, ca.Aperture)]
    pseg = ca.Segment(pels)                                     # (an aperture only warns for a ParameterBeam)

    def both():
        calls, spy = _spy()
        pcalls = []
        orig_p = Segment._lattice_stretch_parameter
        old = segment._HOST
        segment._HOST = spy
        try:
            Segment._lattice_stretch_parameter = lambda self, plan, i, incoming: (lambda r: (pcalls.append(r is not None), r)[1])(orig_p(self, plan, i, incoming))
            with torch.no_grad():
                out = seg.track(beam)
                got = [b.reading.clone() for b in bpms]
                pout = pseg.track(pbeam)
                pgot = [b.reading.clone() for b in bpms]
            assert calls == [N] and pcalls == [True], (calls, pcalls)
            Segment._lattice_stretch_parameter = lambda self, plan, i, incoming: None
            with torch.no_grad():
                ref = _walk(seg, beam)
                want = [b.reading.clone() for b in bpms]
                pref = pseg.track(pbeam)
                pwant = [b.reading.clone() for b in bpms]
        finally:
            segment._HOST = old
            Segment._lattice_stretch_parameter = orig_p
        return out, got, pout, pgot, ref, want, pref, pwant

    out, got, pout, pgot, ref, want, pref, pwant = both()
    assert out.particles.shape == ref.particles.shape == (5, 4, N, 7) and torch.equal(out.particles, ref.particles)
    assert out.energy.shape == ref.energy.shape and torch.equal(out.energy, ref.energy) and torch.equal(out.s, ref.s)
    assert out.survival_probabilities.shape == ref.survival_probabilities.shape == (5, 1, N)
    assert torch.equal(out.survival_probabilities, ref.survival_probabilities)
    assert pout.mu.shape == pref.mu.shape == (5, 4, 7) and torch.equal(pout.mu, pref.mu) and torch.equal(pout.cov, pref.cov)
    assert pout.energy.shape == pref.energy.shape and torch.equal(pout.energy, pref.energy)
    eps = torch.finfo(dt).eps
    order = 8 * torch.finfo(torch.float64).eps * 5e-3
    for k, shape in enumerate([(2,), (5, 1, 2), (5, 1, 2), (5, 4, 2)]):
        assert got[k].shape == want[k].shape == shape and pgot[k].shape == pwant[k].shape == shape, (k, got[k].shape, want[k].shape, pgot[k].shape)
        assert torch.all((got[k] - want[k]).abs() <= 2 * eps * (want[k] + bpms[k].misalignment).abs() + order), k
        assert torch.equal(pgot[k], pwant[k]), k
    # an in-place edit of the (5, 1) strength: the expanded copy follows
    ka.mul_(-0.7)
    out2, _, pout2, _, ref2, _, pref2, _ = both()
    assert torch.equal(out2.particles, ref2.particles) and not torch.equal(out2.particles, out.particles)
    assert torch.equal(pout2.mu, pref2.mu) and not torch.equal(pout2.mu, pout.mu)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_cavity_stretch_with_thousands_of_rows(dt):
    """`small_runs` bit 3 (CHX_LATTICE_SHORT_RUNS: a WAVE per (item, row) prepares a stretch WITH cavities when its runs hold
    at most 64 elements) lifts the `_STRETCH_MAX_ROWS` cap — on purpose. 4096 beam energies through a 12-cavity linac with monitors
    (a ParameterBeam: 4096 rows of moments), one stretch call, against the walk item by item: moments, (4096,) outgoing energies and
    every reading bit for bit; and 1500 rows of a cavity phase for 500 shared particles."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment
    from cheetah_amd.accelerator.segment import Segment

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    B = 4096
    assert B > Segment._STRETCH_MAX_ROWS
    energy = torch.linspace(4e7, 9e7, B, **kw)

    def lattice(phase_rows=None):
        els, bpms = [], []
        for i in range(12):
            bpm = ca.BPM(is_active=True, misalignment=t([1e-5 * i, -1e-5]), **kw)
            bpms.append(bpm)
            phase = t(-10.0 + 3 * i) if (phase_rows is None or i != 5) else phase_rows
            els += [ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw),
                    ca.Cavity(t(1.0377), voltage=t(18e6), phase=phase, frequency=t(1.3e9),
                              cavity_type="standing_wave" if i % 2 else "traveling_wave", **kw), bpm]
        return ca.Segment(els), bpms

    seg, bpms = lattice()
    pbeam = ca.ParameterBeam.from_parameters(energy=energy, mu_x=t(1e-4), sigma_p=t(1e-3), **kw)
    pcalls = []
    orig_p = Segment._lattice_stretch_parameter
    try:
        Segment._lattice_stretch_parameter = lambda self, plan, i, incoming: (lambda r: (pcalls.append(r is not None), r)[1])(orig_p(self, plan, i, incoming))
        with torch.no_grad():
            pout = seg.track(pbeam)
            pgot = [b.reading.clone() for b in bpms]
        assert pcalls == [True], pcalls                              # ONE stretch call took all 4096 rows
        Segment._lattice_stretch_parameter = lambda self, plan, i, incoming: None
        with torch.no_grad():
            pref = seg.track(pbeam)
            pwant = [b.reading.clone() for b in bpms]
    finally:
        Segment._lattice_stretch_parameter = orig_p
    assert pout.mu.shape == pref.mu.shape == (B, 7) and torch.equal(pout.mu, pref.mu) and torch.equal(pout.cov, pref.cov)
    assert pout.energy.shape == (B,) and torch.equal(pout.energy, pref.energy) and torch.equal(pout.s, pref.s)
    assert float((pout.energy - energy).min()) > 1e8
    for g, w in zip(pgot, pwant):
        assert g.shape == w.shape == (B, 2) and torch.equal(g, w)
    # particles: 1500 phases of one cavity over 500 shared particles
    rows = torch.linspace(-40.0, 40.0, 1500, **kw)
    seg2, bpms2 = lattice(rows)
    beam = ca.ParticleBeam.from_parameters(num_particles=500, energy=t(6e7), mu_x=t(1e-4), sigma_p=t(1e-3), **kw)
    calls, spy = _spy()
    old = segment._HOST
    segment._HOST = spy
    try:
        with torch.no_grad():
            out = seg2.track(beam)
            got = [b.reading.clone() for b in bpms2]
        assert calls == [500], calls
        with torch.no_grad():
            ref = _walk(seg2, beam)
            want = [b.reading.clone() for b in bpms2]
    finally:
        segment._HOST = old
    assert out.particles.shape == ref.particles.shape == (1500, 500, 7) and torch.equal(out.particles, ref.particles)
    assert torch.equal(out.energy, ref.energy) and out.energy.shape == (1500,)
    eps = torch.finfo(dt).eps
    for k, (g, w) in enumerate(zip(got, want)):
        assert g.shape == w.shape
        assert torch.all((g - w).abs() <= 2 * eps * (w + bpms2[k].misalignment).abs() + 1e-17), k
