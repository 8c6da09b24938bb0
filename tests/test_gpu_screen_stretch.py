"""Active Screens as items of the one-call lattice stretch (chx_lattice_track_screens / chx_parameter_lattice_track_screens;
screen.py:187-344, segment.py:545-574): the screen's record of the beam and its image come out of the same two launches that track
the beam.

 * against the REFERENCE's own run (tests/golden/screen_stretch.npz, generator tests/golden/generate_golden_screen_stretch.py): six
   drawn lattices with two or three active screens each (mid-lattice and last; cloud-in-cell and histogram; misaligned; binning 2)
   between monitors, apertures and cavities — outgoing beam, every screen's read beam and image, for particles and a ParameterBeam;
   and the README's control step at three sets of magnet settings;
 * against this engine's own walk element by element (`Segment._STRETCH_SCREENS = False`): records bit for bit, images to the
   order of the float atomics;
 * the extent the kernels derive from the pixel size against the reference's tensor expression, bit for bit;
 * the screen's semantics around the record: in-place edits of the beams afterwards, settings changed between track and reading,
   beams above the eager-image size.
Measured on the MI355X (worst case over the lattices; relative to the coordinate's scale / the image's maximum): float64 read beams
and outgoing particles 6.0e-15, images 1.1e-14, ParameterBeam moments 4.7e-16, its images 1.2e-7 (the reference samples the density
at float32-rounded positions); float32 1.4e-6 / 4.7e-6 / 3.7e-7 / 4.2e-7 — the bounds below are >= 4 x those."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "screen_stretch.npz")


def _build(ca, spec, fk):
    kind, kw = spec
    args = {}
    for k, v in kw.items():
        if k == "resolution":
            args[k] = tuple(v)
        elif isinstance(v, (float, list)):
            args[k] = torch.tensor(v, **fk)
        else:
            args[k] = v
    return getattr(ca, kind)(**args, **fk)


class _Spy:
    """Counts the calls of one function of a host module."""

    def __init__(self, host, name, calls):
        self._host, self._name, self._calls = host, name, calls

    def __getattr__(self, name):
        fn = getattr(self._host, name)
        if name != self._name:
            return fn
        return lambda *a: (self._calls.append(len(a)), fn(*a))[1]


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_screen_lattices_vs_reference(dt):
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    g = np.load(GOLDEN)
    fk = {"dtype": dt, "device": "cuda"}
    t = lambda a: torch.tensor(np.asarray(a), **fk)  # noqa: E731
    f64 = dt == torch.float64
    calls, pcalls = [], []
    th = segment._lib.torch_host()
    old = segment._TORCH_HOST
    worst = {"rows": 0.0, "image": 0.0, "pb": 0.0, "pb_image": 0.0}
    try:
        for i in range(int(g["n_lattices"])):
            specs = json.loads(str(g[f"lat{i}_spec"]))
            seg = ca.Segment([_build(ca, s, fk) for s in specs])
            screens = [e for e in seg.elements if isinstance(e, ca.Screen)]
            bpms = [e for e in seg.elements if isinstance(e, ca.BPM)]
            beam = ca.ParticleBeam(t(g[f"lat{i}_in"]), t(g[f"lat{i}_energy"]), particle_charges=t(g[f"lat{i}_q"]),
                                   survival_probabilities=t(g[f"lat{i}_w"]), **fk)
            calls.clear()
            segment._TORCH_HOST = _Spy(th, "lattice_track_screens", calls)
            with torch.no_grad():
                out = seg.track(beam)
            assert len(calls) == 1, (i, calls)            # the whole lattice, screens included, is ONE stretch call
            ref = g[f"lat{i}_out"]
            scale = np.abs(ref).max(axis=0)
            err = (np.abs(out.particles.double().cpu().numpy() - ref) / scale).max()
            worst["rows"] = max(worst["rows"], err)
            assert err < (1e-13 if f64 else 6e-6), (i, err)
            assert float(out.energy) == pytest.approx(float(g[f"lat{i}_energy_out"]), rel=1e-13 if f64 else 1e-6)
            assert float(out.s) == pytest.approx(float(g[f"lat{i}_s_out"]), rel=1e-13 if f64 else 1e-6)
            w_ref, w_got = g[f"lat{i}_w_out"], out.survival_probabilities.double().cpu().numpy()
            assert (np.abs(w_got - w_ref) > 1e-6).sum() <= (0 if f64 else 4), i
            if bpms:
                got_r = torch.stack([b.reading for b in bpms]).double().cpu().numpy()
                ref_r = g[f"lat{i}_readings"]
                live = np.isfinite(ref_r).all(axis=1)
                assert np.abs(got_r[live] - ref_r[live]).max() < (1e-15 if f64 else 3e-7) * (np.abs(ref[:, [0, 2]]).max() + np.abs(ref_r[live]).max())
            for k, scr in enumerate(screens):
                rb = scr.get_read_beam()
                rows_ref = g[f"lat{i}_scr{k}_rows"]
                sc = np.abs(rows_ref).max(axis=0)
                e_rows = (np.abs(rb.particles.double().cpu().numpy() - rows_ref) / sc).max()
                worst["rows"] = max(worst["rows"], e_rows)
                assert e_rows < (1e-13 if f64 else 6e-6), (i, k, e_rows)
                assert np.allclose(rb.particle_charges.double().cpu().numpy(), g[f"lat{i}_scr{k}_q"], rtol=1e-15 if f64 else 1e-7, atol=0)
                w_at = rb.survival_probabilities.double().cpu().numpy()
                assert (np.abs(w_at - g[f"lat{i}_scr{k}_w"]) > 1e-6).sum() <= (0 if f64 else 4), (i, k)
                assert float(rb.energy) == pytest.approx(float(g[f"lat{i}_scr{k}_energy"]), rel=1e-13 if f64 else 1e-6)
                assert float(rb.s) == pytest.approx(float(g[f"lat{i}_scr{k}_s"]), rel=1e-13 if f64 else 1e-6, abs=1e-12)
                img, img_ref = scr.reading.double().cpu().numpy(), g[f"lat{i}_scr{k}_image"]
                assert img.shape == img_ref.shape, (i, k, img.shape, img_ref.shape)
                if scr.method == "cloud-in-cell" or f64:
                    e_img = np.abs(img - img_ref).max() / img_ref.max()
                    worst["image"] = max(worst["image"], e_img)
                    assert e_img < (1e-13 if f64 else 2e-5), (i, k, e_img)
                else:       # a float32 coordinate may fall into the neighbouring histogram bin
                    assert np.abs(img - img_ref).sum() <= 8 * np.abs(g[f"lat{i}_scr{k}_q"]).max(), (i, k)
                assert np.isclose(img.sum(), img_ref.sum(), rtol=1e-12 if f64 else 1e-5)
            if f"lat{i}_pb_mu" not in g.files:
                continue
            pb = ca.ParameterBeam(t(g[f"lat{i}_pb_mu_in"]), t(g[f"lat{i}_pb_cov_in"]), t(g[f"lat{i}_energy"]),
                                  total_charge=t(2e-10), **fk)
            pcalls.clear()
            segment._TORCH_HOST = _Spy(th, "parameter_lattice_track_screens", pcalls)
            with warnings.catch_warnings(), torch.no_grad():
                warnings.simplefilter("ignore")
                pout = seg.track(pb)
            assert len(pcalls) == 1, (i, pcalls)
            size = np.sqrt(np.abs(np.diag(g[f"lat{i}_pb_cov"])))[:6].max()
            e_pb = max(np.abs(pout.mu.double().cpu().numpy() - g[f"lat{i}_pb_mu"])[:6].max() / size,
                       np.abs(pout.cov.double().cpu().numpy() - g[f"lat{i}_pb_cov"])[:6, :6].max() / size ** 2)
            worst["pb"] = max(worst["pb"], e_pb)
            assert e_pb < (1e-13 if f64 else 3e-6), (i, e_pb)
            assert float(pout.energy) == pytest.approx(float(g[f"lat{i}_pb_energy"]), rel=1e-13 if f64 else 1e-6)
            assert float(pout.s) == pytest.approx(float(g[f"lat{i}_pb_s"]), rel=1e-13 if f64 else 1e-6)
            for k, scr in enumerate(screens):
                rb = scr.get_read_beam()
                mu_ref, cov_ref = g[f"lat{i}_pb_scr{k}_mu"], g[f"lat{i}_pb_scr{k}_cov"]
                sz = np.sqrt(np.abs(np.diag(cov_ref)))[:6].max() + np.abs(mu_ref[:6]).max()
                assert np.abs(rb.mu.double().cpu().numpy() - mu_ref)[:6].max() / sz < (1e-13 if f64 else 3e-6), (i, k)
                assert np.abs(rb.cov.double().cpu().numpy() - cov_ref)[:6, :6].max() / sz ** 2 < (1e-13 if f64 else 3e-6), (i, k)
                assert float(rb.energy) == pytest.approx(float(g[f"lat{i}_pb_scr{k}_energy"]), rel=1e-13 if f64 else 1e-6)
                assert float(rb.s) == pytest.approx(float(g[f"lat{i}_pb_scr{k}_s"]), rel=1e-13 if f64 else 1e-6, abs=1e-12)
                assert float(rb.total_charge) == pytest.approx(2e-10, rel=1e-7)
                img, img_ref = scr.reading.double().cpu().numpy(), g[f"lat{i}_pb_scr{k}_image"]
                if f64:
                    assert img.shape == img_ref.shape, (i, k, img.shape, img_ref.shape)
                else:
                    # the number of samples is ceil((right - left) / step) of the TENSORS' values (screen.py:283-287): float32
                    # geometry may round the quotient above the integer, one more row or column than the float64 fixture
                    assert all(0 <= a - b <= 1 for a, b in zip(img.shape, img_ref.shape)), (i, k, img.shape, img_ref.shape)
                    img = img[:img_ref.shape[0], :img_ref.shape[1]]
                # (the reference samples the density at float32-rounded positions, screen.py:283-287; see test_gpu_parameter_beam.py)
                e_img = np.abs(img - img_ref).max() / img_ref.max()
                worst["pb_image"] = max(worst["pb_image"], e_img)
                assert e_img < (5e-7 if f64 else 2e-6), (i, k, e_img)
    finally:
        segment._TORCH_HOST = old
    print("screen stretch vs reference", dt, worst)


def _ares(ca, fk, res=(306, 255)):
    t = lambda v: torch.tensor(v, **fk)  # noqa: E731
    return ca.Segment([
        ca.Marker(name="AREASOLA1"), ca.Drift(t(0.17504), **fk),
        ca.Quadrupole(t(0.122), k1=t(8.2), name="AREAMQZM1", **fk), ca.Drift(t(0.428), **fk),
        ca.Quadrupole(t(0.122), k1=t(-14.3), name="AREAMQZM2", **fk), ca.Drift(t(0.204), **fk),
        ca.VerticalCorrector(t(0.02), angle=t(9e-5), name="AREAMCVM1", **fk), ca.Drift(t(0.204), **fk),
        ca.Quadrupole(t(0.122), k1=t(3.142), name="AREAMQZM3", **fk), ca.Drift(t(0.179), **fk),
        ca.HorizontalCorrector(t(0.02), angle=t(-1e-4), name="AREAMCHM1", **fk), ca.Drift(t(0.45), **fk),
        ca.Screen(resolution=res, pixel_size=t([2.8390e-5, 2.0002e-5]), is_active=True, method="cloud-in-cell", name="AREABSCR1", **fk)])


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_control_step_vs_reference(dt):
    """README.md:43-88: five magnet settings written IN PLACE, track, the screen's image — the reference's numbers."""
    import cheetah_amd as ca

    g = np.load(GOLDEN)
    fk = {"dtype": dt, "device": "cuda"}
    t = lambda a: torch.tensor(np.asarray(a), **fk)  # noqa: E731
    f64 = dt == torch.float64
    seg = _ares(ca, fk)
    beam = ca.ParticleBeam(t(g["control_in"]), t(g["control_energy"]), particle_charges=t(g["control_q"]), **fk)
    pb = ca.ParameterBeam(t(g["control_pb_mu_in"]), t(g["control_pb_cov_in"]), t(g["control_energy"]), **fk)
    settings = [seg.AREAMQZM1.k1, seg.AREAMQZM2.k1, seg.AREAMCVM1.angle, seg.AREAMQZM3.k1, seg.AREAMCHM1.angle]
    with torch.no_grad():
        for k, a in enumerate(g["control_actions"]):
            for target, v in zip(settings, a):
                target.copy_(t(v))
            out = seg.track(beam)
            ref = g[f"control{k}_out"]
            err = (np.abs(out.particles.double().cpu().numpy() - ref) / np.abs(ref).max(axis=0)).max()
            assert err < (1e-13 if f64 else 6e-6), (k, err)
            img, img_ref = seg.AREABSCR1.reading.double().cpu().numpy(), g[f"control{k}_image"]
            assert img.shape == img_ref.shape
            assert np.abs(img - img_ref).max() / img_ref.max() < (1e-12 if f64 else 5e-3), k
            assert np.isclose(img.sum(), img_ref.sum(), rtol=1e-12 if f64 else 1e-5)
            pout = seg.track(pb)
            cov_ref = g[f"control{k}_pb_cov"]
            size = np.sqrt(np.abs(np.diag(cov_ref)))[:6].max() + np.abs(g[f"control{k}_pb_mu"][:6]).max()
            assert np.abs(pout.mu.double().cpu().numpy() - g[f"control{k}_pb_mu"])[:6].max() / size < (1e-12 if f64 else 3e-6), k
            assert np.abs(pout.cov.double().cpu().numpy() - cov_ref)[:6, :6].max() / size ** 2 < (1e-12 if f64 else 3e-6), k
            got_shape, ref_shape = tuple(seg.AREABSCR1.reading.shape), g[f"control{k}_pb_image"].shape
            assert got_shape == ref_shape if f64 else all(0 <= a - b <= 1 for a, b in zip(got_shape, ref_shape)), (got_shape, ref_shape)


def test_screen_extent_bits():
    """The extent the stretch kernels derive from the pixel size = the reference's tensor expression (screen.py:139-148), bit for
    bit: the image's cell indices rest on it."""
    import cheetah_amd as ca
    from cheetah_amd import _lib, _ops

    rng = np.random.default_rng(5)
    for dt in (torch.float32, torch.float64):
        for _ in range(200):
            res = (int(rng.integers(1, 5000)), int(rng.integers(1, 5000)))
            ps = torch.tensor(10.0 ** rng.uniform(-7, -2, size=2), dtype=dt, device="cuda")
            scr = ca.Screen(resolution=res, pixel_size=ps, dtype=dt, device="cuda")
            out = torch.empty(4, dtype=dt, device="cuda")
            _ops.check(_lib.lib().chx_screen_extent(ps.data_ptr(), res[0], res[1], _ops.dtype_code(dt), out.data_ptr(), _ops.stream_ptr()),
                       "chx_screen_extent")
            assert torch.equal(out, scr._compute_extent()), (dt, res, ps)


def _drawn_lattice(ca, fk, seed):
    rng = np.random.default_rng(seed)
    t = lambda v: torch.tensor(v, **fk)  # noqa: E731
    els = []
    for k in range(4):
        els += [ca.Drift(t(rng.uniform(0.1, 0.5)), **fk), ca.Quadrupole(t(0.15), k1=t(rng.uniform(-8, 8)), **fk),
                ca.HorizontalCorrector(t(0.05), angle=t(rng.uniform(-1e-4, 1e-4)), **fk)]
        if k == 1:
            els.append(ca.Cavity(t(0.6), voltage=t(8e6), phase=t(-12.0), frequency=t(1.3e9), **fk))
            els.append(ca.BPM(is_active=True, **fk))
        if k == 2:
            els.append(ca.Aperture(x_max=t(9e-4), y_max=t(7e-4), shape="elliptical", is_active=True, **fk))
        els.append(ca.Screen(resolution=(80, 60), pixel_size=t([6e-5, 5e-5]), misalignment=t([1e-4 * (k - 1), -5e-5 * k]),
                             binning=2 if k == 3 else 1, method="histogram" if k == 2 else "cloud-in-cell", is_active=True,
                             name=f"s{k}", **fk))
    els.append(ca.Drift(t(0.2), **fk))
    return ca.Segment(els)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_stretch_equals_walk(dt):
    """Four screens (mid-lattice, behind a cavity and an aperture, one histogram, one binned) in one stretch call against the walk
    element by element: outgoing beam, records and monitor readings bit for bit; images to the order of the atomics."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator.segment import Segment

    fk = {"dtype": dt, "device": "cuda"}
    torch.manual_seed(3)
    beam = ca.ParticleBeam.from_parameters(num_particles=30_000, sigma_x=torch.tensor(3e-4, **fk), sigma_y=torch.tensor(2e-4, **fk),
                                           energy=torch.tensor(8e7, **fk), **fk)
    beam.survival_probabilities = torch.rand(30_000, **fk)
    seg = _drawn_lattice(ca, fk, 11)
    with torch.no_grad():
        out = seg.track(beam)
        got = [(s.get_read_beam(), s.reading) for s in seg.elements if isinstance(s, ca.Screen)]
        bpm_got = [b.reading.clone() for b in seg.elements if isinstance(b, ca.BPM)]
        Segment._STRETCH_SCREENS = False
        try:
            seg2 = _drawn_lattice(ca, fk, 11)
            ref_out = seg2.track(beam)
            ref = [(s.get_read_beam(), s.reading) for s in seg2.elements if isinstance(s, ca.Screen)]
            bpm_ref = [b.reading.clone() for b in seg2.elements if isinstance(b, ca.BPM)]
        finally:
            Segment._STRETCH_SCREENS = True
    assert torch.equal(out.particles, ref_out.particles) and torch.equal(out.survival_probabilities, ref_out.survival_probabilities)
    assert torch.equal(out.energy, ref_out.energy) and torch.equal(out.s, ref_out.s)
    for a, b in zip(bpm_got, bpm_ref):
        assert torch.allclose(a, b, rtol=1e-6 if dt == torch.float32 else 1e-13, atol=0)
    assert len(got) == len(ref) == 4
    for (rb, img), (rb2, img2) in zip(got, ref):
        assert torch.equal(rb.particles, rb2.particles)
        assert torch.equal(rb.particle_charges, rb2.particle_charges) and torch.equal(rb.survival_probabilities, rb2.survival_probabilities)
        assert torch.equal(rb.energy, rb2.energy) and torch.equal(rb.s, rb2.s)
        assert img.shape == img2.shape
        assert torch.allclose(img, img2, rtol=1e-4 if dt == torch.float32 else 1e-11, atol=float(img2.max()) * (1e-6 if dt == torch.float32 else 1e-13))


def test_record_is_private_and_lazy():
    """screen.py:190 / tests/test_screen.py:198-240 of the reference: editing the incoming or outgoing beam in place afterwards does
    not reach the screen; a setting changed between `track` and the first `reading` is honoured (the reference forms the image when
    it is asked for); beams above the eager-image size are recorded and read on demand."""
    import cheetah_amd as ca

    fk = {"dtype": torch.float32, "device": "cuda"}
    seg = _ares(ca, fk)
    torch.manual_seed(0)
    beam = ca.ParticleBeam.from_twiss(beta_x=torch.tensor(3.14, **fk), beta_y=torch.tensor(42.0, **fk), num_particles=5000, **fk)
    scr = seg.AREABSCR1
    with torch.no_grad():
        out = seg.track(beam)
        assert scr.__dict__["_eager"] is not None and scr.__dict__["_incoming"] is None       # deposited by the stretch call
        img = scr.reading.clone()
        rows = scr.get_read_beam().particles.clone()
        charges = scr.get_read_beam().particle_charges.clone()
        out.particles.mul_(0.7)
        out.energy.mul_(0.5)
        beam.particle_charges.mul_(4.0)
        beam.survival_probabilities.mul_(0.9)
        rb = scr.get_read_beam()
        assert torch.equal(rb.particles, rows) and torch.equal(rb.particle_charges, charges)
        assert float(rb.survival_probabilities.min()) == 1.0 and torch.equal(scr.reading, img)
        beam.particle_charges.div_(4.0)
        beam.survival_probabilities.fill_(1.0)
        # a setting changed before the first reading
        seg.track(beam)
        scr.binning = 3
        img3 = scr.reading
        assert img3.shape == (255 // 3, 306 // 3)
        assert torch.allclose(img3.sum(), img.sum(), rtol=1e-5)
        scr.binning = 1
        seg.track(beam)
        scr.pixel_size.mul_(2.0)            # in place: no epoch, the image key sees the version
        wide = scr.reading
        scr.pixel_size.div_(2.0)
        ref = ca.Screen(resolution=(306, 255), pixel_size=scr.pixel_size * 2.0, is_active=True, **fk)
        ref.track(seg.track(beam))
        assert torch.allclose(wide, ref.reading, rtol=1e-4, atol=1e-6 * float(ref.reading.max()))
        # the kernels follow an in-place change of the pixel size on the NEXT track
        scr.pixel_size.mul_(2.0)
        seg.track(beam)
        assert scr.__dict__["_eager"] is not None
        assert torch.allclose(scr.reading, ref.reading, rtol=1e-4, atol=1e-6 * float(ref.reading.max()))
        scr.pixel_size.div_(2.0)
        # a large beam: recorded, the image on demand
        old = ca.Screen._EAGER_IMAGE_PARTICLES
        ca.Screen._EAGER_IMAGE_PARTICLES = 1000
        try:
            seg.track(beam)
            assert scr.__dict__["_eager"] is None and scr.__dict__["_cached_reading"] is None
            assert torch.allclose(scr.reading, img, rtol=1e-4, atol=1e-6 * float(img.max()))
        finally:
            ca.Screen._EAGER_IMAGE_PARTICLES = old


def test_screen_stretch_declines():
    """What does not ride in the stretch call keeps the walk's results: gradients, a blocking screen (a vectorised beam rides since
    round 6)."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    fk = {"dtype": torch.float32, "device": "cuda"}
    calls = []
    th = segment._lib.torch_host()
    old = segment._TORCH_HOST
    segment._TORCH_HOST = _Spy(th, "lattice_track_screens", calls)
    try:
        seg = _ares(ca, fk, res=(64, 48))
        torch.manual_seed(0)
        beam = ca.ParticleBeam.from_twiss(beta_x=torch.tensor(3.14, **fk), beta_y=torch.tensor(42.0, **fk), num_particles=3000, **fk)
        with torch.no_grad():
            seg.track(beam)
        assert len(calls) == 1
        plain = seg.AREABSCR1.reading.clone()
        # a blocking screen
        seg.AREABSCR1.is_blocking = True
        with torch.no_grad():
            out = seg.track(beam)
        assert len(calls) == 1 and float(out.survival_probabilities.abs().max()) == 0.0
        assert torch.allclose(seg.AREABSCR1.reading, plain, rtol=1e-4, atol=1e-6 * float(plain.max()))
        seg.AREABSCR1.is_blocking = False
        # gradients through a setting
        seg.AREAMQZM1.k1 = torch.nn.Parameter(torch.tensor(8.2, **fk))
        out = seg.track(beam)
        assert len(calls) == 1 and out.particles.requires_grad
        seg.AREABSCR1.get_read_beam().sigma_x.backward()
        assert seg.AREAMQZM1.k1.grad is not None
        with torch.no_grad():
            seg.track(beam)                                   # ... and the plan is back under no_grad
        assert len(calls) == 2
        # three beams in one ParticleBeam: a stretch call too since round 6 (test_screens_of_a_vectorised_beam_ride_in_the_stretch),
        # every beam's image the plain beam's
        many = ca.ParticleBeam(beam.particles.unsqueeze(0).repeat(3, 1, 1), beam.energy, particle_charges=beam.particle_charges, **fk)
        with torch.no_grad():
            seg.track(many)
        images = seg.AREABSCR1.reading
        assert len(calls) == 3 and images.shape == (3, 48, 64)
        for b in range(3):
            assert torch.allclose(images[b], plain, rtol=1e-4, atol=1e-6 * float(plain.max()))
    finally:
        segment._TORCH_HOST = old


def _trainable_lattice(ca, fk, values):
    t = lambda v: torch.tensor(v, **fk)  # noqa: E731
    k1a, k1b, ang = (torch.nn.Parameter(t(v)) for v in values)
    seg = ca.Segment([
        ca.Drift(t(0.6), **fk), ca.Quadrupole(t(0.2), k1=k1a, misalignment=t([1e-4, -5e-5]), **fk), ca.Drift(t(0.4), **fk),
        ca.HorizontalCorrector(t(0.05), angle=ang, **fk), ca.Quadrupole(t(0.15), k1=k1b, tilt=t(0.05), **fk), ca.Drift(t(0.8), **fk),
        ca.Screen(resolution=(64, 48), pixel_size=t([8e-5, 8e-5]), misalignment=t([5e-5, 0.0]), is_active=True, name="scr", **fk)])
    return seg, (k1a, k1b, ang)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_differentiable_stretch_equals_general_path(dt):
    """[run with trainable settings | active Screen] as ONE differentiable node (cheetah_amd._chxtorch RunScreenTrack +
    MomentEntryMappedNode) against the general differentiable path (RunMapPlanned / Apply / clone / MomentEntryMapped): the same
    losses and gradients for beam properties of the read beam and of the outgoing beam (algebraic backward), for a loss on the
    image and on the outgoing particles (particle-sized backward), and with two forward passes before the backward passes."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    fk = {"dtype": dt, "device": "cuda"}
    torch.manual_seed(1)
    beam = ca.ParticleBeam.from_parameters(num_particles=20_000, sigma_x=torch.tensor(2e-4, **fk), sigma_y=torch.tensor(1.5e-4, **fk),
                                           sigma_px=torch.tensor(2e-5, **fk), sigma_py=torch.tensor(2e-5, **fk), energy=torch.tensor(9e7, **fk), **fk)
    beam.survival_probabilities = 0.5 + 0.5 * torch.rand(20_000, **fk)
    calls = []
    th = segment._lib.torch_host()
    old = segment._TORCH_HOST
    segment._TORCH_HOST = _Spy(th, "run_screen_track", calls)

    def losses(seg):
        out = seg.track(beam)
        rb = seg.scr.get_read_beam()
        weights = torch.linspace(0.0, 1.0, 64 * 48, **fk).reshape(48, 64)
        return [rb.sigma_x, rb.sigma_y + 3.0 * rb.mu_x, rb.cov_xpx * 1e4 + out.sigma_x, out.mu_y * 2.0 + rb.sigma_px,
                (seg.scr.reading * weights).sum() * 1e12, out.particles[:, :4].square().mean() * 1e6]

    rel = 2e-4 if dt == torch.float32 else 1e-9
    try:
        values = (4.0, -6.0, 8e-5)
        n_losses = 6
        for which in range(n_losses):
            seg_a, params_a = _trainable_lattice(ca, fk, values)
            calls.clear()
            la = losses(seg_a)[which]
            assert len(calls) == 1, (which, calls)                      # the one-node path was taken
            la.backward()
            segment.Segment._STRETCH_SCREENS = False
            try:
                seg_b, params_b = _trainable_lattice(ca, fk, values)
                lb = losses(seg_b)[which]
                lb.backward()
            finally:
                segment.Segment._STRETCH_SCREENS = True
            assert len(calls) == 1
            assert float(la) == pytest.approx(float(lb), rel=rel), which
            for pa, pb in zip(params_a, params_b):
                assert pa.grad is not None and pb.grad is not None, which
                assert float(pa.grad) == pytest.approx(float(pb.grad), rel=10 * rel, abs=1e-12 * abs(float(lb))), (which, float(pa.grad), float(pb.grad))
        # two forward passes (another beam in between) before the backward passes: every node keeps its own state
        seg, params = _trainable_lattice(ca, fk, values)
        other = ca.ParticleBeam(beam.particles * 1.5, beam.energy, particle_charges=beam.particle_charges, **fk)
        seg.track(beam)
        l1 = seg.scr.get_read_beam().sigma_x
        seg.track(other)
        l2 = seg.scr.get_read_beam().sigma_x
        l1.backward()
        g1 = [float(p.grad) for p in params]
        for p in params:
            p.grad = None
        l2.backward()
        g2 = [float(p.grad) for p in params]
        seg_c, params_c = _trainable_lattice(ca, fk, values)
        seg_c.track(beam)
        seg_c.scr.get_read_beam().sigma_x.backward()
        for a, c in zip(g1, [float(p.grad) for p in params_c]):
            assert a == pytest.approx(c, rel=1e-6)
        assert any(abs(a - b) > 1e-3 * abs(a) for a, b in zip(g1, g2))
        # a setting edited in place between forward and backward is refused like any saved tensor
        seg.track(beam)
        loss = seg.scr.get_read_beam().sigma_x
        with torch.no_grad():
            params[0].add_(0.5)
        with pytest.raises(RuntimeError, match="modified by an inplace operation"):
            loss.backward()
    finally:
        segment._TORCH_HOST = old


def _walk_copy(build, beam, track=lambda seg, b: seg.track(b)):
    """The same lattice built twice: tracked with screens as stretch items, and with screens kept out of the stretches (the walk)."""
    from cheetah_amd.accelerator.segment import Segment

    seg_a = build()
    out_a = track(seg_a, beam)
    Segment._STRETCH_SCREENS = False
    try:
        seg_b = build()
        out_b = track(seg_b, beam)
    finally:
        Segment._STRETCH_SCREENS = True
    return seg_a, out_a, seg_b, out_b


def _same_screens(ca, seg_a, seg_b, dt):
    sa = [e for e in seg_a.modules() if isinstance(e, ca.Screen) and e.is_active]
    sb = [e for e in seg_b.modules() if isinstance(e, ca.Screen) and e.is_active]
    assert len(sa) == len(sb) and sa
    for a, b in zip(sa, sb):
        ra, rb = a.get_read_beam(), b.get_read_beam()
        assert torch.equal(ra.particles, rb.particles) and torch.equal(ra.particle_charges, rb.particle_charges)
        assert torch.equal(ra.survival_probabilities, rb.survival_probabilities) and torch.equal(ra.energy, rb.energy) and torch.equal(ra.s, rb.s)
        ia, ib = a.reading, b.reading
        assert ia.shape == ib.shape
        assert torch.allclose(ia, ib, rtol=1e-4 if dt == torch.float32 else 1e-11, atol=float(ib.max()) * (1e-6 if dt == torch.float32 else 1e-13))


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [1, 63, 257, 10_001])
def test_screen_stretch_shapes_and_positions(dt, n):
    """Screens first, last, back to back, six of them (more than one stretch call takes), inside a nested Segment, odd particle
    counts (the record's parts start at every alignment): equal to the walk."""
    import cheetah_amd as ca

    fk = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **fk)  # noqa: E731
    torch.manual_seed(n)
    full = ca.ParticleBeam.from_parameters(num_particles=max(n, 100), sigma_x=t(2e-4), sigma_y=t(2e-4), energy=t(1e8), **fk)
    beam = ca.ParticleBeam(full.particles[:n].contiguous(), full.energy, particle_charges=full.particle_charges[:n].contiguous(),
                           survival_probabilities=torch.rand(n, **fk), **fk)

    def scr(k, **kw):
        return ca.Screen(resolution=(40, 30), pixel_size=t([5e-5, 6e-5]), misalignment=t([2e-5 * k, -1e-5 * k]), is_active=True, name=f"s{k}",
                         **kw, **fk)

    def build():
        inner = ca.Segment([ca.Drift(t(0.2), **fk), scr(3), ca.Quadrupole(t(0.1), k1=t(-5.0), **fk)], name="inner")
        return ca.Segment([scr(0), ca.Drift(t(0.3), **fk), ca.Quadrupole(t(0.2), k1=t(6.0), **fk), scr(1), scr(2), ca.Drift(t(0.1), **fk), inner,
                           ca.HorizontalCorrector(t(0.05), angle=t(1e-4), **fk), scr(4), ca.Drift(t(0.2), **fk), scr(5), ca.Drift(t(0.4), **fk),
                           scr(6)])

    with torch.no_grad():
        seg_a, out_a, seg_b, out_b = _walk_copy(build, beam)
        assert torch.equal(out_a.particles, out_b.particles) and torch.equal(out_a.s, out_b.s)
        assert out_a.particles.data_ptr() != beam.particles.data_ptr()
        _same_screens(ca, seg_a, seg_b, dt)
        # tracked again (the plans exist now), and with a screen switched off in between
        out_a2 = seg_a.track(beam)
        assert torch.equal(out_a2.particles, out_b.particles)
        seg_a.s1.is_active = False
        seg_b.s1.is_active = False
        out_a3 = seg_a.track(beam)
        assert torch.equal(out_a3.particles, out_b.particles)
        _same_screens(ca, seg_a, seg_b, dt)


def test_screen_stretch_keeps_the_walks_errors_and_copies():
    """A float64 screen under a float32 beam raises like the reference's scatter does; a trainable misalignment and a histogram / kde
    screen keep working; a tracked lattice can be deep-copied, pickled and moved."""
    import copy
    import pickle

    import cheetah_amd as ca

    fk = {"dtype": torch.float32, "device": "cuda"}
    t = lambda v: torch.tensor(v, **fk)  # noqa: E731
    torch.manual_seed(0)
    beam = ca.ParticleBeam.from_parameters(num_particles=4000, sigma_x=t(2e-4), sigma_y=t(2e-4), **fk)
    wide = ca.Screen(resolution=(32, 32), pixel_size=torch.tensor([1e-4, 1e-4], dtype=torch.float64, device="cuda"), is_active=True,
                     dtype=torch.float64, device="cuda")
    seg = ca.Segment([ca.Drift(t(0.5), **fk), wide])
    with torch.no_grad():
        seg.track(beam)
        with pytest.raises(RuntimeError, match="scatter"):
            wide.reading
    for method in ("histogram", "kde"):
        def build(method=method):
            return ca.Segment([ca.Drift(t(0.5), **fk), ca.Quadrupole(t(0.2), k1=t(3.0), **fk),
                               ca.Screen(resolution=(32, 24), pixel_size=t([1e-4, 1e-4]), method=method, is_active=True, name="scr", **fk)])

        with torch.no_grad():
            seg_a, out_a, seg_b, out_b = _walk_copy(build, beam)
        assert torch.equal(out_a.particles, out_b.particles)
        assert torch.allclose(seg_a.scr.reading, seg_b.scr.reading, rtol=1e-4, atol=1e-6 * float(seg_b.scr.reading.max()))
    # trainable misalignment: the screen stays out of the stretch, the gradient reaches it through the image
    mis = torch.nn.Parameter(t([1e-5, 2e-5]))
    seg = ca.Segment([ca.Drift(t(0.5), **fk), ca.Screen(resolution=(32, 24), pixel_size=t([1e-4, 1e-4]), misalignment=mis, is_active=True,
                                                       name="scr", **fk)])
    seg.track(beam)
    assert seg.scr.get_read_beam().particles.requires_grad
    seg.scr.get_read_beam().mu_x.backward()
    assert mis.grad is not None and float(mis.grad[0]) == pytest.approx(-1.0, rel=1e-4)
    # copies of a tracked lattice
    def build():
        return ca.Segment([ca.Drift(t(0.5), **fk), ca.Screen(resolution=(32, 24), pixel_size=t([1e-4, 1e-4]), is_active=True, name="scr", **fk)])

    with torch.no_grad():
        seg = build()
        seg.track(beam)
        image = seg.scr.reading.clone()
        for other in (copy.deepcopy(seg), pickle.loads(pickle.dumps(seg))):
            assert torch.equal(other.scr.get_read_beam().particles, seg.scr.get_read_beam().particles)
            assert torch.allclose(other.scr.reading, image)
            other.track(beam)
            assert torch.allclose(other.scr.reading, image, rtol=1e-4, atol=1e-6 * float(image.max()))
        moved = copy.deepcopy(seg).double()
        assert moved.scr.reading.dtype == torch.float64 and torch.allclose(moved.scr.reading.float(), image, rtol=1e-5)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [700, 40_001, 1_000_003])
def test_moment_sums_of_the_recorded_beam(dt, n):
    """Round 6: the particle pass of the differentiable [run | Screen] stretch leaves the one-pass sums of the recorded beam's 29
    moments about the beam's first row as it stands at the screen, added into at most 64 sets (chx_lattice_screen.mom_partials);
    chx_lattice_screen_moments adds the sets and finalises. Against chx_moments of the recorded rows (pinned to the reference's weighted statistics,
    utils/statistics.py:4-62, by tests/golden/moment_outliers.npz): beams off axis by many sigma, rows with zero weight, a far
    outlier in a workgroup's first slot; one (n < 1e6) and two particles per lane; n not a multiple of the tile."""
    import cheetah_amd as ca
    from cheetah_amd import _lib, _ops

    fk = {"dtype": dt, "device": "cuda"}
    torch.manual_seed(n)
    beam = ca.ParticleBeam.from_parameters(num_particles=n, sigma_x=torch.tensor(2e-4, **fk), sigma_y=torch.tensor(1.5e-4, **fk),
                                           mu_x=torch.tensor(3e-2, **fk), mu_y=torch.tensor(-5e-3, **fk), energy=torch.tensor(9e7, **fk), **fk)
    w = 0.25 + 0.75 * torch.rand(n, **fk)
    w[::7] = 0.0
    beam.survival_probabilities = w
    if n > 600:
        beam.particles[512, 2] -= 0.3                              # far outliers (not the first row: that one is the sums' centre)
        beam.particles[256, 4] += 0.2
    seg, params = _trainable_lattice(ca, fk, (4.0, -6.0, 8e-5))
    seg.track(beam)
    rb = seg.scr._incoming_beam()            # the screen's record (the read beam of this lattice is shifted by its misalignment)
    rows, wr = rb.particles, rb.survival_probabilities
    tag = getattr(rows, "_chx_partials", None)
    assert tag is not None and tag[1] == rows._version and tag[2] is wr
    nblk = _lib.lib().chx_lattice_moment_blocks(n, 1)
    assert tag[0].numel() == 29 * 64 + 6 and nblk == min(64, -(-n // (512 if n >= 1_000_000 else 256)))
    got = torch.empty(29, dtype=torch.float64, device="cuda")
    picked = torch.empty((), **fk)
    _ops.check(_lib.lib().chx_lattice_screen_moments(tag[0].data_ptr(), nblk, _ops.dtype_code(dt), got.data_ptr(), 8, 1, picked.data_ptr(),
                                                     _ops.stream_ptr()), "chx_lattice_screen_moments")
    want = _ops._moments_raw(rows.detach().reshape(1, n, 7).contiguous(), wr.reshape(1, n).contiguous(), 1, n).reshape(29)
    sig = want[[8, 14, 19, 23, 26, 28]].sqrt()
    assert torch.equal(got[:2], want[:2]) or torch.allclose(got[:2], want[:2], rtol=1e-14, atol=0)
    assert ((got[2:8] - want[2:8]).abs() <= 1e-12 * (sig + want[2:8].abs())).all()
    k = 8
    for i in range(6):
        for j in range(i, 6):
            assert abs(float(got[k] - want[k])) <= 1e-11 * float(sig[i] * sig[j]), (i, j, float(got[k]), float(want[k]))
            k += 1
    assert float(picked) == pytest.approx(float(want[8].sqrt()), rel=2e-7 if dt == torch.float32 else 1e-12)
    # and the property itself goes that way: same value as the sums', gradient as the general path's
    sx = rb.sigma_x
    assert float(sx) == float(picked)
    sx.backward()
    assert all(p.grad is not None for p in params)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_property_of_the_record_is_one_node_on_the_settings(dt):
    """Round 6: a beam property of the record of a [run | Screen] stretch hangs on the run's SETTINGS as one autograd node whose
    backward is one launch (cheetah_amd._chxtorch RunMomentEntry -> chx_run_vjp_entry: the builders' VJP forms dL/dC from the
    property's gradient itself) instead of on the composed map, whose node carried the gradient on (two nodes, two launches).
    Same values, same gradients — a strength shared by two quadrupoles, a tilted one, a corrector angle; a second
    property read from the memoised moments; a run longer than the fused kernel takes (two launches behind the same entry point)."""
    import cheetah_amd as ca
    from cheetah_amd import _ops

    fk = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **fk)  # noqa: E731
    torch.manual_seed(3)
    for extra_drifts in (0, 20):
        k1, ang = torch.nn.Parameter(t(4.0)), torch.nn.Parameter(t(6e-5))
        els = [ca.Drift(t(0.6), **fk), ca.Quadrupole(t(0.2), k1=k1, **fk), ca.Drift(t(0.4), **fk), ca.HorizontalCorrector(t(0.05), angle=ang, **fk),
               ca.Quadrupole(t(0.15), k1=k1, tilt=t(0.05), misalignment=t([1e-4, -5e-5]), **fk), ca.Drift(t(0.8), **fk)]
        els += [ca.Drift(t(0.01), **fk) for _ in range(extra_drifts)]
        seg = ca.Segment(els + [ca.Screen(resolution=(64, 48), pixel_size=t([8e-5, 8e-5]), is_active=True, name="scr", **fk)])
        base = ca.ParticleBeam.from_parameters(num_particles=30_000, sigma_x=t(2e-4), sigma_y=t(1.5e-4), sigma_px=t(2e-5), sigma_py=t(2e-5),
                                               mu_x=t(1e-4), energy=t(9e7), **fk)
        beam = base
        got = {}
        for one_node in (True, False):
            _ops.ONE_NODE_PROPERTY[0] = one_node
            try:
                for prop in ("sigma_x", "sigma_y", "mu_x", "sigma_tau", "sigma_px"):
                    for p in (k1, ang):
                        p.grad = None
                    seg.track(beam)
                    rb = seg.scr.get_read_beam()
                    v = getattr(rb, prop)
                    assert ("RunMomentEntry" in v.grad_fn.name()) == one_node, v.grad_fn.name()
                    second = rb.sigma_y if prop == "sigma_x" else None          # (from the memoised moments of the same record)
                    (v if second is None else v + 2.0 * second).backward()
                    got[(one_node, prop)] = (v.detach().clone(), k1.grad.clone(), ang.grad.clone())
            finally:
                _ops.ONE_NODE_PROPERTY[0] = True
        rel = 1e-5 if dt == torch.float32 else 1e-11
        for prop in ("sigma_x", "sigma_y", "mu_x", "sigma_tau", "sigma_px"):
            a, b = got[(True, prop)], got[(False, prop)]
            # (two tracks: the particle pass adds its sums with atomics, whose order — hence the last bits in float64 — is not fixed)
            assert torch.allclose(a[0], b[0], rtol=1e-12 if dt == torch.float64 else 2e-7, atol=0), prop
            for ga, gb, name in zip(a[1:], b[1:], ("k1", "angle")):
                assert torch.isfinite(ga).all() and abs(float(ga - gb)) <= rel * abs(float(gb)) + 1e-30, (extra_drifts, prop, name, float(ga), float(gb))
            assert float(a[1]) != 0.0 or prop in ("sigma_tau",)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_histogram_screen_rides_in_the_particle_pass(dt):
    """Round 6: the 'histogram' image (the ARES lattice file's default method; screen.py:292-311) is deposited by the stretch's
    particle pass like the cloud-in-cell one — torch.histogramdd's bin search on the edges torch.linspace gives (chx_hist2d's
    arithmetic, bit-identical pixel indices) — instead of a memset and a deposit launch at the first reading. Equal to the reading
    formed from the record (the walk's path); an in-place edit of the pixel size, which changes the edges, is followed."""
    import cheetah_amd as ca
    from cheetah_amd import _ops

    fk = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **fk)  # noqa: E731
    seg = ca.Segment([ca.Drift(t(0.3), **fk), ca.Quadrupole(t(0.15), k1=t(5.0), **fk), ca.Drift(t(0.4), **fk),
                      ca.Screen(resolution=(96, 64), pixel_size=t([2.5e-5, 3.0e-5]), misalignment=t([4e-5, -2e-5]), binning=2,
                                method="histogram", is_active=True, name="scr", **fk), ca.Drift(t(0.1), **fk)])
    torch.manual_seed(5)
    beam = ca.ParticleBeam.from_parameters(num_particles=40_000, sigma_x=t(2e-4), sigma_y=t(2.5e-4), **fk)
    beam.survival_probabilities = torch.rand(40_000, **fk)
    # particles exactly on bin edges and on the outer edges (right-inclusive last bin), just outside, far outside
    ex, ey = seg.scr.pixel_bin_edges
    with torch.no_grad():
        seg.track(beam)
        assert seg.scr.__dict__["_eager"] is not None            # deposited by the stretch call
        img = seg.scr.reading.clone()
        rb = seg.scr.get_read_beam()
        want = _ops.hist2d(seg.scr._incoming_beam().particles, ex, ey, charge=beam.particle_charges, survival=beam.survival_probabilities,
                           shift=seg.scr.misalignment)
        assert img.shape == want.shape == (32, 48)
        assert torch.allclose(img, want, rtol=1e-5 if dt == torch.float32 else 1e-12, atol=0)
        assert float(img.sum()) > 0 and int((img != 0).sum()) == int((want != 0).sum())      # the same pixels are lit
        # pixel size edited in place: other edges on the next track
        seg.scr.pixel_size.mul_(1.5)
        seg.track(beam)
        ex2, ey2 = seg.scr.pixel_bin_edges
        assert not torch.equal(ex2, ex)
        want2 = _ops.hist2d(seg.scr._incoming_beam().particles, ex2, ey2, charge=beam.particle_charges,
                            survival=beam.survival_probabilities, shift=seg.scr.misalignment)
        assert seg.scr.__dict__["_eager"] is not None
        assert torch.allclose(seg.scr.reading, want2, rtol=1e-5 if dt == torch.float32 else 1e-12, atol=0)
        assert float((want2 - want).abs().max()) > 0.1 * float(want.max())      # (other edges: another image)
    assert rb.particles.shape == (40_000, 7)


def test_host_step_checks_what_python_hands_it():
    """cheetah_amd._chxtorch trusts its caller for nothing that would write out of bounds on the device: shapes, lengths, dtypes,
    contiguity and devices of the tensors of a stretch call are checked against the plan — a real call's arguments,
    replayed with one of them spoiled, raise instead of launching."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    fk = {"dtype": torch.float32, "device": "cuda"}
    seg = _ares(ca, fk)
    beam = ca.ParticleBeam.from_twiss(beta_x=torch.tensor(3.14, **fk), beta_y=torch.tensor(42.0, **fk), num_particles=2000, **fk)
    pbeam = ca.ParameterBeam.from_twiss(beta_x=torch.tensor(3.14, **fk), beta_y=torch.tensor(42.0, **fk), **fk)
    th = segment._lib.torch_host()
    seen = {}

    class Recorder:
        def __getattr__(self, name):
            fn = getattr(th, name)

            def call(*a):
                seen[name] = a
                return fn(*a)
            return call

    old = segment._TORCH_HOST
    segment._TORCH_HOST = Recorder()
    try:
        with torch.no_grad():
            seg.track(beam)
            seg.track(pbeam)
    finally:
        segment._TORCH_HOST = old
    a = list(seen["lattice_track_screens"])
    good = th.lattice_track_screens(*a)
    assert good[0].shape == (2000, 7)
    x, energy, charges = a[1], a[2], a[4]
    spoiled = [
        (1, x[:, :6].contiguous()), (1, x.double()), (1, x.t().contiguous().t()), (1, x.cpu()), (1, x.reshape(-1)),
        (2, torch.stack([energy, energy])), (2, energy.double()), (4, charges[:-1]), (4, charges.double()), (4, charges[::2]),
        (5, torch.ones(1999, **fk)),
    ]
    for index, bad in spoiled:
        b = list(a)
        b[index] = bad
        with pytest.raises((ValueError, TypeError)):
            th.lattice_track_screens(*b)
    b = list(a)
    b[1] = "not a tensor"
    with pytest.raises(TypeError):
        th.lattice_track_screens(*b)
    pa = list(seen["parameter_lattice_track_screens"])
    assert th.parameter_lattice_track_screens(*pa)[0].shape == (7,)
    for index, bad in ((1, pa[1][:6]), (2, pa[2][:6]), (2, pa[2].double()), (3, pa[3].double()), (1, pa[1].cpu())):
        b = list(pa)
        b[index] = bad
        with pytest.raises((ValueError, TypeError)):
            th.parameter_lattice_track_screens(*b)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("lead", [(3,), (2, 2)])
def test_screens_of_a_vectorised_beam_ride_in_the_stretch(dt, lead):
    """Round 6: a vectorised ParticleBeam (B beams of N particles under one lattice setting) through [run | monitor | aperture |
    active Screen]+ is ONE stretch call too: every screen's record holds the B beams ((B, N, 7) rows, (B, N) charges and survival
    probabilities), every cloud-in-cell image B images, the monitors' readings and the thinned survival probabilities carry the
    beam's vector dims. Against the elements tracked one by one (screen.py:187-344, bpm.py:77-87, aperture.py:90-135). A 'histogram'
    screen refuses vectorised beams like the reference's (screen.py:292-294): its stretch is taken without the screen."""
    import cheetah_amd as ca
    from cheetah_amd.accelerator import _planner

    fk = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **fk)  # noqa: E731

    def lattice(method):
        return [ca.Drift(t(0.3), **fk), ca.Quadrupole(t(0.15), k1=t(5.0), **fk), ca.BPM(is_active=True, **fk), ca.Drift(t(0.4), **fk),
                ca.Aperture(x_max=t(6e-4), y_max=t(5e-4), shape="elliptical", is_active=True, **fk),
                ca.Quadrupole(t(0.15), k1=t(-4.0), tilt=t(0.1), **fk), ca.Drift(t(0.2), **fk),
                ca.Screen(resolution=(48, 40), pixel_size=t([5e-5, 5e-5]), method=method, is_active=True, misalignment=t([4e-5, -2e-5]), **fk),
                ca.HorizontalCorrector(t(0.05), angle=t(1e-4), **fk), ca.Drift(t(0.3), **fk)]

    torch.manual_seed(5)
    n, B = 5000, int(torch.tensor(lead).prod())
    one = ca.ParticleBeam.from_parameters(num_particles=n, sigma_x=t(3e-4), sigma_y=t(2.5e-4), sigma_px=t(3e-5), sigma_py=t(2e-5), energy=t(8e7), **fk)
    shifts = torch.linspace(-2e-4, 2e-4, B, **fk).reshape(lead + (1, 1)) * torch.tensor([1.0, 0, 0.4, 0, 0, 0, 0], **fk)
    beam = ca.ParticleBeam(one.particles + shifts, one.energy, particle_charges=one.particle_charges,
                           survival_probabilities=0.4 + 0.6 * torch.rand(lead + (n,), **fk), **fk)
    els_a, els_b = lattice("cloud-in-cell"), lattice("cloud-in-cell")
    before = dict(_planner.TAKEN)
    out = ca.Segment(els_a).track(beam)
    assert _planner.TAKEN["lattice_stretch"] == before["lattice_stretch"] + 1 and _planner.TAKEN["element"] == before["element"]
    ref = beam
    for e in els_b:
        ref = e.track(ref)
    tol = 1e-12 if dt == torch.float64 else 2e-6
    assert out.particles.shape == lead + (n, 7) and out.survival_probabilities.shape == lead + (n,)
    assert torch.allclose(out.particles, ref.particles, rtol=tol, atol=tol * 1e-3)
    assert torch.equal(out.survival_probabilities, ref.survival_probabilities)
    assert els_a[2].reading.shape == lead + (2,) and torch.allclose(els_a[2].reading, els_b[2].reading, rtol=1e-5, atol=1e-9)
    img, want = els_a[7].reading, els_b[7].reading
    assert img.shape == lead + (40, 48) and torch.allclose(img, want, rtol=1e-4 if dt == torch.float32 else 1e-10, atol=float(want.max()) * 1e-6)
    assert float((img.reshape(B, -1)[0] - img.reshape(B, -1)[-1]).abs().max()) > 1e-2 * float(img.max())   # (the beams differ: so do their images)
    rb, wb = els_a[7].get_read_beam(), els_b[7].get_read_beam()
    assert rb.particles.shape == lead + (n, 7) and rb.particle_charges.shape[-1] == n
    assert torch.allclose(rb.particles, wb.particles, rtol=tol, atol=tol * 1e-3)
    assert torch.allclose(rb.sigma_x, wb.sigma_x, rtol=1e-5) and rb.sigma_x.shape == lead
    # the 'histogram' method: no image for a vectorised beam, in the stretch as in the walk
    els_h = lattice("histogram")
    before = dict(_planner.TAKEN)
    out_h = ca.Segment(els_h).track(beam)
    assert _planner.TAKEN["lattice_stretch"] >= before["lattice_stretch"] + 1          # (the stretch without the screen, not the walk)
    assert torch.allclose(out_h.particles, ref.particles, rtol=tol, atol=tol * 1e-3)
    with pytest.raises(NotImplementedError, match="does not support vectorization"):
        els_h[7].reading
