"""Fused observables against the REAL reference (tests/golden/observables.npz, written by
tests/golden/generate_golden_observables.py):

* `Segment.track_screen_reading` — one Screen image per lattice setting without the (B, N, 7) tracked array
  (`chx_cic_deposit_mapped`); reference: /root/reference/cheetah/accelerator/screen.py:327-339 on the vectorised beam;
* `Segment.get_beam_attrs_along_segment` — every moment attribute after every element from ONE pass over the particles
  (`chx_compose_prefix` + `chx_track_moments`); reference: segment.py:658-700;
* `Segment.track_moments(exact=False)` — mu' = R mu, Sigma' = R Sigma R^T.
"""
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


def tdt(tag):
    return torch.float64 if tag == "f64" else torch.float32


def subcell(ca, dt, k1, screen_active):
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    return ca.Segment([
        ca.Marker(name="AREASOLA1", **kw), ca.Drift(t(0.17504), **kw), ca.Quadrupole(t(0.122), k1=k1, name="AREAMQZM1", **kw),
        ca.Drift(t(0.428), **kw), ca.Quadrupole(t(0.122), k1=t(-14.3), name="AREAMQZM2", **kw), ca.Drift(t(0.204), **kw),
        ca.VerticalCorrector(t(0.02), angle=t(9e-5), name="AREAMCVM1", **kw), ca.Drift(t(0.204), **kw),
        ca.Quadrupole(t(0.122), k1=t(3.142), name="AREAMQZM3", **kw), ca.Drift(t(0.179), **kw),
        ca.HorizontalCorrector(t(0.02), angle=t(-1e-4), name="AREAMCHM1", **kw), ca.Drift(t(0.45), **kw),
        ca.Screen(resolution=(96, 64), pixel_size=t([2.0e-5, 1.5e-5]), misalignment=t([3e-5, -2e-5]), name="AREABSCR1",
                  method="cloud-in-cell", is_active=screen_active, **kw),
    ])


def make_beam(ca, g, tag):
    dt = tdt(tag)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()  # noqa: E731
    return ca.ParticleBeam(d(g[f"in_{tag}"]), torch.tensor(1e8, dtype=dt, device="cuda"), particle_charges=d(g[f"charges_{tag}"]),
                           survival_probabilities=d(g[f"survival_{tag}"]), species=ca.Species("electron", dtype=dt, device="cuda"))


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_screen_images_per_setting_without_tracked_particles(ca, golden, tag):
    g = golden("observables.npz")
    dt = tdt(tag)
    beam = make_beam(ca, g, tag)
    k1 = torch.from_numpy(g[f"k1_{tag}"]).to(dt).cuda()
    seg = subcell(ca, dt, k1, True)
    img = seg.track_screen_reading(beam)                       # fused: no (16, N, 7) array
    assert img.shape == (16, 64, 96)
    rows = [int(r) for r in g["rows"]]
    ref = g[f"images_{tag}"].astype(np.float64)
    got = img[rows].double().cpu().numpy()
    scale = np.abs(ref).max()
    # float atomics: order-dependent sums; the occupancy pattern (which pixels received charge) must match exactly
    assert np.array_equal(got != 0, ref != 0)
    assert np.max(np.abs(got - ref)) <= (1e-11 if tag == "f64" else 2e-5) * scale
    assert np.allclose(img.double().sum(dim=(-1, -2)).cpu().numpy(), g[f"image_sums_{tag}"], rtol=1e-11 if tag == "f64" else 1e-5)
    # identical to the unfused product path (track, then Screen.reading): same cells, same addends
    out = seg.track(beam)
    unfused = seg.AREABSCR1.reading
    assert torch.equal(unfused != 0, img != 0)
    assert torch.allclose(unfused, img, rtol=1e-12 if tag == "f64" else 2e-5, atol=0)
    assert np.allclose(out.sigma_x.double().cpu().numpy(), g[f"sigma_x_{tag}"], rtol=1e-10 if tag == "f64" else 2e-5)
    # inactive screen / other endings fall back to track + reading
    seg.AREABSCR1.is_active = False
    assert float(seg.track_screen_reading(beam).abs().sum()) >= 0.0


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_beam_attrs_along_segment_from_one_pass(ca, golden, tag):
    g = golden("observables.npz")
    dt = tdt(tag)
    beam = make_beam(ca, g, tag)
    seg = subcell(ca, dt, torch.tensor(8.2, dtype=dt, device="cuda"), False)
    names = tuple(str(g["attrs"]).split(","))
    assert seg._attrs_along_fused(names, beam) is not None, "the all-linear lattice must take the fused path"
    vals = seg.get_beam_attrs_along_segment(names, beam)
    rtol = 1e-9 if tag == "f64" else 3e-5
    for name, v in zip(names, vals):
        ref = g[f"along_{name}_{tag}"].astype(np.float64)
        got = v.double().cpu().numpy()
        assert got.shape == ref.shape == (14,), (name, got.shape, ref.shape)
        tol = rtol * np.abs(ref).max() if name.startswith(("mu_", "cov_", "alpha_")) else 0.0
        r = rtol
        if tag == "f32" and name.startswith(("emittance", "beta_", "alpha_")):
            # emittance / Twiss are differences of products of second moments: the REFERENCE's own fp32 values are off its
            # fp64 values by up to 1.1e-3 (alpha_y) along this lattice; judged against the fp64 reference instead
            ref, r = g[f"along_{name}_f64"], 3e-3
        assert np.allclose(got, ref, rtol=r, atol=tol), (name, got, ref)
    # a single name returns a tensor; the element-by-element path (forced through an attribute the fused path does not
    # offer) agrees with the fused one
    sx = seg.get_beam_attrs_along_segment("sigma_x", beam)
    assert torch.equal(sx, vals[0])
    both = seg.get_beam_attrs_along_segment(("sigma_x", "x"), beam)     # "x" needs the particles -> generic path
    assert both[1].shape == (14, beam.particles.shape[0])
    assert torch.allclose(both[0].double(), sx.double(), rtol=1e-12 if tag == "f64" else 2e-6)
    # resolution: split lattice (segment.py:644-647)
    sx5, by5, s5 = seg.get_beam_attrs_along_segment(("sigma_x", "beta_y", "s"), beam, resolution=0.05)
    for got, name in ((sx5, "sigma_x"), (by5, "beta_y"), (s5, "s")):
        ref = g[f"along5cm_{name}_{tag}"].astype(np.float64)
        assert got.shape == ref.shape
        if tag == "f32" and name == "beta_y":
            ref = g["along5cm_beta_y_f64"]
        assert np.allclose(got.double().cpu().numpy(), ref, rtol=3e-3 if tag == "f32" and name == "beta_y" else rtol * 3), name


def test_algebraic_moment_transport_option(ca, golden):
    g = golden("observables.npz")
    for tag in ("f64", "f32"):
        dt = tdt(tag)
        beam = make_beam(ca, g, tag)
        k1 = torch.from_numpy(g[f"k1_{tag}"]).to(dt).cuda()
        seg = subcell(ca, dt, k1, False)
        exact = seg.track_moments(beam)
        alg = seg.track_moments(beam, exact=False)
        tol = 1e-12 if tag == "f64" else 2e-6
        assert torch.allclose(alg.sigma_x.double(), exact.sigma_x.double(), rtol=tol)
        assert torch.allclose(alg.sigma_y.double(), exact.sigma_y.double(), rtol=tol)
        assert torch.allclose(alg.cov.double(), exact.cov.double(), rtol=tol * 10, atol=tol * float(exact.cov.abs().max()))
        assert np.allclose(alg.sigma_x.double().cpu().numpy(), g[f"sigma_x_{tag}"], rtol=1e-10 if tag == "f64" else 2e-5)
        assert float(alg.s) == pytest.approx(float(exact.s))


def test_mapped_deposit_is_bit_identical_to_track_then_deposit(ca):
    """chx_cic_deposit_mapped against chx_apply_affine7 + chx_cic_indices: the cell indices and fractions of the
    on-the-fly coordinates equal those of the tracked particles bit for bit (3-D, per-row maps, shift)."""
    from cheetah_amd import _ops

    torch.manual_seed(5)
    for dt in (torch.float32, torch.float64):
        x = torch.randn(1000, 7, dtype=dt, device="cuda") * 1e-3
        x[:, 6] = 1
        R = torch.eye(7, dtype=dt, device="cuda").repeat(5, 1, 1) + 0.1 * torch.randn(5, 7, 7, dtype=dt, device="cuda")
        R[:, 6] = 0
        R[:, 6, 6] = 1
        ext = torch.tensor([[-3e-3, 3e-3], [-2e-3, 2e-3], [-4e-3, 4e-3]], dtype=dt, device="cuda")
        shift = torch.tensor([1e-4, -2e-4, 0.0], dtype=dt, device="cuda")
        q = torch.rand(1000, dtype=dt, device="cuda")
        tracked = _ops.apply_map(x, R)
        want = _ops.cic_deposit(tracked, (0, 2, 4), (16, 12, 10), ext, charge=q, shift=shift, mode="direct")
        got = _ops.cic_deposit_mapped(x, R, (0, 2, 4), (16, 12, 10), ext, charge=q, shift=shift)
        assert got.shape == want.shape == (5, 16, 12, 10)
        assert torch.equal(got != 0, want != 0)
        assert torch.allclose(got, want, rtol=1e-12 if dt == torch.float64 else 1e-5, atol=0)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_beam_attrs_along_segment_parameter_beam_and_nested_cells(ca, dt):
    """A ParameterBeam and a lattice of nested cells take the fused path too (prefix maps through chx_parameter_track /
    chx_track_moments); against the walk element by element (segment.py:658-700): positions are the ends of the segment's OWN
    elements, a nested cell counts once."""
    from cheetah_amd.accelerator.segment import Segment

    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731

    def cell(i):
        return [ca.Quadrupole(t(0.2), k1=t(2.2 if i % 2 == 0 else -2.2), **kw), ca.Drift(t(0.8), **kw),
                ca.HorizontalCorrector(t(0.05), angle=t(1e-5 * i), **kw)]

    flat = ca.Segment([e for i in range(12) for e in cell(i)])
    nested = ca.Segment([ca.Segment(cell(2 * i) + cell(2 * i + 1)) for i in range(6)] + [ca.Drift(t(0.3), **kw)])
    names = ("beta_x", "alpha_y", "sigma_x", "mu_x", "emittance_y", "s", "energy")
    pbeam = ca.ParameterBeam.from_twiss(beta_x=t(3.14), beta_y=t(12.0), alpha_x=t(0.4), energy=t(1e8), **kw)
    beam = ca.ParticleBeam.from_twiss(beta_x=t(3.14), beta_y=t(12.0), alpha_x=t(0.4), energy=t(1e8), num_particles=20_000, **kw)
    tol = 2e-4 if dt == torch.float32 else 1e-9
    for seg, incoming, n_pos in ((flat, pbeam, 37), (nested, pbeam, 8), (nested, beam, 8)):
        with torch.no_grad():
            assert seg._attrs_along_fused(names, incoming) is not None
            got = seg.get_beam_attrs_along_segment(names, incoming)
            orig = Segment._attrs_along_fused
            Segment._attrs_along_fused = lambda self, names, incoming: None
            try:
                want = seg.get_beam_attrs_along_segment(names, incoming)
            finally:
                Segment._attrs_along_fused = orig
        for n, g, w in zip(names, got, want):
            assert g.shape == w.shape == (n_pos,), (n, g.shape, w.shape)
            scale = w.abs().max().clamp_min(1e-30)
            assert ((g - w).abs().max() / scale) < tol, (n, float((g - w).abs().max() / scale))


def test_beam_attrs_along_segment_with_active_bpms(ca):
    """Active BPMs let the beam pass: the fused path takes them as identities and leaves their readings (the means at their
    position minus the misalignment), like the walk element by element does."""
    from cheetah_amd.accelerator.segment import Segment

    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    els, bpms = [], []
    for i in range(10):
        bpm = ca.BPM(is_active=True, misalignment=t([1e-5 * i, -1e-5]), **kw)
        bpms.append(bpm)
        els += [ca.Quadrupole(t(0.2), k1=t(2.2 if i % 2 == 0 else -2.2), **kw), ca.Drift(t(0.8), **kw),
                ca.VerticalCorrector(t(0.05), angle=t(2e-5 * i), **kw), bpm]
    seg = ca.Segment(els)
    names = ("beta_x", "sigma_y", "mu_y", "s")
    for incoming in (ca.ParticleBeam.from_twiss(beta_x=t(3.14), beta_y=t(12.0), energy=t(1e8), num_particles=20_000, **kw),
                     ca.ParameterBeam.from_twiss(beta_x=t(3.14), beta_y=t(12.0), energy=t(1e8), **kw)):
        with torch.no_grad():
            assert seg._attrs_along_fused(names, incoming) is not None
            got = seg.get_beam_attrs_along_segment(names, incoming)
            got_r = torch.stack([b.reading.clone() for b in bpms])
            orig = Segment._attrs_along_fused
            Segment._attrs_along_fused = lambda self, names, incoming: None
            try:
                want = seg.get_beam_attrs_along_segment(names, incoming)
                want_r = torch.stack([b.reading.clone() for b in bpms])
            finally:
                Segment._attrs_along_fused = orig
        for n, g, w in zip(names, got, want):
            assert g.shape == w.shape == (41,)
            assert ((g - w).abs().max() / w.abs().max().clamp_min(1e-30)) < 2e-4, n
        assert got_r.shape == (10, 2) and torch.allclose(got_r, want_r, rtol=1e-4, atol=2e-9)
        assert (got_r[5] - got_r[1]).abs().max() > 1e-6
