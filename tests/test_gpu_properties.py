"""Reference-independent invariants of the tracking path, restated from the reference's own property
tests and run against cheetah_amd on the GPU."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

f32, f64 = torch.float32, torch.float64


@pytest.fixture(scope="module")
def ca():
    import cheetah_amd

    return cheetah_amd


def t(v, dt=f64):
    return torch.tensor(v, dtype=dt, device="cuda")


def beam(ca, n=20_000, dt=f64, **kw):
    torch.manual_seed(42)
    return ca.ParticleBeam.from_parameters(num_particles=n, dtype=dt, device="cuda", **kw)


def test_split_equals_whole(ca):
    """tests/test_split.py:7-30: an element split into <= 1.5 cm slices tracks like the original."""
    b = beam(ca)
    for el in (ca.Quadrupole(t(0.2), k1=t(4.2), dtype=f64, device="cuda"), ca.Drift(t(0.37))):
        whole = el.track(b).particles
        parts = ca.Segment(el.split(t(0.015))).track(b).particles
        assert len(el.split(t(0.015))) >= 13
        assert torch.allclose(parts, whole, rtol=1e-9, atol=1e-14)


def test_quadrupole_without_strength_is_a_drift(ca):
    """tests/test_quadrupole.py:7-27, :268-292 (|R_quad(k1=0) - R_drift| <= 2e-7 in fp32)."""
    for dt, tol in ((f32, 2e-7), (f64, 1e-15)):
        q = ca.Quadrupole(t(1.0, dt), k1=t(0.0, dt), dtype=dt, device="cuda")
        d = ca.Drift(t(1.0, dt))
        e, sp = t(1e8, dt), ca.Species("electron", dtype=dt, device="cuda")
        assert (q.first_order_transfer_map(e, sp) - d.first_order_transfer_map(e, sp)).abs().max() <= tol
        b = beam(ca, dt=dt)
        assert torch.allclose(q.track(b).particles, d.track(b).particles, rtol=1e-6 if dt == f32 else 1e-13, atol=1e-12)


def test_dipole_limits(ca):
    """tests/test_dipole.py:8-45: angle = 0 is a drift; angle = 0 with k1 is a quadrupole."""
    b = beam(ca)
    dip = ca.Dipole(t(1.0), angle=t(0.0), dtype=f64, device="cuda")
    assert torch.allclose(dip.track(b).particles, ca.Drift(t(1.0)).track(b).particles, rtol=1e-13, atol=1e-18)
    dipq = ca.Dipole(t(0.5), angle=t(0.0), k1=t(2.5), dtype=f64, device="cuda")
    quad = ca.Quadrupole(t(0.5), k1=t(2.5), dtype=f64, device="cuda")
    assert torch.allclose(dipq.track(b).particles, quad.track(b).particles, rtol=1e-12, atol=1e-18)


def test_tilted_dipole_is_rotated_dipole(ca):
    """tests/test_dipole.py:176-220."""
    b = beam(ca)
    tilt = 0.37
    tilted = ca.Dipole(t(1.0), angle=t(0.2), tilt=t(tilt), dtype=f64, device="cuda").track(b).particles
    rot = torch.eye(7, dtype=f64, device="cuda")
    c, s = math.cos(tilt), math.sin(tilt)
    rot[0, 0], rot[0, 2], rot[1, 1], rot[1, 3] = c, s, c, s
    rot[2, 0], rot[2, 2], rot[3, 1], rot[3, 3] = -s, c, -s, c
    seg = ca.Segment([ca.CustomTransferMap(rot), ca.Dipole(t(1.0), angle=t(0.2), dtype=f64, device="cuda"),
                      ca.CustomTransferMap(rot.T.contiguous())])
    assert torch.allclose(seg.track(b).particles, tilted, rtol=1e-12, atol=1e-17)


def test_quadrupole_tilt_periodicity(ca):
    """tests/test_quadrupole.py:87-110: tilts pi/4 and 5 pi/4 give the same beam."""
    b = beam(ca)
    a = ca.Quadrupole(t(0.3), k1=t(3.0), tilt=t(math.pi / 4), dtype=f64, device="cuda").track(b).particles
    c = ca.Quadrupole(t(0.3), k1=t(3.0), tilt=t(5 * math.pi / 4), dtype=f64, device="cuda").track(b).particles
    assert torch.allclose(a, c, rtol=1e-12, atol=1e-17)


def test_negative_drift_inverts(ca):
    """tests/test_drift.py:95-115."""
    b = beam(ca)
    back = ca.Drift(t(-0.8)).track(ca.Drift(t(0.8)).track(b))
    assert torch.allclose(back.particles, b.particles, rtol=1e-12, atol=1e-18)
    assert float(back.s) == pytest.approx(0.0, abs=1e-15)


def test_merged_maps_equal_unmerged(ca):
    """tests/test_speed_optimizations.py:7-80 / test_merge.py."""
    b = beam(ca)
    kw = {"dtype": f64, "device": "cuda"}
    seg = ca.Segment([ca.Drift(t(0.3)), ca.Quadrupole(t(0.2), k1=t(4.2), **kw), ca.HorizontalCorrector(t(0.02), angle=t(1e-4), **kw),
                      ca.Cavity(t(1.0), voltage=t(1e6), phase=t(10.0), frequency=t(1.3e9), name="cav", **kw),
                      ca.Drift(t(0.4)), ca.Dipole(t(0.5), angle=t(0.1), **kw), ca.Marker(**kw), ca.Drift(t(0.1))])
    ref = seg.track(b)
    merged = seg.transfer_maps_merged(b)
    assert len(merged.elements) == 3  # [combined, cavity, combined]
    out = merged.track(b)
    assert torch.allclose(out.particles, ref.particles, rtol=1e-11, atol=1e-17)
    assert torch.allclose(out.energy, ref.energy) and float(out.s) == pytest.approx(float(ref.s))
    step = b
    for e in seg.elements:
        step = e.track(step)
    assert torch.allclose(step.particles, ref.particles, rtol=1e-11, atol=1e-17)


@pytest.mark.parametrize("dt", [f32, f64])
def test_vectorised_shapes_and_broadcasting(ca, dt):
    """tests/test_vectorized.py:145-183, :362-371."""
    b = beam(ca, n=1000, dt=dt)
    kw = {"dtype": dt, "device": "cuda"}
    q = ca.Quadrupole(t(0.2, dt), k1=t([[4.2], [-4.2], [0.0]], dt), tilt=t([0.0, 0.3], dt), **kw)
    out = q.track(b)
    assert out.particles.shape == (3, 2, 1000, 7) and out.particles.dtype == dt and out.particles.is_cuda
    assert out.particle_charges.shape == (1000,) and out.survival_probabilities.shape == (1000,)
    assert out.sigma_x.shape == (3, 2) and out.mu_x.dtype == dt
    # every (i, j) entry equals the scalar element
    for i, k1 in enumerate((4.2, -4.2, 0.0)):
        for j, tilt in enumerate((0.0, 0.3)):
            single = ca.Quadrupole(t(0.2, dt), k1=t(k1, dt), tilt=t(tilt, dt), **kw).track(b).particles
            assert torch.equal(out.particles[i, j], single)
    seg = ca.Segment([ca.Drift(t(0.5, dt)), q, ca.Drift(t([0.1, 0.2], dt))])
    o2 = seg.track(b)
    assert o2.particles.shape == (3, 2, 1000, 7) and o2.s.shape == (2,)
    # vectorised beam (2,) x element (3,1)
    vb = ca.ParticleBeam(b.particles.unsqueeze(0).repeat(2, 1, 1), b.energy, species=b.species)
    o3 = ca.Quadrupole(t(0.2, dt), k1=t([[1.0], [2.0], [3.0]], dt), **kw).track(vb)
    assert o3.particles.shape == (3, 2, 1000, 7)
    with pytest.raises(RuntimeError):
        ca.Quadrupole(t(0.2, dt), k1=t([1.0, 2.0, 3.0], dt), tilt=t([0.0, 0.3], dt), **kw).track(b)


def test_inputs_never_mutated_and_outputs_are_new(ca):
    b = beam(ca)
    before = b.particles.clone()
    seg = ca.Segment([ca.Drift(t(0.3)), ca.Screen(is_active=True, name="scr", dtype=f64, device="cuda"),
                      ca.SpaceChargeKick(t(0.1), grid_shape=(8, 8, 8), dtype=f64, device="cuda"), ca.Drift(t(0.3))])
    out = seg.track(b)
    assert torch.equal(b.particles, before)
    assert out.particles.data_ptr() != b.particles.data_ptr()
    img1 = seg.scr.reading
    assert seg.scr.reading is img1  # cached until the next track
    seg.track(b)
    assert seg.scr.reading is not img1 and torch.allclose(seg.scr.reading, img1)


def test_screen_modes_agree_at_bin_centres(ca):
    """tests/test_cloud_in_cell.py:30-211: CIC equals the histogram when every particle sits at a bin centre."""
    res, px = (32, 24), (1e-4, 2e-4)
    torch.manual_seed(1)
    n = 5000
    ix = torch.randint(0, res[0], (n,), device="cuda")
    iy = torch.randint(0, res[1], (n,), device="cuda")
    parts = torch.zeros(n, 7, dtype=f64, device="cuda")
    parts[:, 0] = (ix.double() + 0.5) * px[0] - res[0] * px[0] / 2
    parts[:, 2] = (iy.double() + 0.5) * px[1] - res[1] * px[1] / 2
    parts[:, 6] = 1
    b = ca.ParticleBeam(parts, t(1e8), species=ca.Species("electron", dtype=f64, device="cuda"))
    imgs = []
    for method in ("histogram", "cloud-in-cell"):
        scr = ca.Screen(resolution=res, pixel_size=t(px), method=method, is_active=True, dtype=f64, device="cuda")
        scr.track(b)
        imgs.append(scr.reading)
    assert imgs[0].shape == (res[1], res[0])
    assert torch.allclose(imgs[0], imgs[1], rtol=1e-9, atol=1e-30)
    assert float(imgs[0].sum()) == pytest.approx(n * 1.602176634e-19, rel=1e-12)


@pytest.mark.parametrize("energy", [2.5e8, 1e6], ids=["ultra-relativistic", "non-relativistic"])
@pytest.mark.parametrize("dt", [f32, f64])
def test_space_charge_cold_uniform_beam_doubles(ca, energy, dt):
    """tests/test_space_charge_kick.py:14-71 (physics known answer, cf. ImpactX expanding-beam test):
    a cold uniform bunch doubles in all three sizes after
    L = beta gamma kappa sqrt(R0^3 / (N_b r_e)), kappa = 1 + sqrt(2)/4 log(3 + 2 sqrt 2); rtol 2e-2."""
    torch.manual_seed(0)
    R0 = 0.001
    rest_energy = 510998.95069
    electron_radius = 2.8179403205e-15  # scipy physical_constants["classical electron radius"] (CODATA 2022)
    gamma = energy / rest_energy
    beta = math.sqrt(1 - 1 / gamma**2)
    kw = {"dtype": dt, "device": "cuda"}
    incoming = ca.ParticleBeam.uniform_3d_ellipsoid(
        num_particles=100_000, total_charge=t(1e-8, dt), energy=t(energy, dt), radius_x=t(R0, dt), radius_y=t(R0, dt),
        radius_tau=t(R0 / gamma / beta, dt), sigma_px=t(1e-15, dt), sigma_py=t(1e-15, dt), sigma_p=t(1e-15, dt), **kw)
    kappa = 1 + (math.sqrt(2) / 4) * math.log(3 + 2 * math.sqrt(2))
    Nb = 1e-8 / 1.602176634e-19
    L = beta * gamma * kappa * math.sqrt(R0**3 / (Nb * electron_radius))
    seg = ca.Segment([ca.Drift(t(L / 6, dt)), ca.SpaceChargeKick(t(L / 3, dt), **kw), ca.Drift(t(L / 3, dt)),
                      ca.SpaceChargeKick(t(L / 3, dt), **kw), ca.Drift(t(L / 3, dt)),
                      ca.SpaceChargeKick(t(L / 3, dt), **kw), ca.Drift(t(L / 6, dt))])
    out = seg.track(incoming)
    for n in ("sigma_x", "sigma_y", "sigma_tau"):
        assert float(getattr(out, n)) == pytest.approx(2 * float(getattr(incoming, n)), rel=2e-2), n


def test_pass_through_elements_return_independent_beams(ca):
    """marker.py:52-53, bpm.py:87, screen.py:239 return `incoming.clone()`: editing the outgoing beam in place must not
    reach the incoming beam (and vice versa), for single elements and for segments made only of pass-through elements; the
    Screen reading survives edits of both (/root/reference/tests/test_screen.py:198-240)."""
    for make in (lambda: ca.Marker(dtype=f64, device="cuda"), lambda: ca.BPM(is_active=True, dtype=f64, device="cuda"),
                 lambda: ca.BPM(dtype=f64, device="cuda"), lambda: ca.Screen(dtype=f64, device="cuda"),
                 lambda: ca.Screen(is_active=True, dtype=f64, device="cuda"),
                 lambda: ca.Aperture(is_active=False, dtype=f64, device="cuda"),
                 lambda: ca.Segment([ca.Marker(dtype=f64, device="cuda"), ca.BPM(dtype=f64, device="cuda")])):
        el = make()
        b = beam(ca)
        before = {k: getattr(b, k).clone() for k in ("particles", "energy", "particle_charges", "survival_probabilities")}
        out = el.track(b)
        assert out is not b and torch.equal(out.particles, b.particles)
        out.particles.mul_(0.7)
        out.energy.mul_(0.5)
        out.particle_charges.mul_(0.3)
        out.survival_probabilities.mul_(0.1)
        for k, v in before.items():
            if isinstance(el, ca.Segment) and k != "particles":
                continue   # a segment of skippable elements is one linear map: like the reference's _track_first_order
                           # (element.py:184-191) the new beam shares energy / charges / survival with the incoming one
            assert torch.equal(getattr(b, k), v), (type(el).__name__, k)
    # a segment that ends in pass-through elements after a real map: new coordinates, no extra copy needed, still independent
    b = beam(ca)
    seg = ca.Segment([ca.Drift(t(0.3)), ca.Marker(dtype=f64, device="cuda"), ca.Screen(is_active=True, name="scr", dtype=f64, device="cuda")])
    before = b.particles.clone()
    out = seg.track(b)
    read = seg.scr.get_read_beam().particles.clone()
    img = seg.scr.reading.clone()
    out.particles.mul_(2.0)
    b.particles.mul_(3.0)
    assert torch.equal(seg.scr.get_read_beam().particles, read) and torch.equal(seg.scr.reading, img)
    assert torch.equal(b.particles, 3.0 * before)
    # a centred screen hands out its snapshot (nothing to shift); the shortcut ends with the first edit of the misalignment,
    # in place or by assignment
    b = beam(ca)
    scr = ca.Screen(is_active=True, dtype=f64, device="cuda")
    scr.track(b)
    assert torch.equal(scr.get_read_beam().particles, b.particles)
    scr.misalignment.add_(torch.tensor([1e-3, -2e-3], dtype=f64, device="cuda"))
    scr.track(b)
    rb = scr.get_read_beam().particles
    assert torch.equal(rb[:, 0], b.particles[:, 0] - 1e-3) and torch.equal(rb[:, 2], b.particles[:, 2] + 2e-3)
    scr2 = ca.Screen(is_active=True, dtype=f64, device="cuda")
    scr2.misalignment = torch.tensor([5e-4, 0.0], dtype=f64, device="cuda")
    scr2.track(b)
    assert torch.equal(scr2.get_read_beam().particles[:, 0], b.particles[:, 0] - 5e-4)
    scr3 = ca.Screen(is_active=True, dtype=torch.float32, device="cuda").double()    # moved: general path, still no shift
    scr3.track(b)
    assert torch.equal(scr3.get_read_beam().particles, b.particles)
    # ParameterBeam through a marker
    pb = ca.ParameterBeam.from_parameters(dtype=f64, device="cuda")
    mu0 = pb.mu.clone()
    o = ca.Marker(dtype=f64, device="cuda").track(pb)
    o.mu.mul_(2.0)
    assert torch.equal(pb.mu, mu0)
