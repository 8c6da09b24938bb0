"""Small physics / contract checks per element, mirroring the reference's tests/test_dipole.py, test_quadrupole.py,
test_drift.py, test_space_charge_kick.py, test_superimposed.py, test_species.py (the parts that need no Ocelot)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
KW = {"device": "cuda"}


def t(v, dtype=None):
    return torch.tensor(v, device="cuda", dtype=dtype)


def test_dipole_off_focussing_and_vectorised():
    import cheetah_amd as ca

    beam = ca.ParameterBeam.from_parameters(sigma_px=t(2e-7), sigma_py=t(2e-7), **KW)
    dipole, drift = ca.Dipole(length=t(1.0), angle=t(0.0), **KW), ca.Drift(length=t(1.0), **KW)
    off, ref = dipole.track(beam), drift.track(beam)
    dipole.angle = t(1.0)
    on = dipole.track(beam)
    assert torch.allclose(off.sigma_x, ref.sigma_x) and not torch.allclose(on.sigma_x, ref.sigma_x)
    dipole = ca.Dipole(length=t([1.0]), k1=t([10.0]), **KW)
    quad = ca.Quadrupole(length=t([1.0]), k1=t([10.0]), **KW)
    vbeam = ca.ParameterBeam.from_parameters(sigma_px=t([2e-7]), sigma_py=t([2e-7]), **KW)
    assert torch.allclose(dipole.track(vbeam).sigma_x, quad.track(vbeam).sigma_x)
    dipole.k1 = t([0.0])
    assert not torch.allclose(dipole.track(vbeam).sigma_x, quad.track(vbeam).sigma_x)
    incoming = ca.ParticleBeam.from_parameters(num_particles=100, energy=t(1e9), mu_x=t(1e-5), **KW)
    for cls in (ca.Dipole, ca.RBend):
        out = ca.Segment([cls(length=t([0.5, 0.5, 0.5]), angle=t([0.1, 0.2, 0.1]), **KW), ca.Drift(t(0.5), **KW)]).track(incoming)
        assert out.particles.shape == (3, 100, 7) and out.mu_x.shape == (3,)
        assert torch.allclose(out.particles[0], out.particles[2]) and not torch.allclose(out.particles[0], out.particles[1])
    seg = ca.Segment([ca.Dipole(length=t([0.5, 0.5, 0.5]).reshape(3, 1), angle=t([0.1, 0.2, 0.1]).reshape(1, 3), **KW),
                      ca.Drift(length=t([0.5, 1.0]).reshape(2, 1, 1), **KW)])
    assert seg.track(incoming).particles.shape == (2, 3, 3, 100, 7)
    bad = ca.Segment([ca.Dipole(length=t([0.5, 0.5, 0.5]).reshape(3, 1), angle=t([0.1, 0.2, 0.1]).reshape(1, 3), **KW),
                      ca.Drift(length=t([0.5, 1.0]).reshape(2, 1), **KW)])
    with pytest.raises(RuntimeError):
        bad.track(incoming)


@pytest.mark.parametrize("method", ["linear", "second_order", "drift_kick_drift"])
def test_dipole_tilt_sanity(method):
    """Tilted dipole == rotate beam, untilted dipole, rotate back (test_dipole.py:176-222)."""
    import cheetah_amd as ca

    f64 = torch.float64
    kw = {"device": "cuda", "dtype": f64}
    TILT = 0.52
    torch.manual_seed(0)
    beam = ca.ParticleBeam.from_parameters(num_particles=5_000, sigma_x=t(2e-4, f64), sigma_px=t(4e-6, f64), energy=t(1.07e8, f64), **kw)
    tilted = ca.Dipole(length=t(1.0601, f64), angle=t(1e-3, f64), tilt=t(TILT, f64), fringe_integral=t(1e3, f64),
                       tracking_method=method, **kw)
    out_a = tilted.track(beam).particles
    c, s = math.cos(TILT), math.sin(TILT)
    rot = torch.eye(7, **kw)
    rot[0, 0] = rot[1, 1] = rot[2, 2] = rot[3, 3] = c
    rot[0, 2] = rot[1, 3] = s
    rot[2, 0] = rot[3, 1] = -s
    plain = tilted.clone()
    plain.tilt = t(0.0, f64)
    rotated = ca.ParticleBeam(beam.particles @ rot.T, beam.energy, species=beam.species)
    out_b = plain.track(rotated).particles @ rot  # rotation_matrix(-TILT).T == rotation_matrix(TILT)
    assert torch.allclose(out_a, out_b, rtol=1e-9, atol=1e-14)


def test_quadrupole_off_misalignment_clone_and_precision():
    import cheetah_amd as ca

    beam = ca.ParameterBeam.from_parameters(sigma_px=t(2e-7), sigma_py=t(2e-7), **KW)
    quad, drift = ca.Quadrupole(length=t(1.0), k1=t(0.0), **KW), ca.Drift(length=t(1.0), **KW)
    assert torch.allclose(quad.track(beam).sigma_x, drift.track(beam).sigma_x)
    quad.k1 = t(1.0)
    assert not torch.allclose(quad.track(beam).sigma_x, drift.track(beam).sigma_x)
    mis = ca.Quadrupole(length=t(1.0), k1=t(1.0), misalignment=t([0.1, 0.1]).unsqueeze(0), **KW)
    assert not torch.allclose(mis.track(beam).mu_x, quad.track(beam).mu_x)
    for m in ("linear", "second_order", "drift_kick_drift"):
        q = ca.Quadrupole(length=t(1.0), k1=t(1.0), tracking_method=m, **KW)
        assert q.clone().tracking_method == m
        vq = ca.Quadrupole(length=t([[0.2, 0.25], [0.3, 0.35], [0.4, 0.45]]), k1=t([[4.2, 4.2], [4.3, 4.3], [4.4, 4.4]]),
                           misalignment=torch.zeros(2, device="cuda"), tilt=t(0.0), tracking_method=m, **KW)
        out = vq.track(ca.ParticleBeam.from_parameters(num_particles=2_000, sigma_x=t([[1e-5, 2e-5], [2e-5, 3e-5], [3e-5, 4e-5]]), **KW))
        assert out.mu_x.shape == (3, 2) and out.sigma_p.shape == (3, 2) and out.energy.shape == torch.Size([])
    for dtype in (torch.float32, torch.float64):
        kw = {"device": "cuda", "dtype": dtype}
        L, k1, tilt = t(0.5, dtype), t(0.0, dtype), t(math.pi / 4, dtype)
        energy, sp = t(1e9, dtype), ca.Species("electron", **kw)
        tm_d = ca.Drift(length=L, **kw).first_order_transfer_map(energy, sp)
        assert torch.allclose(tm_d, ca.Quadrupole(length=L, k1=k1, **kw).first_order_transfer_map(energy, sp), atol=2e-7)
        assert torch.allclose(tm_d, ca.Quadrupole(length=L, k1=k1, tilt=tilt, **kw).first_order_transfer_map(energy, sp), atol=2e-7)


def test_drift_divergence_inversion_and_length_parameter():
    import cheetah_amd as ca

    drift = ca.Drift(length=t(1.0), **KW)
    for beam in (ca.ParameterBeam.from_parameters(sigma_px=t(2e-7), sigma_py=t(2e-7), **KW),
                 ca.ParticleBeam.from_parameters(num_particles=10_000, sigma_px=t(2e-7), sigma_py=t(2e-7), **KW)):
        out = drift.track(beam)
        assert out.sigma_x > beam.sigma_x and out.sigma_y > beam.sigma_y
        assert torch.isclose(out.sigma_px, beam.sigma_px) and torch.isclose(out.sigma_py, beam.sigma_py)
    beam = ca.ParticleBeam.from_parameters(num_particles=10_000, **KW)
    back = ca.Segment([ca.Drift(t(1.3), **KW), ca.Drift(t(-1.3), **KW)]).track(beam)
    assert torch.allclose(back.particles, beam.particles, atol=1e-9)
    L = torch.nn.Parameter(t(1.0))
    out = ca.Drift(length=L, **KW).track(beam)
    assert out.particles.grad_fn is not None
    out.sigma_x.backward()
    assert L.grad is not None and float(L.grad) != 0.0


def test_space_charge_vectorised_expansion_length_and_aperture():
    import cheetah_amd as ca

    R0, energy = t(0.001), t(2.5e8)
    rest = t(510998.95069)
    gamma = energy / rest
    beta = (1 - gamma.square().reciprocal()).sqrt()
    torch.manual_seed(0)
    incoming = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=100_000, total_charge=t(1e-8).repeat(3, 2), energy=energy,
                                                    radius_x=R0, radius_y=R0, radius_tau=R0 / gamma / beta, sigma_px=t(1e-15),
                                                    sigma_py=t(1e-15), sigma_p=t(1e-15), **KW)
    kappa = 1 + (t(2.0).sqrt() / 4) * (3 + 2 * t(2.0).sqrt()).log()
    Nb = incoming.total_charge / 1.602176634e-19
    L = beta * gamma * kappa * (R0.pow(3) / (Nb * 2.8179403205e-15)).sqrt()
    seg = ca.Segment([ca.Drift(L / 6, **KW), ca.SpaceChargeKick(L / 3, **KW), ca.Drift(L / 3, **KW), ca.SpaceChargeKick(L / 3, **KW),
                      ca.Drift(L / 3, **KW), ca.SpaceChargeKick(L / 3, **KW), ca.Drift(L / 6, **KW)])
    out = seg.track(incoming)
    assert out.particles.shape == (3, 2, 100_000, 7)
    for n in ("sigma_x", "sigma_y", "sigma_tau"):
        assert torch.allclose(getattr(out, n), 2 * getattr(incoming, n), rtol=2e-2), n
    one = t(1.0)
    seg = ca.Segment([ca.Drift(one / 6, **KW), ca.SpaceChargeKick(one / 3, **KW), ca.Drift(one / 3, **KW), ca.SpaceChargeKick(one / 3, **KW),
                      ca.Drift(one / 3, **KW), ca.SpaceChargeKick(one / 3, **KW), ca.Drift(one / 6, **KW)])
    assert seg.length.shape == torch.Size([]) and torch.allclose(seg.length, one)
    beam = ca.ParticleBeam.from_parameters(num_particles=10_000, sigma_px=t(2e-7), sigma_py=t(2e-7), **KW)
    before = beam.particles.clone()
    seg.track(beam)
    assert torch.equal(before, beam.particles)
    seg = ca.Segment([ca.Drift(t(0.2), **KW), ca.Aperture(x_max=t(1e-4), y_max=t(1e-4), is_active=False, name="aperture", **KW),
                      ca.Drift(t(0.25), **KW), ca.SpaceChargeKick(t(0.5), **KW), ca.Drift(t(0.25), **KW)])
    beam = ca.ParticleBeam.from_parameters(num_particles=10_000, total_charge=t(1e-9), mu_x=t(5e-5), sigma_px=t(1e-4), sigma_py=t(1e-4), **KW)
    without = seg.track(beam)
    seg.aperture.is_active = True
    with_ap = seg.track(beam)
    assert not torch.allclose(with_ap.particles, without.particles)
    assert with_ap.survival_probabilities.sum(dim=-1).max() < 10_000


def test_superimposed_and_species():
    import cheetah_amd as ca

    base = ca.Quadrupole(length=t(0.4), k1=t(1.1), **KW)
    sup = ca.Superimposed(base, ca.Marker(name="centre", **KW), name="sup", **KW)
    assert torch.allclose(sup.length, t(0.4)) and [type(e).__name__ for e in sup.flattened().elements] == ["Quadrupole", "Marker", "Quadrupole"]
    energy, sp = t(1e8), ca.Species("electron", **KW)
    assert torch.allclose(sup.first_order_transfer_map(energy, sp), base.first_order_transfer_map(energy, sp), rtol=1e-5, atol=1e-7)
    with pytest.raises(AssertionError):
        ca.Superimposed(base, ca.Drift(t(0.1), **KW), **KW)
    for name, mass in (("electron", 510998.95069), ("positron", 510998.95069), ("proton", 938272089.43),
                       ("antiproton", 938272089.43), ("deuteron", 1875612945.0)):
        assert float(ca.Species(name, dtype=torch.float64).mass_eV) == pytest.approx(mass, rel=1e-9)
    custom = ca.Species("muon", num_elementary_charges=torch.tensor(-1.0), mass_eV=torch.tensor(105658375.5))
    assert float(custom.charge_coulomb) == pytest.approx(-1.602176634e-19)
    for bad in ({}, {"num_elementary_charges": torch.tensor(1.0)}, {"mass_eV": torch.tensor(1e6)},
                {"num_elementary_charges": torch.tensor(1.0), "charge_coulomb": torch.tensor(1.6e-19), "mass_eV": torch.tensor(1e6)}):
        with pytest.raises(AssertionError):
            ca.Species("custom", **bad)
