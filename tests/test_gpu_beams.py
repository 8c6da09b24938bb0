"""Beam-class contract mirroring the reference's tests/test_particle_beam.py / test_parameter_beam.py: factories reproduce
the requested parameters (all moments through chx_moments), Twiss round trip, indexing of vectorised beams, ellipsoid
factory, ParticleBeam <-> ParameterBeam conversion."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
KW = {"device": "cuda"}


def t(v, dtype=None):
    return torch.tensor(v, device="cuda", dtype=dtype)


def test_create_from_parameters_has_those_parameters():
    import cheetah_amd as ca

    torch.manual_seed(0)
    want = dict(mu_x=1e-5, mu_px=1e-7, mu_y=2e-5, mu_py=2e-7, sigma_x=1.75e-7, sigma_px=2e-7, sigma_y=1.75e-7, sigma_py=2e-7,
                sigma_tau=1e-6, sigma_p=1e-6, cov_xpx=1e-15, cov_xy=1e-15, cov_xpy=-1.1e-15, cov_xtau=1.2e-15, cov_xp=1e-15,
                cov_pxy=-1.3e-15, cov_pxpy=1.4e-15, cov_pxtau=-1.5e-15, cov_pxp=1e-15, cov_ypy=1e-15, cov_ytau=1.6e-15,
                cov_yp=1e-15, cov_pytau=-1.7e-15, cov_pyp=1e-15, cov_taup=1e-15)
    beam = ca.ParticleBeam.from_parameters(num_particles=1_000_000, energy=t(1e7), total_charge=t(1e-9),
                                           **{k: t(v) for k, v in want.items()}, **KW)
    assert beam.num_particles == 1_000_000
    for k, v in want.items():
        assert np.isclose(getattr(beam, k).cpu().numpy(), v), k
    assert np.isclose(beam.energy.cpu().numpy(), 1e7) and np.isclose(beam.total_charge.cpu().numpy(), 1e-9)
    pb = ca.ParameterBeam.from_parameters(energy=t(1e7), total_charge=t(1e-9), **{k: t(v) for k, v in want.items()}, **KW)
    for k, v in want.items():
        assert np.isclose(getattr(pb, k).cpu().numpy(), v), k


@pytest.mark.parametrize("cls_name", ["ParticleBeam", "ParameterBeam"])
def test_from_twiss_to_twiss(cls_name):
    import cheetah_amd as ca

    f64 = torch.float64
    torch.manual_seed(0)
    extra = {"num_particles": 5_000_000} if cls_name == "ParticleBeam" else {}
    beam = getattr(ca, cls_name).from_twiss(
        beta_x=t(5.91253676811640894, f64), alpha_x=t(3.55631307633660354, f64), emittance_x=t(3.494768647122823e-09, f64),
        beta_y=t(5.91253676811640982, f64), alpha_y=t(1.0, f64), emittance_y=t(3.497810737006068e-09, f64),
        dispersion_x=t(2e-2, f64), sigma_p=t(1e-3, f64), energy=t(6e6, f64), dtype=f64, device="cuda", **extra)
    assert np.isclose(beam.beta_x.cpu().numpy(), 5.91253676811640894)
    assert np.isclose(beam.alpha_x.cpu().numpy(), 3.55631307633660354)
    assert np.isclose(beam.emittance_x.cpu().numpy(), 3.494768647122823e-09)
    assert np.isclose(beam.beta_y.cpu().numpy(), 5.91253676811640982)
    assert np.isclose(beam.alpha_y.cpu().numpy(), 1.0)
    assert np.isclose(beam.emittance_y.cpu().numpy(), 3.497810737006068e-09)
    assert np.isclose(beam.sigma_p.cpu().numpy(), 1e-3)
    assert np.isclose(beam.dispersion_x.cpu().numpy(), 2e-2, rtol=1e-3)
    assert np.isclose(beam.energy.cpu().numpy(), 6e6)


def test_uniform_ellipsoid_dtype_device_and_vectorisation():
    import cheetah_amd as ca

    beam = ca.ParticleBeam.uniform_3d_ellipsoid(**KW)
    feats = [f for f in beam.defining_features if f != "species"]
    assert set(feats) >= {"particles", "energy", "particle_charges", "survival_probabilities", "s"}
    for f in feats:
        assert getattr(beam, f).dtype == torch.float32 and getattr(beam, f).is_cuda, f
    dbl = ca.ParticleBeam.uniform_3d_ellipsoid(dtype=torch.float64, **KW)
    for f in feats:
        assert getattr(dbl, f).dtype == torch.float64, f
    rx, ry, rt = t([1e-3, 2e-3]), t([1e-4, 2e-4]), t([1e-5, 2e-5])
    spx, spy, sp = t([2e-7, 1e-7]), t([3e-7, 2e-7]), t([1e-6, 2e-6])
    energy, charge = t([1e7, 2e7]), t([1e-9, 3e-9])
    torch.manual_seed(0)
    beam = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=1_000_000, radius_x=rx, radius_y=ry, radius_tau=rt, sigma_px=spx,
                                                sigma_py=spy, sigma_p=sp, energy=energy, total_charge=charge, **KW)
    assert beam.num_particles == 1_000_000
    assert torch.all(beam.x.abs().transpose(0, 1) <= rx) and torch.all(beam.y.abs().transpose(0, 1) <= ry)
    assert torch.all(beam.tau.abs().transpose(0, 1) <= rt)
    assert torch.allclose(beam.sigma_px, spx) and torch.allclose(beam.sigma_py, spy) and torch.allclose(beam.sigma_p, sp)
    assert torch.allclose(beam.energy, energy) and torch.allclose(beam.total_charge, charge)
    assert ca.ParticleBeam.from_parameters(num_particles=10_000, mu_x=t(1e-5), sigma_x=t([1.75e-7, 2.75e-7]), **KW).particles.shape == (2, 10_000, 7)


def test_indexing_of_vectorised_beams():
    import cheetah_amd as ca

    torch.manual_seed(1)
    quad = ca.Quadrupole(length=t(0.2).unsqueeze(0), k1=torch.rand((5, 2), device="cuda"), **KW)
    out = quad.track(ca.ParticleBeam.from_parameters(num_particles=1_000, sigma_x=t(1e-5), **KW))
    sub = out[:3]
    assert sub.particles.shape == (3, 2, 1_000, 7) and sub.energy.shape == (3, 2)
    assert sub.particle_charges.shape == (3, 2, 1_000) and sub.survival_probabilities.shape == (3, 2, 1_000)
    assert torch.all(sub.particles == out.particles[:3]) and torch.all(sub.energy == out.energy)
    assert torch.all(sub.particle_charges == out.particle_charges)
    quad = ca.Quadrupole(length=t(0.2), k1=t(0.1), **KW)
    out = quad.track(ca.ParticleBeam.from_parameters(num_particles=1_000, sigma_x=t(1e-5),
                                                     energy=torch.rand((5, 2), device="cuda") * 154e6 + 1e6, **KW))
    sub = out[:3]
    assert sub.particles.shape == (3, 2, 1_000, 7) and sub.energy.shape == (3, 2)
    assert torch.allclose(sub.particles, out.particles[:3]) and torch.allclose(sub.energy, out.energy[:3])
    with pytest.raises(RuntimeError):
        _ = ca.ParticleBeam.from_parameters(sigma_x=torch.rand((5, 2), device="cuda"),
                                            energy=torch.rand((4, 2), device="cuda") * 154e6 + 1e6, **KW)[:3]
    with pytest.raises(IndexError):
        _ = ca.ParticleBeam.from_parameters(num_particles=100, energy=torch.rand((5, 2), device="cuda") * 154e6 + 1e6, **KW)[6]


def test_vectorised_conversion_to_parameter_beam_and_back():
    import cheetah_amd as ca

    f64 = torch.float64
    torch.manual_seed(2)
    orig = ca.ParticleBeam.from_parameters(num_particles=10_000, mu_x=t((2e-4, 3e-4), f64), sigma_x=t((2e-5, 3e-5), f64),
                                           energy=t((1e7, 2e7), f64), dtype=f64, device="cuda")
    orig.survival_probabilities = orig.survival_probabilities.repeat(3, 1, 1)
    orig.survival_probabilities[0, 0, : 10_000 // 3] = 0.3
    orig.survival_probabilities[1, 0, : 10_000 // 3] = 0.6
    back = orig.as_parameter_beam().as_particle_beam(num_particles=4_000_000)
    assert isinstance(back, ca.ParticleBeam)
    for n in ("mu_x", "mu_px", "mu_y", "mu_py", "mu_tau", "mu_p", "sigma_x", "sigma_px", "sigma_y", "sigma_py", "sigma_tau",
              "sigma_p", "energy", "total_charge"):
        assert torch.allclose(getattr(orig, n), getattr(back, n), rtol=2e-3, atol=1e-9), n
    pb = ca.ParameterBeam.from_twiss(beta_x=t(3.14, f64), beta_y=t(42.0, f64), energy=t(1e8, f64), total_charge=t(1e-10, f64),
                                     dtype=f64, device="cuda")
    rec = pb.as_particle_beam(num_particles=4_000_000).as_parameter_beam()
    assert torch.allclose(pb.cov, rec.cov, rtol=5e-3, atol=1e-22) and torch.allclose(pb.mu, rec.mu, atol=1e-8)
    assert torch.isclose(pb.energy, rec.energy) and torch.isclose(pb.total_charge, rec.total_charge)
