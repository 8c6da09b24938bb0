"""Vectorisation / broadcasting contract of the host layer, mirroring the reference's tests/test_vectorized.py
(shapes of outgoing particles, energy, charges, survival and moments for vector-valued element parameters, beam
energies and beam parameters) for every element type the package provides, ParticleBeam and ParameterBeam."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
KW = {"device": "cuda", "dtype": torch.float32}
N = 20_000


def t(v):
    return torch.tensor(v, **KW)


def _beam(ca, cls, **kwargs):
    extra = {"num_particles": N} if cls is ca.ParticleBeam else {}
    return cls.from_parameters(**extra, **kwargs, **KW)


def _check_moment_shapes(ca, out, shape, energy_shape=()):
    if isinstance(out, ca.ParticleBeam):
        assert out.particles.shape == (*shape, N, 7)
        assert out.particle_charges.shape == (N,)
    for n in ("mu_x", "mu_px", "mu_y", "mu_py", "sigma_x", "sigma_px", "sigma_y", "sigma_py", "sigma_tau", "sigma_p"):
        assert getattr(out, n).shape == shape, n
    assert out.energy.shape == torch.Size(energy_shape)
    assert out.total_charge.shape == torch.Size([])


def _elements_with_length(ca):
    L = t(1.0)
    els = [ca.Drift(L, tracking_method=m, **KW) for m in ("linear", "second_order", "drift_kick_drift")]
    els += [ca.Quadrupole(L, k1=t(1.3), tracking_method=m, **KW) for m in ("linear", "second_order", "drift_kick_drift")]
    els += [ca.Dipole(L, angle=t(0.2), tracking_method=m, **KW) for m in ("linear", "second_order", "drift_kick_drift")]
    els += [ca.RBend(L, angle=t(0.2), **KW), ca.HorizontalCorrector(L, angle=t(1e-4), **KW),
            ca.VerticalCorrector(L, angle=t(1e-4), **KW), ca.CombinedCorrector(L, horizontal_angle=t(1e-4), **KW),
            ca.Cavity(L, voltage=t(1e6), phase=t(10.0), frequency=t(1.3e9), **KW), ca.Cavity(L, **KW),
            ca.Solenoid(L, k=t(0.5), **KW), ca.Undulator(L, **KW), ca.Sextupole(L, k2=t(2.0), **KW),
            ca.Sextupole(L, k2=t(2.0), tracking_method="linear", **KW),
            ca.TransverseDeflectingCavity(L, voltage=t(1e6), frequency=t(1e9), **KW)]
    return els


def test_segment_length_shapes():
    import cheetah_amd as ca

    seg = ca.Segment([ca.Drift(t([0.6, 0.5]), **KW), ca.Quadrupole(t([0.2, 0.25]), k1=t([4.2, 4.2]), **KW),
                      ca.Drift(t([0.4, 0.3]), **KW)])
    assert seg.length.shape == (2,)
    seg = ca.Segment([ca.Drift(t([[0.6, 0.5], [0.4, 0.3], [0.4, 0.3]]), **KW),
                      ca.Quadrupole(t([[0.2, 0.25], [0.3, 0.35], [0.3, 0.35]]), k1=t([[4.2, 4.2], [4.3, 4.3], [4.3, 4.3]]), **KW),
                      ca.Drift(t([[0.4, 0.3], [0.2, 0.1], [0.2, 0.1]]), **KW)])
    assert seg.length.shape == (3, 2)


@pytest.mark.parametrize("beam_cls", ["ParticleBeam", "ParameterBeam"])
def test_track_segment_shapes(beam_cls):
    import cheetah_amd as ca

    cls = getattr(ca, beam_cls)
    seg = ca.Segment([ca.Drift(t([0.6, 0.5]), **KW), ca.Quadrupole(t([0.2, 0.25]), k1=t([4.2, 4.2]), **KW),
                      ca.Drift(t([0.4, 0.3]), **KW)])
    _check_moment_shapes(ca, seg.track(_beam(ca, cls, sigma_x=t([1e-5, 2e-5]))), (2,))
    quad = ca.Quadrupole(t([0.2, 0.25]), k1=t([4.2, 4.2]), **KW)
    _check_moment_shapes(ca, quad.track(_beam(ca, cls, sigma_x=t([1e-5, 2e-5]))), (2,))
    seg = ca.Segment([ca.Drift(t([[0.6, 0.5], [0.4, 0.3], [0.2, 0.1]]), **KW),
                      ca.Quadrupole(t([[0.2, 0.25], [0.3, 0.35], [0.4, 0.45]]), k1=t([[4.2, 4.2], [4.3, 4.3], [4.4, 4.4]]), **KW),
                      ca.Drift(t([[0.4, 0.3], [0.6, 0.5], [0.8, 0.7]]), **KW)])
    out = seg.track(_beam(ca, cls, sigma_x=t([[1e-5, 2e-5], [2e-5, 3e-5], [3e-5, 4e-5]])))
    _check_moment_shapes(ca, out, (3, 2))
    assert seg.track_moments(_beam(ca, ca.ParticleBeam, sigma_x=t([[1e-5, 2e-5], [2e-5, 3e-5], [3e-5, 4e-5]]))).mu.shape == (3, 2, 7)


def test_enormous_scan_parameter_beam():
    import cheetah_amd as ca

    seg = ca.Segment([ca.Drift(t(0.175), **KW), ca.Quadrupole(t(0.122), k1=torch.linspace(-30.0, 30.0, 200_000, **KW).repeat(3, 1), **KW),
                      ca.Drift(t(0.428), **KW), ca.Quadrupole(t(0.122), k1=t(-14.3), **KW), ca.Drift(t(0.45), **KW)])
    out = seg.track(ca.ParameterBeam.from_parameters(**KW))
    assert out.mu_x.shape == (3, 200_000) and out.sigma_y.shape == (3, 200_000) and out.energy.shape == torch.Size([])
    assert torch.isfinite(out.cov).all()


@pytest.mark.parametrize("beam_cls", ["ParticleBeam", "ParameterBeam"])
@pytest.mark.parametrize("cavity_type", ["standing_wave", "traveling_wave"])
def test_cavity_with_zero_and_non_zero_voltage(beam_cls, cavity_type):
    import cheetah_amd as ca

    cav = ca.Cavity(t(3.0441), voltage=t([0.0, 48198468.0, 0.0]), phase=t(48198468.0), frequency=t(2.8560e09),
                    cavity_type=cavity_type, **KW)
    out = cav.track(_beam(ca, getattr(ca, beam_cls), sigma_x=t(1e-5)))
    _check_moment_shapes(ca, out, (3,), energy_shape=(3,))
    assert torch.isfinite(out.sigma_x).all()


@pytest.mark.parametrize("beam_cls", ["ParticleBeam", "ParameterBeam"])
def test_vectorized_undulator_and_solenoid(beam_cls):
    import cheetah_amd as ca

    cls = getattr(ca, beam_cls)
    _check_moment_shapes(ca, ca.Undulator(t([0.4, 0.7]), **KW).track(_beam(ca, cls, sigma_x=t(1e-5))), (2,))
    _check_moment_shapes(ca, ca.Solenoid(t([0.4, 0.7]), k=t([4.2, 3.1]), **KW).track(_beam(ca, cls, sigma_x=t(1e-5))), (2,))


def test_broadcasting_two_different_inputs_for_every_element_with_length():
    """tests/test_vectorized.py:340-399: length (3,1) against beam energies (2,) -> (3,2,N,7); charges and energy keep
    their own shapes."""
    import cheetah_amd as ca

    for el in _elements_with_length(ca):
        incoming = _beam(ca, ca.ParticleBeam, energy=t([154e6, 14e9]))
        el.length = t([[0.6], [0.5], [0.4]])
        out = el.track(incoming)
        label = f"{type(el).__name__}/{el.tracking_method}"
        assert out.particles.shape == (3, 2, N, 7), label
        assert out.particle_charges.shape == (N,), label
        assert out.energy.shape == (2,), label
        assert torch.isfinite(out.particles).all(), label


def test_broadcasting_corrector_angles_and_solenoid_misalignment():
    import cheetah_amd as ca

    for cls in (ca.HorizontalCorrector, ca.VerticalCorrector):
        out = cls(t(0.15), angle=t([[1e-5], [2e-5], [3e-5]]), **KW).track(_beam(ca, ca.ParticleBeam, energy=t([154e6, 14e9])))
        assert out.particles.shape == (3, 2, N, 7) and out.particle_charges.shape == (N,) and out.energy.shape == (2,)
    sol = ca.Solenoid(t(0.15), misalignment=t([[[1e-5, 2e-5], [2e-5, 3e-5]], [[3e-5, 4e-5], [4e-5, 5e-5]],
                                                [[5e-5, 6e-5], [6e-5, 7e-5]]]), **KW)
    out = sol.track(_beam(ca, ca.ParticleBeam, energy=t([154e6, 14e9])))
    assert out.particles.shape == (3, 2, N, 7) and out.energy.shape == (2,)


def test_vectorized_parameter_beam_creation():
    import cheetah_amd as ca

    beam = ca.ParameterBeam.from_parameters(mu_x=t([2e-4, 3e-4]), sigma_x=t([1e-5, 2e-5]), **KW)
    assert beam.mu_x.shape == (2,) and torch.allclose(beam.mu_x, t([2e-4, 3e-4]))
    assert beam.sigma_x.shape == (2,) and torch.allclose(beam.sigma_x, t([1e-5, 2e-5]))


@pytest.mark.parametrize("shape", ["rectangular", "elliptical"])
def test_vectorized_aperture_broadcasting(shape):
    """tests/test_vectorized.py:461-515 incl. its survival fractions (elliptical 0.0235 / 0.42 / 0.552)."""
    import cheetah_amd as ca

    torch.manual_seed(0)
    incoming = ca.ParticleBeam.from_parameters(num_particles=100_000, sigma_py=t(1e-4), sigma_px=t(2e-4),
                                               energy=t([154e6, 14e9]), **KW)
    seg = ca.Segment([ca.Drift(t(0.5), **KW), ca.Aperture(x_max=t([[1e-5], [2e-4], [3e-4]]), y_max=t(2e-4), shape=shape, **KW),
                      ca.Drift(t(0.5), **KW)])
    out = seg.track(incoming)
    assert out.particles.shape == (2, 100_000, 7) and out.energy.shape == (2,)
    assert out.particle_charges.shape == (100_000,) and out.survival_probabilities.shape == (3, 2, 100_000)
    frac = out.survival_probabilities.mean(dim=-1)[:, 0].cpu().numpy()
    if shape == "elliptical":
        assert np.allclose(frac, [0.0235, 0.42, 0.552], atol=7e-3)
    else:
        assert frac[0] < frac[1] < frac[2] < 1.0


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("B,N", [(16, 2048), (16, 2049), (3, 1001), (37, 70_003), (9, 1024 * 5 + 64), (1025, 4), (64, 1)])
def test_shared_beam_apply_every_alignment(oracle, tag, B, N):
    """One beam, B maps (`chx_apply_affine7` with Bx = 1): the wave-staged kernel when every output row starts on a 16-byte
    boundary (N * 7 * sizeof % 16 == 0), the workgroup-staged one otherwise; ragged last tiles, fewer rows than a chunk,
    tiles with idle waves. Bit for bit against the oracle's fma chain."""
    import numpy as np
    import torch

    from cheetah_amd import _ops

    dt = torch.float32 if tag == "f32" else torch.float64
    rng = np.random.default_rng(B * 1000 + N)
    x = rng.normal(size=(N, 7)).astype(np.float32 if tag == "f32" else np.float64) * 1e-3
    x[:, 6] = 1.0
    R = np.tile(np.eye(7), (B, 1, 1)) + 0.1 * rng.normal(size=(B, 7, 7))
    R[:, 6] = 0.0
    R[:, 6, 6] = 1.0
    R = R.astype(x.dtype)
    got = _ops.apply_map(torch.tensor(x, device="cuda"), torch.tensor(R, device="cuda"))
    assert got.shape == (B, N, 7) and got.dtype == dt
    want = oracle.apply(x[None], R, mode=1)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("B,N", [(1, 4_000_001), (3, 1_500_004), (3, 1_500_001), (2, 2_000_000)])
def test_streaming_size_apply_paths(oracle, B, N, tag):
    """Launches of more than 96 MB (they stream from HBM) take the barrier-free wave-staged kernel when the batch rows are
    16-byte aligned — including a ragged last wave and the few elements behind the last whole 16-byte chunk of a single
    row — and 256-row workgroup tiles otherwise. Per-row beams and maps, bit for bit against the oracle."""
    import numpy as np
    import torch

    from cheetah_amd import _ops

    rng = np.random.default_rng(N)
    ndt = np.float32 if tag == "f32" else np.float64
    x = (rng.normal(size=(B, N, 7)) * 1e-3).astype(ndt)
    x[..., 6] = 1.0
    R = (np.tile(np.eye(7), (B, 1, 1)) + 0.1 * rng.normal(size=(B, 7, 7))).astype(ndt)
    R[:, 6] = 0.0
    R[:, 6, 6] = 1.0
    got = _ops.apply_map(torch.tensor(x, device="cuda"), torch.tensor(R, device="cuda")).cpu().numpy()
    want = oracle.apply(x, R, mode=1)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_random_vectorised_lattices_vs_reference(tag):
    """Eight drawn beamlines whose settings carry vector dimensions of shapes (3,), (2, 1), (2, 3) (tests/golden/
    generate_golden_random_vectorized.py), a shared or a vectorised incoming beam: the SHAPES of the outgoing particles,
    energy and survival probabilities are the reference's, and so are the values."""
    import json
    import os

    import numpy as np
    import torch

    import cheetah_amd as ca

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lattices_random_vectorized.npz"))
    dt = torch.float64 if tag == "f64" else torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    for i in range(int(g["n_lattices"])):
        spec = json.loads(str(g[f"spec_{i}"]))
        elements = []
        for kind, args in spec:
            targs = {k: (torch.tensor(v, **kw) if isinstance(v, (float, list)) else v) for k, v in args.items()}
            elements.append(getattr(ca, kind)(**targs, **kw))
        beam = ca.ParticleBeam(torch.tensor(g[f"in_{i}"], **kw), torch.tensor(g[f"energy_{i}"], **kw),
                               particle_charges=torch.tensor(g[f"charges_{i}"], **kw), species=ca.Species("electron", **kw))
        out = ca.Segment(elements).track(beam)
        ref = g[f"out_{i}"]
        names = [k for k, _ in spec]
        assert tuple(out.particles.shape) == ref.shape, (i, names)
        assert tuple(out.energy.shape) == g[f"energy_out_{i}"].shape, (i, names)
        assert tuple(out.survival_probabilities.shape) == g[f"survival_{i}"].shape, (i, names)
        got = out.particles.cpu().numpy().astype(np.float64)
        scale = np.maximum(np.abs(ref).reshape(-1, 7).max(axis=0), 1e-30)
        assert (np.abs(got - ref) / scale).max() < (1e-10 if tag == "f64" else 3e-4), (i, names)
        assert np.allclose(out.energy.cpu().numpy(), g[f"energy_out_{i}"], rtol=1e-12 if tag == "f64" else 1e-6)
        surv = out.survival_probabilities.cpu().numpy()
        assert np.sum(surv != g[f"survival_{i}"]) <= (0 if tag == "f64" else 2), (i, names)
