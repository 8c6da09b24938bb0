"""ParameterBeam path against the reference: its own consistency goldens
(tests/test_elements.py:356-431, torch.allclose defaults) re-run at 1e-12, active cavities, README segment with a
screen reading, and a vectorised k1 scan."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
f64 = torch.float64


def dev(a):
    a = np.asarray(a)
    return torch.as_tensor(np.ascontiguousarray(a)).reshape(a.shape).cuda()


def t(v):
    return torch.tensor(v, dtype=torch.float32).to(f64).cuda()  # reference tests: fp32 tensors, then .to(float64)


def t64(v):
    return torch.tensor(v, dtype=f64, device="cuda")


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def ca():
    import cheetah_amd

    return cheetah_amd


def incoming(ca, g):
    mass, nq = g["species"]
    sp = ca.Species("custom", num_elementary_charges=t64(nq), mass_eV=t64(mass))
    return ca.ParameterBeam(dev(g["mu"]), dev(g["cov"]), dev(g["energy"]), total_charge=dev(g["total_charge"]), species=sp)


def test_reference_parameter_beam_consistency(ca, golden):
    g = golden("parameter_beam.npz")
    kw = {"dtype": f64, "device": "cuda"}
    elements = {
        "Drift_ParameterBeam_linear": ca.Drift(length=t([1.0, -1.0])),
        "Quadrupole_ParameterBeam_linear": ca.Quadrupole(length=t(1.0), k1=t([1.0, -2.0]), tilt=t(0.42), misalignment=t([0.01, -0.02])),
        "Dipole_ParameterBeam_linear": ca.Dipole(length=t(1.0), angle=t([1.0, -2.0]), tilt=t(0.42), **kw),
        "RBend_ParameterBeam_linear": ca.RBend(length=t(1.0), angle=t([1.0, -2.0]), tilt=t(0.42), **kw),
        "HorizontalCorrector_ParameterBeam_default": ca.HorizontalCorrector(length=t(1.0), angle=t([1.0, -2.0])),
        "VerticalCorrector_ParameterBeam_default": ca.VerticalCorrector(length=t(1.0), angle=t([1.0, -2.0])),
        "CombinedCorrector_ParameterBeam_default": ca.CombinedCorrector(length=t(1.0), horizontal_angle=t([1.0, -2.0]), vertical_angle=t([1.0, -2.0])),
        "Cavity_ParameterBeam_default": ca.Cavity(length=t(1.0), **kw),
        "CustomTransferMap_ParameterBeam_identity": ca.CustomTransferMap(torch.eye(7, **kw)),
        "Marker_ParameterBeam_default": ca.Marker(**kw),
        "Screen_ParameterBeam_default": ca.Screen(**kw),
        "Segment_ParameterBeam_default": ca.Segment([ca.Drift(length=t(1.0))]),
    }
    for name, el in elements.items():
        out = el.track(incoming(ca, g))
        emu, ecov = g[f"{name}__mu"], g[f"{name}__cov"]
        assert tuple(out.mu.shape) == emu.shape and tuple(out.cov.shape) == ecov.shape, name
        assert np.allclose(out.mu.cpu().numpy(), emu) and np.allclose(out.cov.cpu().numpy(), ecov), name
        assert rel(out.mu.cpu().numpy(), emu) < 1e-12 and rel(out.cov.cpu().numpy(), ecov) < 1e-11, name
        assert np.allclose(out.energy.cpu().numpy(), g[f"{name}__energy"]) and np.allclose(out.s.cpu().numpy(), g[f"{name}__s"])


def test_active_cavity_parameter_beam(ca, golden):
    g = golden("parameter_beam.npz")
    kw = {"dtype": f64, "device": "cuda"}
    beam = ca.ParameterBeam(dev(g["b_mu"]), dev(g["b_cov"]), t64(6e6), total_charge=t64(1e-10))
    for i, (ctype, V) in enumerate((("standing_wave", 18.15975e6), ("traveling_wave", 18.15975e6),
                                    ("standing_wave", [18.15975e6, -1.0e6, 5e6]))):
        cav = ca.Cavity(t64(1.0377), voltage=t64(V), phase=t64(30.0), frequency=t64(1.3e9), cavity_type=ctype, **kw)
        out = cav.track(beam)
        assert rel(out.mu.cpu().numpy(), g[f"cav{i}_mu"]) < 1e-12
        assert rel(out.cov.cpu().numpy(), g[f"cav{i}_cov"]) < 1e-11
        assert np.allclose(out.energy.cpu().numpy(), g[f"cav{i}_energy"], rtol=1e-14)


def test_readme_segment_screen_and_scan(ca, golden):
    g = golden("parameter_beam.npz")
    kw = {"dtype": f64, "device": "cuda"}
    seg = ca.Segment([
        ca.Drift(t64(0.175)), ca.Quadrupole(t64(0.122), k1=t64(8.2), **kw), ca.Drift(t64(0.428)),
        ca.Quadrupole(t64(0.122), k1=t64(-14.3), **kw), ca.Drift(t64(0.204)), ca.VerticalCorrector(t64(0.02), angle=t64(9e-5), **kw),
        ca.Drift(t64(0.204)), ca.Quadrupole(t64(0.122), k1=t64(3.142), **kw), ca.Drift(t64(0.179)),
        ca.HorizontalCorrector(t64(0.02), angle=t64(-1e-4), **kw), ca.Drift(t64(0.45)),
        ca.Screen(resolution=(200, 160), pixel_size=t64([2e-7, 3e-7]), misalignment=t64([-4.4e-5, 9.5e-5]), is_active=True, name="scr", **kw),
    ])
    beam = ca.ParameterBeam(dev(g["seg_in_mu"]), dev(g["seg_in_cov"]), t64(1e8))
    out = seg.track(beam)
    assert rel(out.mu.cpu().numpy(), g["seg_out_mu"]) < 1e-12 and rel(out.cov.cpu().numpy(), g["seg_out_cov"]) < 1e-11
    assert float(out.s) == pytest.approx(float(g["seg_out_s"]))
    for n in ("sigma_x", "sigma_y", "beta_x", "alpha_x", "emittance_x", "emittance_y"):
        assert float(getattr(out, n)) == pytest.approx(float(g["seg_" + n]), rel=1e-8), n
    img = seg.scr.reading.cpu().numpy()
    ref = g["seg_reading"]
    assert img.shape == ref.shape
    assert ref.max() > 0 and np.allclose(img, ref, rtol=2e-4, atol=1e-6 * ref.max())  # reference grid has fp32 jitter
    # vectorised scan on the ARES EA subcell
    k1 = dev(g["scan_k1"])
    seg2 = ca.Segment([
        ca.Marker(**kw), ca.Drift(t64(0.17504)), ca.Quadrupole(t64(0.122), k1=k1, **kw), ca.Drift(t64(0.428)),
        ca.Quadrupole(t64(0.122), k1=t64(-14.3), **kw), ca.Drift(t64(0.204)), ca.VerticalCorrector(t64(0.02), angle=t64(9e-5), **kw),
        ca.Drift(t64(0.204)), ca.Quadrupole(t64(0.122), k1=t64(3.142), **kw), ca.Drift(t64(0.179)),
        ca.HorizontalCorrector(t64(0.02), angle=t64(-1e-4), **kw), ca.Drift(t64(0.45)), ca.Screen(**kw)])
    out2 = seg2.track(beam)
    assert rel(out2.mu.cpu().numpy(), g["scan_mu"]) < 1e-12 and rel(out2.cov.cpu().numpy(), g["scan_cov"]) < 1e-11


def test_beam_types_agree(ca):
    """tests/test_compare_beam_type.py: ParameterBeam tracking equals the moments of the tracked ParticleBeam."""
    torch.manual_seed(0)
    kw = {"dtype": f64, "device": "cuda"}
    pb = ca.ParameterBeam.from_twiss(beta_x=t64(5.9), alpha_x=t64(3.5), emittance_x=t64(3.5e-9), beta_y=t64(5.9),
                                     alpha_y=t64(2e-7), emittance_y=t64(3.5e-9), energy=t64(6e6), **kw)
    particles = pb.as_particle_beam(200_000)
    for n in ("sigma_x", "sigma_px", "sigma_y", "sigma_py", "sigma_tau", "sigma_p", "cov_xpx"):
        assert float(getattr(particles, n)) == pytest.approx(float(getattr(pb, n)), rel=1e-6), n
    seg = ca.Segment([ca.Drift(t64(0.5)), ca.Quadrupole(t64(0.2), k1=t64(5.0), **kw), ca.Drift(t64(1.0)),
                      ca.Dipole(t64(0.3), angle=t64(0.05), **kw), ca.Drift(t64(0.2))])
    a, b = seg.track(pb), seg.track(particles)
    for n in ("mu_x", "mu_y", "sigma_x", "sigma_px", "sigma_y", "sigma_py", "sigma_tau", "sigma_p"):
        assert float(getattr(b, n)) == pytest.approx(float(getattr(a, n)), rel=1e-6, abs=1e-12), n


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_parameter_beam_through_random_lattices_vs_reference(golden, tag):
    """The drawn beamlines of lattices_random.npz with a ParameterBeam: mu, cov and energy behind each against the reference
    (cavities on: the cavity's moment update; apertures: transparent for a ParameterBeam)."""
    import json

    import numpy as np
    import torch

    import cheetah_amd as ca

    g = golden("lattices_random.npz")
    dt = torch.float64 if tag == "f64" else torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    for i in range(int(g["n_lattices"])):
        spec = json.loads(str(g[f"spec_{i}"]))
        elements = []
        for kind, args in spec:
            targs = {k: (torch.tensor(v, **kw) if isinstance(v, (float, list)) else v) for k, v in args.items()}
            elements.append(getattr(ca, kind)(**targs, **kw))
        beam = ca.ParameterBeam(torch.tensor(g[f"pmu_in_{i}"], **kw), torch.tensor(g[f"pcov_in_{i}"], **kw),
                                torch.tensor(float(g[f"energy_{i}"]), **kw), species=ca.Species("electron", **kw))
        out = ca.Segment(elements).track(beam)
        mu, cov = out.mu.cpu().numpy().astype(np.float64), out.cov.cpu().numpy().astype(np.float64)
        rmu, rcov = g[f"pmu_out_{i}"], g[f"pcov_out_{i}"]
        sig = np.sqrt(np.abs(np.diag(rcov)[:6]))
        tol = 1e-9 if tag == "f64" else 2e-3
        assert np.all(np.abs(mu[:6] - rmu[:6]) <= tol * (sig + np.abs(rmu[:6])) + 1e-30), (i, [k for k, _ in spec])
        # entries of the tiny tau block (sigma_tau ~ 4e-13 for this beam) carry the rounding of the large ones
        floor = (1e-14 if tag == "f64" else 1e-6) * np.abs(rcov).max()
        assert np.all(np.abs(cov[:6, :6] - rcov[:6, :6]) <= tol * np.outer(sig, sig) + floor), (i, [k for k, _ in spec])
        assert float(out.energy) == pytest.approx(float(g[f"penergy_out_{i}"]), rel=1e-12 if tag == "f64" else 1e-6)


def test_drawn_lattices_with_diagnostics_vs_reference():
    """parameter_beams_random.npz (tests/golden/generate_golden_random_parameter_beams.py): ten drawn beamlines, some with
    vectorised quadrupole strengths / corrector angles / cavity voltages ((3,) and (2, 1)) and a vectorised energy, an active BPM
    in the middle and an active Screen with drawn resolution, pixel size, binning and misalignment at the end: outgoing mu, cov,
    energy, s, total charge (values AND shapes), the BPM reading, the Screen's read beam and its image against the reference
    in float64. (For a vectorised ParameterBeam the reference's Screen raises NotImplementedError, screen.py:253-258; this
    engine returns one image per vector entry — a superset, not asserted here.)"""
    import json
    import os

    import cheetah_amd as ca

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parameter_beams_random.npz"))
    kw = {"dtype": f64, "device": "cuda"}
    for i in range(int(g["n_cases"])):
        spec = json.loads(str(g[f"spec_{i}"]))
        elements = []
        for kind, args in spec:
            targs = {k: (tuple(v) if k == "resolution" else torch.tensor(v, **kw) if isinstance(v, (float, list)) else v)
                     for k, v in args.items()}
            elements.append(getattr(ca, kind)(**targs, **kw))
        seg = ca.Segment(elements)
        beam = ca.ParameterBeam(t64(g[f"mu_in_{i}"]), t64(g[f"cov_in_{i}"]), t64(g[f"energy_in_{i}"]),
                                total_charge=t64(g[f"charge_in_{i}"]), species=ca.Species("electron", **kw))
        out = seg.track(beam)
        mu_ref, cov_ref = g[f"mu_out_{i}"], g[f"cov_out_{i}"]
        assert tuple(out.mu.shape) == mu_ref.shape and tuple(out.cov.shape) == cov_ref.shape, (i, out.mu.shape, mu_ref.shape)
        assert tuple(out.energy.shape) == g[f"energy_out_{i}"].shape, (i, out.energy.shape, g[f"energy_out_{i}"].shape)
        sig = np.sqrt(np.abs(np.diagonal(cov_ref, axis1=-2, axis2=-1)))[..., :6]
        assert (np.abs(out.mu.cpu().numpy() - mu_ref)[..., :6] / (np.abs(mu_ref[..., :6]) + sig)).max() < 1e-10, i
        assert (np.abs(out.cov.cpu().numpy() - cov_ref)[..., :6, :6] / (sig[..., :, None] * sig[..., None, :])).max() < 1e-9, i
        assert np.allclose(out.energy.cpu().numpy(), g[f"energy_out_{i}"], rtol=1e-13)
        assert np.allclose(out.s.cpu().numpy(), g[f"s_out_{i}"], rtol=1e-13) and tuple(out.s.shape) == g[f"s_out_{i}"].shape
        assert np.allclose(out.total_charge.cpu().numpy(), g[f"charge_out_{i}"], rtol=1e-13)
        bpm = seg.bpm.reading.cpu().numpy()
        assert bpm.shape == g[f"bpm_{i}"].shape and np.allclose(bpm, g[f"bpm_{i}"], rtol=1e-9, atol=1e-14), i
        if f"image_{i}" in g.files:
            rb = seg.screen.get_read_beam()
            assert np.allclose(rb.mu.cpu().numpy(), g[f"read_mu_{i}"], rtol=1e-9, atol=1e-14)
            img, ref = seg.screen.reading.cpu().numpy(), g[f"image_{i}"]
            assert img.shape == ref.shape, (i, img.shape, ref.shape)
            # the reference samples the density on a dtype-less torch.arange grid (screen.py:283-287), i.e. at float32-rounded
            # positions even for a float64 beam; the kernel samples at the exact pixel origins: 1e-7 of the pixel position,
            # times the density's slope
            assert np.abs(img - ref).max() <= 3e-6 * ref.max(), (i, np.abs(img - ref).max() / ref.max())
