"""Splitting and re-merging every element, merge rules of Segment, tracking with non-leaf settings (mirrors the
reference's tests/test_split.py, test_merge.py and test_differentiable.py:96-147)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

KW = {"dtype": torch.float64, "device": "cuda"}


def t(v):
    return torch.tensor(v, **KW)


def elements(ca):
    return {
        "Drift": ca.Drift(t(0.3), **KW), "Drift_dkd": ca.Drift(t(0.3), tracking_method="drift_kick_drift", **KW),
        "Quadrupole": ca.Quadrupole(t(0.2), k1=t(4.2), tilt=t(0.1), misalignment=t([1e-4, -2e-4]), **KW),
        "Quadrupole_dkd": ca.Quadrupole(t(0.2), k1=t(4.2), tracking_method="drift_kick_drift", **KW),
        "Quadrupole_2nd": ca.Quadrupole(t(0.2), k1=t(4.2), tracking_method="second_order", **KW),
        "Dipole": ca.Dipole(t(0.5), angle=t(0.1), dipole_e1=t(0.05), dipole_e2=t(0.02), tilt=t(0.1), **KW),
        "RBend": ca.RBend(t(0.5), angle=t(0.1), **KW),
        "HorizontalCorrector": ca.HorizontalCorrector(t(0.1), angle=t(1e-4), **KW),
        "VerticalCorrector": ca.VerticalCorrector(t(0.1), angle=t(1e-4), **KW),
        "CombinedCorrector": ca.CombinedCorrector(t(0.1), horizontal_angle=t(1e-4), vertical_angle=t(-2e-4), **KW),
        "Solenoid": ca.Solenoid(t(0.3), k=t(0.5), **KW), "Undulator": ca.Undulator(t(0.3), **KW),
        "Sextupole": ca.Sextupole(t(0.2), k2=t(3.0), **KW),
        "Cavity": ca.Cavity(t(0.6), voltage=t(1e6), phase=t(10.0), frequency=t(1.3e9), **KW),
        "TransverseDeflectingCavity": ca.TransverseDeflectingCavity(t(0.3), voltage=t(1e5), **KW),
        "Marker": ca.Marker(**KW), "BPM": ca.BPM(**KW), "Aperture": ca.Aperture(x_max=t(1.0), y_max=t(1.0), **KW),
    }


def beam(ca):
    torch.manual_seed(0)
    return ca.ParticleBeam.from_parameters(num_particles=3000, sigma_p=t(1e-4), **KW)


def test_split_and_remerge_end_at_the_same_place():
    import cheetah_amd as ca

    incoming = beam(ca)
    for name, original in elements(ca).items():
        pieces = original.split(resolution=t(0.015))
        assert all(p.length.dtype == original.length.dtype for p in pieces), name
        split = ca.Segment(pieces)
        merged = split.with_consecutive_elements_merged()
        assert all(m.length.dtype == original.length.dtype for m in merged.elements), name
        rtol = 1e-2 if original.tracking_method == "second_order" else 1e-5
        out_original = original.track(incoming).particles
        out_split, out_merged = split.track(incoming).particles, merged.track(incoming).particles
        scale = out_original.abs().max(dim=0).values
        assert torch.all((out_split - out_original).abs() <= rtol * out_original.abs() + 1e-9 * scale + 1e-20), name
        assert torch.all((out_merged - out_split).abs() <= rtol * out_split.abs() + 1e-9 * scale + 1e-20), name
        assert float(sum(p.length for p in pieces)) == pytest.approx(float(original.length), rel=1e-12, abs=1e-15), name


def test_merge_rules_of_segment():
    import cheetah_amd as ca

    drifts = [ca.Drift(t(0.5), name=f"d{i}", **KW) for i in range(6)]
    merged = ca.Segment(drifts).with_consecutive_elements_merged(except_for=["d4"])
    assert [e.name for e in merged.elements] == ["d", "d4", "d5"]
    d = [ca.Drift(t(0.5), name=f"drift_{i}", **KW) for i in range(1, 6)]
    parent = ca.Segment([ca.Segment(d[0:2], name="sub1"), ca.Segment(d[2:4], name="sub2"), d[4]], name="parent")
    merged_parent = parent.with_consecutive_elements_merged()
    for sub in merged_parent.elements[:2]:
        assert isinstance(sub, ca.Segment) and len(sub.elements) == 1 and float(sub.elements[0].length) == 1.0
    mixed = ca.Segment([ca.Drift(t(0.5), name="d1", **KW), ca.Drift(t(0.5), name="d2", **KW),
                        ca.Quadrupole(t(0.2), name="q1", **KW), ca.Drift(t(0.5), name="d3", **KW),
                        ca.Drift(t(0.5), name="d4", **KW)]).with_consecutive_elements_merged()
    assert [e.name for e in mixed.elements] == ["d", "q1", "d"]


def test_nonleaf_settings_parameters_and_beam_gradients():
    import cheetah_amd as ca

    kw = {"dtype": torch.float32, "device": "cuda"}
    b = ca.ParticleBeam.from_parameters(num_particles=1000, **kw)
    for name, element in elements(ca).items():
        element = element.to(torch.float32)
        seg = ca.Segment([ca.Drift(torch.tensor(1.0, requires_grad=True, **kw), **kw), element])
        out = seg.track(b)
        assert out.particles.shape == (1000, 7), name
    dipole_with_buffer = ca.Dipole(length=torch.tensor(1.0))
    assert len(list(dipole_with_buffer.parameters())) == 0
    parameter = torch.nn.Parameter(torch.tensor(0.2))
    initial = ca.Dipole(length=torch.tensor(1.0), angle=parameter)
    assigned = ca.Dipole(length=torch.tensor(1.0))
    assigned.angle = parameter
    assert list(initial.parameters()) == list(assigned.parameters()) and len(list(initial.parameters())) == 1
    assert any(p is parameter for p in initial.parameters())
    gbeam = ca.ParticleBeam.from_parameters(num_particles=100, mu_x=torch.tensor(0.0, requires_grad=True, **kw),
                                            mu_y=torch.tensor(0.0, requires_grad=True, **kw), energy=torch.tensor(1e6, **kw),
                                            **kw)
    assert gbeam.x.requires_grad and gbeam.y.requires_grad


def test_cavity_vectorised_large_phase_runs():
    """tests/test_cavity.py:7-32: used to trip a scalar `assert Ei > 0` in the reference."""
    import cheetah_amd as ca

    kw = {"dtype": torch.float32, "device": "cuda"}
    v = lambda x: torch.tensor([x, x, x], **kw)  # noqa: E731
    cavity = ca.Cavity(length=v(3.0441), voltage=v(48198468.0), phase=v(48198468.0), frequency=v(2.8560e09), name="k26_2a", **kw)
    out = cavity.track(ca.ParticleBeam.from_parameters(num_particles=100_000, sigma_x=torch.tensor(1e-5, **kw), **kw))
    assert out.particles.shape == (3, 100_000, 7) and torch.isfinite(out.particles).all()
