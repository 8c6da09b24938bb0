"""BPM / Screen / Segment behaviours mirroring the reference's tests/test_bpm.py, test_screen.py (dtype conversion,
readings unaffected by later in-place edits of the beams) and test_segment.py (beam generators, attribute scans,
set_attrs_on_every_element with unsupported tracking methods)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
KW = {"device": "cuda"}


def t(v, dtype=None):
    return torch.tensor(v, device="cuda", dtype=dtype)


def test_bpm_reading_dtype_and_misalignment():
    import cheetah_amd as ca

    seg = ca.Segment([ca.Drift(t(1.0), dtype=torch.float32, **KW), ca.BPM(name="bpm", is_active=True, dtype=torch.float32, **KW)])
    beam = ca.ParameterBeam.from_parameters(dtype=torch.float32, **KW)
    assert seg.bpm.reading.dtype == torch.float32 and torch.isnan(seg.bpm.reading).all()
    seg.track(beam)
    assert seg.bpm.reading.dtype == torch.float32 and torch.isfinite(seg.bpm.reading).all()
    seg = seg.double()
    assert seg.bpm.reading.dtype == torch.float64
    bpm = ca.BPM(name="bpm", is_active=True, misalignment=t([0.1, 0.2]), **KW)
    bpm.track(ca.ParameterBeam.from_parameters(mu_x=t(0.0), mu_y=t(0.0), **KW))
    assert torch.allclose(bpm.reading, -t([0.1, 0.2]))
    pbeam = ca.ParticleBeam.from_parameters(num_particles=10_000, mu_x=t(3e-4), mu_y=t(-1e-4), **KW)
    bpm.track(pbeam)
    assert torch.allclose(bpm.reading, t([3e-4 - 0.1, -1e-4 - 0.2]), atol=1e-6)


def test_screen_reading_dtype_conversion():
    import cheetah_amd as ca

    seg = ca.Segment([ca.Drift(t(1.0), dtype=torch.float32, **KW), ca.Screen(name="screen", is_active=True, dtype=torch.float32, **KW)])
    beam = ca.ParameterBeam.from_parameters(dtype=torch.float32, **KW)
    assert seg.screen.reading.dtype == torch.float32
    cloned = seg.clone()
    cloned.track(beam)
    cloned = cloned.double()
    assert cloned.screen.reading.dtype == torch.float64
    seg.track(beam)
    assert seg.screen.reading.dtype == torch.float32
    seg = seg.double()
    assert seg.screen.reading.dtype == torch.float64


def test_screen_read_beam_is_a_snapshot():
    import cheetah_amd as ca

    torch.manual_seed(0)
    incoming = ca.ParticleBeam.from_parameters(num_particles=20_000, **KW)
    screen = ca.Screen(is_active=True, resolution=(64, 48), pixel_size=t([1e-4, 1e-4]), **KW)
    outgoing = screen.track(incoming)
    image = screen.reading.clone()
    original = screen.get_read_beam().clone()
    incoming.particles *= 2.0
    incoming.energy *= 3.0
    incoming.particle_charges *= 4.0
    incoming.survival_probabilities *= 0.9
    outgoing.particles *= 0.7
    outgoing.particle_charges *= 0.3
    after = screen.get_read_beam()
    assert torch.all(original.particles == after.particles) and original.energy == after.energy
    assert torch.all(original.particle_charges == after.particle_charges)
    assert torch.all(original.survival_probabilities == after.survival_probabilities)
    screen.__dict__["_cached_reading"] = None
    assert torch.allclose(screen.reading, image, rtol=1e-5, atol=0)  # float atomics: same image up to summation order
    pin = ca.ParameterBeam.from_parameters(**KW)
    pscreen = ca.Screen(is_active=True, **KW)
    pout = pscreen.track(pin)
    orig = pscreen.get_read_beam().clone()
    pin.mu *= 2.0
    pin.cov *= 3.0
    pout.mu *= 0.7
    assert torch.all(orig.mu == pscreen.get_read_beam().mu) and torch.all(orig.cov == pscreen.get_read_beam().cov)


def test_beam_generators_and_attribute_scans():
    import cheetah_amd as ca

    seg = ca.Segment([ca.Drift(t(0.5), **KW), ca.Quadrupole(t(0.3), **KW), ca.Drift(t(0.2), **KW)])
    pbeam = ca.ParameterBeam.from_parameters(**KW)
    beams = list(seg.beam_along_segment_generator(incoming=pbeam))
    assert len(beams) == 4 and all(isinstance(b, ca.ParameterBeam) for b in beams)
    beams = list(seg.beam_along_segment_generator(incoming=pbeam, resolution=0.1))
    assert len(beams) == 11 and all(isinstance(b, ca.ParameterBeam) for b in beams)
    beam = ca.ParticleBeam.from_parameters(num_particles=3_000, **KW)
    for names in ("beta_x", ("beta_x",), ("s", "beta_x"), ("x", "mu_x")):
        res = seg.get_beam_attrs_along_segment(names, beam)
        if isinstance(names, str):
            assert isinstance(res, torch.Tensor) and len(res) == 4
        else:
            assert isinstance(res, tuple) and len(res) == len(names)
            for r, n in zip(res, names):
                assert r.shape == ((4, 3_000) if n == "x" else (4,)), n


@pytest.mark.parametrize("target", ["linear", "second_order", "drift_kick_drift", "unsupported"])
def test_setting_tracking_method_over_a_segment(target):
    import cheetah_amd as ca

    seg = ca.Segment([ca.Drift(t(0.5), name="d1", **KW), ca.Quadrupole(t(0.3), name="q1", **KW), ca.Drift(t(0.2), name="d2", **KW),
                      ca.Dipole(t(0.5), name="b1", **KW), ca.Sextupole(t(0.4), k2=t(0.1), name="s1", **KW), ca.Marker(name="m1", **KW)])
    before = {e.name: e.tracking_method for e in seg.elements}
    with pytest.warns(ca.PhysicsWarning, match=r"Invalid tracking method '.+' for element .+ of type .+, supported methods "
                                               r"are \[.+\]. Keeping the previous tracking method .+."):
        seg.set_attrs_on_every_element(tracking_method=target)
    for e in seg.elements:
        want = target if target in e.supported_tracking_methods else before[e.name]
        assert e.tracking_method == want, e.name
