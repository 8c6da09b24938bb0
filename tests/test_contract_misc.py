"""Small pieces of the reference's class contract that need no kernel: cloning, dtype conversion of beams, Species
construction rules, Superimposed structure, lattice pruning (mirrors tests/test_clone.py, test_beam.py, test_species.py,
test_superimposed.py, test_speed_optimizations.py:238-265, test_tracking_lengthless_elements.py of the reference)."""
import pytest
import torch


def all_elements(ca):
    t = torch.tensor
    return [
        ca.Drift(length=t(1.0)), ca.Quadrupole(length=t(0.2), k1=t(4.2), tilt=t(0.1), misalignment=t([1e-3, 2e-3])),
        ca.Dipole(length=t(0.5), angle=t(0.1), dipole_e1=t(0.05)), ca.RBend(length=t(0.5), angle=t(0.1)),
        ca.HorizontalCorrector(length=t(0.1), angle=t(1e-4)), ca.VerticalCorrector(length=t(0.1), angle=t(1e-4)),
        ca.CombinedCorrector(length=t(0.1), horizontal_angle=t(1e-4), vertical_angle=t(2e-4)),
        ca.Cavity(length=t(1.0), voltage=t(1e6), phase=t(10.0), frequency=t(1.3e9)), ca.Solenoid(length=t(0.3), k=t(0.5)),
        ca.Undulator(length=t(1.0)), ca.Sextupole(length=t(0.2), k2=t(3.0)),
        ca.TransverseDeflectingCavity(length=t(1.0), voltage=t(1e6)), ca.Marker(), ca.BPM(),
        ca.Aperture(x_max=t(1e-3), y_max=t(2e-3)), ca.Screen(resolution=(100, 80), pixel_size=t([1e-4, 1e-4])),
        ca.SpaceChargeKick(effect_length=t(0.1)), ca.CustomTransferMap(predefined_transfer_map=torch.eye(7)),
        ca.Segment([ca.Drift(length=t(1.0)), ca.Quadrupole(length=t(0.2), k1=t(1.0))]),
    ]


def test_clone_copies_tensors_and_metadata():
    import cheetah_amd as ca

    for element in all_elements(ca):
        element.metadata = {"control_system": {"pv_base": "A:Q1:"}}
        clone = element.clone()
        assert type(clone) is type(element) and clone.name == element.name
        for feature in element.defining_tensors:
            original, copied = getattr(element, feature), getattr(clone, feature)
            assert torch.allclose(original, copied, equal_nan=True), (type(element).__name__, feature)
            assert original.data_ptr() != copied.data_ptr(), (type(element).__name__, feature)
        assert clone.metadata == element.metadata and clone.metadata is not element.metadata
        clone.metadata["control_system"]["pv_base"] = "B:Q2:"
        assert element.metadata["control_system"]["pv_base"] == "A:Q1:"


@pytest.mark.parametrize("beam_class", ["ParameterBeam", "ParticleBeam"])
def test_beam_clone_dtype_conversion_and_transformed_dtype(beam_class):
    import cheetah_amd as ca

    BeamClass = getattr(ca, beam_class)
    beam = BeamClass.from_parameters(species=ca.Species("positron"))
    clone = beam.clone()
    features = [f for f in beam.defining_features if f != "species"]
    for feature in features:
        assert torch.allclose(getattr(beam, feature), getattr(clone, feature))
        assert getattr(beam, feature).data_ptr() != getattr(clone, feature).data_ptr()
        assert getattr(beam, feature).dtype == torch.float32
    assert clone.species.name == "positron" and clone.species is not beam.species
    assert clone.species.mass_eV == beam.species.mass_eV
    beam.to(torch.float64)                                  # in place, like nn.Module.to
    for feature in features:
        assert getattr(beam, feature).dtype == torch.float64, feature
    if beam_class == "ParticleBeam":
        return      # its transformed_to reads the beam moments from the kernel: checked in the GPU test below
    b64 = BeamClass.from_parameters(mu_x=torch.tensor(1e-5, dtype=torch.float64), dtype=torch.float64)
    moved = b64.transformed_to(mu_x=torch.tensor(-2e-5, dtype=torch.float64))
    for feature in features:
        assert getattr(moved, feature).dtype == torch.float64, feature


def test_species_rules():
    import cheetah_amd as ca
    from scipy.constants import physical_constants

    for name in ("proton", "electron", "deuteron"):
        assert ca.Species(name).mass_eV == physical_constants[f"{name} mass energy equivalent in MeV"][0] * 1e6
    assert ca.Species("electron").mass_eV == ca.Species("positron").mass_eV
    assert ca.Species("proton").mass_eV == ca.Species("antiproton").mass_eV
    muon = ca.Species(name="muon", num_elementary_charges=torch.tensor(-1.0), mass_eV=torch.tensor(105.6583755e6))
    assert muon.mass_eV == physical_constants["muon mass energy equivalent in MeV"][0] * 1e6
    assert muon.charge_coulomb == -1 * physical_constants["elementary charge"][0]
    for kwargs in ({"mass_eV": 1e6}, {"num_elementary_charges": 1}, {}, {"num_elementary_charges": 1, "charge_coulomb": 1.0},
                   {"mass_eV": 1e6, "mass_kg": 1e-27}):
        with pytest.raises(AssertionError):
            ca.Species(name="muon", **kwargs)


def test_superimposed_structure():
    import cheetah_amd as ca

    t = torch.tensor
    sup = ca.Superimposed(base_element=ca.Quadrupole(length=t(1.0)), superimposed_element=ca.BPM())
    kinds = [type(e) for e in sup._segment.elements]
    assert kinds == [ca.Quadrupole, ca.BPM, ca.Quadrupole]
    assert sup._segment.elements[0].length == t(0.5) and sup._segment.elements[2].length == t(0.5) and sup.length == t(1.0)
    seg = ca.Segment([ca.Drift(length=t(1.0)),
                      ca.Superimposed(base_element=ca.Quadrupole(length=t(1.0), k1=t(1.0)), superimposed_element=ca.BPM()),
                      ca.Drift(length=t(1.0))])
    assert [type(e) for e in seg.flattened().elements] == [ca.Drift, ca.Quadrupole, ca.BPM, ca.Quadrupole, ca.Drift]
    with pytest.raises(AssertionError, match="The superimposed element must have zero length."):
        ca.Superimposed(base_element=ca.Quadrupole(length=t(1.0)), superimposed_element=ca.Dipole(length=t(0.5)))


def test_without_inactive_zero_length_elements():
    import cheetah_amd as ca

    t = torch.tensor
    segment = ca.Segment(elements=[
        ca.Drift(length=t(1.0)), ca.Dipole(length=t(0.0), angle=t(0.0)), ca.Dipole(length=t(0.0), angle=t(0.0), name="my_dipole"),
        ca.Dipole(length=t([0.0, 0.1]), angle=t(0.0)), ca.Drift(length=t(0.0)), ca.Drift(length=t(-0.1)),
        ca.Dipole(length=t(0.0), angle=t([0.5, 0.0]))])
    pruned = segment.without_inactive_zero_length_elements()
    pruned_except = segment.without_inactive_zero_length_elements(except_for=["my_dipole"])
    assert (len(segment.elements), len(pruned.elements), len(pruned_except.elements)) == (7, 4, 5)
    assert torch.allclose(segment.length, pruned.length) and torch.allclose(segment.length, pruned_except.length)


@pytest.mark.gpu
def test_tracking_lengthless_elements_and_superimposed_map():
    import cheetah_amd as ca

    kw = {"device": "cuda", "dtype": torch.float32}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    beam_in = ca.ParticleBeam.from_parameters(num_particles=100, **kw)
    out = ca.Segment([ca.Marker(name="start")]).track(beam_in)
    assert torch.allclose(out.particles, beam_in.particles)
    seg = ca.Segment([ca.Cavity(length=t(0.1), voltage=t(1e6), name="C2", **kw), ca.Marker(name="start"),
                      ca.Cavity(length=t(0.1), voltage=t(1e6), name="C1", **kw)])
    assert torch.isfinite(seg.track(beam_in).particles).all()
    quad = ca.Quadrupole(length=t(1.0), k1=t(4.2), **kw)
    sup = ca.Superimposed(base_element=quad, superimposed_element=ca.BPM())
    sp = ca.Species("electron", **kw)
    assert torch.allclose(sup.first_order_transfer_map(t(1.0e9), sp), quad.first_order_transfer_map(t(1.0e9), sp), atol=1e-6)
    b64 = ca.ParticleBeam.from_parameters(mu_x=torch.tensor(1e-5, dtype=torch.float64, device="cuda"), dtype=torch.float64, device="cuda")
    moved = b64.transformed_to(mu_x=torch.tensor(-2e-5, dtype=torch.float64, device="cuda"))
    assert all(getattr(moved, f).dtype == torch.float64 for f in moved.defining_features if f != "species")
    assert float(moved.mu_x) == pytest.approx(-2e-5, rel=1e-9)
    off = ca.Undulator(length=t(1.0), **kw)                      # test_undulator.py: an undulator that is off is a drift
    assert not off.is_active
    assert torch.allclose(off.track(beam_in).particles, ca.Drift(length=t(1.0), **kw).track(beam_in).particles, atol=1e-7)


def test_subcell_follows_the_reference_walk_for_any_names():
    """Segment.subcell (segment.py:94-141 of the reference) is written by position here; the reference's walk — an element called
    `start` opens the cut (and is skipped when include_start is off, every time it appears), the first other element called `end`
    closes it, `start` behind `end` gives an empty segment — restated as the checker, over random lattices with repeated names."""
    import itertools
    import random

    import cheetah_amd as ca

    def walk(elements, start, end, include_start, include_end):
        sub, inside = [], start is None
        for e in elements:
            if e.name == start:
                inside = True
                if include_start:
                    sub.append(e)
                continue
            if e.name == end:
                if include_end and inside:
                    sub.append(e)
                break
            if inside:
                sub.append(e)
        return [id(e) for e in sub]

    rng = random.Random(1)
    checked = 0
    for _ in range(60):
        names = [rng.choice("abcde") + ("" if rng.random() < 0.4 else str(rng.randint(0, 3))) for _ in range(rng.randint(1, 7))]
        seg = ca.Segment([ca.Marker(name=n) for n in names])
        for start, end, i_s, i_e in itertools.product([None] + names, [None] + names, (True, False), (True, False)):
            got = [id(e) for e in seg.subcell(start, end, include_start=i_s, include_end=i_e).elements]
            assert got == walk(seg.elements, start, end, i_s, i_e), (names, start, end, i_s, i_e)
            checked += 1
    assert checked > 3000
    with pytest.raises(ValueError):
        seg.subcell(start="not there")
