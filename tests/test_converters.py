"""Lattice / beam import: Bmad and Elegant lattice files, the ARES NX-tables export, Ocelot cells
(duck-typed), Astra / Elegant / openPMD particle data. Host-side parsing, so these run without a GPU. Every lattice is
compared element by element with what the reference's converters produce from the same file
(tests/golden/converters/*.json, written by tests/golden/generate_golden_converters.py with the reference's LatticeJSON
writer); the warnings asserted are the ones the reference's own tests assert (tests/test_bmad_conversion.py,
test_elegant_conversion.py, test_reading_nx_tables.py, test_astra_import.py, test_ocelot_import.py)."""
import json
import os
import types
import warnings

import numpy as np
import pytest
import torch

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "converters")
F64 = {"dtype": torch.float64}


def path(name):
    return os.path.join(HERE, name)


def as_json(segment, tmp_path):
    out = os.path.join(tmp_path, "converted.json")
    segment.to_lattice_json(out)
    with open(out) as f:
        return json.load(f)


def assert_same_lattice(mine: dict, ref: dict):
    assert mine["root"] == ref["root"]
    assert mine["lattices"] == ref["lattices"]
    assert list(mine["elements"]) == list(ref["elements"])
    for name, (kind, params) in ref["elements"].items():
        my_kind, my_params = mine["elements"][name]
        assert my_kind == kind, name
        assert set(my_params) == set(params), (name, set(my_params) ^ set(params))
        for key, value in params.items():
            if isinstance(value, (int, float, list)) and not isinstance(value, bool):
                assert np.allclose(np.asarray(my_params[key], dtype=float), np.asarray(value, dtype=float), rtol=1e-12,
                                   atol=0, equal_nan=True), (name, key, my_params[key], value)
            else:
                assert my_params[key] == value, (name, key)


def load_ref(name):
    with open(path(name + ".json")) as f:
        return json.load(f)


def test_bmad_tutorial_lattice(tmp_path):
    import cheetah_amd as ca

    with pytest.warns(ca.NotUnderstoodPropertyWarning, match="( d | g | dg )"):
        converted = ca.Segment.from_bmad(path("bmad_tutorial_lattice.bmad"), **F64)
    assert [e.name for e in converted.elements] == ["d", "b", "n", "q", "s", "v"]
    assert float(converted.q.length) == pytest.approx(0.6) and float(converted.v.length) == pytest.approx(-0.6)
    assert_same_lattice(as_json(converted, tmp_path), load_ref("bmad_tutorial_lattice"))


def test_bmad_includes_wildcards_inheritance_and_all_types(tmp_path):
    import cheetah_amd as ca

    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        converted = ca.Segment.from_bmad(path("extra.bmad"), sanitize_names=False, **F64)
    assert any(issubclass(w.category, ca.UnknownElementWarning) and "taylor" in str(w.message) for w in caught)
    assert_same_lattice(as_json(converted, tmp_path), load_ref("extra_bmad"))
    assert converted.inner.qd.k1 == -4.2 and float(converted.inner.qd.length) == 0.25   # inherited, then wildcard-set


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_bmad_dtype_passing(dtype):
    import cheetah_amd as ca

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        converted = ca.Segment.from_bmad(path("bmad_tutorial_lattice.bmad"), dtype=dtype)
    assert converted.d.length.dtype == dtype and converted.b.dipole_e1.dtype == dtype and converted.s.k2.dtype == dtype


def test_elegant_fodo(tmp_path):
    import cheetah_amd as ca

    with pytest.warns(ca.NoBeamPropertiesInLatticeWarning, match="c.*charge"), pytest.warns(
            ca.DirtyNameWarning, match="long-name-quad"), pytest.warns(ca.DirtyNameWarning, match="a:q3"), pytest.warns(
            ca.NotUnderstoodPropertyWarning, match="nonsense"):
        converted = ca.Segment.from_elegant(path("fodo.lte"), "fodo", **F64)
    assert [e.name for e in converted.elements] == ["c", "q1", "d1", "m1", "b1", "d1", "q2", "d2", "s1", "csrbend",
                                                    "long-name-quad", "d3", "a:q3"]
    assert float(converted.b1.gap) == pytest.approx(0.04) and float(converted.b1.fringe_integral) == 0.5
    assert float(getattr(converted, "a:q3").k1) == 1.5
    assert_same_lattice(as_json(converted, tmp_path), load_ref("fodo"))


def test_elegant_reversed_line(tmp_path):
    import cheetah_amd as ca

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        converted = ca.Segment.from_elegant(path("fodo.lte"), "reversed_fodo", sanitize_names=False, **F64).flattened()
        forward = ca.Segment.from_elegant(path("fodo.lte"), "fodo", sanitize_names=False, **F64)
    assert [e.name for e in converted.elements] == [e.name for e in forward.elements][::-1]
    assert_same_lattice(as_json(converted, tmp_path), load_ref("reversed_fodo"))


def test_elegant_cavity_and_ematrix(tmp_path):
    import cheetah_amd as ca

    with pytest.warns(ca.NotUnderstoodPropertyWarning, match="(end[12]_focus|body_focus_model|change_p0)"), pytest.warns(
            ca.PhysicsWarning, match="srs"):
        converted = ca.Segment.from_elegant(path("cavity.lte"), "cavity", **F64)
    assert float(converted.c1.phase) == 0.0 and float(converted.c1.voltage) == 16.175e6
    expected = torch.eye(7, **F64)
    expected[1, 0], expected[1, 2], expected[1, 6] = 0.04, 0.003, -0.0027
    expected[3, 0], expected[3, 2], expected[3, 6] = 0.003, -0.04, -0.15
    assert torch.allclose(converted.c1e.predefined_transfer_map, expected)
    assert_same_lattice(as_json(converted, tmp_path), load_ref("cavity"))


def test_elegant_all_types_and_rpn(tmp_path):
    import cheetah_amd as ca

    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        converted = ca.Segment.from_elegant(path("extra.lte"), "everything", **F64)
    assert any(issubclass(w.category, ca.UnknownElementWarning) and "twiss" in str(w.message) for w in caught)
    assert float(converted.dr.length) == 0.75      # "0.5 0.25 +" in reverse-Polish notation
    assert_same_lattice(as_json(converted, tmp_path), load_ref("extra_lte"))


def test_nx_tables(tmp_path):
    import cheetah_amd as ca

    converted = ca.Segment.from_nx_tables(path("Stage4v3_9.txt"))
    assert converted.name == "Stage4v3_9" and len(converted.elements) == 226
    assert_same_lattice(as_json(converted, tmp_path), load_ref("Stage4v3_9"))


def test_lattice_text_errors_and_expressions():
    from cheetah_amd.converters import lattice_text as lt

    ctx = lt.initial_context()
    ctx.update({"abs": -0.6, "q": {"element_type": "quadrupole", "l": 0.6}})
    assert lt.evaluate("abs(abs)", ctx) == pytest.approx(0.6)           # separate function / variable namespaces
    assert lt.evaluate("-q[l]", ctx) == pytest.approx(-0.6)
    assert lt.evaluate("0.6  -0.1", ctx) == pytest.approx(0.5)
    assert lt.evaluate("2 ^ 3 ^ 2", ctx) == pytest.approx(512.0)        # right associative
    assert lt.evaluate("-2 ^ 2", ctx) == pytest.approx(-4.0)
    # reverse-Polish (Elegant). Compound strings follow Elegant's semantics; the reference reads them through its infix
    # parser first and gets "1 2 + 4 /" = 1.5 and "3 4 * sqrt" = 6 (converters/utils/infix.py) - a deliberate difference
    assert lt.evaluate("3 4 * sqrt", ctx) == pytest.approx(12**0.5)
    assert lt.evaluate("1 2 + 4 /", ctx) == pytest.approx(0.75)
    assert lt.evaluate("pi 2 /", ctx) == pytest.approx(np.pi / 2)
    assert lt.evaluate("1e-3", ctx) == 1e-3 and lt.evaluate("12", ctx) == 12 and isinstance(lt.evaluate("12", ctx), int)
    assert lt.evaluate("traveling_wave", ctx) == "traveling_wave"
    with pytest.warns(Warning, match="not_defined"):
        assert lt.evaluate("not_defined + 1", ctx) == "not_defined + 1"
    with pytest.raises(ValueError, match="not understood"):
        lt._execute("this is not a statement", ctx)


def test_ocelot_cell_duck_typed():
    """No Ocelot in this image: stand-in classes with Ocelot's class names and attributes (tests/test_ocelot_import.py
    builds the same kinds of cells with the real package)."""
    import cheetah_amd as ca

    def make(kind, bases=(), **attrs):
        return type(kind, bases, {})

    Element = type("Element", (), {})
    classes = {k: type(k, (Element,), {}) for k in ("Drift", "Quadrupole", "Sextupole", "Solenoid", "Hcor", "Vcor", "Bend",
                                                      "Cavity", "TWCavity", "Monitor", "Marker", "Undulator", "Aperture",
                                                      "Multipole")}
    classes["SBend"] = type("SBend", (classes["Bend"],), {})
    classes["RBend"] = type("RBend", (classes["Bend"],), {})

    def el(kind, **attrs):
        obj = classes[kind]()
        for k, v in attrs.items():
            setattr(obj, k, v)
        return obj

    bend = dict(l=0.3, angle=0.1, e1=0.05, e2=0.06, tilt=0.2, fint=0.4, fintx=0.5, gap=0.03)
    cell = [
        el("Drift", id="d1", l=0.5), el("Quadrupole", id="q1", l=0.2, k1=3.0), el("Sextupole", id="s1", l=0.1, k2=7.0),
        el("Solenoid", id="sol", l=0.2, k=0.5), el("Hcor", id="h1", l=0.02, angle=1e-4), el("Vcor", id="v1", l=0.02, angle=2e-4),
        el("SBend", id="b1", **bend), el("RBend", id="b2", **bend), el("Cavity", id="c1", l=1.0, v=0.02, freq=1.3e9, phi=10.0),
        el("TWCavity", id="c2", l=4.0, v=0.06, freq=2.998e9, phi=-5.0), el("Monitor", id="ARBSCR1"), el("Monitor", id="BPM.1"),
        el("Monitor", id="other"), el("Marker", id="m"), el("Undulator", id="u", l=1.0, lperiod=0.02, Kx=1.1, Ky=0.0),
        el("Aperture", id="ap", xmax=1e-3, ymax=2e-3, type="elip"), el("Multipole", id="mp", l=0.07),
    ]
    with pytest.warns(ca.DefaultParameterWarning, match="default screen"), pytest.warns(
            ca.UnknownElementWarning, match="mp"), pytest.warns(ca.DirtyNameWarning, match="BPM.1"):
        seg = ca.Segment.from_ocelot(cell, name="cell", **F64)
    kinds = [type(e).__name__ for e in seg.elements]
    assert kinds == ["Drift", "Quadrupole", "Sextupole", "Solenoid", "HorizontalCorrector", "VerticalCorrector", "Dipole",
                     "Dipole", "Cavity", "Cavity", "Screen", "BPM", "Marker", "Marker", "Undulator", "Aperture", "Drift"]
    assert float(seg.c1.voltage) == pytest.approx(2e7) and seg.c1.cavity_type == "standing_wave"
    assert seg.c2.cavity_type == "traveling_wave" and float(seg.b2.dipole_e2) == 0.06     # RBend derives from Bend
    assert float(seg.b1.fringe_integral_exit) == 0.5 and seg.ap.shape == "elliptical" and float(seg.mp.length) == 0.07
    assert tuple(seg.ARBSCR1.resolution) == (2448, 2040)
    from cheetah_amd.converters.ocelot import subcell_of_ocelot

    assert [e.id for e in subcell_of_ocelot(cell, "q1", "sol")] == ["q1", "s1", "sol"]


def test_astra_import():
    import cheetah_amd as ca

    g = np.load(path("beams.npz"))
    beam = ca.ParticleBeam.from_astra(path("synthetic.astra"), **F64)
    assert beam.num_particles == 293 and beam.particles.dtype == torch.float64
    assert np.allclose(beam.particles.numpy(), g["astra_particles"], rtol=1e-13, atol=1e-20)
    assert float(beam.energy) == pytest.approx(float(g["astra_energy"]), rel=1e-15)
    assert np.allclose(beam.particle_charges.numpy(), g["astra_charges"], rtol=1e-15)
    pbeam = ca.ParameterBeam.from_astra(path("synthetic.astra"), **F64)
    assert np.allclose(pbeam.mu.numpy(), g["astra_mu"], rtol=1e-12, atol=1e-22)
    assert np.allclose(pbeam.cov.numpy(), g["astra_cov"], rtol=1e-11, atol=1e-30)
    assert float(pbeam.total_charge) == pytest.approx(float(g["astra_total_charge"]), rel=1e-13)
    assert ca.ParticleBeam.from_astra(path("synthetic.astra")).particles.dtype == torch.get_default_dtype()


def test_elegant_coordinates_and_openpmd_group():
    import cheetah_amd as ca
    from cheetah_amd.converters.elegant import elegant_to_cheetah_coordinates

    g = np.load(path("beams.npz"))
    out = elegant_to_cheetah_coordinates(torch.tensor(g["elegant_in"]), torch.tensor(g["elegant_pc"]))
    assert np.allclose(out.numpy(), g["elegant_out"], rtol=1e-13, atol=1e-20)
    two_pages = elegant_to_cheetah_coordinates(torch.tensor(g["elegant_in"]).repeat(2, 1, 1), torch.tensor([199.5, 199.5], **F64))
    assert torch.equal(two_pages[0], two_pages[1]) and torch.allclose(two_pages[0], out[0])
    with pytest.raises(ImportError, match="sdds"):
        ca.ParticleBeam.from_elegant("nonexistent.sdds")

    group = types.SimpleNamespace(species="electron", **{k: g["pmd_" + k] for k in ("x", "y", "px", "py", "t", "energy",
                                                                                    "weight", "status")})
    beam = ca.ParticleBeam.from_openpmd_particlegroup(group, torch.tensor(5e7, **F64), **F64)
    assert np.allclose(beam.particles.numpy(), g["pmd_particles"], rtol=1e-13, atol=1e-22)
    assert np.allclose(beam.particle_charges.numpy(), g["pmd_charges"]) and np.allclose(
        beam.survival_probabilities.numpy(), g["pmd_survival"])
    data = beam.openpmd_data()      # what the reference hands to pmd_beamphysics.ParticleGroup(data=...)
    assert np.allclose(data["x"], g["pmd_x"]) and np.allclose(data["px"], g["pmd_px"], rtol=1e-12)
    assert np.allclose(data["t"], g["pmd_t"], rtol=1e-9, atol=1e-25) and data["species"] == "electron"
    total = np.sqrt(data["px"]**2 + data["py"]**2 + data["pz"]**2 + beam.species.mass_eV.item()**2)
    assert np.allclose(total, g["pmd_energy"], rtol=1e-12)
    with pytest.raises(ImportError, match="openPMD"):
        ca.ParticleBeam.from_openpmd_file("x.h5", torch.tensor(5e7))
    with pytest.raises(ImportError, match="openPMD"):
        beam.save_as_openpmd_h5("x.h5")
