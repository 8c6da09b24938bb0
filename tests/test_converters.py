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
    with pytest.raises(FileNotFoundError):
        ca.ParticleBeam.from_elegant("nonexistent.sdds")     # read by converters.sdds_file when `sdds` is not installed

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


# ---- SDDS container (Elegant particle files) ------------------------------------------------------------------------------
_SDDS_STRUCT = {"short": "h", "ushort": "H", "long": "i", "ulong": "I", "long64": "q", "ulong64": "Q", "float": "f", "double": "d"}


def write_sdds(file, parameters, columns, pages, mode="ascii", endian="<", column_major=False, fixed=None, declare_endian="comment"):
    """Minimal SDDS writer for the tests. parameters / columns: [(name, type)]; pages: [(parameter values, rows)], a row is a
    tuple of column values; fixed: {parameter name: text} written as fixed_value (those take no value in the pages)."""
    import struct

    fixed = fixed or {}
    head = ["SDDS1" if not column_major else "SDDS3"]
    if mode == "binary" and declare_endian == "comment":
        head.append("!# little-endian" if endian == "<" else "!# big-endian")
    head.append('&description text="test beam, made by tests/test_converters.py", contents="phase space" &end')
    for name, t in parameters:
        extra = f', fixed_value={fixed[name]}' if name in fixed else ""
        head.append(f'&parameter name={name}, type={t}, description="parameter {name}, with a comma"{extra} &end')
    for name, t in columns:
        head.append(f"&column name={name}, type={t},\n    units=arb &end")          # a command over two lines
    data_cmd = f"&data mode={mode}"
    if column_major:
        data_cmd += ", column_major_order=1"
    if mode == "binary" and declare_endian == "field":
        data_cmd += ", endian=" + ("little" if endian == "<" else "big")
    head.append(data_cmd + " &end")
    out = ("\n".join(head) + "\n").encode("latin-1")
    free = [(n, t) for n, t in parameters if n not in fixed]
    if mode == "ascii":
        text = ""
        for values, rows in pages:
            text += "! page\n"
            for (n, t), v in zip(free, values):
                text += (f'"{v}"' if t == "string" and " " in v else str(v) if t != "double" else repr(float(v))) + "\n"
            if columns:
                text += f"{len(rows)}\n"
                for row in rows:
                    text += " ".join(f'"{v}"' if t == "string" else (repr(float(v)) if t in ("double", "float") else str(v))
                                     for (n, t), v in zip(columns, row)) + "\n"
        out += text.encode("latin-1")
    else:
        def pack(t, v):
            if t == "string":
                b = v.encode("latin-1")
                return struct.pack(endian + "i", len(b)) + b
            if t == "character":
                return v.encode("latin-1")
            return struct.pack(endian + _SDDS_STRUCT[t], v)

        for values, rows in pages:
            out += struct.pack(endian + "i", len(rows))
            for (n, t), v in zip(free, values):
                out += pack(t, v)
            if column_major:
                for c, (n, t) in enumerate(columns):
                    for row in rows:
                        out += pack(t, row[c])
            else:
                for row in rows:
                    for (n, t), v in zip(columns, row):
                        out += pack(t, v)
    with open(file, "wb") as f:
        f.write(out)


def _sdds_case():
    rng = np.random.default_rng(11)
    parameters = [("Step", "long"), ("pCentral", "double"), ("Label", "string"), ("Particles", "long64"), ("Flag", "character")]
    columns = [("x", "double"), ("xp", "double"), ("y", "double"), ("yp", "double"), ("t", "double"), ("p", "double"),
               ("dt", "float"), ("particleID", "ulong"), ("tag", "string"), ("kind", "short")]
    pages = []
    for page, n in enumerate((5, 0, 3)):
        rows = [(*(rng.normal(size=4) * 1e-4), 1e-9 + 1e-13 * rng.normal(), 200.0 + rng.normal(), float(np.float32(rng.normal())),
                 int(rng.integers(0, 2**32 - 1)), f"particle {i} of page {page}", int(rng.integers(-300, 300))) for i in range(n)]
        pages.append(((page + 1, 199.5 + page, f"bunch number {page}", 10**12 + page, "ab"[page % 2]), rows))
    return parameters, columns, pages


@pytest.mark.parametrize("variant", ["ascii", "binary-le", "binary-be", "binary-le-colmajor", "binary-be-field", "ascii-fixed"])
def test_sdds_reader_round_trips_every_layout(tmp_path, variant):
    from cheetah_amd.converters import sdds_file

    parameters, columns, pages = _sdds_case()
    file = os.path.join(tmp_path, "beam.sdds")
    fixed = {"pCentral": "199.5"} if variant == "ascii-fixed" else None
    kw = {"ascii": {}, "binary-le": {"mode": "binary"}, "binary-be": {"mode": "binary", "endian": ">"},
          "binary-le-colmajor": {"mode": "binary", "column_major": True},
          "binary-be-field": {"mode": "binary", "endian": ">", "declare_endian": "field"}, "ascii-fixed": {"fixed": fixed}}[variant]
    pages_written = pages
    if fixed:
        pages_written = [(tuple(v for (n, t), v in zip(parameters, values) if n not in fixed), rows) for values, rows in pages]
    write_sdds(file, parameters, columns, pages_written, **kw)
    data = sdds_file.load(file)
    assert data.parameterName == [n for n, _ in parameters] and data.columnName == [n for n, _ in columns]
    assert data.loaded_pages == 3 and data.mode == ("ascii" if variant.startswith("ascii") else "binary")
    assert data.description["contents"] == "phase space" and "tests/test_converters.py" in data.description["text"]
    assert data.parameterDefinition[1]["description"] == "parameter pCentral, with a comma"
    assert data.columnDefinition[0]["units"] == "arb"
    for i, (name, t) in enumerate(parameters):
        want = [199.5] * 3 if fixed and name in fixed else [values[i] for values, _ in pages]
        assert data.getParameterValueList(name) == want, name
    for c, (name, t) in enumerate(columns):
        got = data.getColumnValueLists(name)
        assert [len(g) for g in got] == [5, 0, 3]
        for page, (_, rows) in enumerate(pages):
            assert got[page] == [row[c] for row in rows], (name, page)      # exact: repr() round-trips doubles


def test_sdds_header_quirks_and_errors(tmp_path):
    from cheetah_amd.converters import sdds_file

    file = os.path.join(tmp_path, "quirks.sdds")
    text = (
        "SDDS2\n"
        "! a comment line\n"
        '&description text="elegant output: \\"watch\\" file & more", contents="x, y" &end\n'
        "&parameter name=Charge, symbol=\"Q\", units=C, type=double,\n"
        "   ! a comment inside a command\n"
        '   description="Beam charge" &end\n'
        "&parameter name=Comment, type=string &end\n"
        "&associate filename=lattice.lte, path=., contents=lattice, sdds=0 &end\n"
        "&column name=x, units=m, type=double &end\n"
        "&column name=n, type=long &end\n"
        "&data mode=ascii, lines_per_row=2, no_row_counts=1, additional_header_lines=1 &end\n"
        "this header line is skipped\n"
        "  1.5e-9\n"
        "a comment, unquoted\n"
        " 1e-3\n 7\n"
        "! comment between rows\n"
        " -2e-3\n 8\n"
        "\n"
        "2.5e-9\n"
        '"second page"\n'
        "\n")
    with open(file, "w") as f:
        f.write(text)
    data = sdds_file.load(file)
    assert data.description["text"] == 'elegant output: "watch" file & more'
    assert data.parameterDefinition[0]["symbol"] == "Q" and data.parameterDefinition[0]["units"] == "C"
    assert data.getParameterValueList("Charge") == [1.5e-9, 2.5e-9]
    assert data.getParameterValueList("Comment") == ["a comment, unquoted", "second page"]
    assert data.getColumnValueLists("x") == [[1e-3, -2e-3], []] and data.getColumnValueLists("n") == [[7, 8], []]

    def load_text(body):
        with open(file, "w") as f:
            f.write(body)
        return sdds_file.load(file)

    with pytest.raises(ValueError, match="not an SDDS file"):
        load_text("# something else\n")
    with pytest.raises(ValueError, match="without a &data"):
        load_text("SDDS1\n&column name=x, type=double &end\n")
    with pytest.raises(NotImplementedError, match="include"):
        load_text("SDDS1\n&include filename=other.sdds &end\n&data mode=ascii &end\n")
    with pytest.raises(ValueError, match="2 of 3 elements of array a"):
        load_text("SDDS1\n&array name=a, type=double &end\n&data mode=ascii &end\n3\n1.0 2.0\n")
    with pytest.raises(ValueError, match="unknown type"):
        load_text("SDDS1\n&column name=x, type=quad &end\n&data mode=ascii &end\n")
    with pytest.raises(ValueError, match="ends after 1 of 3 rows"):
        load_text("SDDS1\n&column name=x, type=double &end\n&data mode=ascii &end\n3\n1.0\n")
    with open(file, "wb") as f:
        f.write(b"SDDS1\n&column name=x, type=double &end\n&data mode=binary &end\n" + b"\x02\x00\x00\x00" + b"\x00" * 8)
    with pytest.raises(ValueError, match="ends inside a page"):
        sdds_file.load(file)


def _extended(v: float) -> bytes:
    """x86 80-bit extended precision in 16 bytes, as SDDS writes a longdouble."""
    import math
    import struct

    if v == 0.0:
        return bytes(16)
    m, e = math.frexp(abs(v))
    return struct.pack("<QH", int(m * 2.0**64), (e - 1 + 16383) | (0x8000 if v < 0 else 0)) + bytes(6)


@pytest.mark.parametrize("mode", ["ascii", "binary"])
def test_sdds_reader_arrays_and_longdouble(tmp_path, mode):
    """`&array` blocks sit between the parameters and the table of every page; `longdouble` is the 16-byte x86 extended
    format in binary data and a plain number in ASCII data."""
    import struct

    from cheetah_amd.converters import sdds_file

    file = os.path.join(tmp_path, "arrays.sdds")
    head = ("SDDS5\n!# little-endian\n&parameter name=Step, type=long &end\n&parameter name=Precise, type=longdouble &end\n"
            "&array name=Grid, type=double, dimensions=2 &end\n&array name=Names, type=string &end\n"
            "&array name=Wide, type=longdouble &end\n"
            "&column name=x, type=double &end\n&column name=w, type=longdouble &end\n"
            f"&data mode={mode} &end\n").encode("latin-1")
    pages = [
        (1, 0.1, [2, 3], [1.5, -2.5, 3.25, 4.0, 5.0, 6.0], ["first name", "b"], [1e300, -2.0 ** -1030, 0.0], [(0.25, 1.0 / 3.0), (-4.0, 2.5e-7)]),
        (2, -7.75, [0, 3], [], [], [3.0], []),
    ]
    if mode == "ascii":
        body = ""
        for step, precise, dims, grid, names, wide, rows in pages:
            body += f"{step}\n{precise!r}\n{dims[0]} {dims[1]}\n"
            body += "".join(" ".join(repr(v) for v in grid[i:i + 4]) + "\n" for i in range(0, len(grid), 4))      # 4 per line
            body += f"{len(names)}\n" + (" ".join(f'"{n}"' for n in names) + "\n" if names else "")
            body += f"{len(wide)}\n" + " ".join(repr(v) for v in wide) + "\n"
            body += f"{len(rows)}\n" + "".join(f"{x!r} {w!r}\n" for x, w in rows)
        raw = head + body.encode("latin-1")
    else:
        raw = head
        for step, precise, dims, grid, names, wide, rows in pages:
            raw += struct.pack("<i", len(rows)) + struct.pack("<i", step) + _extended(precise)
            raw += struct.pack("<2i", *dims) + struct.pack(f"<{len(grid)}d", *grid)
            raw += struct.pack("<i", len(names)) + b"".join(struct.pack("<i", len(n)) + n.encode() for n in names)
            raw += struct.pack("<i", len(wide)) + b"".join(_extended(v) for v in wide)
            raw += b"".join(struct.pack("<d", x) + _extended(w) for x, w in rows)
    with open(file, "wb") as f:
        f.write(raw)
    data = sdds_file.load(file)
    assert data.loaded_pages == 2 and data.arrayName == ["Grid", "Names", "Wide"]
    assert data.getParameterValueList("Step") == [1, 2]
    assert data.getParameterValueList("Precise") == [0.1, -7.75]
    assert data.arrayDimensions[0] == [[2, 3], [0, 3]] and data.arrayDimensions[1] == [[2], [0]]
    assert data.arrayData[0] == [pages[0][3], []]
    assert data.arrayData[1] == [["first name", "b"], []]
    assert data.arrayData[2] == [pages[0][5], [3.0]]                 # subnormal-in-double and large values survive
    assert data.getColumnValueLists("x") == [[0.25, -4.0], []]
    assert data.getColumnValueLists("w") == [[1.0 / 3.0, 2.5e-7], []]

    if mode == "binary":                                              # the layout of a big-endian longdouble is machine dependent
        with open(file, "wb") as f:
            f.write(raw.replace(b"little-endian", b"big-endian"))
        with pytest.raises(NotImplementedError, match="longdouble"):
            sdds_file.load(file)
        inf = sdds_file._extended_to_float(struct.pack("<QH", 1 << 63, 0xFFFF))
        assert inf == -float("inf") and np.isnan(sdds_file._extended_to_float(struct.pack("<QH", 3 << 62, 0x7FFF)))
        assert sdds_file._extended_to_float(struct.pack("<QH", 1 << 63, 0x7FFE)) == float("inf")     # beyond double range


@pytest.mark.parametrize("mode", ["ascii", "binary"])
def test_particle_beam_from_an_elegant_sdds_file(tmp_path, mode):
    import cheetah_amd as ca
    from cheetah_amd.converters.elegant import ELECTRON_MASS_EV, elegant_to_cheetah_coordinates

    rng = np.random.default_rng(3)
    n = 64
    rows6 = np.stack([rng.normal(size=n) * 2e-4, rng.normal(size=n) * 1e-5, rng.normal(size=n) * 1e-4, rng.normal(size=n) * 2e-5,
                      1e-9 + rng.normal(size=n) * 1e-13, 195.7 * (1 + rng.normal(size=n) * 1e-3)], axis=1)
    q = np.full(n, 2.5e-13)
    file = os.path.join(tmp_path, "bunch.sdds")
    columns = [(c, "double") for c in ("x", "xp", "y", "yp", "t", "p")] + [("particleID", "ulong"), ("q", "double")]
    rows = [(*map(float, r), i + 1, float(qi)) for i, (r, qi) in enumerate(zip(rows6, q))]
    write_sdds(file, [("Step", "long"), ("pCentral", "double"), ("Particles", "long")], columns, [((1, 195.7, n), rows)], mode=mode)
    beam = ca.ParticleBeam.from_elegant(file, **F64)
    want = elegant_to_cheetah_coordinates(torch.tensor(rows6[None], **F64), torch.tensor([195.7], **F64))
    assert beam.particles.shape == (1, n, 7) and torch.equal(beam.particles, want)
    assert torch.equal(beam.particle_charges, torch.tensor(q[None], **F64))
    assert float(beam.energy[0]) == pytest.approx(np.sqrt((195.7 * ELECTRON_MASS_EV) ** 2 + ELECTRON_MASS_EV ** 2), rel=1e-15)
    assert beam.species.name == "electron"
    # no pCentral parameter, no charge column: the first particle is the reference, unit charges (elegant.py:500-519)
    write_sdds(file, [("Step", "long")], columns[:6], [((1,), [r[:6] for r in rows])], mode=mode)
    beam2 = ca.ParticleBeam.from_elegant(file, **F64)
    assert float(beam2.particles[0, 0, 5]) == 0.0 and torch.all(beam2.particle_charges == 1.0)
    # other conventions are refused like the reference does
    write_sdds(file, [], [(c, "double") for c in ("r", "pz", "pr", "pphi", "t", "q")], [((), [(0.0,) * 6])], mode=mode)
    with pytest.raises(ValueError, match="spiffe"):
        ca.ParticleBeam.from_elegant(file)
    write_sdds(file, [], [(c, "double") for c in ("x", "y", "z")], [((), [(0.0,) * 3])], mode=mode)
    with pytest.raises(ValueError, match="Elegant beam convention"):
        ca.ParticleBeam.from_elegant(file)


@pytest.mark.parametrize("name", ["elegant_bunch_binary.sdds", "elegant_bunch_ascii.sdds", "elegant_bunch_colmajor_be.sdds"])
def test_elegant_sdds_import_against_spec_bytes_and_reference_conversion(name, tmp_path):
    """The pin of `ParticleBeam.from_elegant` / converters.sdds_file: the three committed files were laid
    out byte by byte from the SDDS specification by tests/golden/generate_golden_sdds.py — a script that does not import
    cheetah_amd — with the header elegant's bunch files carry; `sdds_beam.npz` holds what the REFERENCE's conversion
    (`elegant_to_cheetah_coordinates` and the energy / charge expressions of `convert_beam`, elegant.py:497-567) makes of the
    numbers in them. Exact in float64; every parameter and column of the container is checked as well."""
    import cheetah_amd as ca
    from cheetah_amd.converters import sdds_file

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "converters")
    g = np.load(os.path.join(here, "sdds_beam.npz"))
    data = sdds_file.load(os.path.join(here, name))
    assert data.parameterName == ["Step", "pCentral", "Charge", "Particles", "IDSlotsPerBunch", "SVNVersion", "Label"]
    assert data.columnName == ["x", "xp", "y", "yp", "t", "p", "dt", "particleID"]
    assert data.loaded_pages == 1 and data.mode == ("ascii" if "ascii" in name else "binary")
    assert data.getParameterValueList("Step") == [int(g["steps"][0])]
    assert data.getParameterValueList("pCentral") == [float(g["p_central"][0])]
    assert data.getParameterValueList("Charge") == [float(g["bunch_charge"][0])]
    assert data.getParameterValueList("Particles") == [12] and data.getParameterValueList("IDSlotsPerBunch") == [100]
    assert data.getParameterValueList("SVNVersion") == ["unknown"]                      # fixed_value: not stored in the pages
    assert data.getParameterValueList("Label") == [str(g["labels"][0])]
    assert data.columnDefinition[1]["symbol"] == "x'" and data.parameterDefinition[1]["units"] == "m$be$nc"
    for c in range(7):
        assert data.columnData[c][0] == [float(v) for v in g["elegant_rows"][0][:, c]], data.columnName[c]
    assert data.getColumnValueLists("particleID")[0] == [int(v) for v in g["ids"][0]]
    beam = ca.ParticleBeam.from_elegant(os.path.join(here, name), **F64)
    assert beam.particles.shape == (1, 12, 7)
    assert np.array_equal(beam.particles.numpy(), g["particles"])                       # the reference's conversion, bit for bit
    assert np.array_equal(beam.energy.numpy(), g["energy"]) and np.array_equal(beam.particle_charges.numpy(), g["charges"])
    assert beam.species.name == "electron"
    if "ascii" in name:
        # the same file without its pCentral parameter: the first particle's momentum is the reference (elegant.py:500-506)
        text = open(os.path.join(here, name), encoding="latin-1").read()
        text = text.replace('&parameter name=pCentral, symbol="p$bcen$n", units="m$be$nc", description="Reference beta*gamma", type=double, &end\n', "")
        text = text.replace(f"\n{float(g['p_central'][0])!r}\n", "\n", 1)
        stripped = os.path.join(tmp_path, "no_pcentral.sdds")
        with open(stripped, "w", encoding="latin-1") as f:
            f.write(text)
        beam2 = ca.ParticleBeam.from_elegant(stripped, **F64)
        assert np.array_equal(beam2.particles.numpy(), g["particles_first"]) and np.array_equal(beam2.energy.numpy(), g["energy_first"])
