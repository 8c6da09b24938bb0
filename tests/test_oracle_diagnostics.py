"""The CPU oracle walks the lattices of tests/golden/diagnostics_stretch.npz, scan_stretch.npz and long_lattice.npz element by element — maps
(chxo_build_rmatrix), `x @ R.mT`, Cavity.track, the weighted means a BPM reads (bpm.py:77-87), the aperture mask
(aperture.py:104-128) — and must reproduce the REFERENCE's float64 run: the third side of the triangle reference / oracle / HIP
(the HIP path against the same files: tests/test_gpu_diagnostics_stretch_golden.py, tests/test_gpu_fast_run.py)."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _element_map(oracle, kind, kw, E):
    if kind == "Drift":
        return oracle.build_rmatrix("drift", [kw["length"]], E)
    if kind == "Quadrupole":
        mis = kw.get("misalignment", [0.0, 0.0])
        return oracle.build_rmatrix("quadrupole", [kw["length"], kw["k1"], kw.get("tilt", 0.0), mis[0], mis[1]], E)   # [L, k1, tilt, mis_x, mis_y]
    if kind == "HorizontalCorrector":
        return oracle.build_rmatrix("hcor", [kw["length"], kw["angle"]], E)
    if kind == "VerticalCorrector":
        return oracle.build_rmatrix("vcor", [kw["length"], kw["angle"]], E)
    if kind == "Dipole":   # [L, angle, k1, e1, e2, tilt, fint, fint_exit, gap]
        return oracle.build_rmatrix("dipole", [kw["length"], kw["angle"], 0.0, kw.get("dipole_e1", 0.0), 0.0, 0.0, 0.0, 0.0, 0.0], E)
    if kind == "CustomTransferMap":
        return np.asarray(kw["predefined_transfer_map"], dtype=np.float64).reshape(1, 7, 7)
    if kind == "Marker":
        return oracle.build_rmatrix("identity", [], E)
    raise AssertionError(kind)


def _walk(oracle, specs, x, w, E):
    """(particles, survival, energy, s, readings) of the reference's element-by-element semantics in float64."""
    x, w, s, readings = x[None].copy(), w.copy(), 0.0, []
    for kind, kw in specs:
        if kind == "BPM":
            m = oracle.moments(x, w[None])
            mis = kw.get("misalignment", [0.0, 0.0])
            readings.append([m["mu"][0, 0] - mis[0], m["mu"][0, 2] - mis[1]])
        elif kind == "Aperture":
            w = oracle.aperture_mask(x[0], w, kw["x_max"], kw["y_max"], kw["shape"])
        elif kind == "Cavity":
            ck = "cavity_sw" if kw["cavity_type"] == "standing_wave" else "cavity_tw"
            params = [kw["length"], kw["voltage"], kw["phase"], kw["frequency"]]
            R = oracle.build_rmatrix(ck, params, E)
            coeffs, e_out = oracle.cavity_coeffs(params, E)
            x = oracle.cavity_track(x, R, coeffs)
            E = float(e_out[0])
            s += kw["length"]
        else:
            x = oracle.apply(x, _element_map(oracle, kind, kw, E))
            s += kw.get("length", 0.0)
    return x[0], w, E, s, np.asarray(readings)


def test_oracle_walks_the_diagnostics_lattices_like_the_reference(oracle):
    g = np.load(os.path.join(GOLDEN, "diagnostics_stretch.npz"))
    for i in range(int(g["n_lattices"])):
        specs = json.loads(str(g[f"lat{i}_spec"]))
        x, w, E, s, readings = _walk(oracle, specs, g[f"lat{i}_in"], g[f"lat{i}_w"], float(g[f"lat{i}_energy"]))
        ref = g[f"lat{i}_out"]
        err = (np.abs(x - ref).max(axis=0) / np.abs(ref).max(axis=0)).max()
        assert err < 1e-11, (i, err)
        assert np.array_equal(w, g[f"lat{i}_w_out"]), i
        assert E == pytest.approx(float(g[f"lat{i}_energy_out"]), rel=1e-13)
        assert s == pytest.approx(float(g[f"lat{i}_s_out"]), rel=1e-12)
        ref_r = g[f"lat{i}_readings"]
        live = np.isfinite(ref_r).all(axis=1)
        assert np.array_equal(np.isfinite(readings).all(axis=1), live)
        size = np.abs(ref[:, [0, 2]]).max() + np.abs(ref_r[live]).max()
        assert np.abs(readings[live] - ref_r[live]).max() / size < 1e-12, i


def test_oracle_walks_the_long_lattice_like_the_reference(oracle):
    g = np.load(os.path.join(GOLDEN, "long_lattice.npz"))
    specs = json.loads(str(g["spec"]))
    E = float(g["energy"])
    # merged like segment.py:534-574: one composed map for the 700 skippable elements
    R = oracle.compose([_element_map(oracle, k, kw, E) for k, kw in specs])
    out = oracle.apply(g["in"][None], R)[0]
    ref = g["out"]
    assert (np.abs(out - ref).max(axis=0) / np.abs(ref).max(axis=0)).max() < 1e-11
    many = oracle.apply(g["in"][None] * np.array([0.5, 1.0, 1.5]).reshape(3, 1, 1) * np.array([1, 1, 1, 1, 1, 1, 0.0]) + np.array([0, 0, 0, 0, 0, 0, 1.0]), R)
    mref = g["many_out"]
    assert (np.abs(many - mref).max(axis=(0, 1)) / np.abs(mref).max(axis=(0, 1))).max() < 1e-11


def test_oracle_walks_the_scans_row_by_row_like_the_reference(oracle):
    """tests/golden/scan_stretch.npz: four lattice settings in one reference track = four scalar walks of the oracle, one per row of
    the (4,) settings; a monitor in front of the first vectorised element holds ONE reading, equal for all rows."""
    g = np.load(os.path.join(GOLDEN, "scan_stretch.npz"))
    rows = int(g["rows"])
    for i in range(int(g["n_lattices"])):
        specs = json.loads(str(g[f"lat{i}_spec"]))
        ref, w_ref = g[f"lat{i}_out"], g[f"lat{i}_w_out"]
        for b in range(rows):
            row = [[k, {q: (v[b] if isinstance(v, list) and len(v) == rows and q in ("k1", "angle") else v) for q, v in kw.items()}] for k, kw in specs]
            x, w, E, s, readings = _walk(oracle, row, g[f"lat{i}_in"], g[f"lat{i}_w"], float(g[f"lat{i}_energy"]))
            err = (np.abs(x - ref[b]).max(axis=0) / np.abs(ref[b]).max(axis=0)).max()
            assert err < 1e-11, (i, b, err)
            assert np.array_equal(w, w_ref[b] if w_ref.ndim == 2 else w_ref), (i, b)
            assert E == pytest.approx(float(g[f"lat{i}_energy_out"]), rel=1e-13) and s == pytest.approx(float(g[f"lat{i}_s_out"]), rel=1e-12)
            size = np.abs(ref[b][:, [0, 2]]).max()
            for k in range(int(g[f"lat{i}_n_bpms"])):
                r_ref = g[f"lat{i}_reading{k}"]
                r_ref = r_ref[b] if r_ref.ndim == 2 else r_ref
                if np.isfinite(r_ref).all():
                    assert np.abs(readings[k] - r_ref).max() / (size + np.abs(r_ref).max()) < 1e-12, (i, b, k)
                else:
                    assert not np.isfinite(readings[k]).all()


def test_oracle_walks_the_energy_scans_row_by_row_like_the_reference(oracle):
    """The `lat*_escan_*` arrays of tests/golden/scan_stretch.npz: a (4,) beam energy on top of the (4,) settings = four scalar walks
    of the oracle, each at its own energy (the outgoing energies through the cavities included)."""
    g = np.load(os.path.join(GOLDEN, "scan_stretch.npz"))
    rows = int(g["rows"])
    for i in range(int(g["n_lattices"])):
        specs = json.loads(str(g[f"lat{i}_spec"]))
        ref, w_ref, e_ref = g[f"lat{i}_escan_out"], g[f"lat{i}_escan_w_out"], g[f"lat{i}_escan_energy_out"]
        for b in range(rows):
            row = [[k, {q: (v[b] if isinstance(v, list) and len(v) == rows and q in ("k1", "angle") else v) for q, v in kw.items()}] for k, kw in specs]
            x, w, E, s, readings = _walk(oracle, row, g[f"lat{i}_in"], g[f"lat{i}_w"], float(g[f"lat{i}_escan_energy"][b]))
            sub = ref.shape[-2]                                   # (the file keeps the first `sub` particles of every row)
            err = (np.abs(x[:sub] - ref[b]).max(axis=0) / np.abs(ref[b]).max(axis=0)).max()
            assert err < 1e-11, (i, b, err)
            assert np.array_equal(w[:sub], w_ref[b] if w_ref.ndim == 2 else w_ref), (i, b)
            assert E == pytest.approx(float(e_ref[b]), rel=1e-13)
            size = np.abs(ref[b][:, [0, 2]]).max()
            for k in range(int(g[f"lat{i}_n_bpms"])):
                r_ref = g[f"lat{i}_escan_reading{k}"]
                r_ref = r_ref[b] if r_ref.ndim == 2 else r_ref
                if np.isfinite(r_ref).all():
                    assert np.abs(readings[k] - r_ref).max() / (size + np.abs(r_ref).max()) < 1e-12, (i, b, k)
