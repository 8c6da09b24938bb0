"""Pins the CPU oracle (oracle/) against golden vectors generated from the real reference.

Runs without a GPU.  Tolerances: fp64 map entries / tracked coordinates 1e-12 relative to the
row scale; integer indices exact.
"""
import json

import numpy as np
import pytest


def rel_err(a, b, scale=None):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    s = np.max(np.abs(b)) if scale is None else scale
    return np.max(np.abs(a - b)) / (s if s > 0 else 1.0)


def test_maps_match_reference(golden, oracle):
    g = golden("maps.npz")
    n = int(g["n_cases"])
    assert n > 100
    worst = 0.0
    for i in range(n):
        kind = str(g[f"kind_{i}"])
        mass, nq = g[f"species_{i}"]
        R = oracle.build_rmatrix(kind, g[f"params_{i}"], g[f"energy_{i}"], mass, nq)
        Rref = g[f"R_{i}"].reshape(-1, 7, 7)
        assert R.shape == Rref.shape, (i, kind)
        # entrywise: abs error relative to max(|entry|, 1e-3 * max|R|)
        denom = np.maximum(np.abs(Rref), 1e-3 * np.max(np.abs(Rref)))
        err = np.max(np.abs(R - Rref) / denom)
        worst = max(worst, err)
        assert err < 2e-10, (i, kind, g[f"params_{i}"], err)
    print("worst map error", worst)


def test_known_answers_survey(oracle):
    # SURVEY.md section 8 a3/a4/a6 known answers (fp64 arithmetic, electron, E = 1e8). They were
    # produced from torch.tensor(<python float>) inputs, i.e. parameters rounded to fp32 first.
    f = lambda v: float(np.float32(v))  # noqa: E731
    R = oracle.build_rmatrix("drift", [1.0], 1e8)[0]
    assert R[4, 5] == pytest.approx(-2.6112674614598606e-05, rel=1e-13)
    R = oracle.build_rmatrix("quadrupole", [f(0.2), f(4.2), 0, 0, 0], 1e8)[0]
    assert R[0, 0] == pytest.approx(0.9171694353948605, rel=1e-14)
    assert R[0, 1] == pytest.approx(0.19444685526181435, rel=1e-14)
    assert R[1, 0] == pytest.approx(-0.8166767550118259, rel=1e-14)
    assert R[2, 2] == pytest.approx(1.085182604045398, rel=1e-14)
    assert R[3, 2] == pytest.approx(0.863718333391727, rel=1e-14)
    assert R[4, 5] == pytest.approx(-5.222535000741556e-06, rel=1e-13)
    R = oracle.build_rmatrix("dipole", [1.0, f(0.1), 0, 0, 0, 0, 0, 0, 0], 1e8)[0]
    assert R[0, 0] == pytest.approx(0.9950041651292624, rel=1e-14)
    assert R[0, 5] == pytest.approx(0.049959000231723, rel=1e-12)
    assert R[4, 5] == pytest.approx(0.0016397644060950995, rel=1e-11)
    R = oracle.build_rmatrix("cavity_sw", [f(1.0377), f(18.15975e6), 30.0, f(1.3e9)], 6e6)[0]
    assert R[0, 0] == pytest.approx(0.2509527579711751, rel=1e-13)
    assert R[5, 4] == pytest.approx(11.389552849339733, rel=1e-13)


def _track_fixture(g, oracle, name, kind, params, dtype=np.float64):
    x = g["incoming_particles_f32"].astype(dtype)[None]
    mass, nq = g["species"]
    # the reference test builds elements from fp32 tensors and then calls .to(float64)
    params = np.asarray(params, dtype=np.float32).astype(np.float64)
    R = oracle.build_rmatrix(kind, params, g["incoming_energy"], mass, nq).astype(dtype)
    if kind.startswith("cavity"):  # Cavity.track is not a pure matrix apply even at V = 0 (T566 term)
        coeffs, _ = oracle.cavity_coeffs(params, g["incoming_energy"], mass, nq)
        out = oracle.cavity_track(x, R, coeffs)
    else:
        out = oracle.apply(x, R)
    exp = g[f"{name}__particles"].reshape(-1, int(g["keep"]), 7)
    return out[:, : int(g["keep"])], exp


@pytest.mark.parametrize(
    "name,kind,params",
    [
        ("Drift_ParticleBeam_linear", "drift", [[1.0], [-1.0]]),
        ("Quadrupole_ParticleBeam_linear", "quadrupole", [[1.0, 1.0, 0.42, 0.01, -0.02], [1.0, -2.0, 0.42, 0.01, -0.02]]),
        ("Dipole_ParticleBeam_linear", "dipole", [[1.0, 1.0, 0, 0, 0, 0.42, 0, 0, 0], [1.0, -2.0, 0, 0, 0, 0.42, 0, 0, 0]]),
        ("HorizontalCorrector_ParticleBeam_default", "hcor", [[1.0, 1.0], [1.0, -2.0]]),
        ("VerticalCorrector_ParticleBeam_default", "vcor", [[1.0, 1.0], [1.0, -2.0]]),
        ("CombinedCorrector_ParticleBeam_default", "ccor", [[1.0, 1.0, 1.0], [1.0, -2.0, -2.0]]),
        ("Cavity_ParticleBeam_default", "cavity_sw", [[1.0, 0.0, 0.0, 0.0]]),
        ("Marker_ParticleBeam_default", "identity", []),
        ("Segment_ParticleBeam_default", "drift", [[1.0]]),
    ],
)
def test_reference_consistency_goldens(golden, oracle, name, kind, params):
    """The reference's own regression goldens (tests/test_elements.py:356-431), torch.allclose defaults."""
    g = golden("consistency.npz")
    out, exp = _track_fixture(g, oracle, name, kind, params)
    assert out.shape == exp.shape
    assert np.allclose(out, exp, rtol=1e-5, atol=1e-8)
    # and much tighter than the reference's own tolerance
    assert rel_err(out, exp) < 1e-12


def test_rbend_consistency(golden, oracle):
    # RBend(length=1, angle=[1,-2], tilt=0.42): rbend.py:104-116 sets e1 = e2 = angle/2 (+ rbend_e)
    g = golden("consistency.npz")
    p = [[1.0, a, 0, a / 2, a / 2, 0.42, 0, 0, 0] for a in (1.0, -2.0)]
    out, exp = _track_fixture(g, oracle, "RBend_ParticleBeam_linear", "dipole", p)
    assert rel_err(out, exp) < 1e-12


def test_segment_readme_c1(golden, oracle):
    g = golden("segment_readme.npz")
    E = g["energy"]
    maps = [
        oracle.build_rmatrix("drift", [0.175], E),
        oracle.build_rmatrix("quadrupole", [0.122, 8.2, 0, 0, 0], E),
        oracle.build_rmatrix("drift", [0.428], E),
        oracle.build_rmatrix("quadrupole", [0.122, -14.3, 0, 0, 0], E),
        oracle.build_rmatrix("drift", [0.204], E),
        oracle.build_rmatrix("vcor", [0.02, 9e-5], E),
        oracle.build_rmatrix("drift", [0.204], E),
        oracle.build_rmatrix("quadrupole", [0.122, 3.142, 0, 0, 0], E),
        oracle.build_rmatrix("drift", [0.179], E),
        oracle.build_rmatrix("hcor", [0.02, -1e-4], E),
        oracle.build_rmatrix("drift", [0.45], E),
        oracle.build_rmatrix("identity", [], E),
    ]
    R = oracle.compose(maps)
    assert rel_err(R[0], g["R_merged"]) < 1e-13
    out = oracle.apply(g["in_particles"][None], R)
    assert rel_err(out[0], g["out_particles"]) < 1e-12
    m = oracle.moments(out, g["survival"])
    names = ["x", "px", "y", "py", "tau", "p"]
    for j, n in enumerate(names):
        assert m["mu"][0, j] == pytest.approx(float(g[f"mom_mu_{n}"]), rel=1e-9, abs=1e-18)
        assert np.sqrt(m["cov"][0, j, j]) == pytest.approx(float(g[f"mom_sigma_{n}"]), rel=1e-11)
    assert m["cov"][0, 0, 1] == pytest.approx(float(g["mom_cov_xpx"]), rel=1e-10)
    # screen reading (cloud-in-cell, 512x256 @ 1e-5)
    res, px = g["resolution"], g["pixel_size"]
    extent = np.array([[-res[0] * px[0] / 2, res[0] * px[0] / 2], [-res[1] * px[1] / 2, res[1] * px[1] / 2]])
    grid = oracle.cic_deposit(out, (0, 2), (int(res[0]), int(res[1])), extent, charge=g["charges"],
                              survival=g["survival"], abs_charge=True)
    img = grid[0].T
    ref = np.zeros(tuple(g["img_shape"]))
    ref[g["img_idx"][:, 0], g["img_idx"][:, 1]] = g["img_val"]
    assert img.shape == ref.shape
    assert np.allclose(img, ref, rtol=1e-9, atol=1e-30)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_fodo100_c2(golden, oracle, tag):
    g = golden("fodo100.npz")
    dt = np.float64 if tag == "f64" else np.float32
    E = g[f"energy_{tag}"]
    mass, nq = g[f"species_{tag}"]  # the fixture beam was created in fp32: its species mass is fp32-rounded
    f = lambda v: float(np.float32(v)) if tag == "f32" else v  # noqa: E731
    cell = [oracle.build_rmatrix("quadrupole", [f(0.2), f(4.2), 0, 0, 0], E, mass, nq),
            oracle.build_rmatrix("drift", [f(0.8)], E, mass, nq),
            oracle.build_rmatrix("quadrupole", [f(0.2), f(-4.2), 0, 0, 0], E, mass, nq),
            oracle.build_rmatrix("drift", [f(0.8)], E, mass, nq)]
    R = oracle.compose(cell * 25)
    tol = 1e-10 if tag == "f64" else 2e-4  # the reference accumulates 100 matmuls in the working dtype
    assert rel_err(R[0], g[f"R_{tag}"]) < tol
    x = g[f"in_{tag}"][None].astype(dt)
    out = oracle.apply(x, R.astype(dt))
    scale = np.max(np.abs(g[f"out_merged_{tag}"]), axis=0)
    err = np.max(np.abs(out[0] - g[f"out_merged_{tag}"]) / np.maximum(scale, 1e-30))
    assert err < (1e-11 if tag == "f64" else 5e-4)
    # element by element
    y = x
    for m in cell * 25:
        y = oracle.apply(y, m.astype(dt))
    err = np.max(np.abs(y[0] - g[f"out_elementwise_{tag}"]) / np.maximum(scale, 1e-30), axis=0)
    # Reference quirk: every tracked beam gets `species.clone()` (element.py:190), which re-creates a
    # known species with the full-precision mass (species.py:126-134) while the fixture's incoming
    # beam (created in fp32, then .to(float64)) carries the fp32-rounded mass: from the 2nd element
    # on R45 differs by 5e-8 relative -> tau (column 4) is only comparable to ~1e-7 here.
    tol = np.full(7, 1e-11 if tag == "f64" else 5e-4)
    tol[4] = max(tol[4], 1e-7)
    assert (err < tol).all(), err


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_k1scan_c3(golden, oracle, tag):
    g = golden("k1scan.npz")
    dt = np.float64 if tag == "f64" else np.float32
    E, k1 = g[f"energy_{tag}"], g[f"k1_{tag}"].astype(np.float64)
    B = len(k1)
    q1 = np.zeros((B, 5))
    q1[:, 0], q1[:, 1] = 0.122, k1
    maps = [oracle.build_rmatrix("identity", [], E), oracle.build_rmatrix("drift", [0.17504], E),
            oracle.build_rmatrix("quadrupole", q1, E), oracle.build_rmatrix("drift", [0.428], E),
            oracle.build_rmatrix("quadrupole", [0.122, -14.3, 0, 0, 0], E), oracle.build_rmatrix("drift", [0.204], E),
            oracle.build_rmatrix("vcor", [0.02, 9e-5], E), oracle.build_rmatrix("drift", [0.204], E),
            oracle.build_rmatrix("quadrupole", [0.122, 3.142, 0, 0, 0], E), oracle.build_rmatrix("drift", [0.179], E),
            oracle.build_rmatrix("hcor", [0.02, -1e-4], E), oracle.build_rmatrix("drift", [0.45], E),
            oracle.build_rmatrix("identity", [], E)]
    R = oracle.compose(maps)
    tol = 1e-12 if tag == "f64" else 2e-5
    assert rel_err(R, g[f"R_{tag}"]) < tol
    out = oracle.apply(g[f"in_{tag}"][None].astype(dt), R.astype(dt))
    assert rel_err(out[:, :256], g[f"out_{tag}"]) < (1e-12 if tag == "f64" else 5e-5)
    m = oracle.moments(out)
    assert np.allclose(np.sqrt(m["cov"][:, 0, 0]), g[f"sigma_x_{tag}"], rtol=1e-10 if tag == "f64" else 2e-4)
    assert np.allclose(np.sqrt(m["cov"][:, 2, 2]), g[f"sigma_y_{tag}"], rtol=1e-10 if tag == "f64" else 2e-4)


def test_cavity_track(golden, oracle):
    g = golden("cavity.npz")
    n = int(g["n_cases"])
    for i in range(n):
        meta = json.loads(str(g[f"c{i}_meta"]))
        dt = np.float64 if meta["dtype"] == "f64" else np.float32
        kind = "cavity_sw" if meta["type"] == "standing_wave" else "cavity_tw"
        params = g[f"c{i}_params"]
        R = oracle.build_rmatrix(kind, params, meta["E"])
        tol = 1e-11 if dt == np.float64 else 3e-5
        assert rel_err(R, g[f"c{i}_R"].reshape(R.shape)) < tol, (i, meta)
        coeffs, e_out = oracle.cavity_coeffs(params, meta["E"])
        assert np.allclose(e_out, g[f"c{i}_energy_out"].reshape(-1), rtol=1e-13 if dt == np.float64 else 1e-6)
        out = oracle.cavity_track(g[f"c{i}_in"][None].astype(dt), R.astype(dt), coeffs)
        exp = g[f"c{i}_out"].reshape(out.shape)
        scale = np.max(np.abs(exp), axis=(0, 1))
        err = np.max(np.abs(out - exp) / scale)
        assert err < (1e-10 if dt == np.float64 else 2e-3), (i, meta, err)


def test_cavity_known_answer_survey(oracle):
    f = lambda v: float(np.float32(v))  # noqa: E731  (known answer computed from fp32-rounded inputs)
    x = np.array([[[f(1e-3), f(2e-4), f(-5e-4), f(1e-4), f(3e-5), f(1e-3), 1.0]]])
    p = [f(1.0377), f(18.15975e6), 30.0, f(1.3e9)]
    R = oracle.build_rmatrix("cavity_sw", p, 6e6)
    c, e = oracle.cavity_coeffs(p, 6e6)
    out = oracle.cavity_track(x, R, c)[0, 0]
    exp = [3.48217829953666018e-04, -2.75323654123137607e-04, -7.68438429943421066e-05,
           2.19369130523794999e-04, 2.83182080108386010e-05, 6.14198296013636065e-04, 1.0]
    assert np.allclose(out, exp, rtol=1e-12)
    assert e[0] == pytest.approx(21726804.82637446, rel=1e-14)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_moments(golden, oracle, tag):
    g = golden("moments.npz")
    x, w = g[f"particles_{tag}"], g[f"survival_{tag}"]
    m = oracle.moments(x, w)
    names = ["x", "px", "y", "py", "tau", "p"]
    rt = 1e-10 if tag == "f64" else 2e-3
    for j, n in enumerate(names):
        sig = np.sqrt(m["cov"][:, j, j])
        assert np.allclose(sig, g[f"sigma_{n}_{tag}"], rtol=rt)
        assert np.allclose(m["mu"][:, j], g[f"mu_{n}_{tag}"], rtol=rt, atol=rt * np.max(sig))
    assert np.allclose(m["cov"][:, 0, 1], g[f"cov_xpx_{tag}"], rtol=rt, atol=1e-3 * rt * 1e-8)
    assert np.allclose(m["cov"][:, 2, 3], g[f"cov_ypy_{tag}"], rtol=10 * rt, atol=1e-16)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_cic_bit_exact_vs_reference(golden, oracle, tag):
    """CPU oracle accumulates in the same order as the reference's scatter_add_ passes."""
    g = golden("cic.npz")
    pos, q, ext = g[f"pos_{tag}"], g[f"q_{tag}"], g[f"ext_{tag}"]
    B, N = pos.shape[:2]
    x = np.zeros((B, N, 7), dtype=pos.dtype)
    x[..., 0], x[..., 2], x[..., 4] = pos[..., 0], pos[..., 1], pos[..., 2]
    for nd, bins, cols in ((1, (17,), (0,)), (2, (16, 12), (0, 2)), (3, (8, 6, 10), (0, 2, 4))):
        grid = oracle.cic_deposit(x, cols, bins, ext[:nd], charge=q)
        ref = g[f"grid{nd}d_{tag}"]
        assert grid.shape == ref.shape
        assert np.array_equal(grid, ref), (nd, np.max(np.abs(grid - ref)))
    grid = oracle.cic_deposit(x, (0, 2), (9, 7), g[f"ext_v_{tag}"], charge=q)
    assert np.array_equal(grid, g[f"grid2d_v_{tag}"])


def test_screen_indices_and_images(golden, oracle):
    g = golden("screen.npz")
    n = int(g["n_cases"])
    for i in range(n):
        meta = json.loads(str(g[f"s{i}_meta"]))
        dt = np.float32 if meta["dtype"] == "f32" else np.float64
        x = g[f"s{i}_particles"][None]
        q, s = g[f"s{i}_q"], g[f"s{i}_surv"]
        mis = np.asarray(meta["misalignment"], dtype=dt)
        ref = np.zeros(tuple(g[f"s{i}_img_shape"]), dtype=dt)
        ref[g[f"s{i}_img_idx"][:, 0], g[f"s{i}_img_idx"][:, 1]] = g[f"s{i}_img_val"]
        if meta["method"] == "histogram":
            img, ij = oracle.hist2d(x, g[f"s{i}_edges_x"], g[f"s{i}_edges_y"], charge=q, survival=s, shift=mis)
            assert np.array_equal(ij[0], g[f"s{i}_ij"]), (i, meta)  # bit-exact pixel indices
            assert np.array_equal((img[0] != 0), (ref != 0))
            assert np.allclose(img[0], ref, rtol=1e-5 if dt == np.float32 else 1e-12, atol=0)
        else:
            res, b = meta["resolution"], meta["binning"]
            ext = g[f"s{i}_extent"].reshape(2, 2)
            grid = oracle.cic_deposit(x, (0, 2), (res[0] // b, res[1] // b), ext, charge=q, survival=s,
                                      shift=mis, abs_charge=True)
            img = grid[0].T
            assert img.shape == ref.shape
            assert np.array_equal(img != 0, ref != 0), (i, meta)  # same pixels hit
            assert np.array_equal(img, ref), (i, meta, np.max(np.abs(img - ref)))


@pytest.mark.parametrize("gi", [0, 1])
def test_space_charge_fp64(golden, oracle, gi):
    g = golden("space_charge.npz")
    k = f"g{gi}_f64"
    grid = tuple(int(v) for v in g[f"{k}_grid"])
    out, d = oracle.space_charge_kick(g[f"{k}_in"][None], g["energy"], g[f"{k}_charges"], g[f"{k}_survival"],
                                      g["effect_length"], grid_shape=grid, details=True)
    assert np.allclose(d["half"][0], g[f"{k}_half"][0], rtol=1e-12)
    # cells that only receive tiny (1-f) weights amplify the 1-ulp difference of sigma -> extent
    rho_ref = g[f"{k}_rho"]
    assert np.allclose(d["rho"][0] / np.prod(d["cell"][0]), rho_ref, rtol=1e-9, atol=1e-13 * rho_ref.max())
    assert rel_err(d["phi"][0], g[f"{k}_phi"]) < 1e-10
    assert rel_err(d["forces"][0, :512], g[f"{k}_forces"]) < 1e-8
    exp = g[f"{k}_out"]
    kick_scale = np.max(np.abs(exp - g[f"{k}_in"]), axis=0)
    err = np.max(np.abs(out[0] - exp), axis=0)
    # error relative to the size of the kick itself (columns 1,3,5 change)
    for c in (1, 3, 5):
        assert err[c] < 1e-6 * kick_scale[c] + 1e-20, (c, err[c], kick_scale[c])
    for c in (0, 2, 4, 6):
        assert err[c] <= 1e-15 * max(1.0, np.max(np.abs(exp[:, c]))) + 1e-18


def test_si_roundtrip(golden, oracle):
    g = golden("space_charge.npz")
    x = g["g0_f64_in"][None]
    xp = oracle.to_xyz_pxpypz(x, g["energy"])
    assert rel_err(xp[0, :256], g["g0_f64_xp"]) < 1e-14
    back = oracle.from_xyz_pxpypz(xp, g["energy"])
    assert np.allclose(back, x, rtol=1e-9, atol=1e-14)


def test_maps_match_reference_on_a_random_sweep(golden, oracle):
    """340 drawn settings (tests/golden/generate_golden_random_maps.py: magnitudes over several decades, exact zeros mixed in,
    electrons and protons, 5 MeV ... 20 GeV) of every map builder against the reference's own matrices."""
    g = golden("maps_random.npz")
    kinds = sorted(k[2:] for k in g.files if k.startswith("R_"))
    assert kinds == ["cavity_sw", "cavity_tw", "ccor", "dipole", "drift", "hcor", "quadrupole", "solenoid", "vcor"]
    for kind in kinds:
        P, E, S, R = g[f"params_{kind}"], g[f"energy_{kind}"], g[f"species_{kind}"], g[f"R_{kind}"]
        assert len(R) >= 25
        for i in range(len(R)):
            Ro = oracle.build_rmatrix(kind, P[i], E[i], S[i][0], S[i][1])[0]
            denom = np.maximum(np.abs(R[i]), 1e-3 * np.max(np.abs(R[i])))
            assert np.max(np.abs(Ro - R[i]) / denom) < 2e-10, (kind, i, P[i], E[i])
