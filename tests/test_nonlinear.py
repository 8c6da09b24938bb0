"""drift_kick_drift (Bmad-X) and second_order tracking.

CPU (`-m "not gpu"`): the oracle against the Bmad-X / Bmad-Tao results the reference's tests use, the
reference's own outputs and its consistency goldens (tests/golden/dkd.npz, second_order.npz).
GPU (`-m gpu`): the HIP kernels (through the C-ABI, and through the Element API) against the oracle and the
same goldens. Tolerances follow the reference's tests: fp64 atol = rtol = 1e-14 against Bmad-X
(test_drift.py:64-69), fp32 atol 1e-5 / rtol 1e-6.
"""
import numpy as np
import pytest
import torch

KINDS = ["drift", "quadrupole", "dipole", "tdc"]


def _bcast(par, En):
    shp = np.broadcast_shapes(par.shape[:-1], En.shape)
    parb = np.broadcast_to(par, shp + par.shape[-1:]).reshape(-1, par.shape[-1])
    Eb = np.broadcast_to(En, shp).reshape(-1)
    return shp, np.ascontiguousarray(parb), np.ascontiguousarray(Eb)


def _to_bmad(o, e_ref, m):
    p0 = np.sqrt(e_ref**2 - m**2)
    energy = e_ref + o[5] * p0
    p = np.sqrt(energy**2 - m**2)
    return np.array([o[0], o[1], o[2], o[3], -(p / energy) * o[4], (p - p0) / p0])


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("kind", KINDS)
def test_oracle_dkd_matches_bmadx(golden, oracle, kind):
    g = golden("dkd.npz")
    m, nq = g["species"]
    out, e = oracle.dkd_track(kind, g["incoming"], g[f"bmadx_{kind}_params"], g["energy"], m, nq,
                              int(g[f"bmadx_{kind}_steps"]), int(g[f"bmadx_{kind}_fringe"]))
    assert np.allclose(out[0], g[f"bmadx_{kind}"], rtol=1e-14, atol=1e-14)     # Bmad-X itself
    assert np.allclose(out[0], g[f"ref64_{kind}"], rtol=1e-14, atol=1e-14)     # the reference run here
    assert e[0] == g[f"ref64_{kind}_energy"]
    # fp32 storage, fp64 arithmetic: inside the reference's fp32 tolerance of the fp64 truth
    o32, _ = oracle.dkd_track(kind, g["incoming"].astype(np.float32), g[f"bmadx_{kind}_params"], g["energy"], m, nq,
                              int(g[f"bmadx_{kind}_steps"]), int(g[f"bmadx_{kind}_fringe"]))
    assert np.allclose(o32[0], g[f"bmadx_{kind}"], rtol=1e-6, atol=1e-5)
    assert np.allclose(o32[0], g[f"ref32_{kind}"], rtol=1e-6, atol=1e-5)


def test_oracle_dkd_matches_bmad_tao_species(golden, oracle):
    """tests/test_compare_bmad.py: six species through Drift / Dipole / Quadrupole against Bmad (Tao)."""
    g = golden("dkd.npz")
    c = g["tao_coords"]
    for s in g["tao_species_names"]:
        m, nq = g[f"tao_{s}_species"]
        tau, delta, en = g[f"tao_{s}_tau_delta_energy"]
        xin = np.array([[c[0], c[1], c[2], c[3], tau, delta, 1.0]])
        for ename, kind in (("Drift", "drift"), ("Dipole", "dipole"), ("Quadrupole", "quadrupole")):
            out, e = oracle.dkd_track(kind, xin, g[f"tao_{ename}_params"], en, m, nq, 1, int(g[f"tao_{ename}_fringe"]))
            # torch.allclose(..., atol=1e-14) of the reference's test carries the default rtol = 1e-5
            assert np.allclose(_to_bmad(out[0, 0], e[0], m), g[f"tao_{s}_{ename}"], rtol=1e-7, atol=1e-14), (s, ename)


def test_oracle_dkd_extra_cases(golden, oracle):
    g = golden("dkd.npz")
    for name in g["extra_names"]:
        m, nq = g[f"x_{name}_species"]
        shp, par, En = _bcast(g[f"x_{name}_params"], g[f"x_{name}_energy"])
        out, e = oracle.dkd_track(str(g[f"x_{name}_kind"]), g[f"x_{name}_in"], par, En, m, nq,
                                  int(g[f"x_{name}_steps"]), int(g[f"x_{name}_fringe"]))
        ref = g[f"x_{name}_out"].reshape(out.shape)
        assert np.allclose(out, ref, rtol=1e-13, atol=1e-14), name
        assert np.allclose(e, np.broadcast_to(g[f"x_{name}_energy_out"], shp).reshape(-1), rtol=1e-15), name


def _second_order_cases(g):
    for name in g["names"]:
        shp, par, En = _bcast(g[f"{name}_params"], g[f"{name}_energy"])
        yield str(name), str(g[f"{name}_kind"]), shp, par, En, g[f"{name}_species"][0]


def _relerr(a, b):
    return np.max(np.abs(a - b)) / np.max(np.abs(b))


def test_oracle_second_order_matches_reference(golden, oracle):
    g = golden("second_order.npz")
    for name, kind, shp, par, En, m in _second_order_cases(g):
        T = oracle.build_ttensor(kind, par, En, m)
        Tr = g[f"{name}_T"].reshape(T.shape)
        for b in range(T.shape[0]):
            assert _relerr(T[b], Tr[b]) < 1e-13, (name, b)
        out = oracle.apply_second_order(g["incoming"], T)
        ref = g[f"{name}_out"].reshape(out.shape)
        for j in range(6):
            assert _relerr(out[..., j], ref[..., j]) < 1e-12, (name, j)


# ------------------------------------------------------------------------------------------------ GPU
def _dev(a, dt):
    return torch.tensor(np.asarray(a), dtype=dt, device="cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_hip_dkd_matches_bmadx_and_oracle(golden, oracle, kind, tag):
    from cheetah_amd import _ops

    g = golden("dkd.npz")
    dt = torch.float64 if tag == "f64" else torch.float32
    m, nq = g["species"]
    par = g[f"bmadx_{kind}_params"].reshape(1, -1)
    steps, fringe = int(g[f"bmadx_{kind}_steps"]), int(g[f"bmadx_{kind}_fringe"])
    x = _dev(g["incoming"], dt)
    out, e = _ops.dkd_track(_ops.DKD_KIND[kind], x, _dev(par, dt), torch.Size(()), _dev(g["energy"], dt), float(m),
                            float(nq), steps, fringe)
    out = out.cpu().numpy()
    xin = g["incoming"].astype(np.float64 if tag == "f64" else np.float32)
    ora, e_ora = oracle.dkd_track(kind, xin, par, g["energy"], m, nq, steps, fringe)
    if tag == "f64":
        assert np.allclose(out, g[f"bmadx_{kind}"], rtol=1e-14, atol=1e-14)
        assert np.allclose(out, ora[0], rtol=1e-14, atol=1e-14)
        assert float(e) == e_ora[0]
    else:
        assert np.allclose(out, g[f"bmadx_{kind}"], rtol=1e-6, atol=1e-5)
        # same fp32 inputs, fp64 arithmetic on both sides: identical up to the final rounding to fp32
        assert np.allclose(out, ora[0], rtol=3e-7, atol=1e-9)
    assert np.all(out[:, 6] == 1.0)


@pytest.mark.gpu
def test_hip_dkd_elements_match_reference_cases(golden):
    """The extra reference-run cases through the Element API (vectorised parameters / energies, species, fringes)."""
    import cheetah_amd as ca

    g = golden("dkd.npz")
    f64 = torch.float64
    kw = {"dtype": f64, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    dk = {"tracking_method": "drift_kick_drift"}
    elements = {
        "drift_vec": ca.Drift(length=t([1.0, -1.0, 0.3]), **dk, **kw),
        "drift_proton": ca.Drift(length=t(2.0), **dk, **kw),
        "quad_vec": ca.Quadrupole(length=t(1.0), k1=t([1.0, -2.0, 0.0]), tilt=t(0.42), misalignment=t([0.01, -0.02]), **dk, **kw),
        "quad_steps5_proton": ca.Quadrupole(length=t(0.4), k1=t(-3.0), num_steps=5, **dk, **kw),
        "quad_energy_vec": ca.Quadrupole(length=t(0.2), k1=t([4.2, -4.2]), num_steps=3, **dk, **kw),
        "dipole_vec": ca.Dipole(length=t(1.0), angle=t([1.0, -2.0]), tilt=t(0.42), **dk, **kw),
        "dipole_zero": ca.Dipole(length=t(1.0), angle=t(0.0), **dk, **kw),
        "dipole_entrance": ca.Dipole(length=t(0.7), angle=t(0.3), dipole_e1=t(0.1), dipole_e2=t(0.2), fringe_integral=t(0.4),
                                     fringe_integral_exit=t(0.6), gap=t(0.03), gap_exit=t(0.05), fringe_at="entrance", **dk, **kw),
        "dipole_exit_proton": ca.Dipole(length=t(0.7), angle=t(-0.3), dipole_e1=t(0.1), dipole_e2=t(0.2),
                                        fringe_integral=t(0.4), fringe_integral_exit=t(0.6), gap=t(0.03), gap_exit=t(0.05),
                                        fringe_at="exit", **dk, **kw),
        "dipole_neither": ca.Dipole(length=t(0.7), angle=t(0.3), dipole_e1=t(0.1), fringe_at="neither", **dk, **kw),
        "rbend": ca.RBend(length=t(1.0), angle=t([1.0, -2.0]), tilt=t(0.42), **dk, **kw),
        "tdc_vec": ca.TransverseDeflectingCavity(length=t(1.0), voltage=t([[1e7], [2e7], [0.0]]), phase=t(0.4),
                                                 frequency=t(1e9), **kw),
        "tdc_misaligned": ca.TransverseDeflectingCavity(length=t(0.8), voltage=t(5e6), phase=t(-0.1), frequency=t(2.856e9),
                                                        tilt=t(0.3), misalignment=t([1e-3, -2e-3]), **kw),
    }
    assert set(elements) == set(str(n) for n in g["extra_names"])
    for name, el in elements.items():
        m, nq = g[f"x_{name}_species"]
        sp = ca.Species("custom", num_elementary_charges=t(float(nq)), mass_eV=t(float(m)), **kw)
        beam = ca.ParticleBeam(t(g[f"x_{name}_in"]), t(g[f"x_{name}_energy"]), species=sp)
        assert not el.is_skippable
        out = ca.Segment([el]).track(beam)
        ref = g[f"x_{name}_out"]
        assert tuple(out.particles.shape) == ref.shape, name
        assert np.allclose(out.particles.cpu().numpy(), ref, rtol=1e-13, atol=1e-14), name
        assert tuple(out.energy.shape) == g[f"x_{name}_energy_out"].shape, name
        assert np.allclose(out.energy.cpu().numpy(), g[f"x_{name}_energy_out"], rtol=1e-15), name


@pytest.mark.gpu
def test_hip_dkd_consistency_goldens(golden):
    """tests/test_elements.py:356-431 for the drift_kick_drift / TDC configurations of tests/conftest.py."""
    import cheetah_amd as ca

    g, c = golden("dkd.npz"), golden("consistency.npz")
    f64 = torch.float64
    kw = {"dtype": f64, "device": "cuda"}
    t = lambda v: torch.tensor(v, dtype=torch.float32).to(f64).cuda()  # noqa: E731
    dk = {"tracking_method": "drift_kick_drift"}
    elements = {
        "Drift_ParticleBeam_drift_kick_drift": ca.Drift(length=t([1.0, -1.0]), **dk, **kw),
        "Quadrupole_ParticleBeam_drift_kick_drift": ca.Quadrupole(length=t(1.0), k1=t([1.0, -2.0]), tilt=t(0.42),
                                                                  misalignment=t([0.01, -0.02]), **dk, **kw),
        "Dipole_ParticleBeam_drift_kick_drift": ca.Dipole(length=t(1.0), angle=t([1.0, -2.0]), tilt=t(0.42), **dk, **kw),
        "RBend_ParticleBeam_drift_kick_drift": ca.RBend(length=t(1.0), angle=t([1.0, -2.0]), tilt=t(0.42), **dk, **kw),
        "TransverseDeflectingCavity_ParticleBeam_active": ca.TransverseDeflectingCavity(length=t(1.0), voltage=t(1e6), **kw),
        "TransverseDeflectingCavity_ParticleBeam_inactive": ca.TransverseDeflectingCavity(length=t(1.0), voltage=t(0.0), **kw),
    }
    for name, el in elements.items():
        beam = ca.ParticleBeam(torch.tensor(c["incoming_particles_f32"], device="cuda").to(f64),
                               torch.tensor(c["incoming_energy"], device="cuda").reshape(()),
                               species=ca.Species("electron", **kw))
        out = el.track(beam)
        exp = g[f"{name}__particles"]
        got = out.particles.cpu().numpy()[..., :512, :]
        assert got.shape == exp.shape, name
        assert np.allclose(got, exp), name                       # the reference's own criterion
        assert np.allclose(got, exp, rtol=1e-9, atol=1e-13), name
        assert np.allclose(out.energy.cpu().numpy(), g[f"{name}__energy"]), name


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_hip_second_order_tensors_and_tracking(golden, oracle, tag):
    from cheetah_amd import _ops

    g = golden("second_order.npz")
    dt = torch.float64 if tag == "f64" else torch.float32
    x = _dev(g["incoming"] if tag == "f64" else g["incoming_f32"], dt)
    for name, kind, shp, par, En, m in _second_order_cases(g):
        T = _ops.build_ttensor(_ops.T_KIND[kind], _dev(par, dt), torch.Size((par.shape[0],)), _dev(En, dt), float(m))
        Tn = T.cpu().numpy().astype(np.float64)
        Tr = g[f"{name}_T"].reshape(Tn.shape)
        To = oracle.build_ttensor(kind, par, En, m)
        for b in range(Tn.shape[0]):
            assert _relerr(Tn[b], Tr[b]) < (1e-12 if tag == "f64" else 2e-6), (name, b)
            assert _relerr(Tn[b], To[b]) < (1e-12 if tag == "f64" else 2e-6), (name, b)
        out = _ops.apply_second_order(x, T).cpu().numpy()
        ora = oracle.apply_second_order(x.cpu().numpy(), Tn)
        ref = g[f"{name}_out" if tag == "f64" else f"{name}_out_f32"].reshape(out.shape)
        for j in range(6):
            # same T, same x: only the summation order differs from the oracle
            assert _relerr(out[..., j], ora[..., j]) < (1e-13 if tag == "f64" else 2e-6), (name, j)
            assert _relerr(out[..., j], ref[..., j]) < (1e-11 if tag == "f64" else 2e-5), (name, j)
        assert np.all(out[..., 6] == 1.0), name


@pytest.mark.gpu
def test_hip_second_order_consistency_goldens(golden):
    import cheetah_amd as ca

    g, c = golden("second_order.npz"), golden("consistency.npz")
    f64 = torch.float64
    kw = {"dtype": f64, "device": "cuda"}
    t = lambda v: torch.tensor(v, dtype=torch.float32).to(f64).cuda()  # noqa: E731
    so = {"tracking_method": "second_order"}
    elements = {
        "Drift_ParticleBeam_second_order": ca.Drift(length=t([1.0, -1.0]), **so, **kw),
        "Quadrupole_ParticleBeam_second_order": ca.Quadrupole(length=t(1.0), k1=t([1.0, -2.0]), tilt=t(0.42),
                                                              misalignment=t([0.01, -0.02]), **so, **kw),
        "Dipole_ParticleBeam_second_order": ca.Dipole(length=t(1.0), angle=t([1.0, -2.0]), tilt=t(0.42), **so, **kw),
        "RBend_ParticleBeam_second_order": ca.RBend(length=t(1.0), angle=t([1.0, -2.0]), tilt=t(0.42), **so, **kw),
        "Sextupole_ParticleBeam_second_order": ca.Sextupole(length=t(1.0), k2=t([1.0, -2.0]), tilt=t(0.42),
                                                            misalignment=t([0.01, -0.02]), **kw),
    }
    for name, el in elements.items():
        assert el.tracking_method == "second_order" and not el.is_skippable
        beam = ca.ParticleBeam(torch.tensor(c["incoming_particles_f32"], device="cuda").to(f64),
                               torch.tensor(c["incoming_energy"], device="cuda").reshape(()),
                               species=ca.Species("electron", **kw))
        out = ca.Segment([el]).track(beam)
        exp = g[f"{name}__particles"]
        got = out.particles.cpu().numpy()[..., :512, :]
        assert got.shape == exp.shape, name
        assert np.allclose(got, exp), name
        for j in range(6):
            # the Sextupole pickle predates a change of the reference itself (today's reference is 9e-8 / 1.2e-7
            # away from it in x / y); the others reproduce to the rounding of the stored parameters
            assert _relerr(got[..., j], exp[..., j]) < (2e-7 if name.startswith("Sextupole") else 1e-9), (name, j)


@pytest.mark.gpu
def test_nonlinear_entry_points_reject_bad_arguments():
    from cheetah_amd import _lib

    lib = _lib.lib()
    assert [lib.chx_dkd_num_params(k) for k in range(5)] == [1, 5, 9, 7, -1]
    assert [lib.chx_t_num_params(k) for k in range(6)] == [1, 5, 9, 5, 4, -1]
    x = torch.zeros(4, 7, device="cuda")
    p = torch.ones(1, 5, device="cuda")
    e = torch.full((1,), 1e8, device="cuda")
    args = (x.data_ptr(), p.data_ptr(), e.data_ptr(), 510998.95, -1.0)
    assert lib.chx_dkd_track(7, *args, 1, 3, 1, 1, 1, 1, 4, 0, x.data_ptr(), None, None) == -1      # kind
    assert lib.chx_dkd_track(1, *args, 0, 3, 1, 1, 1, 1, 4, 0, x.data_ptr(), None, None) == -1      # num_steps
    assert lib.chx_dkd_track(1, *args, 1, 3, 1, 1, 1, 1, 4, 5, x.data_ptr(), None, None) == -2      # dtype
    assert lib.chx_dkd_track(1, *args, 1, 3, 2, 3, 1, 1, 4, 0, x.data_ptr(), None, None) == -1      # broadcast
    assert lib.chx_apply_second_order(x.data_ptr(), None, x.data_ptr(), 1, 1, 1, 4, 0, None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_random_nonlinear_lattices_vs_reference(tag):
    """Ten drawn beamlines (tests/golden/generate_golden_random_nonlinear.py) in which every element takes a drawn tracking
    method out of the ones it supports — linear, second_order, drift_kick_drift with `num_steps`, fringe fields at either end —
    plus sextupoles and transverse deflecting cavities, rebuilt from JSON and tracked here against the reference (float64)."""
    import json
    import os

    import cheetah_amd as ca

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lattices_random_nonlinear.npz"))
    dt = torch.float64 if tag == "f64" else torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    for i in range(int(g["n_lattices"])):
        spec = json.loads(str(g[f"spec_{i}"]))
        elements = []
        for kind, args in spec:
            targs = {k: (torch.tensor(v, **kw) if isinstance(v, (float, list)) else v) for k, v in args.items()}
            elements.append(getattr(ca, kind)(**targs, **kw))
        beam = ca.ParticleBeam(torch.tensor(g[f"in_{i}"], **kw), torch.tensor(float(g[f"energy_{i}"]), **kw),
                               species=ca.Species("electron", **kw))
        out = ca.Segment(elements).track(beam)
        ref = g[f"out_{i}"]
        got = out.particles.cpu().numpy().astype(np.float64)
        scale = np.maximum(np.abs(ref).max(axis=0), 1e-30)
        err = (np.abs(got - ref) / scale).max()
        # fp64: the Bmad-X element maps are evaluated with series for the cancelling forms here (more accurate than the
        # reference's direct formulas, whose own tolerance against Bmad-X is 1e-14 absolute); fp32: fp64 arithmetic on
        # fp32-rounded coordinates and settings
        assert err < (1e-11 if tag == "f64" else 1e-3), (i, [(k, a.get("tracking_method")) for k, a in spec], err)   # measured: 1e-12 / 3e-4
        assert float(out.energy) == pytest.approx(float(g[f"energy_out_{i}"]), rel=1e-12 if tag == "f64" else 1e-6)


@pytest.mark.gpu
def test_vectorised_nonlinear_lattices_vs_reference():
    """Six drawn lines of lattices_random_nonlinear.npz with VECTORISED settings ((3,), (2, 1), (2,)) on elements of every
    tracking method and / or vectorised particles, for electrons, positrons and protons: the shape of the outgoing particles
    and every vector entry's coordinates against the reference in float64."""
    import json
    import os

    import cheetah_amd as ca

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lattices_random_nonlinear.npz"))
    kw = {"dtype": torch.float64, "device": "cuda"}
    for i in range(int(g["n_vectorised"])):
        spec = json.loads(str(g[f"v_spec_{i}"]))
        elements = []
        for kind, args in spec:
            targs = {k: (torch.tensor(v, **kw) if isinstance(v, (float, list)) else v) for k, v in args.items()}
            elements.append(getattr(ca, kind)(**targs, **kw))
        beam = ca.ParticleBeam(torch.tensor(g[f"v_in_{i}"], **kw), torch.tensor(float(g[f"v_energy_{i}"]), **kw),
                               species=ca.Species(str(g[f"v_species_{i}"]), **kw))
        out = ca.Segment(elements).track(beam)
        ref = g[f"v_out_{i}"]
        assert tuple(out.particles.shape) == ref.shape, (i, out.particles.shape, ref.shape)
        assert tuple(out.energy.shape) == g[f"v_energy_out_{i}"].shape, (i, out.energy.shape)
        got = out.particles.cpu().numpy()
        scale = np.maximum(np.abs(ref).reshape(-1, 7).max(axis=0), 1e-30)
        err = (np.abs(got - ref) / scale).max()
        assert err < 1e-11, (i, str(g[f"v_species_{i}"]), [(k, a.get("tracking_method")) for k, a in spec], err)
        assert np.allclose(out.energy.cpu().numpy(), g[f"v_energy_out_{i}"], rtol=1e-12)
