"""SCANS of lattice settings tracked with particles against the REFERENCE's own run (tests/golden/scan_stretch.npz, generator
tests/golden/generate_golden_scan_stretch.py): six drawn lattices with active monitors, apertures and cavities, about a third of the
quadrupole strengths and corrector angles (4,) tensors. Each is ONE stretch call here (chx_lattice_track_diag with Bm = 4 rows of
maps and one shared incoming beam; chx_parameter_lattice_track for the ParameterBeam): the (4, 1200, 7) outgoing particles, survival
probabilities and their SHAPE, energy, s and every monitor's reading with the reference's shape — (2,) in front of the first
vectorised element, (4, 2) behind; and both once more under a (4,) BEAM ENERGY (`lat*_escan_*`) and, for the lattices with cavities, under a phase scan of every cavity (`lat*_cscan_*`); and a 2-D grid scan by broadcasting (`lat*_gscan_*`: shapes (3, 1) and (1, 2)) (segment.py:545-574, bpm.py:77-87, aperture.py:90-135, cavity.py:100-251). Measured on the MI355X
(worst of the six lattices): float64 particles 5.8e-15, readings 6.8e-17, ParameterBeam 4.6e-16; float32 4.1e-7 / 1.3e-8 / 3.2e-7;
the bounds are those of test_gpu_diagnostics_stretch_golden.py."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scan_stretch.npz")


def _build(ca, spec, fk):
    kind, kw = spec
    args = {k: (torch.tensor(v, **fk) if isinstance(v, (float, list)) else v) for k, v in kw.items()}
    return getattr(ca, kind)(**args, **fk)


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_scans_with_particles_vs_reference(dt):
    import cheetah_amd as ca
    from cheetah_amd.accelerator import segment

    g = np.load(GOLDEN)
    fk = {"dtype": dt, "device": "cuda"}
    t = lambda a: torch.tensor(np.asarray(a), **fk)  # noqa: E731
    stretch_calls = []
    host = segment._lib.host()

    class Spy:
        def __getattr__(self, name):
            fn = getattr(host, name)
            return fn if name != "lattice_track" else (lambda *a: (stretch_calls.append(len(a)), fn(*a))[1])

    f64 = dt == torch.float64
    worst = {"particles": 0.0, "readings": 0.0, "pb": 0.0}
    old = segment._HOST
    segment._HOST = Spy()
    try:
        for i in range(int(g["n_lattices"])):
            specs = json.loads(str(g[f"lat{i}_spec"]))
            seg = ca.Segment([_build(ca, s, fk) for s in specs])
            bpms = [e for e in seg.elements if isinstance(e, ca.BPM)]
            assert len(bpms) == int(g[f"lat{i}_n_bpms"])
            beam = ca.ParticleBeam(t(g[f"lat{i}_in"]), t(g[f"lat{i}_energy"]), particle_charges=t(g[f"lat{i}_q"]),
                                   survival_probabilities=t(g[f"lat{i}_w"]), **fk)
            stretch_calls.clear()
            with torch.no_grad():
                out = seg.track(beam)
            assert stretch_calls == [21], (i, stretch_calls)          # the whole scan is one stretch call
            ref = g[f"lat{i}_out"]
            assert tuple(out.particles.shape) == ref.shape
            scale = np.abs(ref).max(axis=(0, 1))
            err = (np.abs(out.particles.double().cpu().numpy() - ref) / scale).max()
            worst["particles"] = max(worst["particles"], err)
            assert err < (1e-13 if f64 else 3e-6), (i, err)
            w_ref = g[f"lat{i}_w_out"]
            w_got = out.survival_probabilities.double().cpu().numpy()
            assert w_got.shape == w_ref.shape, (i, w_got.shape, w_ref.shape)
            if f64:
                assert np.array_equal(w_got, w_ref), (i, int((w_got != w_ref).sum()))
            else:       # a float32 coordinate may fall on the other side of an aperture edge
                assert (np.abs(w_got - w_ref) > 1e-6).sum() <= 8, (i, int((np.abs(w_got - w_ref) > 1e-6).sum()))
            assert float(out.energy) == pytest.approx(float(g[f"lat{i}_energy_out"]), rel=1e-13 if f64 else 1e-6)
            assert float(out.s) == pytest.approx(float(g[f"lat{i}_s_out"]), rel=1e-13 if f64 else 1e-6)
            size = np.abs(ref[..., [0, 2]]).max()
            for k, b in enumerate(bpms):
                r_ref = g[f"lat{i}_reading{k}"]
                r_got = b.reading.double().cpu().numpy()
                assert r_got.shape == r_ref.shape, (i, k, r_got.shape, r_ref.shape)
                live = np.isfinite(r_ref)
                assert np.array_equal(np.isfinite(r_got), live)
                if live.any():
                    e = np.abs(r_got[live] - r_ref[live]).max() / (size + np.abs(r_ref[live]).max())
                    worst["readings"] = max(worst["readings"], e)
                    assert e < (1e-15 if f64 else 3e-7), (i, k, e)
            # the same scan for a ParameterBeam
            pb = ca.ParameterBeam(t(g[f"lat{i}_pb_mu_in"]), t(g[f"lat{i}_pb_cov_in"]), t(g[f"lat{i}_energy"]), **fk)
            with warnings.catch_warnings(), torch.no_grad():
                warnings.simplefilter("ignore")
                pout = seg.track(pb)
            mu_ref, cov_ref = g[f"lat{i}_pb_mu"], g[f"lat{i}_pb_cov"]
            assert tuple(pout.mu.shape) == mu_ref.shape and tuple(pout.cov.shape) == cov_ref.shape
            e_mu = np.abs(pout.mu.double().cpu().numpy() - mu_ref).max() / np.abs(mu_ref[..., :6]).max()
            e_cov = np.abs(pout.cov.double().cpu().numpy() - cov_ref).max() / np.abs(cov_ref).max()
            worst["pb"] = max(worst["pb"], e_mu, e_cov)
            assert max(e_mu, e_cov) < (5e-15 if f64 else 1.5e-6), (i, e_mu, e_cov)
            for k, b in enumerate(bpms):
                r_ref = g[f"lat{i}_pb_reading{k}"]
                r_got = b.reading.double().cpu().numpy()
                assert r_got.shape == r_ref.shape, (i, k, r_got.shape, r_ref.shape)
                assert np.abs(r_got - r_ref).max() < (5e-15 if f64 else 1.5e-6) * (np.abs(mu_ref[..., :6]).max() + np.abs(r_ref).max())
            # a (4,) BEAM ENERGY on top of the vectorised settings: row b at energy b through the cavities
            escan = t(g[f"lat{i}_escan_energy"])
            ebeam = ca.ParticleBeam(t(g[f"lat{i}_in"]), escan, particle_charges=t(g[f"lat{i}_q"]), survival_probabilities=t(g[f"lat{i}_w"]), **fk)
            stretch_calls.clear()
            with torch.no_grad():
                eout = seg.track(ebeam)
            assert stretch_calls == [21], (i, stretch_calls)
            eref = g[f"lat{i}_escan_out"]                           # (the first SUB particles of every row are kept in the file)
            SUB = eref.shape[-2]
            assert tuple(eout.particles.shape) == (4, 1200, 7)
            err = (np.abs(eout.particles[..., :SUB, :].double().cpu().numpy() - eref) / np.abs(eref).max(axis=(0, 1))).max()
            worst["escan"] = max(worst.get("escan", 0.0), err)
            assert err < (1e-13 if f64 else 3e-6), (i, err)
            ew_ref, ew_got = g[f"lat{i}_escan_w_out"], eout.survival_probabilities[..., :SUB].double().cpu().numpy()
            assert ew_got.shape == ew_ref.shape and eout.survival_probabilities.shape[-1] == 1200
            if f64:
                assert np.array_equal(ew_got, ew_ref)
            e_ref = g[f"lat{i}_escan_energy_out"]
            assert tuple(eout.energy.shape) == e_ref.shape and np.allclose(eout.energy.double().cpu().numpy(), e_ref, rtol=1e-13 if f64 else 1e-6, atol=0)
            esize = np.abs(eref[..., [0, 2]]).max()
            for k, b in enumerate(bpms):
                r_ref, r_got = g[f"lat{i}_escan_reading{k}"], b.reading.double().cpu().numpy()
                assert r_got.shape == r_ref.shape, (i, k, r_got.shape, r_ref.shape)
                live = np.isfinite(r_ref)
                assert np.array_equal(np.isfinite(r_got), live)
                if live.any():
                    assert np.abs(r_got[live] - r_ref[live]).max() / (esize + np.abs(r_ref[live]).max()) < (1e-15 if f64 else 3e-7), (i, k)
            epb = ca.ParameterBeam(t(g[f"lat{i}_pb_mu_in"]), t(g[f"lat{i}_pb_cov_in"]), escan, **fk)
            with warnings.catch_warnings(), torch.no_grad():
                warnings.simplefilter("ignore")
                epout = seg.track(epb)
            emu, ecov = g[f"lat{i}_escan_pb_mu"], g[f"lat{i}_escan_pb_cov"]
            assert tuple(epout.mu.shape) == emu.shape and tuple(epout.cov.shape) == ecov.shape
            e1 = np.abs(epout.mu.double().cpu().numpy() - emu).max() / np.abs(emu[..., :6]).max()
            e2 = np.abs(epout.cov.double().cpu().numpy() - ecov).max() / np.abs(ecov).max()
            worst["escan_pb"] = max(worst.get("escan_pb", 0.0), e1, e2)
            assert max(e1, e2) < (5e-15 if f64 else 1.5e-6), (i, e1, e2)
            assert np.allclose(epout.energy.double().cpu().numpy(), g[f"lat{i}_escan_pb_energy"], rtol=1e-13 if f64 else 1e-6, atol=0)
            for k, b in enumerate(bpms):
                r_ref, r_got = g[f"lat{i}_escan_pb_reading{k}"], b.reading.double().cpu().numpy()
                assert r_got.shape == r_ref.shape, (i, k, r_got.shape, r_ref.shape)
                assert np.abs(r_got - r_ref).max() < (5e-15 if f64 else 1.5e-6) * (np.abs(emu[..., :6]).max() + np.abs(r_ref).max())
            # a PHASE scan of every cavity + a voltage scan of the first, some rows losing energy (cavity.py:157 over the batch)
            if f"lat{i}_cscan_spec" in g.files:
                cspecs = json.loads(str(g[f"lat{i}_cscan_spec"]))
                cseg = ca.Segment([_build(ca, sp, fk) for sp in cspecs])
                cbpms = [e for e in cseg.elements if isinstance(e, ca.BPM)]
                stretch_calls.clear()
                with torch.no_grad():
                    cout = cseg.track(beam)
                assert stretch_calls == [21], (i, stretch_calls)
                cref = g[f"lat{i}_cscan_out"]
                assert tuple(cout.particles.shape) == (4, 1200, 7)
                err = (np.abs(cout.particles[..., :SUB, :].double().cpu().numpy() - cref) / np.abs(cref).max(axis=(0, 1))).max()
                worst["cscan"] = max(worst.get("cscan", 0.0), err)
                assert err < (1e-13 if f64 else 3e-6), (i, err)
                ce_ref = g[f"lat{i}_cscan_energy_out"]
                assert tuple(cout.energy.shape) == ce_ref.shape == (4,)
                assert np.allclose(cout.energy.double().cpu().numpy(), ce_ref, rtol=1e-13 if f64 else 1e-6, atol=0)
                cw_ref, cw_got = g[f"lat{i}_cscan_w_out"], cout.survival_probabilities[..., :SUB].double().cpu().numpy()
                assert cw_got.shape == cw_ref.shape and (not f64 or np.array_equal(cw_got, cw_ref))
                csize = np.abs(cref[..., [0, 2]]).max()
                for k, b in enumerate(cbpms):
                    r_ref, r_got = g[f"lat{i}_cscan_reading{k}"], b.reading.double().cpu().numpy()
                    assert r_got.shape == r_ref.shape, (i, k, r_got.shape, r_ref.shape)
                    live = np.isfinite(r_ref)
                    assert np.array_equal(np.isfinite(r_got), live)
                    if live.any():
                        assert np.abs(r_got[live] - r_ref[live]).max() / (csize + np.abs(r_ref[live]).max()) < (1e-15 if f64 else 3e-7), (i, k)
                with warnings.catch_warnings(), torch.no_grad():
                    warnings.simplefilter("ignore")
                    cpout = cseg.track(pb)
                cmu, ccov = g[f"lat{i}_cscan_pb_mu"], g[f"lat{i}_cscan_pb_cov"]
                assert tuple(cpout.mu.shape) == cmu.shape and tuple(cpout.cov.shape) == ccov.shape
                e1 = np.abs(cpout.mu.double().cpu().numpy() - cmu).max() / np.abs(cmu[..., :6]).max()
                e2 = np.abs(cpout.cov.double().cpu().numpy() - ccov).max() / np.abs(ccov).max()
                worst["cscan_pb"] = max(worst.get("cscan_pb", 0.0), e1, e2)
                assert max(e1, e2) < (5e-15 if f64 else 1.5e-6), (i, e1, e2)
                assert np.allclose(cpout.energy.double().cpu().numpy(), g[f"lat{i}_cscan_pb_energy"], rtol=1e-13 if f64 else 1e-6, atol=0)
            # a 2-D GRID scan written by broadcasting: strengths of shape (3, 1) and (1, 2) (and a cavity phase of shape (1, 2))
            if f"lat{i}_gscan_spec" in g.files:
                gspecs = json.loads(str(g[f"lat{i}_gscan_spec"]))
                gseg = ca.Segment([_build(ca, sp, fk) for sp in gspecs])
                gbpms = [e for e in gseg.elements if isinstance(e, ca.BPM)]
                stretch_calls.clear()
                with torch.no_grad():
                    gout = gseg.track(beam)
                assert stretch_calls == [21], (i, stretch_calls)
                gref = g[f"lat{i}_gscan_out"]
                assert tuple(gout.particles.shape) == (3, 2, 1200, 7) and gref.shape == (3, 2, SUB, 7)
                err = (np.abs(gout.particles[..., :SUB, :].double().cpu().numpy() - gref) / np.abs(gref).max(axis=(0, 1, 2))).max()
                worst["gscan"] = max(worst.get("gscan", 0.0), err)
                assert err < (1e-13 if f64 else 3e-6), (i, err)
                ge_ref = g[f"lat{i}_gscan_energy_out"]
                assert tuple(gout.energy.shape) == ge_ref.shape, (i, gout.energy.shape, ge_ref.shape)
                assert np.allclose(gout.energy.double().cpu().numpy(), ge_ref, rtol=1e-13 if f64 else 1e-6, atol=0)
                gw_ref, gw_got = g[f"lat{i}_gscan_w_out"], gout.survival_probabilities[..., :SUB].double().cpu().numpy()
                assert gw_got.shape == gw_ref.shape and gout.survival_probabilities.shape[-1] == 1200, (i, gw_got.shape, gw_ref.shape)
                if f64:
                    assert np.array_equal(gw_got, gw_ref)
                gsize = np.abs(gref[..., [0, 2]]).max()
                for k, b in enumerate(gbpms):
                    r_ref, r_got = g[f"lat{i}_gscan_reading{k}"], b.reading.double().cpu().numpy()
                    assert r_got.shape == r_ref.shape, (i, k, r_got.shape, r_ref.shape)
                    live = np.isfinite(r_ref)
                    assert np.array_equal(np.isfinite(r_got), live)
                    if live.any():
                        assert np.abs(r_got[live] - r_ref[live]).max() / (gsize + np.abs(r_ref[live]).max()) < (1e-15 if f64 else 3e-7), (i, k)
                with warnings.catch_warnings(), torch.no_grad():
                    warnings.simplefilter("ignore")
                    gpout = gseg.track(pb)
                gmu, gcov = g[f"lat{i}_gscan_pb_mu"], g[f"lat{i}_gscan_pb_cov"]
                assert tuple(gpout.mu.shape) == gmu.shape and tuple(gpout.cov.shape) == gcov.shape
                e1 = np.abs(gpout.mu.double().cpu().numpy() - gmu).max() / np.abs(gmu[..., :6]).max()
                e2 = np.abs(gpout.cov.double().cpu().numpy() - gcov).max() / np.abs(gcov).max()
                worst["gscan_pb"] = max(worst.get("gscan_pb", 0.0), e1, e2)
                assert max(e1, e2) < (5e-15 if f64 else 1.5e-6), (i, e1, e2)
                assert tuple(gpout.energy.shape) == g[f"lat{i}_gscan_pb_energy"].shape
                for k, b in enumerate(gbpms):
                    r_ref, r_got = g[f"lat{i}_gscan_pb_reading{k}"], b.reading.double().cpu().numpy()
                    assert r_got.shape == r_ref.shape, (i, k, r_got.shape, r_ref.shape)
                    assert np.abs(r_got - r_ref).max() < (5e-15 if f64 else 1.5e-6) * (np.abs(gmu[..., :6]).max() + np.abs(r_ref).max())
    finally:
        segment._HOST = old
    print(f"grid scans vs reference ({dt}): worst particles {worst.get('gscan', 0):.2e}, parameter beam {worst.get('gscan_pb', 0):.2e}")
    print(f"cavity scans vs reference ({dt}): worst particles {worst.get('cscan', 0):.2e}, parameter beam {worst.get('cscan_pb', 0):.2e}")
    print(f"energy scans vs reference ({dt}): worst particles {worst.get('escan', 0):.2e}, parameter beam {worst.get('escan_pb', 0):.2e}")
    print(f"scans vs reference ({dt}): worst particles {worst['particles']:.2e}, readings {worst['readings']:.2e}, "
          f"parameter beam {worst['pb']:.2e}")
