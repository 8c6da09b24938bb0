"""Lattices with ACTIVE beam position monitors, apertures and cavities between the magnets against the REFERENCE's own run
(tests/golden/diagnostics_stretch.npz, generator tests/golden/generate_golden_diagnostics_stretch.py): eight drawn lattices, each
tracked here as ONE stretch call (chx_lattice_track_diag / chx_parameter_lattice_track) — outgoing particles, survival
probabilities, energy, path length and every monitor's reading (bpm.py:77-87, aperture.py:90-135, cavity.py:100-251,
segment.py:545-574); a ParameterBeam; three beams in one ParticleBeam. Measured on the MI355X (worst of the eight lattices, relative to a coordinate's scale / the beam size): float64 particles 2.4e-14,
readings 1.9e-16, ParameterBeam moments 1.2e-15; float32 7.3e-7 / 7.2e-8 / 3.4e-7 — the bounds below are 4 x those; float64 survival
probabilities behind the apertures: identical."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "diagnostics_stretch.npz")


def _build(ca, spec, fk):
    kind, kw = spec
    args = {k: (torch.tensor(v, **fk) if isinstance(v, (float, list)) else v) for k, v in kw.items()}
    return getattr(ca, kind)(**args, **fk)


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_diagnostics_lattices_vs_reference(dt):
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

    worst = {"particles": 0.0, "readings": 0.0, "pb": 0.0}
    old = segment._HOST
    segment._HOST = Spy()
    try:
        for i in range(int(g["n_lattices"])):
            specs = json.loads(str(g[f"lat{i}_spec"]))
            seg = ca.Segment([_build(ca, s, fk) for s in specs])
            bpms = [e for e in seg.elements if isinstance(e, ca.BPM)]
            beam = ca.ParticleBeam(t(g[f"lat{i}_in"]), t(g[f"lat{i}_energy"]), particle_charges=t(g[f"lat{i}_q"]),
                                   survival_probabilities=t(g[f"lat{i}_w"]), **fk)
            stretch_calls.clear()
            with torch.no_grad():
                out = seg.track(beam)
            assert len(stretch_calls) == 1, (i, stretch_calls)          # the whole lattice is one stretch
            ref = g[f"lat{i}_out"]
            scale = np.abs(ref).max(axis=0)
            err = (np.abs(out.particles.double().cpu().numpy() - ref) / scale).max()
            worst["particles"] = max(worst["particles"], err)
            assert err < (1e-13 if dt == torch.float64 else 3e-6), (i, err)
            w_ref = g[f"lat{i}_w_out"]
            w_got = out.survival_probabilities.double().cpu().numpy()
            if dt == torch.float64:
                assert np.array_equal(w_got, w_ref), (i, int((w_got != w_ref).sum()))
            else:       # a float32 coordinate may fall on the other side of an aperture edge
                assert (np.abs(w_got - w_ref) > 1e-6).sum() <= 4, (i, int((np.abs(w_got - w_ref) > 1e-6).sum()))
            assert float(out.energy) == pytest.approx(float(g[f"lat{i}_energy_out"]), rel=1e-13 if dt == torch.float64 else 1e-6)
            assert float(out.s) == pytest.approx(float(g[f"lat{i}_s_out"]), rel=1e-13 if dt == torch.float64 else 1e-6)
            got_r = torch.stack([b.reading for b in bpms]).double().cpu().numpy()
            ref_r = g[f"lat{i}_readings"]
            live = np.isfinite(ref_r).all(axis=1)                          # (every particle lost in front of a monitor: 0 / 0)
            size = np.abs(ref[:, [0, 2]]).max() + np.abs(ref_r[live]).max()
            err_r = np.abs(got_r[live] - ref_r[live]).max() / size
            worst["readings"] = max(worst["readings"], err_r)
            assert err_r < (1e-15 if dt == torch.float64 else 3e-7), (i, err_r)
            assert np.array_equal(np.isfinite(got_r).all(axis=1), live)
            # ParameterBeam
            pb = ca.ParameterBeam(t(g[f"lat{i}_pb_mu_in"]), t(g[f"lat{i}_pb_cov_in"]), t(g[f"lat{i}_energy"]), **fk)
            with warnings.catch_warnings(), torch.no_grad():
                warnings.simplefilter("ignore")
                pout = seg.track(pb)
            mu_ref, cov_ref = g[f"lat{i}_pb_mu"], g[f"lat{i}_pb_cov"]
            e_mu = np.abs(pout.mu.double().cpu().numpy() - mu_ref).max() / np.abs(mu_ref[:6]).max()
            e_cov = np.abs(pout.cov.double().cpu().numpy() - cov_ref).max() / np.abs(cov_ref).max()
            worst["pb"] = max(worst["pb"], e_mu, e_cov)
            assert max(e_mu, e_cov) < (5e-15 if dt == torch.float64 else 1.5e-6), (i, e_mu, e_cov)
            pr = torch.stack([b.reading for b in bpms]).double().cpu().numpy()
            assert np.abs(pr - g[f"lat{i}_pb_readings"]).max() < (5e-15 if dt == torch.float64 else 1.5e-6) * (np.abs(mu_ref[:6]).max() + np.abs(g[f"lat{i}_pb_readings"]).max())
            assert float(pout.energy) == pytest.approx(float(g[f"lat{i}_pb_energy"]), rel=1e-13 if dt == torch.float64 else 1e-6)
            # three beams in one ParticleBeam
            if f"lat{i}_many_out" in g.files:
                parts = t(g[f"lat{i}_in"]).unsqueeze(0) * torch.tensor([0.6, 1.0, 1.7], **fk).reshape(3, 1, 1)
                parts[..., 6] = 1.0
                many = ca.ParticleBeam(parts.contiguous(), t(g[f"lat{i}_energy"]), particle_charges=t(g[f"lat{i}_q"]),
                                       survival_probabilities=t(g[f"lat{i}_w"]), **fk)
                stretch_calls.clear()
                with torch.no_grad():
                    mout = seg.track(many)
                assert len(stretch_calls) == 1
                mref = g[f"lat{i}_many_out"]
                merr = (np.abs(mout.particles.double().cpu().numpy() - mref) / np.abs(mref).max(axis=(0, 1))).max()
                assert merr < (1e-13 if dt == torch.float64 else 3e-6), (i, merr)
                mw = torch.broadcast_to(mout.survival_probabilities, (3, 1500)).double().cpu().numpy()
                if dt == torch.float64:
                    assert np.array_equal(mw, g[f"lat{i}_many_w_out"])
                mr = torch.stack([b.reading for b in bpms]).double().cpu().numpy()
                rr = g[f"lat{i}_many_readings"]
                ok = np.isfinite(rr)
                assert mr.shape == rr.shape and np.array_equal(np.isfinite(mr), ok)
                assert np.abs(mr[ok] - rr[ok]).max() < (1e-15 if dt == torch.float64 else 3e-7) * (np.abs(mref[..., [0, 2]]).max() + np.abs(rr[ok]).max())
    finally:
        segment._HOST = old
    print(f"diagnostics lattices vs reference ({dt}): worst particles {worst['particles']:.2e}, readings {worst['readings']:.2e}, "
          f"parameter beam {worst['pb']:.2e}")
