"""Degenerate inputs against the reference (tests/golden/edge_cases.npz, generator tests/golden/generate_golden_edge_cases.py):
one-particle and all-dead beams, zero-length elements, switched-off strengths, gamma -> 1, phase advances of 100 rad, a negative
drift length, a closed aperture, NaN / inf coordinates. Where the reference returns finite numbers this engine returns the same
numbers; where it returns NaN / inf (0 / 0 in the map of a zero-length dipole or cavity, a NaN coordinate spreading through the
map), the NaNs and the signed infinities sit in the same places. One case is only required to be non-finite in the same places:
at gamma = 1 exactly (beta = 0) the reference's matmul yields NaN where the fma chain here yields inf."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NON_FINITE_ONLY = {"gamma_exactly_one"}
STATS = ["mu_x", "sigma_x", "sigma_p", "emittance_x", "total_charge"]


def test_degenerate_inputs_vs_reference():
    import cheetah_amd as ca

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "edge_cases.npz"))
    kw = {"dtype": torch.float64, "device": "cuda"}
    report = []
    for name in [str(n) for n in g["names"]]:
        spec = json.loads(str(g[f"{name}_spec"]))
        elements = []
        for kind, args in spec:
            targs = {k: (torch.tensor(v, **kw) if isinstance(v, float) else v) for k, v in args.items()}
            elements.append(getattr(ca, kind)(**targs, **kw))
        beam = ca.ParticleBeam(torch.tensor(g[f"{name}_in"], **kw), torch.tensor(float(g[f"{name}_energy"]), **kw),
                               particle_charges=torch.tensor(g[f"{name}_charges"], **kw),
                               survival_probabilities=torch.tensor(g[f"{name}_survival"], **kw), species=ca.Species("electron", **kw))
        assert str(g[f"{name}_raises"]) == "", name            # the reference tracked every case of the fixture
        out = ca.Segment(elements).track(beam)
        got, ref = out.particles.cpu().numpy(), g[f"{name}_out"]
        assert got.shape == ref.shape, name
        same_nan = np.array_equal(np.isnan(got), np.isnan(ref))
        same_inf = np.array_equal(np.isinf(got), np.isinf(ref)) and np.array_equal(np.sign(got[np.isinf(got)]), np.sign(ref[np.isinf(ref)]))
        finite = np.isfinite(ref) & np.isfinite(got)
        scale = np.maximum(np.abs(np.where(np.isfinite(ref), ref, 0.0)).max(axis=0), 1e-300)
        err = (np.abs(np.where(finite, got - ref, 0.0)) / scale).max()
        report.append((name, same_nan, same_inf, err))
        if name in NON_FINITE_ONLY:
            same_nan = same_inf = np.array_equal(np.isfinite(got), np.isfinite(ref))
        assert same_nan and same_inf, (name, int(np.isnan(got).sum()), int(np.isnan(ref).sum()), int(np.isinf(got).sum()), int(np.isinf(ref).sum()))
        assert err < 1e-9, (name, err)
        assert np.array_equal(out.survival_probabilities.cpu().numpy(), g[f"{name}_out_survival"]), name
        e_ref = float(g[f"{name}_out_energy"])
        assert float(out.energy) == pytest.approx(e_ref, rel=1e-12), name
        for s in STATS:
            if f"{name}_{s}" not in g.files:
                continue
            if name == "hundred_radians" and s == "emittance_x":
                continue    # sqrt(s_xx s_pp - s_xp^2) with all three ~1e72 and a 1e-6 relative difference: rounding decides
            v, r = float(getattr(out, s)), float(g[f"{name}_{s}"])
            assert (np.isnan(v) and np.isnan(r)) or (np.isinf(v) and np.isinf(r) and np.sign(v) == np.sign(r)) \
                or v == pytest.approx(r, rel=1e-7, abs=1e-16 if s == "emittance_x" else 1e-300), (name, s, v, r)
            # (emittance: sqrt of a difference of products that is exactly zero for two particles — rounding residue 1e-23)
