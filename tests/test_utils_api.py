"""`cheetah_amd.utils`, `track_methods` and the small API aliases (mirror of cheetah.utils / cheetah.track_methods)."""
import types

import numpy as np
import pytest
import torch

F64 = {"dtype": torch.float64}


def test_top_level_names_of_the_reference_exist():
    import cheetah_amd as ca

    for name in ("Beam", "ParticleBeam", "ParameterBeam", "Species", "Segment", "Element", "converters", "latticejson",
                 "track_methods", "utils", "PhysicsWarning", "DirtyNameWarning", "VisualizationWarning"):
        assert hasattr(ca, name), name
    assert issubclass(ca.ParticleBeam, ca.Beam) and issubclass(ca.UnknownElementWarning, ca.PhysicsWarning)
    for name in ("compute_relativistic_factors", "unbiased_weighted_covariance", "unbiased_weighted_covariance_matrix",
                 "unbiased_weighted_std", "unbiased_weighted_variance", "match_distribution_moments", "elementwise_linspace",
                 "cloud_in_cell_charge_deposition", "merge_element_names", "UniqueNameGenerator",
                 "squash_index_for_unavailable_dims"):
        assert hasattr(ca.utils, name), name


def test_statistics_helpers_against_numpy():
    from cheetah_amd import utils

    rng = np.random.default_rng(0)
    x, y, w = rng.normal(size=(3, 500)), rng.normal(size=(3, 500)), rng.uniform(0.1, 1.0, size=(3, 500))
    tx, ty, tw = (torch.tensor(a) for a in (x, y, w))
    for b in range(3):
        ref = np.cov(np.stack([x[b], y[b]]), aweights=w[b])
        assert float(utils.unbiased_weighted_covariance(tx, ty, tw, dim=-1)[b]) == pytest.approx(ref[0, 1], rel=1e-12)
        assert float(utils.unbiased_weighted_variance(tx, tw, dim=-1)[b]) == pytest.approx(ref[0, 0], rel=1e-12)
        assert float(utils.unbiased_weighted_std(ty, tw, dim=-1)[b]) == pytest.approx(ref[1, 1] ** 0.5, rel=1e-12)
    samples = torch.tensor(rng.normal(size=(2, 400, 4)))
    weights = torch.tensor(rng.uniform(0.1, 1.0, size=(2, 400)))
    cov = utils.unbiased_weighted_covariance_matrix(samples, weights)
    for b in range(2):
        assert np.allclose(cov[b].numpy(), np.cov(samples[b].numpy().T, aweights=weights[b].numpy()), rtol=1e-11)
    target_mu = torch.tensor([1.0, -2.0, 0.5, 0.0], **F64)
    A = torch.tensor(rng.normal(size=(4, 4)))
    target_cov = A @ A.T + torch.eye(4, **F64)
    matched = utils.match_distribution_moments(samples, target_mu, target_cov, weights)
    total = weights.sum(-1, keepdim=True)
    assert torch.allclose((matched * weights.unsqueeze(-1)).sum(-2) / total, target_mu.expand(2, 4), atol=1e-12)
    assert torch.allclose(utils.unbiased_weighted_covariance_matrix(matched, weights), target_cov.expand(2, 4, 4), atol=1e-10)


def test_small_helpers():
    from cheetah_amd import utils

    g, ig2, beta = utils.compute_relativistic_factors(torch.tensor(1e8, **F64), torch.tensor(510998.95, **F64))
    assert float(g) == pytest.approx(195.6951, rel=1e-6) and float(ig2) == pytest.approx(1 / float(g) ** 2)
    assert float(beta) == pytest.approx((1 - float(ig2)) ** 0.5)
    ramp = utils.elementwise_linspace(torch.tensor([0.0, 1.0]), torch.tensor([1.0, 3.0]), 5)
    assert ramp.shape == (2, 5) and torch.allclose(ramp[1], torch.linspace(1.0, 3.0, 5))
    assert utils.squash_index_for_unavailable_dims((2, 3, 1), (4, 1)) == (3, 0)
    assert utils.squash_index_for_unavailable_dims((2,), ()) == ()
    gen = utils.UniqueNameGenerator("unnamed")
    assert (gen(), gen()) == ("unnamed_0", "unnamed_1")
    assert utils.merge_element_names("QUAD_A", "QUAD_B") == "QUAD_" and utils.merge_element_names("a", "b") == "a_b"


def test_beams_from_ocelot_particle_array_duck_typed():
    import cheetah_amd as ca

    rng = np.random.default_rng(1)
    parray = types.SimpleNamespace(rparticles=rng.normal(size=(6, 200)) * 1e-4, E=0.1, q_array=np.full(200, 1e-15))
    beam = ca.ParticleBeam.from_ocelot(parray, **F64)
    assert beam.particles.shape == (200, 7) and float(beam.energy) == pytest.approx(1e8)
    assert np.allclose(beam.particles[:, :6].numpy(), parray.rparticles.T) and torch.all(beam.particles[:, 6] == 1)
    pbeam = ca.ParameterBeam.from_ocelot(parray, **F64)
    assert np.allclose(pbeam.mu[:6].numpy(), parray.rparticles.mean(axis=1))
    assert np.allclose(pbeam.cov[:6, :6].numpy(), np.cov(parray.rparticles), rtol=1e-12)
    assert float(pbeam.total_charge) == pytest.approx(2e-13)


@pytest.mark.gpu
def test_track_methods_wrappers_and_aliases():
    import cheetah_amd as ca
    from cheetah_amd import track_methods as tmeth

    kw = {"dtype": torch.float64, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    sp = ca.Species("electron", **kw)
    E = t(1e8)
    drift = ca.Drift(t(0.7), **kw)
    assert torch.allclose(tmeth.drift_matrix(t(0.7), E, sp), drift.first_order_transfer_map(E, sp), rtol=1e-14)
    with pytest.warns(DeprecationWarning):
        assert torch.equal(drift.transfer_map(E, sp), drift.first_order_transfer_map(E, sp))
    quad = ca.Quadrupole(t(0.3), k1=t([2.0, -3.0]), **kw)
    R = tmeth.base_rmatrix(t(0.3), t([2.0, -3.0]), t(0.0), sp, E)
    assert R.shape == (2, 7, 7) and torch.allclose(R, quad.first_order_transfer_map(E, sp), rtol=1e-12, atol=1e-15)
    bend = ca.Dipole(t(0.5), angle=t(0.2), k1=t(0.4), **kw)
    Rb = tmeth.base_rmatrix(t(0.5), t(0.4), t(0.2 / 0.5), sp, E)
    assert torch.allclose(Rb, bend.first_order_transfer_map(E, sp), rtol=1e-12, atol=1e-15)
    T = tmeth.base_ttensor(t(0.3), t(2.0), t(0.0), t(0.0), sp, E)
    Tq = ca.Quadrupole(t(0.3), k1=t(2.0), tracking_method="second_order", **kw).second_order_transfer_map(E, sp)
    assert T.shape == (7, 7, 7) and torch.all(T[:, 6, :] == 0)
    mask = torch.ones(7, 7, 7, dtype=torch.bool, device="cuda")
    mask[:, 6, :] = False
    assert torch.allclose(T[mask], Tq[mask], rtol=1e-12, atol=1e-18)
    Ts = tmeth.base_ttensor(t(0.2), t(0.0), t(5.0), t(0.0), sp, E)
    assert float(Ts[1, 0, 0]) != 0.0
    Tg = tmeth.base_ttensor(t(0.2), t(1.0), t(5.0), t(0.3), sp, E)       # every strength at once: CHX_T_GENERAL
    assert Tg.shape == (7, 7, 7) and torch.isfinite(Tg).all() and float(Tg[0, 2, 2]) != 0.0
    rot = tmeth.rotation_matrix(t([0.3]))
    assert torch.allclose(rot[0] @ rot[0].mT, torch.eye(7, **kw), atol=1e-15)
    entry, exit_ = tmeth.misalignment_matrix(t([1e-3, -2e-3]))
    assert torch.allclose(entry @ exit_, torch.eye(7, **kw), atol=1e-18)
    c_entry, c_exit = tmeth.combined_rotation_misalignment_matrix(t(0.3), t([1e-3, -2e-3]))
    assert torch.allclose(c_exit @ c_entry, torch.eye(7, **kw), atol=1e-15)
    # the quadrupole's own dressing is exactly this pair
    qm = ca.Quadrupole(t(0.3), k1=t(2.0), tilt=t(0.3), misalignment=t([1e-3, -2e-3]), **kw)
    assert torch.allclose(c_exit @ tmeth.base_rmatrix(t(0.3), t(2.0), t(0.0), sp, E) @ c_entry,
                          qm.first_order_transfer_map(E, sp), rtol=1e-11, atol=1e-15)
    grid = ca.utils.cloud_in_cell_charge_deposition(t([[0.3, 0.4], [0.6, 0.7]]), (4, 4), t([[0.0, 1.0], [0.0, 1.0]]), t([1.0, 2.0]))
    assert grid.shape == (4, 4) and float(grid.sum()) == pytest.approx(3.0)
    pb = ca.ParameterBeam.from_parameters(sigma_x=t(1e-4), energy=E, **kw).linspaced(11)
    assert isinstance(pb, ca.ParticleBeam) and pb.num_particles == 11


def test_submodule_layout_of_the_reference():
    """Imports the reference's tests and user code make (tests/test_autograd.py:4, test_cloud_in_cell.py:5,
    test_compare_bmad.py:8, test_segment.py:7, test_infix.py:3, test_rpn.py:3) resolve after switching packages."""
    import importlib

    for mod, names in {
        "cheetah_amd.utils.autograd": ["log1pdiv", "si1mdiv", "sicos1mdiv", "sipsicos3mdiv", "sicoskuddelmuddel15mdiv",
                                       "cossqrtmcosdivdiff", "simsidivdiff", "si2msi2divdiff", "sqrta2minusbdiva"],
        "cheetah_amd.utils.bmadx": ["bmad_to_cheetah_z_pz", "cheetah_to_bmad_z_pz", "cheetah_to_bmad_coords",
                                    "bmad_to_cheetah_coords"],
        "cheetah_amd.utils.cloud_in_cell": ["cloud_in_cell_charge_deposition"],
        "cheetah_amd.utils.kde": ["kde_histogram_1d", "kde_histogram_2d"],
        "cheetah_amd.utils.statistics": ["unbiased_weighted_covariance", "unbiased_weighted_std", "match_distribution_moments"],
        "cheetah_amd.utils.warnings": ["PhysicsWarning", "DirtyNameWarning", "UnknownElementWarning"],
        "cheetah_amd.utils.physics": ["compute_relativistic_factors"],
        "cheetah_amd.utils.names": ["UniqueNameGenerator", "merge_element_names"],
        "cheetah_amd.utils.vector": ["squash_index_for_unavailable_dims"],
        "cheetah_amd.utils.elementwise_linspace": ["elementwise_linspace"],
        "cheetah_amd.converters.utils.infix": ["evaluate_expression"],
        "cheetah_amd.converters.utils.rpn": ["evaluate_expression"],
    }.items():
        m = importlib.import_module(mod)
        for n in names:
            assert hasattr(m, n), (mod, n)
    import cheetah_amd as ca

    assert ca.utils.warnings.PhysicsWarning is ca.warnings.PhysicsWarning is ca.utils.PhysicsWarning
    assert callable(ca.utils.elementwise_linspace) and callable(ca.utils.kde_histogram_2d)


def test_special_functions_refuse_host_tensors():
    """No CPU fallback: the element-wise special functions only run through chx_special."""
    from cheetah_amd.utils.autograd import log1pdiv

    with pytest.raises(Exception):
        log1pdiv(torch.tensor([0.5], dtype=torch.float64))


def test_infix_and_rpn_evaluators():
    """The cases of the reference's tests/test_infix.py and tests/test_rpn.py."""
    from cheetah_amd.converters.utils import infix, rpn

    assert infix.evaluate_expression("2 + 3") == 5
    assert infix.evaluate_expression("(10 * 2) + (4 ^ 2)") == 36
    assert infix.evaluate_expression("a + b", {"a": 10, "b": 5}) == 15
    for bad, ctx in (("2 +", None), ("a + b", {"a": 10})):
        with pytest.raises(SyntaxError):
            infix.evaluate_expression(bad, ctx)
    nested = {"a": 10, "b": {"beep": 10, "boop": 100, "test": 5}, "test": 3}
    assert rpn.evaluate_expression("2 3 +") == 5
    assert rpn.evaluate_expression("10 2 * 4 2 ^ + sqrt") == 6
    assert rpn.evaluate_expression("10 2 * 3 4 * + #should be valid") == 32
    assert rpn.evaluate_expression("10 2 * pi 4 * +", {"pi": 3}) == 32
    assert rpn.evaluate_expression("a b[test] - b[boop] *", nested) == 500
    assert rpn.evaluate_expression("b[test] b[boop] *", nested) == 500
    for bad in ("'2 3 +'", "ldsp2h +dldsp17h +lblxsph/2-lbxsph/2"):
        with pytest.raises(SyntaxError):
            rpn.evaluate_expression(bad)


def test_public_attribute_names_of_the_reference():
    """Every public attribute name of every class the reference exports (tests/golden/api_surface.json, written by
    generate_golden_api_surface.py from the imported reference) exists on the class of the same name here. Plotting and
    mesh methods exist as well, but only to say that drawing is outside this engine."""
    import json
    import os

    import cheetah_amd as ca

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "api_surface.json")) as f:
        surface = json.load(f)
    assert len(surface) >= 25
    for cls_name, names in surface.items():
        cls = getattr(ca, cls_name)
        missing = [n for n in names if not hasattr(cls, n)]
        assert not missing, (cls_name, missing)
    seg = ca.Segment([ca.Drift(torch.tensor(1.0))])
    for call in (seg.plot_overview, seg.elements[0].plot, seg.elements[0].to_mesh):
        with pytest.raises(NotImplementedError, match="outside this tracking engine"):
            call()


@pytest.mark.gpu
def test_general_ttensor_vs_reference_golden():
    """`track_methods.base_ttensor` with k2 together with k1 and / or hx (CHX_T_GENERAL) against the reference
    (/root/reference/cheetah/track_methods.py:80-281) on 48 drawn and special settings (tests/golden/ttensor_general.npz):
    tensors to 1e-11 of their scale, gradients of a fixed contraction with respect to length, k1, k2, hx and energy."""
    import os

    import numpy as np

    import cheetah_amd as ca
    from cheetah_amd import track_methods as tmeth

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ttensor_general.npz"))
    kw = {"dtype": torch.float64, "device": "cuda"}
    sp = ca.Species("electron", **kw)
    coef = torch.tensor(g["coef"], **kw)
    for row, T_ref, G_ref in zip(g["settings"], g["T"], g["grads"]):
        args = [torch.tensor(v, **kw).requires_grad_(True) for v in row]
        T = tmeth.base_ttensor(args[0], args[1], args[2], args[3], sp, args[4])
        T_ref = torch.tensor(T_ref, **kw)
        a = (row[1] + row[3] ** 2) * row[0] ** 2                 # kx^2 L^2
        if a != 0 and abs(a) < 1e-8:
            # j3 (track_methods.py:134-143) has no limit at kx2 -> 0: a rounding-sized kx2 gives ~1e32 in T[4,5,5] on both sides,
            # digits that mean nothing; everything else is compared, the gradients of this row are not
            keep = torch.ones_like(T_ref, dtype=torch.bool)
            keep[4, 5, 5] = False
            scale = float(T_ref[keep].abs().max())
            assert float((T - T_ref)[keep].abs().max()) <= 1e-6 * scale, row     # the reference's closed forms cancel here
            continue
        scale = float(T_ref.abs().max()) or 1.0
        assert float((T - T_ref).abs().max()) <= 1e-11 * scale, (row, float((T - T_ref).abs().max()) / scale)
        grads = torch.autograd.grad((T * coef).sum(), args)
        for name, got, want, x in zip(("L", "k1", "k2", "hx", "E"), grads, G_ref, row):
            # the contraction's sensitivities span many decades (energy in eV against strengths of order one): compare in
            # units of the largest relative sensitivity of the row
            if not np.isfinite(want):
                # the reference's own backward is NaN at its special points (e.g. d/dk1 at exactly k1 = 0 through
                # `sicos1mdiv`, utils/autograd.py); the dual-number builder gives a finite one-sided value there
                assert np.isfinite(float(got))
                continue
            unit = max(abs(w * v) for w, v in zip(G_ref, row) if np.isfinite(w)) or 1.0
            assert abs(float(got) - want) * abs(x) <= 2e-8 * unit + 1e-30, (row, name, float(got), want)
