"""Aperture mask kernel, Superimposed and the lattice utilities of Segment
(split / merge / inactive_elements_as_drifts / subcell / ... , segment.py:73-367, 584-724).
CPU: the numpy oracle of the mask vs the reference's survival probabilities (bit-exact, incl. boundary particles).
GPU: chx_aperture_mask vs the same goldens, and the edited lattices tracked through the HIP path."""
import numpy as np
import pytest
import torch


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("shape", ["rectangular", "elliptical"])
def test_oracle_aperture_mask_bit_exact(golden, oracle, tag, shape):
    g = golden("aperture_lattice.npz")
    x, s = g[f"x_{tag}"], g[f"surv_{tag}"]
    xm, ym = g[f"limits_{tag}"].astype(x.dtype)
    assert np.array_equal(oracle.aperture_mask(x, s, xm, ym, shape), g[f"{shape}_{tag}"])
    xv = np.array([2.5e-4, 1e-4, np.inf], dtype=x.dtype)
    yv = np.array([[1.5e-4], [3e-4]], dtype=x.dtype)
    assert np.array_equal(oracle.aperture_mask(x, s, xv, yv, shape), g[f"{shape}_vec_{tag}"])
    lost = g[f"{shape}_{tag}"] == 0
    assert 0 < lost.sum() < lost.size


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("shape", ["rectangular", "elliptical"])
def test_hip_aperture_mask_bit_exact(golden, tag, shape):
    import cheetah_amd as ca

    g = golden("aperture_lattice.npz")
    dt = torch.float32 if tag == "f32" else torch.float64
    kw = {"dtype": dt, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    beam = ca.ParticleBeam(t(g[f"x_{tag}"]), t(1e8), survival_probabilities=t(g[f"surv_{tag}"]), species=ca.Species("electron", **kw))
    xm, ym = g[f"limits_{tag}"]
    out = ca.Aperture(x_max=t(xm), y_max=t(ym), shape=shape, **kw).track(beam)
    assert np.array_equal(out.survival_probabilities.cpu().numpy(), g[f"{shape}_{tag}"])
    assert out.particles.data_ptr() == beam.particles.data_ptr()      # particles untouched, not copied
    outv = ca.Aperture(x_max=t([2.5e-4, 1e-4, float("inf")]), y_max=t([[1.5e-4], [3e-4]]), shape=shape, **kw).track(beam)
    assert np.array_equal(outv.survival_probabilities.cpu().numpy(), g[f"{shape}_vec_{tag}"])
    off = ca.Aperture(x_max=t(xm), y_max=t(ym), shape=shape, is_active=False, **kw)
    assert off.is_skippable and torch.equal(off.track(beam).survival_probabilities, beam.survival_probabilities)
    with pytest.raises(AssertionError):
        ca.Aperture(x_max=t(xm), y_max=t(ym), shape="triangular", **kw).track(beam)


def _lattice(ca, kw):
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    return ca.Segment([
        ca.Drift(t(0.3), name="d1", **kw), ca.Drift(t(0.2), name="d2", **kw),
        ca.Quadrupole(t(0.1), k1=t(4.2), name="q1a", **kw), ca.Quadrupole(t(0.15), k1=t(-2.0), name="q1b", **kw),
        ca.Marker(name="m", **kw),
        ca.HorizontalCorrector(t(0.05), angle=t(0.0), name="hc", **kw),
        ca.Solenoid(t(0.2), k=t(0.8), name="s1", **kw), ca.Solenoid(t(0.1), k=t(0.4), name="s2", **kw),
        ca.Superimposed(ca.Quadrupole(t(0.4), k1=t(1.1), name="qs", **kw),
                        ca.Aperture(x_max=t(1e-4), y_max=t(2e-4), name="ap", **kw), name="sup", **kw),
        ca.Drift(t(0.5), name="d3", **kw),
    ])


@pytest.mark.gpu
def test_lattice_utilities_match_reference(golden):
    import cheetah_amd as ca

    g = golden("aperture_lattice.npz")
    kw = {"dtype": torch.float64, "device": "cuda"}
    seg = _lattice(ca, kw)
    m, nq = g["lat_species"]
    sp = ca.Species("custom", num_elementary_charges=torch.tensor(float(nq), **kw), mass_eV=torch.tensor(float(m), **kw), **kw)
    beam = ca.ParticleBeam(torch.tensor(g["lat_in"], **kw), torch.tensor(g["lat_energy"], **kw), species=sp)

    def close(a, b):
        return np.allclose(a.cpu().numpy(), b, rtol=1e-11, atol=1e-15)

    out = seg.track(beam)
    assert close(out.particles, g["lat_out"]) and np.array_equal(out.survival_probabilities.cpu().numpy(), g["lat_out_survival"])
    assert 0 < (g["lat_out_survival"] == 0).sum() < 512
    merged = seg.with_consecutive_elements_merged()
    assert merged.element_names == list(g["lat_merged_names"])
    assert np.allclose([float(e.length) for e in merged.elements], g["lat_merged_lengths"], rtol=1e-15)
    assert float(merged.elements[1].k1) == pytest.approx(float(g["lat_merged_q_k1"]), rel=1e-15)
    assert float(merged.elements[4].k) == pytest.approx(float(g["lat_merged_s_k"]), rel=1e-15)
    assert close(merged.track(beam).particles, g["lat_merged_out"])
    drifts = seg.inactive_elements_as_drifts()
    assert [type(e).__name__ for e in drifts.elements] == list(g["lat_drifts_types"])
    assert close(drifts.track(beam).particles, g["lat_drifts_out"])
    split = ca.Segment(seg.split(torch.tensor(0.12, **kw)))
    assert split.element_names == list(g["lat_split_names"])
    assert close(split.track(beam).particles, g["lat_split_out"])
    assert np.allclose(seg.get_beam_attrs_along_segment("sigma_x", beam).cpu().numpy(), g["lat_sigma_x_along"], rtol=1e-10)
    mu_x, s = seg.get_beam_attrs_along_segment(("mu_x", "s"), beam, resolution=0.25)
    assert np.allclose(mu_x.cpu().numpy(), g["lat_mu_x_res"], rtol=1e-9, atol=1e-16) and np.allclose(s.cpu().numpy(), g["lat_s_res"], rtol=1e-14)
    assert seg.subcell("q1a", "s1", include_end=False).element_names == list(g["lat_subcell_names"])
    assert seg.without_inactive_markers().element_names == list(g["lat_nomarkers_names"])
    assert seg.without_inactive_zero_length_elements().element_names == list(g["lat_nozero_names"])
    assert seg.reversed().element_names == list(g["lat_reversed_names"])
    pre, el, post = seg.partition_at("hc")
    assert pre.element_names == ["d1", "d2", "q1a", "q1b", "m"] and el.name == "hc" and post.element_names[0] == "s1"
    assert seg.element_index("m") == 4
    with pytest.raises(ValueError):
        seg.element_index("nope")
    with pytest.raises(ValueError):
        seg.subcell("nope")
    seg.set_attrs_on_every_element(filter_type=ca.Drift, tracking_method="second_order")
    assert all(e.tracking_method == "second_order" for e in seg.elements if isinstance(e, ca.Drift))
