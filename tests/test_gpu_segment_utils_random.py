"""The lattice utilities of Segment on DRAWN lattices against the reference (tests/golden/segment_utils_random.npz, generator
tests/golden/generate_golden_random_segment_utils.py): for eight named beamlines with nested Segments, Superimposed elements,
cavities and inactive diagnostics — flattened, reversed, the three `without_…` / `…_as_drifts` filters, merged consecutive
elements, merged transfer maps (with `except_for`), subcells between drawn names with the include flags, partitions, splits at a
drawn resolution, name look-ups, beam attributes along the lattice and attribute assignment by element type: the derived
lattice's element classes, names and lengths, and the beam tracked through it (reference segment.py:62-368, 584-730)."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KW = {"dtype": torch.float64, "device": "cuda"}


def build(ca, spec):
    kind, kw = spec
    kw = dict(kw)
    name = kw.pop("name")
    if kind == "Segment":
        return ca.Segment([build(ca, c) for c in kw["elements"]], name=name)
    if kind == "Superimposed":
        return ca.Superimposed(build(ca, kw["base_element"]), build(ca, kw["superimposed_element"]), name=name, **KW)
    args = {k: (torch.tensor(v, **KW) if isinstance(v, (float, list)) else v) for k, v in kw.items()}
    return getattr(ca, kind)(**args, name=name, **KW)


def same_lattice(g, key, seg, beam, names=True):
    classes = [type(e).__name__ for e in seg.elements]
    assert classes == [str(c) for c in g[f"{key}_classes"]], (key, classes, list(g[f"{key}_classes"]))
    if names:
        # elements the reference creates on the way get names from a process-wide counter ("unnamed_element_17"): not comparable
        for mine, ref in zip([e.name for e in seg.elements], [str(n) for n in g[f"{key}_names"]]):
            assert mine == ref or ref.startswith("unnamed_element"), (key, mine, ref)
    lengths = np.asarray([float(e.length) for e in seg.elements])
    assert np.allclose(lengths, g[f"{key}_lengths"], rtol=1e-13, atol=1e-15), key
    out = seg.track(beam)
    ref = g[f"{key}_out"]
    scale = np.maximum(np.abs(ref).max(axis=0), 1e-30)
    assert (np.abs(out.particles.cpu().numpy() - ref) / scale).max() < 1e-10, key
    assert float(out.energy) == pytest.approx(float(g[f"{key}_energy"]), rel=1e-13), key


def test_segment_utilities_on_drawn_lattices_vs_reference(golden):
    import cheetah_amd as ca

    g = golden("segment_utils_random.npz")
    for i in range(int(g["n_lattices"])):
        root = json.loads(str(g[f"spec_{i}"]))
        seg = build(ca, root)
        beam = ca.ParticleBeam(torch.tensor(g[f"in_{i}"], **KW), torch.tensor(float(g[f"energy_{i}"]), **KW),
                               particle_charges=torch.tensor(g[f"charges_{i}"], **KW), species=ca.Species("electron", **KW))
        k = f"l{i}"
        same_lattice(g, f"{k}_plain", seg, beam)
        flat = seg.flattened()
        same_lattice(g, f"{k}_flattened", flat, beam)
        same_lattice(g, f"{k}_reversed", seg.reversed(), beam)
        same_lattice(g, f"{k}_no_markers", seg.without_inactive_markers(), beam)
        same_lattice(g, f"{k}_no_zero_length", seg.without_inactive_zero_length_elements(), beam)
        same_lattice(g, f"{k}_as_drifts", seg.inactive_elements_as_drifts(), beam)
        same_lattice(g, f"{k}_flat_as_drifts", flat.inactive_elements_as_drifts(), beam)
        same_lattice(g, f"{k}_consecutive_merged", flat.with_consecutive_elements_merged(), beam)
        same_lattice(g, f"{k}_maps_merged", seg.transfer_maps_merged(incoming_beam=beam), beam)
        same_lattice(g, f"{k}_flat_maps_merged", flat.transfer_maps_merged(incoming_beam=beam), beam)
        keep = [str(n) for n in g[f"{k}_except"]]
        same_lattice(g, f"{k}_flat_maps_merged_except", flat.transfer_maps_merged(incoming_beam=beam, except_for=keep), beam)
        same_lattice(g, f"{k}_flat_no_zero_length_except", flat.without_inactive_zero_length_elements(except_for=keep), beam)
        for j, (start, end, inc_a, inc_b) in enumerate(json.loads(str(g[f"{k}_subcell_args"]))):
            same_lattice(g, f"{k}_subcell{j}", flat.subcell(start=start, end=end, include_start=inc_a, include_end=inc_b), beam)
        at = str(g[f"{k}_partition_at"])
        for mode in ("before", "after", "both"):
            parts = flat.partition_at(at, mode=mode)
            counts = [len(p.elements) if isinstance(p, ca.Segment) else -1 for p in parts]
            assert counts == [int(c) for c in g[f"{k}_partition_{mode}_counts"]], (k, mode)
            names = ["|".join(e.name for e in p.elements) if isinstance(p, ca.Segment) else p.name for p in parts]
            assert names == [str(n) for n in g[f"{k}_partition_{mode}_names"]], (k, mode)
        pieces = seg.split(resolution=torch.tensor(float(g[f"{k}_resolution"]), **KW))
        same_lattice(g, f"{k}_split", ca.Segment(pieces), beam)
        assert list(flat.element_names) == [str(n) for n in g[f"{k}_element_names"]]
        assert [flat.element_index(e.name) for e in flat.elements] == [int(v) for v in g[f"{k}_element_index"]]
        attrs = ("s", "mu_x", "sigma_y", "beta_x", "energy")
        for tag, r in (("", None), ("_res", float(g[f"{k}_resolution"]))):
            got = flat.get_beam_attrs_along_segment(attrs, beam, resolution=r)
            for a, v in zip(attrs, got):
                ref = g[f"{k}_along{tag}_{a}"]
                v = v.cpu().numpy()
                assert v.shape == ref.shape, (k, tag, a, v.shape, ref.shape)
                # beta_x = sigma_x^2 / emittance_x with emittance^2 = sigma_x^2 sigma_px^2 - cov_xpx^2: behind a long drift
                # (beta of kilometres) the difference cancels ~1e6-fold, and the 1e-14 of the one-pass moments shows up as
                # MEASURED 3.2e-8 at beta_x = 3684 m of lattice l3 (every other entry <= 1e-9): bound 4x for beta, 1e-8 otherwise
                assert np.allclose(v, ref, rtol=1.3e-7 if a == "beta_x" else 1e-8, atol=1e-9 * np.abs(ref).max()), (k, tag, a)
        seg2 = build(ca, root)
        seg2.set_attrs_on_every_element(ca.Quadrupole, k1=torch.tensor(1.25, **KW))
        seg2.set_attrs_on_every_element(ca.Drift, is_recursive=False, length=torch.tensor(0.123, **KW))
        same_lattice(g, f"{k}_set_attrs", seg2, beam)
        assert float(seg2.length) == pytest.approx(float(g[f"{k}_set_attrs_total_length"]), rel=1e-13)
