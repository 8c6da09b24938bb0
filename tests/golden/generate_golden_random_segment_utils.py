#!/usr/bin/env python3
"""The reference's Segment utilities on drawn lattices -> tests/golden/segment_utils_random.npz: eight beamlines drawn like the
structured ones (nested Segments, Superimposed, cavities, diagnostics; every element NAMED e0, e1, … depth first so that both
sides can address them), and for each what the reference makes of
  flattened / reversed / without_inactive_markers / without_inactive_zero_length_elements / inactive_elements_as_drifts /
  with_consecutive_elements_merged / transfer_maps_merged (with and without except_for) / subcell (4 drawn name pairs and the
  include_* flags) / partition_at (3 modes) / split(resolution) / element_index / element_names /
  get_beam_attrs_along_segment (with and without resolution) / set_attrs_on_every_element (by type, recursive and not):
element class names, element names, lengths and the tracked beam (64 particles, float64) behind each derived lattice.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_random_segment_utils.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cheetah  # noqa: E402
import generate_golden_random_structured as S  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
f64 = {"dtype": torch.float64}
S.rng = np.random.default_rng(5150)
rng = S.rng


def name_all(spec, counter):
    """depth first: children before their parent, like construction order"""
    kind, kw = spec
    if kind == "Segment":
        for c in kw["elements"]:
            name_all(c, counter)
    if kind == "Superimposed":
        name_all(kw["base_element"], counter)
        name_all(kw["superimposed_element"], counter)
    kw["name"] = f"e{counter[0]}"
    counter[0] += 1


def build(module, spec):
    kind, kw = spec
    kw = dict(kw)
    name = kw.pop("name")
    if kind == "Segment":
        return module.Segment([build(module, c) for c in kw["elements"]], name=name)
    if kind == "Superimposed":
        return module.Superimposed(build(module, kw["base_element"]), build(module, kw["superimposed_element"]), name=name, **f64)
    args = {k: (torch.tensor(v, **f64) if isinstance(v, (float, list)) else v) for k, v in kw.items()}
    return getattr(module, kind)(**args, name=name, **f64)


def describe(arrays, key, seg, beam):
    """classes, names, lengths of the top-level elements and the tracked particles"""
    arrays[f"{key}_classes"] = np.asarray([type(e).__name__ for e in seg.elements])
    arrays[f"{key}_names"] = np.asarray([e.name for e in seg.elements])
    arrays[f"{key}_lengths"] = np.asarray([float(e.length) for e in seg.elements])
    out = seg.track(beam)
    arrays[f"{key}_out"] = out.particles.numpy()
    arrays[f"{key}_energy"] = out.energy.numpy()


if __name__ == "__main__":
    arrays = {"n_lattices": np.asarray(8)}
    for i in range(8):
        top = []
        for _ in range(int(rng.integers(5, 10))):
            r = rng.random()
            if r < 0.3:
                top.append(S.draw_segment(1))
            elif r < 0.45:
                top.append(S.draw_superimposed())
            elif r < 0.55:
                top.append([str(rng.choice(["Marker", "BPM", "Screen"])), {}])   # inactive diagnostics to remove / replace
            elif r < 0.62:
                top.append(["Aperture", {"x_max": S.u(1e-3, 5e-3), "y_max": S.u(1e-3, 5e-3), "is_active": bool(rng.random() < 0.5)}])
            else:
                top.append(S.draw_simple())
        root = ["Segment", {"elements": top}]
        name_all(root, [0])
        seg = build(cheetah, root)
        energy = float(np.exp(rng.uniform(np.log(3e7), np.log(3e9))))
        torch.manual_seed(4000 + i)
        beam = cheetah.ParticleBeam.from_parameters(num_particles=64, sigma_x=torch.tensor(2e-4, **f64), sigma_y=torch.tensor(2e-4, **f64),
                                                    sigma_px=torch.tensor(5e-5, **f64), sigma_py=torch.tensor(5e-5, **f64),
                                                    sigma_tau=torch.tensor(1e-4, **f64), sigma_p=torch.tensor(1e-3, **f64),
                                                    energy=torch.tensor(energy, **f64), **f64)
        arrays[f"spec_{i}"] = np.asarray(json.dumps(root))
        arrays[f"energy_{i}"] = np.asarray(energy)
        arrays[f"in_{i}"] = beam.particles.numpy()
        arrays[f"charges_{i}"] = beam.particle_charges.numpy()
        k = f"l{i}"
        describe(arrays, f"{k}_plain", seg, beam)
        flat = seg.flattened()
        describe(arrays, f"{k}_flattened", flat, beam)
        describe(arrays, f"{k}_reversed", seg.reversed(), beam)
        describe(arrays, f"{k}_no_markers", seg.without_inactive_markers(), beam)
        describe(arrays, f"{k}_no_zero_length", seg.without_inactive_zero_length_elements(), beam)
        describe(arrays, f"{k}_as_drifts", seg.inactive_elements_as_drifts(), beam)
        describe(arrays, f"{k}_flat_as_drifts", flat.inactive_elements_as_drifts(), beam)
        describe(arrays, f"{k}_consecutive_merged", flat.with_consecutive_elements_merged(), beam)
        describe(arrays, f"{k}_maps_merged", seg.transfer_maps_merged(incoming_beam=beam), beam)
        describe(arrays, f"{k}_flat_maps_merged", flat.transfer_maps_merged(incoming_beam=beam), beam)
        flat_names = [e.name for e in flat.elements]
        keep = [str(n) for n in rng.choice(flat_names, size=min(2, len(flat_names)), replace=False)]
        arrays[f"{k}_except"] = np.asarray(keep)
        describe(arrays, f"{k}_flat_maps_merged_except", flat.transfer_maps_merged(incoming_beam=beam, except_for=keep), beam)
        describe(arrays, f"{k}_flat_no_zero_length_except", flat.without_inactive_zero_length_elements(except_for=keep), beam)
        # subcells of the flattened lattice between drawn names
        pairs = []
        for j in range(4):
            a, b = sorted(int(v) for v in rng.integers(0, len(flat_names), 2))
            inc_a, inc_b = bool(rng.random() < 0.6), bool(rng.random() < 0.6)
            start = flat_names[a] if rng.random() < 0.85 else None
            end = flat_names[b] if rng.random() < 0.85 else None
            pairs.append([start, end, inc_a, inc_b])
            describe(arrays, f"{k}_subcell{j}", flat.subcell(start=start, end=end, include_start=inc_a, include_end=inc_b), beam)
        arrays[f"{k}_subcell_args"] = np.asarray(json.dumps(pairs))
        at = flat_names[int(rng.integers(0, len(flat_names)))]
        arrays[f"{k}_partition_at"] = np.asarray(at)
        for mode in ("before", "after", "both"):
            parts = flat.partition_at(at, mode=mode)
            arrays[f"{k}_partition_{mode}_counts"] = np.asarray([len(p.elements) if isinstance(p, cheetah.Segment) else -1 for p in parts])
            arrays[f"{k}_partition_{mode}_names"] = np.asarray(["|".join(e.name for e in p.elements) if isinstance(p, cheetah.Segment)
                                                                  else p.name for p in parts])
        res = float(rng.uniform(0.05, 0.3))
        arrays[f"{k}_resolution"] = np.asarray(res)
        pieces = seg.split(resolution=torch.tensor(res, **f64))
        describe(arrays, f"{k}_split", cheetah.Segment(pieces), beam)
        arrays[f"{k}_element_names"] = np.asarray(list(flat.element_names))
        arrays[f"{k}_element_index"] = np.asarray([flat.element_index(n) for n in flat_names])
        attrs = ("s", "mu_x", "sigma_y", "beta_x", "energy")
        for tag, r in (("", None), ("_res", res)):
            got = flat.get_beam_attrs_along_segment(attrs, beam, resolution=r)
            for a, v in zip(attrs, got):
                arrays[f"{k}_along{tag}_{a}"] = v.numpy()
        # set attributes by type: every Quadrupole's k1 and tracking method (recursive), then only the top level's drifts
        seg2 = build(cheetah, root)
        seg2.set_attrs_on_every_element(cheetah.Quadrupole, k1=torch.tensor(1.25, **f64))
        seg2.set_attrs_on_every_element(cheetah.Drift, is_recursive=False, length=torch.tensor(0.123, **f64))
        describe(arrays, f"{k}_set_attrs", seg2, beam)
        arrays[f"{k}_set_attrs_total_length"] = seg2.length.numpy()
        print(i, len(top), len(flat_names), "split into", len(pieces), "at", res)
    np.savez_compressed(os.path.join(OUT, "segment_utils_random.npz"), **arrays)
    print("wrote", len(arrays), "arrays")
