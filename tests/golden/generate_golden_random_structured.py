#!/usr/bin/env python3
"""Random STRUCTURED lattices tracked by the reference -> tests/golden/lattices_random_structured.npz: ten drawn beamlines whose
element lists contain nested Segments (two levels), Superimposed elements (a zero-length marker / BPM / corrector kick in the
middle of a drift or quadrupole), Undulators, CustomTransferMaps, linear Sextupoles, active cavities and one small
SpaceChargeKick followed by a run of linear elements — for electrons, positrons, protons and a custom ion. Stored per lattice:
the recursive element list as JSON, the species, 128 incoming particles and what `Segment.track` makes of them in float64
(particles, survival probabilities, energy, s), plus mu / cov of a ParameterBeam for the lattices without space charge, and
the LatticeJSON text the reference writes for each lattice with the reference's tracking result after reading it back.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_random_structured.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import cheetah  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
f64 = {"dtype": torch.float64}
rng = np.random.default_rng(20260929)


def u(lo, hi):
    return float(rng.uniform(lo, hi))


def draw_simple():
    kind = str(rng.choice(["Drift", "Quadrupole", "Dipole", "HorizontalCorrector", "VerticalCorrector", "Solenoid", "Undulator",
                           "Sextupole", "CustomTransferMap", "Cavity", "Marker", "BPM"]))
    if kind == "Drift":
        return [kind, {"length": u(0.05, 1.5)}]
    if kind == "Quadrupole":
        kw = {"length": u(0.05, 0.5), "k1": u(-20.0, 20.0)}
        if rng.random() < 0.3:
            kw["tilt"] = u(-0.4, 0.4)
        return [kind, kw]
    if kind == "Dipole":
        kw = {"length": u(0.2, 1.2), "angle": u(-0.3, 0.3)}
        if rng.random() < 0.5:
            kw.update({"dipole_e1": u(-0.15, 0.15), "dipole_e2": u(-0.15, 0.15), "fringe_integral": u(0.2, 0.6), "gap": u(0.01, 0.04)})
        return [kind, kw]
    if kind in ("HorizontalCorrector", "VerticalCorrector"):
        return [kind, {"length": u(0.01, 0.2), "angle": u(-4e-4, 4e-4)}]
    if kind == "Solenoid":
        return [kind, {"length": u(0.05, 0.4), "k": u(-2.5, 2.5)}]
    if kind == "Undulator":
        return [kind, {"length": u(0.2, 2.0)}]
    if kind == "Sextupole":
        return [kind, {"length": u(0.05, 0.3), "k2": u(-40.0, 40.0), "tracking_method": "linear"}]
    if kind == "CustomTransferMap":
        # a thin rotation-and-shear in x/y plus an offset column: not symplectic on purpose (the element takes any 7x7)
        m = np.eye(7)
        a = u(-0.3, 0.3)
        m[0, 0], m[0, 1], m[1, 0], m[1, 1] = np.cos(a), u(0.1, 1.5), u(-0.5, 0.5), np.cos(a)
        m[2, 3], m[3, 2] = u(0.1, 1.5), u(-0.5, 0.5)
        m[4, 5] = u(-1e-3, 1e-3)
        m[0, 6], m[2, 6] = u(-1e-4, 1e-4), u(-1e-4, 1e-4)
        return [kind, {"predefined_transfer_map": m.tolist(), "length": u(0.0, 0.8)}]
    if kind == "Cavity":
        return [kind, {"length": u(0.3, 1.1), "voltage": u(2e6, 1.5e7), "phase": u(-40.0, 40.0), "frequency": 1.3e9,
                       "cavity_type": str(rng.choice(["standing_wave", "traveling_wave"]))}]
    return [kind, {}]


def draw_superimposed():
    base = str(rng.choice(["Drift", "Quadrupole"]))   # the reference's Dipole.split returns the dipole itself: no halves
    if base == "Drift":
        b = [base, {"length": u(0.2, 1.0)}]
    else:
        b = [base, {"length": u(0.1, 0.5), "k1": u(-15.0, 15.0)}]
    top = str(rng.choice(["Marker", "BPM", "HorizontalCorrector"]))
    t = [top, {"length": 0.0, "angle": u(-3e-4, 3e-4)}] if top == "HorizontalCorrector" else [top, {}]
    return ["Superimposed", {"base_element": b, "superimposed_element": t}]


def draw_segment(depth):
    n = int(rng.integers(2, 5))
    children = []
    for _ in range(n):
        r = rng.random()
        if depth < 2 and r < 0.2:
            children.append(draw_segment(depth + 1))
        elif r < 0.35:
            children.append(draw_superimposed())
        else:
            children.append(draw_simple())
    return ["Segment", {"elements": children}]


def build(module, spec, top_level=False):
    kind, kw = spec
    if kind == "Segment":
        return module.Segment([build(module, c) for c in kw["elements"]])
    if kind == "Superimposed":
        return module.Superimposed(build(module, kw["base_element"]), build(module, kw["superimposed_element"]), **f64)
    if kind == "SpaceChargeKick":
        return module.SpaceChargeKick(effect_length=torch.tensor(kw["effect_length"], **f64), grid_shape=tuple(kw["grid_shape"]), **f64)
    args = {k: (torch.tensor(v, **f64) if isinstance(v, (float, list)) else v) for k, v in kw.items()}
    return getattr(module, kind)(**args, **f64)


def species(module, name):
    if name == "ion":   # a fully stripped carbon-12 nucleus, approximately
        return module.Species("ion", num_elementary_charges=torch.tensor(6.0, **f64), mass_eV=torch.tensor(1.1178e10, **f64))
    return module.Species(name, **f64)


SPECIES = ["electron", "electron", "positron", "proton", "ion", "electron", "proton", "electron", "positron", "electron"]

if __name__ == "__main__":
    arrays = {"n_lattices": np.asarray(len(SPECIES))}
    for i, sp_name in enumerate(SPECIES):
        top = []
        for _ in range(int(rng.integers(4, 9))):
            r = rng.random()
            top.append(draw_segment(1) if r < 0.3 else (draw_superimposed() if r < 0.45 else draw_simple()))
        with_sc = i in (2, 5, 7)
        if with_sc:
            # a small grid: the kick itself is pinned by space_charge_random.npz, here it sits in front of a run of linear elements
            at = int(rng.integers(0, len(top)))
            top.insert(at, ["SpaceChargeKick", {"effect_length": u(0.05, 0.4), "grid_shape": [int(rng.choice([8, 12, 16])) for _ in range(3)]}])
            top.insert(at + 1, ["Drift", {"length": u(0.1, 0.5)}])
            top.insert(at + 2, ["Quadrupole", {"length": 0.2, "k1": u(-8.0, 8.0)}])
        mass = float(species(cheetah, sp_name).mass_eV)
        gamma = float(np.exp(rng.uniform(np.log(1.5 if mass > 1e8 else 40.0), np.log(30.0 if mass > 1e8 else 8000.0))))
        energy = gamma * mass
        torch.manual_seed(3000 + i)
        sp = species(cheetah, sp_name)
        beam = cheetah.ParticleBeam.from_parameters(
            num_particles=128, mu_x=torch.tensor(u(-2e-4, 2e-4), **f64), mu_y=torch.tensor(u(-2e-4, 2e-4), **f64),
            sigma_x=torch.tensor(u(5e-5, 5e-4), **f64), sigma_y=torch.tensor(u(5e-5, 5e-4), **f64),
            sigma_px=torch.tensor(u(1e-5, 2e-4), **f64), sigma_py=torch.tensor(u(1e-5, 2e-4), **f64),
            sigma_tau=torch.tensor(u(1e-5, 1e-3), **f64), sigma_p=torch.tensor(u(1e-4, 3e-3), **f64),
            energy=torch.tensor(energy, **f64), total_charge=torch.tensor(u(1e-11, 1e-9), **f64), species=sp, **f64)
        seg = cheetah.Segment([build(cheetah, s) for s in top])
        out = seg.track(beam)
        arrays[f"spec_{i}"] = np.asarray(json.dumps(top))
        arrays[f"species_{i}"] = np.asarray(sp_name)
        arrays[f"energy_{i}"] = np.asarray(energy)
        arrays[f"in_{i}"] = beam.particles.numpy()
        arrays[f"charges_{i}"] = beam.particle_charges.numpy()
        arrays[f"out_{i}"] = out.particles.numpy()
        arrays[f"survival_{i}"] = out.survival_probabilities.numpy()
        arrays[f"energy_out_{i}"] = out.energy.numpy()
        arrays[f"s_out_{i}"] = out.s.numpy()
        arrays[f"length_{i}"] = seg.length.numpy()
        arrays[f"n_flat_{i}"] = np.asarray(len(seg.flattened().elements))
        # the lattice as the reference writes it to LatticeJSON, and what the reference tracks after reading that file back in
        # float64 (the file keeps ~17 digits; a default-dtype load would round every setting to float32)
        import tempfile
        path = os.path.join(tempfile.mkdtemp(), "lattice.json")
        seg.to_lattice_json(path)
        arrays[f"json_{i}"] = np.asarray(open(path).read())
        arrays[f"out_json_{i}"] = cheetah.Segment.from_lattice_json(path, **f64).track(beam).particles.numpy()
        if not with_sc:
            # NOT beam.as_parameter_beam(): the reference's conversion drops the species (particle_beam.py:1168-1178 passes none,
            # so a proton beam becomes an electron ParameterBeam); the moments are the same, the species is passed on here
            conv = beam.as_parameter_beam()
            pb = cheetah.ParameterBeam(conv.mu, conv.cov, beam.energy, total_charge=beam.total_charge, species=sp, **f64)
            arrays[f"converted_species_{i}"] = np.asarray(conv.species.name)
            pout = seg.track(pb)
            arrays[f"pmu_in_{i}"], arrays[f"pcov_in_{i}"] = pb.mu.numpy(), pb.cov.numpy()
            arrays[f"pmu_out_{i}"], arrays[f"pcov_out_{i}"] = pout.mu.numpy(), pout.cov.numpy()
        print(i, sp_name, f"gamma {gamma:.1f}", json.dumps(top)[:140])
    np.savez_compressed(os.path.join(OUT, "lattices_random_structured.npz"), **arrays)
    print("wrote", len(arrays), "arrays")
