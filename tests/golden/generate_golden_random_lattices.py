#!/usr/bin/env python3
"""Random lattices tracked by the reference -> tests/golden/lattices_random.npz: for each of 12 drawn beamlines (6-14 elements:
drifts, quadrupoles with tilt / misalignment, dipoles and rectangular bends with faces, correctors, solenoids, standing- and
travelling-wave cavities on and off, markers, BPMs, inactive screens, apertures on and off) the element list as JSON, 96
incoming particles and what `Segment.track` makes of them in float64 (particles, survival probabilities, energy, s), and the same for a ParameterBeam
(mu, cov, energy).
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_random_lattices.py
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
rng = np.random.default_rng(424242)


def u(lo, hi):
    return float(rng.uniform(lo, hi))


def draw_element():
    kind = str(rng.choice(["Drift", "Drift", "Quadrupole", "Quadrupole", "Dipole", "RBend", "HorizontalCorrector", "VerticalCorrector",
                       "CombinedCorrector", "Solenoid", "Cavity", "Marker", "BPM", "Screen", "Aperture"]))
    if kind == "Drift":
        return kind, {"length": u(0.05, 2.0)}
    if kind == "Quadrupole":
        kw = {"length": u(0.05, 0.5), "k1": u(-25.0, 25.0)}
        if rng.random() < 0.4:
            kw["tilt"] = u(-0.5, 0.5)
        if rng.random() < 0.4:
            kw["misalignment"] = [u(-1e-3, 1e-3), u(-1e-3, 1e-3)]
        return kind, kw
    if kind in ("Dipole", "RBend"):
        kw = {"length": u(0.2, 1.5), "angle": u(-0.4, 0.4)}
        if rng.random() < 0.5:
            kw.update({"fringe_integral": u(0.2, 0.7), "gap": u(0.01, 0.05)})
        if rng.random() < 0.4:
            kw["k1"] = u(-3.0, 3.0)
        if rng.random() < 0.3:
            kw["tilt"] = u(-0.3, 0.3)
        faces = ("dipole_e1", "dipole_e2") if kind == "Dipole" else ("rbend_e1", "rbend_e2")
        if rng.random() < 0.5:
            kw[faces[0]] = u(-0.2, 0.2)
            kw[faces[1]] = u(-0.2, 0.2)
        return kind, kw
    if kind in ("HorizontalCorrector", "VerticalCorrector"):
        return kind, {"length": u(0.01, 0.2), "angle": u(-5e-4, 5e-4)}
    if kind == "CombinedCorrector":
        return kind, {"length": u(0.01, 0.2), "horizontal_angle": u(-5e-4, 5e-4), "vertical_angle": u(-5e-4, 5e-4)}
    if kind == "Solenoid":
        return kind, {"length": u(0.05, 0.5), "k": u(-3.0, 3.0)}
    if kind == "Cavity":
        on = rng.random() < 0.7
        return kind, {"length": u(0.3, 1.2), "voltage": u(1e6, 2e7) if on else 0.0, "phase": u(-60.0, 60.0),
                      "frequency": 1.3e9, "cavity_type": str(rng.choice(["standing_wave", "traveling_wave"]))}
    if kind == "Aperture":
        return kind, {"x_max": u(5e-4, 1e-2), "y_max": u(5e-4, 1e-2), "shape": str(rng.choice(["rectangular", "elliptical"])),
                      "is_active": bool(rng.random() < 0.6)}
    if kind == "Screen":
        return kind, {"is_active": False}
    return kind, {}


def build(module, spec):
    elements = []
    for kind, kw in spec:
        args = {k: (torch.tensor(v, **f64) if isinstance(v, (float, list)) else v) for k, v in kw.items()}
        elements.append(getattr(module, kind)(**args, **f64))
    return module.Segment(elements)


arrays = {"n_lattices": np.asarray(12)}
for i in range(12):
    spec = [draw_element() for _ in range(int(rng.integers(6, 15)))]
    energy = float(np.exp(rng.uniform(np.log(2e7), np.log(5e9))))
    torch.manual_seed(1000 + i)
    beam = cheetah.ParticleBeam.from_parameters(num_particles=96, energy=torch.tensor(energy, **f64), sigma_x=torch.tensor(3e-4, **f64),
                                                sigma_y=torch.tensor(2e-4, **f64), sigma_px=torch.tensor(2e-5, **f64),
                                                sigma_py=torch.tensor(3e-5, **f64), sigma_tau=torch.tensor(1e-4, **f64),
                                                sigma_p=torch.tensor(2e-3, **f64), **f64)
    out = build(cheetah, spec).track(beam)
    assert torch.isfinite(out.particles).all()
    # the same beamline with a ParameterBeam (apertures do not act on it)
    pbeam = cheetah.ParameterBeam.from_twiss(beta_x=torch.tensor(u(1.0, 20.0), **f64), alpha_x=torch.tensor(u(-1.5, 1.5), **f64),
                                             beta_y=torch.tensor(u(1.0, 20.0), **f64), alpha_y=torch.tensor(u(-1.5, 1.5), **f64),
                                             emittance_x=torch.tensor(2e-9, **f64), emittance_y=torch.tensor(3e-9, **f64),
                                             energy=torch.tensor(energy, **f64), **f64)
    pout = build(cheetah, spec).track(pbeam)
    arrays[f"pmu_in_{i}"], arrays[f"pcov_in_{i}"] = pbeam.mu.numpy(), pbeam.cov.numpy()
    arrays[f"pmu_out_{i}"], arrays[f"pcov_out_{i}"] = pout.mu.numpy(), pout.cov.numpy()
    arrays[f"penergy_out_{i}"] = pout.energy.numpy()
    arrays[f"spec_{i}"] = np.asarray(json.dumps(spec))
    arrays[f"in_{i}"] = beam.particles.numpy()
    arrays[f"energy_{i}"] = np.asarray(energy)
    arrays[f"charges_{i}"] = beam.particle_charges.numpy()
    arrays[f"out_{i}"] = out.particles.numpy()
    arrays[f"survival_{i}"] = out.survival_probabilities.numpy()
    arrays[f"energy_out_{i}"] = out.energy.numpy()
    arrays[f"s_out_{i}"] = out.s.numpy()
    print(i, [k for k, _ in spec], "E", f"{energy:.3e}", "->", f"{float(out.energy):.3e}", "alive", float(out.survival_probabilities.sum()))
path = os.path.join(OUT, "lattices_random.npz")
np.savez_compressed(path, **arrays)
print("->", path, f"{os.path.getsize(path) / 1024:.1f} KiB")
