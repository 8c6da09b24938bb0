#!/usr/bin/env python3
"""Random beamlines with VECTORISED settings, tracked by the reference -> tests/golden/lattices_random_vectorized.npz: eight drawn
lines of 4-8 elements in which some settings carry vector dimensions of shapes (3,), (2, 1) or (2, 3) — they broadcast to a
(2, 3) batch — with a cavity (vectorised voltage or phase) and an aperture in some; 24 incoming particles shared by the
batch or, in every other case, a vectorised incoming beam. Float64 in, the tracked particles, survival and energy out.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_random_vectorized.py
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
rng = np.random.default_rng(8675309)
SHAPES = [(), (), (3,), (2, 1), (2, 3)]


def vec(lo, hi):
    shape = SHAPES[int(rng.integers(len(SHAPES)))]
    return rng.uniform(lo, hi, size=shape).tolist() if shape else float(rng.uniform(lo, hi))


def draw_element():
    kind = str(rng.choice(["Drift", "Quadrupole", "Quadrupole", "Dipole", "HorizontalCorrector", "VerticalCorrector", "Solenoid",
                           "Cavity", "Aperture", "Marker"]))
    if kind == "Drift":
        return kind, {"length": vec(0.05, 1.5)}
    if kind == "Quadrupole":
        return kind, {"length": vec(0.05, 0.4), "k1": vec(-20.0, 20.0), "tilt": vec(-0.3, 0.3)}
    if kind == "Dipole":
        return kind, {"length": vec(0.3, 1.0), "angle": vec(-0.3, 0.3), "dipole_e1": vec(-0.1, 0.1), "fringe_integral": vec(0.2, 0.6),
                      "gap": vec(0.01, 0.04)}
    if kind in ("HorizontalCorrector", "VerticalCorrector"):
        return kind, {"length": vec(0.01, 0.2), "angle": vec(-5e-4, 5e-4)}
    if kind == "Solenoid":
        return kind, {"length": vec(0.05, 0.5), "k": vec(-3.0, 3.0)}
    if kind == "Cavity":
        return kind, {"length": vec(0.3, 1.2), "voltage": vec(1e6, 2e7), "phase": vec(-60.0, 60.0), "frequency": 1.3e9,
                      "cavity_type": str(rng.choice(["standing_wave", "traveling_wave"]))}
    if kind == "Aperture":
        return kind, {"x_max": vec(5e-4, 5e-3), "y_max": vec(5e-4, 5e-3), "shape": str(rng.choice(["rectangular", "elliptical"])),
                      "is_active": True}
    return kind, {}


def build(module, spec):
    elements = []
    for kind, kw in spec:
        args = {k: (torch.tensor(v, **f64) if isinstance(v, (float, list)) else v) for k, v in kw.items()}
        elements.append(getattr(module, kind)(**args, **f64))
    return module.Segment(elements)


n_lat = 8
arrays = {"n_lattices": np.asarray(n_lat)}
for i in range(n_lat):
    spec = [draw_element() for _ in range(int(rng.integers(4, 9)))]
    energy = float(np.exp(rng.uniform(np.log(3e7), np.log(3e9))))
    torch.manual_seed(3000 + i)
    beam = cheetah.ParticleBeam.from_parameters(num_particles=24, energy=torch.tensor(energy, **f64), sigma_x=torch.tensor(3e-4, **f64),
                                                sigma_y=torch.tensor(2e-4, **f64), sigma_px=torch.tensor(2e-5, **f64),
                                                sigma_py=torch.tensor(3e-5, **f64), sigma_tau=torch.tensor(1e-4, **f64),
                                                sigma_p=torch.tensor(2e-3, **f64), **f64)
    particles, en = beam.particles, beam.energy
    if i % 2:    # a vectorised incoming beam: particles (3, N, 7), energy (2, 1)
        particles = particles.unsqueeze(0) * torch.tensor([1.0, 0.5, -1.0], **f64).reshape(3, 1, 1)
        particles[..., 6] = 1.0
        en = en * torch.tensor([[1.0], [1.1]], **f64)
        beam = cheetah.ParticleBeam(particles, en, particle_charges=beam.particle_charges, species=beam.species)
    out = build(cheetah, spec).track(beam)
    assert torch.isfinite(out.particles).all(), spec
    arrays[f"spec_{i}"] = np.asarray(json.dumps(spec))
    arrays[f"in_{i}"] = beam.particles.numpy()
    arrays[f"energy_{i}"] = beam.energy.numpy()
    arrays[f"charges_{i}"] = beam.particle_charges.numpy()
    arrays[f"out_{i}"] = out.particles.numpy()
    arrays[f"survival_{i}"] = out.survival_probabilities.numpy()
    arrays[f"energy_out_{i}"] = out.energy.numpy()
    print(i, [k for k, _ in spec], "in", tuple(beam.particles.shape), tuple(beam.energy.shape), "-> out", tuple(out.particles.shape),
          tuple(out.energy.shape), tuple(out.survival_probabilities.shape))
path = os.path.join(OUT, "lattices_random_vectorized.npz")
np.savez_compressed(path, **arrays)
print("->", path, f"{os.path.getsize(path) / 1024:.1f} KiB")
