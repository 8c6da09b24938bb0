#!/usr/bin/env python3
"""Gradients of the reference through drawn beamlines -> tests/golden/lattices_random_grads.npz: eight lines of 5-9 linear elements
(drifts, quadrupoles with tilt, dipoles with pole faces, correctors, solenoids, cavities, markers); up to four of their scalar
settings, the incoming beam energy and the incoming particles are made trainable; the loss mixes first and second powers of
the outgoing coordinates with drawn weights. Stored: the element list, which settings are trainable, the weights, the loss and
d loss / d (each setting, energy, particles) from the reference's autograd in float64.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_random_grads.py
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
rng = np.random.default_rng(55555)


def u(lo, hi):
    return float(rng.uniform(lo, hi))


def draw_element():
    kind = str(rng.choice(["Drift", "Quadrupole", "Quadrupole", "Dipole", "HorizontalCorrector", "VerticalCorrector", "Solenoid", "Cavity",
                           "Marker"]))
    if kind == "Drift":
        return kind, {"length": u(0.05, 1.5)}
    if kind == "Quadrupole":
        return kind, {"length": u(0.05, 0.4), "k1": u(-15.0, 15.0), "tilt": u(-0.3, 0.3)}
    if kind == "Dipole":
        return kind, {"length": u(0.3, 1.0), "angle": u(-0.3, 0.3), "dipole_e1": u(-0.1, 0.1), "dipole_e2": u(-0.1, 0.1),
                      "fringe_integral": u(0.2, 0.6), "gap": u(0.01, 0.04)}
    if kind in ("HorizontalCorrector", "VerticalCorrector"):
        return kind, {"length": u(0.01, 0.2), "angle": u(-5e-4, 5e-4)}
    if kind == "Solenoid":
        return kind, {"length": u(0.05, 0.5), "k": u(-3.0, 3.0)}
    if kind == "Cavity":
        return kind, {"length": u(0.3, 1.2), "voltage": u(1e6, 2e7), "phase": u(-60.0, 60.0), "frequency": 1.3e9,
                      "cavity_type": str(rng.choice(["standing_wave", "traveling_wave"]))}
    return kind, {}


TRAINABLE = {"length", "k1", "tilt", "angle", "dipole_e1", "k", "voltage", "phase"}


def build(module, spec, trainable, kw):
    elements, params = [], []
    for e, (kind, args) in enumerate(spec):
        targs = {}
        for k, v in args.items():
            if isinstance(v, float):
                t = torch.tensor(v, **kw)
                if [e, k] in trainable:
                    t = torch.nn.Parameter(t)
                    params.append(t)
                targs[k] = t
            else:
                targs[k] = v
        elements.append(getattr(module, kind)(**targs, **kw))
    return module.Segment(elements), params


n_lat = 8
arrays = {"n_lattices": np.asarray(n_lat)}
for i in range(n_lat):
    spec = [draw_element() for _ in range(int(rng.integers(5, 10)))]
    candidates = [[e, k] for e, (kind, args) in enumerate(spec) for k in args if k in TRAINABLE]
    rng.shuffle(candidates)
    trainable = sorted(candidates[:4])
    energy = float(np.exp(rng.uniform(np.log(3e7), np.log(3e9))))
    torch.manual_seed(4000 + i)
    beam0 = cheetah.ParticleBeam.from_parameters(num_particles=48, energy=torch.tensor(energy, **f64), sigma_x=torch.tensor(3e-4, **f64),
                                                 sigma_y=torch.tensor(2e-4, **f64), sigma_px=torch.tensor(2e-5, **f64),
                                                 sigma_py=torch.tensor(3e-5, **f64), sigma_tau=torch.tensor(1e-4, **f64),
                                                 sigma_p=torch.tensor(2e-3, **f64), **f64)
    x = torch.nn.Parameter(beam0.particles.clone())
    en = torch.nn.Parameter(torch.tensor(energy, **f64))
    seg, params = build(cheetah, spec, trainable, f64)
    out = seg.track(cheetah.ParticleBeam(x, en, particle_charges=beam0.particle_charges, species=beam0.species))
    w1 = torch.tensor(rng.uniform(-1.0, 1.0, size=6) * np.array([1e3, 1e4, 1e3, 1e4, 1e3, 1e2]), **f64)
    w2 = torch.tensor(rng.uniform(0.0, 1.0, size=6) * np.array([1e6, 1e8, 1e6, 1e8, 1e6, 1e4]), **f64)
    loss = (out.particles[:, :6] * w1).sum(dim=1).mean() + (out.particles[:, :6].square() * w2).sum(dim=1).mean()
    loss.backward()
    arrays[f"spec_{i}"] = np.asarray(json.dumps(spec))
    arrays[f"trainable_{i}"] = np.asarray(json.dumps(trainable))
    arrays[f"in_{i}"] = beam0.particles.numpy()
    arrays[f"energy_{i}"] = np.asarray(energy)
    arrays[f"w1_{i}"], arrays[f"w2_{i}"] = w1.numpy(), w2.numpy()
    arrays[f"loss_{i}"] = loss.detach().numpy()
    arrays[f"grads_{i}"] = np.asarray([float(p.grad) for p in params])
    arrays[f"grad_energy_{i}"] = en.grad.numpy()
    arrays[f"grad_particles_{i}"] = x.grad.numpy()
    print(i, [k for k, _ in spec], "trainable", trainable, "loss", float(loss), "grads", [f"{float(p.grad):.3e}" for p in params],
          "dE", f"{float(en.grad):.3e}")
path = os.path.join(OUT, "lattices_random_grads.npz")
np.savez_compressed(path, **arrays)
print("->", path, f"{os.path.getsize(path) / 1024:.1f} KiB")
