#!/usr/bin/env python3
"""Random beamlines with the non-linear tracking methods, tracked by the reference -> tests/golden/lattices_random_nonlinear.npz:
ten drawn lines of 5-10 elements — drifts, quadrupoles (with `num_steps`), dipoles / rectangular bends (fringe fields at
either or both ends), sextupoles and transverse deflecting cavities — every element with a drawn `tracking_method` out of
the ones it supports (linear, second_order, drift_kick_drift), 64 particles in, the tracked particles out, float64;
plus six lines with VECTORISED settings ((3,), (2, 1), (2,)) and / or vectorised particles, for electrons, positrons and protons.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_random_nonlinear.py
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
rng = np.random.default_rng(31337)


def u(lo, hi):
    return float(rng.uniform(lo, hi))


def pick(options):
    return str(rng.choice(options))


def draw_element():
    kind = pick(["Drift", "Quadrupole", "Quadrupole", "Dipole", "RBend", "Sextupole", "TransverseDeflectingCavity"])
    if kind == "Drift":
        return kind, {"length": u(0.05, 1.5), "tracking_method": pick(["linear", "second_order", "drift_kick_drift"])}
    if kind == "Quadrupole":
        kw = {"length": u(0.05, 0.4), "k1": u(-20.0, 20.0), "tracking_method": pick(["linear", "second_order", "drift_kick_drift"]),
              "num_steps": int(rng.integers(1, 6))}
        if rng.random() < 0.4:
            kw["tilt"] = u(-0.4, 0.4)
        if rng.random() < 0.4:
            kw["misalignment"] = [u(-5e-4, 5e-4), u(-5e-4, 5e-4)]
        return kind, kw
    if kind in ("Dipole", "RBend"):
        kw = {"length": u(0.2, 1.0), "angle": u(-0.3, 0.3), "tracking_method": pick(["linear", "second_order", "drift_kick_drift"])}
        if rng.random() < 0.6:
            kw.update({"fringe_integral": u(0.2, 0.7), "fringe_integral_exit": u(0.2, 0.7), "gap": u(0.01, 0.05),
                       "gap_exit": u(0.01, 0.05), "fringe_at": pick(["both", "entrance", "exit", "neither"])})
        faces = ("dipole_e1", "dipole_e2") if kind == "Dipole" else ("rbend_e1", "rbend_e2")
        if rng.random() < 0.5:
            kw[faces[0]], kw[faces[1]] = u(-0.15, 0.15), u(-0.15, 0.15)
        if rng.random() < 0.3:
            kw["tilt"] = u(-0.3, 0.3)
        if rng.random() < 0.3 and kw["tracking_method"] != "drift_kick_drift":
            kw["k1"] = u(-2.0, 2.0)
        return kind, kw
    if kind == "Sextupole":
        kw = {"length": u(0.05, 0.3), "k2": u(-50.0, 50.0), "tracking_method": pick(["linear", "second_order"])}
        if rng.random() < 0.4:
            kw["tilt"] = u(-0.4, 0.4)
        if rng.random() < 0.4:
            kw["misalignment"] = [u(-5e-4, 5e-4), u(-5e-4, 5e-4)]
        return kind, kw
    kw = {"length": u(0.2, 1.0), "voltage": u(-5e6, 5e6), "phase": u(-180.0, 180.0), "frequency": 2.998e9, "num_steps": int(rng.integers(1, 5))}
    if rng.random() < 0.4:
        kw["tilt"] = u(-0.4, 0.4)
    return kind, kw


def build(module, spec):
    elements = []
    for kind, kw in spec:
        args = {k: (torch.tensor(v, **f64) if isinstance(v, (float, list)) else v) for k, v in kw.items()}
        elements.append(getattr(module, kind)(**args, **f64))
    return module.Segment(elements)


n_lat = 10
arrays = {"n_lattices": np.asarray(n_lat)}
for i in range(n_lat):
    spec = [draw_element() for _ in range(int(rng.integers(5, 11)))]
    energy = float(np.exp(rng.uniform(np.log(2e7), np.log(5e9))))
    torch.manual_seed(2000 + i)
    beam = cheetah.ParticleBeam.from_parameters(num_particles=64, energy=torch.tensor(energy, **f64), sigma_x=torch.tensor(3e-4, **f64),
                                                sigma_y=torch.tensor(2e-4, **f64), sigma_px=torch.tensor(2e-5, **f64),
                                                sigma_py=torch.tensor(3e-5, **f64), sigma_tau=torch.tensor(1e-4, **f64),
                                                sigma_p=torch.tensor(2e-3, **f64), **f64)
    out = build(cheetah, spec).track(beam)
    assert torch.isfinite(out.particles).all(), spec
    arrays[f"spec_{i}"] = np.asarray(json.dumps(spec))
    arrays[f"in_{i}"] = beam.particles.numpy()
    arrays[f"energy_{i}"] = np.asarray(energy)
    arrays[f"out_{i}"] = out.particles.numpy()
    arrays[f"energy_out_{i}"] = out.energy.numpy()
    print(i, [(k, a.get("tracking_method", "-")) for k, a in spec])
# ---- vectorised settings / beams and other species (drawn after the lattices above, which keep their draws)
VEC_KEYS = {"Quadrupole": "k1", "Dipole": "angle", "RBend": "angle", "Sextupole": "k2", "TransverseDeflectingCavity": "voltage",
            "Drift": "length"}
CASES = [("electron", (3,), ()), ("proton", (2, 1), ()), ("positron", (), (3,)), ("electron", (2,), (2,)), ("proton", (), ()),
         ("electron", (3,), ())]
arrays["n_vectorised"] = np.asarray(len(CASES))
for i, (sp_name, setting_shape, beam_shape) in enumerate(CASES):
    spec = [draw_element() for _ in range(int(rng.integers(4, 8)))]
    hit = 0
    for kind, kw in spec:
        if setting_shape and (hit == 0 or rng.random() < 0.3):
            key = VEC_KEYS[kind]
            kw[key] = (kw[key] * rng.uniform(0.6, 1.4, setting_shape)).tolist()
            hit += 1
    sp = cheetah.Species(sp_name, **f64)
    gamma = float(np.exp(rng.uniform(np.log(3.0 if sp_name == "proton" else 60.0), np.log(40.0 if sp_name == "proton" else 6000.0))))
    energy = gamma * float(sp.mass_eV)
    x = rng.standard_normal((*beam_shape, 48, 7)) * np.array([3e-4, 2e-5, 2e-4, 3e-5, 1e-4, 2e-3, 0.0])
    x[..., 6] = 1.0
    beam = cheetah.ParticleBeam(torch.tensor(x), torch.tensor(energy, **f64), species=sp, **f64)
    out = build(cheetah, spec).track(beam)
    assert torch.isfinite(out.particles).all(), spec
    arrays[f"v_spec_{i}"] = np.asarray(json.dumps(spec))
    arrays[f"v_species_{i}"] = np.asarray(sp_name)
    arrays[f"v_in_{i}"], arrays[f"v_energy_{i}"] = x, np.asarray(energy)
    arrays[f"v_out_{i}"], arrays[f"v_energy_out_{i}"] = out.particles.numpy(), out.energy.numpy()
    print("vectorised", i, sp_name, setting_shape, beam_shape, "->", tuple(out.particles.shape),
          [(k, a.get("tracking_method", "-")) for k, a in spec])
path = os.path.join(OUT, "lattices_random_nonlinear.npz")
np.savez_compressed(path, **arrays)
print("->", path, f"{os.path.getsize(path) / 1024:.1f} KiB")
