#!/usr/bin/env python3
"""Degenerate inputs through the reference -> tests/golden/edge_cases.npz: what `Segment.track` and the beam statistics return
(values, NaN / inf patterns) or that they raise, for
  one-particle beams, beams whose survival probabilities are all zero, zero-length elements (drift, quadrupole with k1, dipole
  without angle, solenoid, cavity), switched-off strengths (k1 = 0, angle = 0, k = 0, voltage = 0), gamma barely above 1,
  phase advances of ~100 rad, a negative drift length, an aperture of zero width, a screen the beam misses entirely and a beam
  with a NaN coordinate.
Every case: JSON element list, incoming particles / survival / energy, and either the outgoing particles + survival + energy
+ (mu_x, sigma_x, sigma_p, emittance_x) or the name of the exception the reference raised. float64.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_edge_cases.py
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import cheetah  # noqa: E402

warnings.simplefilter("ignore")
OUT = os.path.dirname(os.path.abspath(__file__))
f64 = {"dtype": torch.float64}
rng = np.random.default_rng(404)


def beam_array(n):
    x = rng.standard_normal((n, 7)) * np.array([2e-4, 3e-5, 2e-4, 3e-5, 1e-4, 1e-3, 0.0])
    x[:, 6] = 1.0
    return x


LINE = [["Drift", {"length": 0.7}], ["Quadrupole", {"length": 0.2, "k1": 6.0}], ["Dipole", {"length": 0.5, "angle": 0.1}],
        ["HorizontalCorrector", {"length": 0.1, "angle": 2e-4}], ["Drift", {"length": 0.3}]]

CASES = {
    "one_particle": (LINE, beam_array(1), None, 1e8),
    "two_particles": (LINE, beam_array(2), None, 1e8),
    "all_dead": (LINE, beam_array(50), np.zeros(50), 1e8),
    "one_survivor": (LINE, beam_array(50), np.eye(50)[7], 1e8),
    "zero_length_drift": ([["Drift", {"length": 0.0}]], beam_array(40), None, 1e8),
    "zero_length_quadrupole": ([["Drift", {"length": 0.2}], ["Quadrupole", {"length": 0.0, "k1": 5.0}], ["Drift", {"length": 0.2}]],
                               beam_array(40), None, 1e8),
    "zero_length_dipole_no_angle": ([["Dipole", {"length": 0.0, "angle": 0.0}], ["Drift", {"length": 0.2}]], beam_array(40), None, 1e8),
    "zero_length_solenoid": ([["Solenoid", {"length": 0.0, "k": 1.0}], ["Drift", {"length": 0.2}]], beam_array(40), None, 1e8),
    "zero_length_cavity_off": ([["Cavity", {"length": 0.0, "voltage": 0.0, "frequency": 1.3e9}], ["Drift", {"length": 0.2}]],
                               beam_array(40), None, 1e8),
    "zero_length_cavity_on": ([["Cavity", {"length": 0.0, "voltage": 1e6, "phase": 10.0, "frequency": 1.3e9}]], beam_array(40), None, 1e8),
    "all_strengths_zero": ([["Quadrupole", {"length": 0.3, "k1": 0.0}], ["Dipole", {"length": 0.4, "angle": 0.0}],
                            ["Solenoid", {"length": 0.2, "k": 0.0}], ["Cavity", {"length": 0.5, "voltage": 0.0, "frequency": 1.3e9}],
                            ["HorizontalCorrector", {"length": 0.1, "angle": 0.0}], ["Sextupole", {"length": 0.2, "k2": 0.0}],
                            ["Undulator", {"length": 0.3}]], beam_array(40), None, 1e8),
    "gamma_barely_above_one": (LINE, beam_array(40), None, 510998.95069 * 1.0001),
    "gamma_exactly_one": ([["Drift", {"length": 0.5}]], beam_array(40), None, 510998.95069),
    "hundred_radians": ([["Quadrupole", {"length": 10.0, "k1": 100.0}], ["Quadrupole", {"length": 10.0, "k1": -100.0}]],
                        beam_array(40) * 1e-3 + np.array([0, 0, 0, 0, 0, 0, 0.999]), None, 1e9),
    "negative_drift": ([["Drift", {"length": -0.4}], ["Quadrupole", {"length": 0.2, "k1": 3.0}]], beam_array(40), None, 1e8),
    "closed_aperture": ([["Drift", {"length": 0.2}], ["Aperture", {"x_max": 0.0, "y_max": 0.0, "is_active": True}], ["Drift", {"length": 0.2}]],
                        beam_array(40), None, 1e8),
    "nan_coordinate": (LINE, np.where(np.arange(280).reshape(40, 7) == 23, np.nan, beam_array(40)), None, 1e8),
    "inf_coordinate": (LINE, np.where(np.arange(280).reshape(40, 7) == 23, np.inf, beam_array(40)), None, 1e8),
    "second_order_zero_strength": ([["Quadrupole", {"length": 0.3, "k1": 0.0, "tracking_method": "second_order"}],
                                    ["Dipole", {"length": 0.4, "angle": 0.0, "tracking_method": "second_order"}]], beam_array(40), None, 1e8),
    "dkd_zero_strength": ([["Quadrupole", {"length": 0.3, "k1": 0.0, "tracking_method": "drift_kick_drift"}],
                           ["Dipole", {"length": 0.4, "angle": 0.0, "tracking_method": "drift_kick_drift"}],
                           ["Drift", {"length": 0.0, "tracking_method": "drift_kick_drift"}]], beam_array(40), None, 1e8),
}


def build(spec):
    els = []
    for kind, kw in spec:
        args = {k: (torch.tensor(v, **f64) if isinstance(v, float) else v) for k, v in kw.items()}
        els.append(getattr(cheetah, kind)(**args, **f64))
    return cheetah.Segment(els)


STATS = ["mu_x", "sigma_x", "sigma_p", "emittance_x", "total_charge"]

if __name__ == "__main__":
    arrays = {"names": np.asarray(list(CASES))}
    for name, (spec, x, survival, energy) in CASES.items():
        arrays[f"{name}_spec"] = np.asarray(json.dumps(spec))
        arrays[f"{name}_in"] = x
        arrays[f"{name}_energy"] = np.asarray(energy)
        n = x.shape[0]
        surv = np.ones(n) if survival is None else survival
        arrays[f"{name}_survival"] = surv
        charges = np.full(n, 1e-12 / n)
        arrays[f"{name}_charges"] = charges
        try:
            beam = cheetah.ParticleBeam(torch.tensor(x), torch.tensor(energy, **f64), particle_charges=torch.tensor(charges),
                                        survival_probabilities=torch.tensor(surv), **f64)
            out = build(spec).track(beam)
            arrays[f"{name}_out"] = out.particles.numpy()
            arrays[f"{name}_out_survival"] = out.survival_probabilities.numpy()
            arrays[f"{name}_out_energy"] = out.energy.numpy()
            for s in STATS:
                try:
                    arrays[f"{name}_{s}"] = np.asarray(getattr(out, s).numpy())
                except Exception as err:  # noqa: BLE001
                    arrays[f"{name}_{s}_raises"] = np.asarray(type(err).__name__)
            arrays[f"{name}_raises"] = np.asarray("")
            o = arrays[f"{name}_out"]
            print(f"{name:30s} finite {np.isfinite(o).mean():.2f}  nan {np.isnan(o).sum():4d}  inf {np.isinf(o).sum():3d}  sigma_x "
                  f"{arrays.get(f'{name}_sigma_x')}")
        except Exception as err:  # noqa: BLE001
            arrays[f"{name}_raises"] = np.asarray(type(err).__name__)
            print(f"{name:30s} raises {type(err).__name__}: {str(err)[:80]}")
    np.savez_compressed(os.path.join(OUT, "edge_cases.npz"), **arrays)
    print("wrote", len(arrays), "arrays")
