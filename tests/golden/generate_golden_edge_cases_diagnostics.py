#!/usr/bin/env python3
"""Degenerate inputs of the Screen and the SpaceChargeKick through the reference -> tests/golden/edge_cases_diagnostics.npz:
screens the beam misses, NaN / inf coordinates on a screen (both methods), a binning that does not divide the resolution, dead
particles only; space-charge kicks of a beam without charge, of two particles, of dead particles only, of particles that all
sit in one point. Each case stores the result (image or kicked particles) or the name of the exception the reference raised.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_edge_cases_diagnostics.py
"""
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
rng = np.random.default_rng(505)
t = lambda v: torch.tensor(v, **f64)  # noqa: E731


def particles(n, sx=2e-4, sy=2e-4, st=1e-4):
    x = rng.standard_normal((n, 7)) * np.array([sx, 3e-5, sy, 3e-5, st, 1e-3, 0.0])
    x[:, 6] = 1.0
    return x


arrays = {}
names = []


def record(name, fn, inputs):
    names.append(name)
    for k, v in inputs.items():
        arrays[f"{name}_{k}"] = np.asarray(v)
    try:
        arrays[f"{name}_result"] = fn()
        arrays[f"{name}_raises"] = np.asarray("")
        r = arrays[f"{name}_result"]
        print(f"{name:34s} shape {r.shape} finite {np.isfinite(r).mean():.2f} sum {np.nansum(r):.4e}")
    except Exception as err:  # noqa: BLE001
        arrays[f"{name}_raises"] = np.asarray(type(err).__name__)
        print(f"{name:34s} raises {type(err).__name__}: {str(err)[:90]}")


def screen_case(name, x, method, resolution=(60, 40), binning=1, survival=None, misalignment=(0.0, 0.0)):
    surv = np.ones(len(x)) if survival is None else survival

    def run():
        beam = cheetah.ParticleBeam(torch.tensor(x), t(1e8), particle_charges=torch.full((len(x),), 1e-15, **f64),
                                    survival_probabilities=torch.tensor(surv), **f64)
        scr = cheetah.Screen(resolution=resolution, pixel_size=t([2e-5, 2e-5]), binning=binning, misalignment=t(list(misalignment)),
                             method=method, is_active=True, **f64)
        scr.track(beam)
        return scr.reading.numpy()

    record(name, run, {"x": x, "survival": surv, "resolution": resolution, "binning": binning, "misalignment": misalignment,
                       "method": method})


for method in ("histogram", "cloud-in-cell"):
    tag = "hist" if method == "histogram" else "cic"
    screen_case(f"screen_missed_{tag}", particles(300) + np.array([5e-2, 0, 5e-2, 0, 0, 0, 0]), method)
    x = particles(300); x[17, 0] = np.nan
    screen_case(f"screen_nan_{tag}", x, method)
    x = particles(300); x[17, 2] = np.inf; x[18, 0] = -np.inf
    screen_case(f"screen_inf_{tag}", x, method)
    screen_case(f"screen_odd_binning_{tag}", particles(300), method, resolution=(61, 41), binning=2)
    screen_case(f"screen_all_dead_{tag}", particles(300), method, survival=np.zeros(300))
    screen_case(f"screen_one_particle_{tag}", particles(1, 1e-5, 1e-5), method)
    x = particles(300)
    x[:, 0] = np.round(x[:, 0] / 2e-5) * 2e-5          # exactly on pixel edges
    x[:, 2] = np.round(x[:, 2] / 2e-5) * 2e-5
    screen_case(f"screen_on_edges_{tag}", x, method)


def sc_case(name, x, charges, survival=None, grid=(8, 8, 8), energy=1e8):
    surv = np.ones(len(x)) if survival is None else survival

    def run():
        beam = cheetah.ParticleBeam(torch.tensor(x), t(energy), particle_charges=torch.tensor(charges), survival_probabilities=torch.tensor(surv),
                                    **f64)
        return cheetah.SpaceChargeKick(effect_length=t(0.3), grid_shape=grid, **f64).track(beam).particles.numpy()

    record(name, run, {"x": x, "charges": charges, "survival": surv, "grid": grid, "energy": energy})


sc_case("sc_no_charge", particles(200), np.zeros(200))
sc_case("sc_two_particles", particles(2), np.full(2, 1e-12))
sc_case("sc_one_particle", particles(1), np.full(1, 1e-12))
sc_case("sc_all_dead", particles(200), np.full(200, 1e-14), survival=np.zeros(200))
sc_case("sc_one_point", np.tile(particles(1), (50, 1)), np.full(50, 1e-14))
sc_case("sc_flat_in_y", particles(200, sy=0.0), np.full(200, 1e-14))
sc_case("sc_negative_and_positive_charges", particles(200), np.where(np.arange(200) % 2 == 0, 1e-14, -1e-14))
sc_case("sc_gamma_barely_above_one", particles(200), np.full(200, 1e-14), energy=510998.95069 * 1.001)
arrays["names"] = np.asarray(names)
np.savez_compressed(os.path.join(OUT, "edge_cases_diagnostics.npz"), **arrays)
print("wrote", len(arrays), "arrays")
