#!/usr/bin/env python3
"""Beam statistics of beams that are hard on a ONE-PASS moment sum -> tests/golden/moment_outliers.npz.

The reference's weighted statistics are two-pass (utils/statistics.py:30-48: mean first, then the centred sum); the device sums
sum w d d^T about a provisional centre in one sweep (chx_moments_fused: the weighted mean of the row's first 64 particles) and
re-centres exactly. Cases that would break a careless centre:
  dead_outlier_slot0      particle 0 sits 1e8 sigma away and has survival probability 0
  alive_outlier_slot0     particle 0 sits 1e5 sigma away and counts
  dead_first_wave         the first 64 particles are all dead and 1e3 sigma away (no weight in the first wave)
  offset_beam             a beam of sigma 1e-7 centred 0.3 away from the origin in every coordinate
  weighted_tail           smoothly varying weights, heavy-tailed coordinates
Per case: particles (5000 x 7), survival probabilities, and the reference's mu_*, sigma_* and cov_xpx / cov_ypy / cov_taup,
in float64 and (particles rounded to) float32.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_moment_outliers.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import cheetah  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(77)
N = 5000
SIG = np.array([2e-4, 3e-5, 2e-4, 3e-5, 1e-4, 1e-3, 0.0])
PROPS = ["mu_x", "mu_px", "mu_y", "mu_py", "mu_tau", "mu_p", "sigma_x", "sigma_px", "sigma_y", "sigma_py", "sigma_tau", "sigma_p",
         "cov_xpx", "cov_ypy", "cov_taup"]


def base():
    x = rng.standard_normal((N, 7)) * SIG
    x[:, 6] = 1.0
    return x


def cases():
    out = {}
    x, w = base(), np.ones(N)
    x[0, :6] += 1e8 * SIG[:6]
    w[0] = 0.0
    out["dead_outlier_slot0"] = (x, w)
    x, w = base(), np.ones(N)
    x[0, :6] -= 1e5 * SIG[:6]
    out["alive_outlier_slot0"] = (x, w)
    x, w = base(), np.ones(N)
    x[:64, :6] += 1e3 * SIG[:6]
    w[:64] = 0.0
    out["dead_first_wave"] = (x, w)
    x, w = rng.standard_normal((N, 7)) * 1e-7, np.ones(N)
    x[:, :6] += 0.3
    x[:, 6] = 1.0
    out["offset_beam"] = (x, w)
    x = rng.standard_t(3, size=(N, 7)) * SIG
    x[:, 6] = 1.0
    w = 0.5 + 0.5 * np.sin(np.arange(N) * 0.01) ** 2
    out["weighted_tail"] = (x, w)
    return out


if __name__ == "__main__":
    arrays = {"names": np.asarray(list(cases()))}
    rng = np.random.default_rng(77)
    for name, (x, w) in cases().items():
        for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
            xt = torch.tensor(x, dtype=dt)
            wt = torch.tensor(w, dtype=dt)
            beam = cheetah.ParticleBeam(xt, torch.tensor(1e8, dtype=dt), survival_probabilities=wt, dtype=dt)
            arrays[f"{name}_{tag}_x"] = xt.numpy()
            arrays[f"{name}_{tag}_w"] = wt.numpy()
            for p in PROPS:
                arrays[f"{name}_{tag}_{p}"] = np.asarray(getattr(beam, p).double().numpy())
            print(name, tag, "sigma_x", float(beam.sigma_x), "mu_x", float(beam.mu_x))
    np.savez_compressed(os.path.join(OUT, "moment_outliers.npz"), **arrays)
    print("wrote", len(arrays), "arrays")
