#!/usr/bin/env python3
"""Golden vectors for the fused observables of row f2 (tests/golden/observables.npz), from the REAL reference:

* Screen images per lattice setting: k1 scan (B = 16) on the ARES EA subcell ending in an ACTIVE cloud-in-cell Screen
  (screen.py:327-339 on a vectorised beam) — 96 x 64 pixel images of 8 of the 16 settings, fp32 and fp64;
* beam attributes along the segment (segment.py:658-700): sigma / mu / emittance / beta / s after every element of the
  13-element lattice and at a resolution of 5 cm.

Run in the build container only (imports /root/reference read-only):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_golden_observables.py
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import cheetah  # noqa: E402  (the reference)

torch.set_num_threads(4)
ATTRS = ("sigma_x", "sigma_y", "sigma_px", "mu_x", "mu_py", "cov_xpx", "emittance_x", "beta_x", "alpha_y", "sigma_tau", "s",
         "energy", "total_charge")
ROWS = (0, 1, 3, 6, 8, 11, 14, 15)


def npy(t):
    return t.detach().cpu().numpy()


def subcell(dt, k1, screen_active, res=(96, 64), px=(2.0e-5, 1.5e-5)):
    f = {"dtype": dt}
    t = lambda v: torch.tensor(v, **f)  # noqa: E731
    return cheetah.Segment(elements=[
        cheetah.Marker(name="AREASOLA1", **f), cheetah.Drift(length=t(0.17504), **f),
        cheetah.Quadrupole(length=t(0.122), k1=k1, name="AREAMQZM1", **f), cheetah.Drift(length=t(0.428), **f),
        cheetah.Quadrupole(length=t(0.122), k1=t(-14.3), name="AREAMQZM2", **f), cheetah.Drift(length=t(0.204), **f),
        cheetah.VerticalCorrector(length=t(0.02), angle=t(9e-5), name="AREAMCVM1", **f), cheetah.Drift(length=t(0.204), **f),
        cheetah.Quadrupole(length=t(0.122), k1=t(3.142), name="AREAMQZM3", **f), cheetah.Drift(length=t(0.179), **f),
        cheetah.HorizontalCorrector(length=t(0.02), angle=t(-1e-4), name="AREAMCHM1", **f), cheetah.Drift(length=t(0.45), **f),
        cheetah.Screen(resolution=res, pixel_size=t(list(px)), misalignment=t([3e-5, -2e-5]), name="AREABSCR1",
                       method="cloud-in-cell", is_active=screen_active, **f),
    ])


def main():
    arrays = {"rows": np.asarray(ROWS), "attrs": np.asarray(",".join(ATTRS))}
    torch.manual_seed(4242)
    N, B = 3000, 16
    beam32 = cheetah.ParticleBeam.from_parameters(num_particles=N, dtype=torch.float32)
    surv = torch.ones(N)
    surv[::5] = 0.25
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        beam = beam32.to(dt)
        beam.survival_probabilities = surv.to(dt)
        k1 = torch.linspace(-12, 12, B, dtype=dt)
        seg = subcell(dt, k1, True)
        out = seg.track(beam)
        img = seg.AREABSCR1.reading                     # (B, H, W)
        assert img.shape == (B, 64, 96), img.shape
        arrays[f"in_{tag}"] = npy(beam.particles)
        arrays[f"charges_{tag}"] = npy(beam.particle_charges)
        arrays[f"survival_{tag}"] = npy(beam.survival_probabilities)
        arrays[f"k1_{tag}"] = npy(k1)
        arrays[f"images_{tag}"] = npy(img)[list(ROWS)]
        arrays[f"image_sums_{tag}"] = npy(img.sum(dim=(-1, -2)))
        arrays[f"sigma_x_{tag}"] = npy(out.sigma_x)
        # along the segment, one setting
        seg1 = subcell(dt, torch.tensor(8.2, dtype=dt), False)
        vals = seg1.get_beam_attrs_along_segment(ATTRS, beam)
        for name, v in zip(ATTRS, vals):
            arrays[f"along_{name}_{tag}"] = npy(v)
        vals = seg1.get_beam_attrs_along_segment(("sigma_x", "beta_y", "s"), beam, resolution=0.05)
        for name, v in zip(("sigma_x", "beta_y", "s"), vals):
            arrays[f"along5cm_{name}_{tag}"] = npy(v)
    path = os.path.join(HERE, "observables.npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote observables.npz: {os.path.getsize(path) / 1024:.1f} KiB")
    print("along sigma_x f64:", arrays["along_sigma_x_f64"])
    print("image sums f64:", arrays["image_sums_f64"][:4], "total charge", float((beam.particle_charges.abs() * beam.survival_probabilities).sum()))


if __name__ == "__main__":
    main()
