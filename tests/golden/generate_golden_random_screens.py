#!/usr/bin/env python3
"""Screen readings of the reference on drawn set-ups -> tests/golden/screens_random.npz: ten screens with non-square
resolutions (16..120 pixels per side, a multiple of the binning), pixel sizes from 5 um to 0.5 mm, binning 1 / 2 / 4,
misalignments of up to a quarter of the screen, methods histogram and cloud-in-cell; 600 particles whose spread is drawn
relative to the screen (some of them fall outside), dead particles mixed in. The particle charges are one power of two, so
that a histogram pixel is an exact multiple of it: equal images = identical pixel indices for every particle.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_random_screens.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import cheetah  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(1234567)
arrays = {"n_cases": np.asarray(10)}
for i in range(10):
    dtype = torch.float32 if i % 2 else torch.float64
    kw = {"dtype": dtype}
    binning = int(rng.choice([1, 2, 4]))
    res = (int(rng.integers(4, 30)) * 4, int(rng.integers(4, 30)) * 4)
    px = [float(np.exp(rng.uniform(np.log(5e-6), np.log(5e-4)))) for _ in range(2)]
    half = [res[0] * px[0] / 2, res[1] * px[1] / 2]
    mis = [float(rng.uniform(-0.25, 0.25) * 2 * half[0]), float(rng.uniform(-0.25, 0.25) * 2 * half[1])]
    method = "histogram" if i < 6 else "cloud-in-cell"
    n = 600
    x = rng.normal(size=(n, 7)) * np.array([half[0] * rng.uniform(0.2, 0.8), 1e-5, half[1] * rng.uniform(0.2, 0.8), 1e-5, 1e-4, 1e-3, 0.0])
    x[:, 6] = 1.0
    # a few particles exactly on pixel edges and on the screen's outer edges
    edges = np.linspace(-half[0], half[0], res[0] // binning + 1)
    x[:8, 0] = edges[rng.integers(0, len(edges), size=8)] + mis[0]
    x[8, 0], x[9, 0] = -half[0] + mis[0], half[0] + mis[0]
    particles = torch.tensor(x, **kw)
    charges = torch.full((n,), 2.0 ** -70, **kw)
    survival = torch.tensor((rng.random(n) > 0.1).astype(np.float64), **kw)
    beam = cheetah.ParticleBeam(particles, torch.tensor(1e8, **kw), particle_charges=charges, survival_probabilities=survival, **kw)
    screen = cheetah.Screen(resolution=res, pixel_size=torch.tensor(px, **kw), binning=binning, misalignment=torch.tensor(mis, **kw),
                            method=method, is_active=True, **kw)
    screen.track(beam)
    img = screen.reading
    arrays[f"meta_{i}"] = np.asarray([res[0], res[1], binning, 0 if method == "histogram" else 1, 0 if dtype == torch.float64 else 1])
    arrays[f"pixel_size_{i}"] = np.asarray(px)
    arrays[f"misalignment_{i}"] = np.asarray(mis)
    arrays[f"xy_{i}"] = particles[:, [0, 2]].numpy()      # in the screen's dtype; the other coordinates do not matter
    arrays[f"survival_{i}"] = survival.numpy()
    arrays[f"image_{i}"] = img.numpy()
    print(i, method, str(dtype), res, binning, "image", tuple(img.shape), "charge on screen / sent", float(img.sum() / (charges * survival).sum()))
path = os.path.join(OUT, "screens_random.npz")
np.savez_compressed(path, **arrays)
print("->", path, f"{os.path.getsize(path) / 1024:.1f} KiB")
