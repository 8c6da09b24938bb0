#!/usr/bin/env python3
"""Screen.reading for a beam focused into a few pixels (every particle in one or two deposit tiles): the hot-tile path
of the sorted deposit (cic_accumulate_hot_kernel) and the LDS combining table of the histogram."""
import json
import sys

import torch

sys.path.insert(0, ".")
import cheetah_amd as ca  # noqa: E402
from benchmarks.run_configs import timeit  # noqa: E402

dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
out = {}
for sigma in (2e-6, 2e-5, 2e-4):
    beam = ca.ParticleBeam.from_parameters(num_particles=1_000_000, sigma_x=torch.tensor(sigma, **kw),
                                           sigma_y=torch.tensor(sigma, **kw), **kw)
    for method in ("cloud-in-cell", "histogram"):
        scr = ca.Screen(resolution=(2448, 2040), pixel_size=torch.tensor((3.5488e-6, 2.5003e-6), **kw), method=method,
                        is_active=True, **kw)

        def f():
            scr.track(beam)
            return scr.reading

        out[f"sigma={sigma:g} {method}"] = round(timeit(f, 10, 2), 4)
print(json.dumps(out))
