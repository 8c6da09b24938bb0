#!/usr/bin/env python3
"""The reference's DEFAULT space-charge grid (32^3) under a large beam — 400 000 particles, four kicks, the chain of this engine with
dense deposit tiles (64 tiles, ~25 000 slots in an occupied one, 16 workgroups per tile) -> tests/golden/sc_dense_tiles.npz: the
REAL reference tracks [Drift, SpaceChargeKick(0.2, 32^3), Drift, Quadrupole, Drift] x 4 in float64; stored are a 2000-particle
sample (every 200th particle) of the outgoing beam and its first / second moments. Inputs: tests/fullsize_inputs.c4_particles(400000)
on both sides. Run in the build container (~1 min):
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_sc_dense_tiles.py
"""
import os
import sys
import time

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import cheetah  # noqa: E402
from tests import fullsize_inputs as fi  # noqa: E402

torch.set_num_threads(8)
f64 = torch.float64
t = lambda v: torch.tensor(v, dtype=f64)  # noqa: E731
N = 400_000

if __name__ == "__main__":
    x = fi.c4_particles(N, seed=20260930)
    beam = cheetah.ParticleBeam(torch.from_numpy(x), t(fi.C4_ENERGY), particle_charges=torch.from_numpy(fi.c4_charges(N) * (N / 1e6) * 2.5), dtype=f64)
    els = []
    for cell in range(4):
        els += [cheetah.Drift(t(0.1), dtype=f64), cheetah.SpaceChargeKick(t(0.2), grid_shape=(32, 32, 32), dtype=f64),
                cheetah.Drift(t(0.1), dtype=f64), cheetah.Quadrupole(t(0.1), k1=t(fi.c4_quad_k1(cell)), dtype=f64), cheetah.Drift(t(0.1), dtype=f64)]
    t0 = time.time()
    out = cheetah.Segment(els).track(beam)
    free = cheetah.Segment([e for e in els if not isinstance(e, cheetah.SpaceChargeKick)]).track(beam)
    p = out.particles
    print(f"reference: {time.time() - t0:.1f} s; sigma_x {float(out.sigma_x):.6e} (without space charge {float(free.sigma_x):.6e})")
    np.savez_compressed(os.path.join(HERE, "sc_dense_tiles.npz"), n=np.asarray(N), out_sample=p[::200].numpy(), free_sample=free.particles[::200].numpy(),
                        mean=p[:, :6].mean(dim=0).numpy(), std=p[:, :6].std(dim=0).numpy(), total_charge=out.total_charge.numpy())
