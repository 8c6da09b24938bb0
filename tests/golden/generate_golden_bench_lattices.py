#!/usr/bin/env python3
"""Golden vectors for the lattices bench.py TIMES, at the shapes it times them: tests/golden/bench_lattices.npz.

Run in the build container only (imports /root/reference read-only through generate_golden.py):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_golden_bench_lattices.py

bench.py's headline (C2) and its DKD_FODO100 / SECOND_ORDER_FODO100 configs track 1e6 particles through
25 x [Quad(0.2, +4.2), Drift(0.8), Quad(0.2, -4.2), Drift(0.8)] with `linear`, `drift_kick_drift` and `second_order`
tracking. The reference is a CPU program: it tracks a SAMPLE of 4096 particles of the same distribution
(`ParticleBeam.from_parameters` defaults) through the same 100 elements, element by element
(`for e in elements: beam = e.track(beam)`, segment.py:545-574 without merging; element.py:180-228), in float32 (its own
arithmetic on the benchmark's dtype) and in float64 on the same float32 inputs (the accurate answer). The GPU test puts the
sample into rows [0, 4096) of a 1e6-particle beam, tracks it through the product path at full size and compares those rows.

Contents are DATA only: the sample and the reference's outputs as numpy arrays.
"""

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from generate_golden import cheetah, np, npy, save, torch  # noqa: E402

SAMPLE = 4096
N_CELLS = 25


def fodo(dtype, method):
    kw = {"dtype": dtype}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    els = []
    for _ in range(N_CELLS):
        els += [cheetah.Quadrupole(length=t(0.2), k1=t(4.2), tracking_method=method, **kw),
                cheetah.Drift(length=t(0.8), tracking_method=method, **kw),
                cheetah.Quadrupole(length=t(0.2), k1=t(-4.2), tracking_method=method, **kw),
                cheetah.Drift(length=t(0.8), tracking_method=method, **kw)]
    return els


def main():
    torch.manual_seed(20260929)
    beam32 = cheetah.ParticleBeam.from_parameters(num_particles=SAMPLE, dtype=torch.float32)
    x32 = beam32.particles.clone()
    out = {"sample": npy(x32), "energy": npy(beam32.energy), "particle_charges": npy(beam32.particle_charges),
           "n_elements": np.asarray(4 * N_CELLS)}
    for method in ("linear", "drift_kick_drift", "second_order"):
        for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            beam = cheetah.ParticleBeam(x32.to(dtype), beam32.energy.to(dtype), particle_charges=beam32.particle_charges.to(dtype),
                                        dtype=dtype)
            for e in fodo(dtype, method):
                beam = e.track(beam)
            out[f"{method}_{tag}"] = npy(beam.particles)
            out[f"{method}_{tag}_energy"] = npy(beam.energy)
            print(method, tag, "sigma_x", float(beam.sigma_x))
    # how far the reference's own float32 run is from its float64 run, per coordinate, in units of the coordinate's scale:
    # the yardstick for the bounds of tests/test_gpu_bench_parity.py
    for method in ("linear", "drift_kick_drift", "second_order"):
        a, b = out[f"{method}_f32"].astype(np.float64), out[f"{method}_f64"]
        print(method, "ref f32 vs f64:", (np.abs(a - b).max(axis=0) / np.abs(b).max(axis=0))[:6])
    save("bench_lattices.npz", **out)


if __name__ == "__main__":
    main()
