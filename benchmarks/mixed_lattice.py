#!/usr/bin/env python3
"""100-element FODO whose drifts are tracked linearly (the default) and whose quadrupoles to second order (or, argv[2] =
drift_kick_drift, with the Bmad-X maps), argv[1] = 1e6 float32 particles:
Segment.track (the linear runs ride in the second-order elements' pass) against the same lattice piece by piece (every run and
every magnet a pass of its own — the walk before the runs were taken along)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402

kw = {"dtype": torch.float32, "device": "cuda"}
tt = lambda v: torch.tensor(v, **kw)  # noqa: E731
method = sys.argv[2] if len(sys.argv) > 2 else "second_order"
so = {"tracking_method": method}
els = []
for _ in range(25):
    els += [ca.Quadrupole(tt(0.2), k1=tt(4.2), **so, **kw), ca.Drift(tt(0.8), **kw),
            ca.Quadrupole(tt(0.2), k1=tt(-4.2), **so, **kw), ca.Drift(tt(0.8), **kw)]
seg = ca.Segment(els)
pieces = [e if e.tracking_method == method else ca.Segment([e]) for e in els]
beam = ca.ParticleBeam.from_parameters(num_particles=int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000, **kw)


def piecewise():
    b = beam
    for p in pieces:
        b = p.track(b)
    return b


with torch.no_grad():
    for name, run in (("Segment.track", lambda: seg.track(beam)), ("piece by piece", piecewise)):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
        print(f"mixed FODO100 (linear drifts, {method} quadrupoles), {beam.num_particles} particles, {name}: {best:.3f} ms per track")
    a, b = seg.track(beam), piecewise()
    print("equal:", torch.equal(a.particles, b.particles), torch.equal(a.s, b.s), torch.equal(a.energy, b.energy))
