#!/usr/bin/env python3
"""A scan of lattice settings tracked with PARTICLES (one beam shared by the B rows) through lattices with active monitors: the
stretch call (chx_lattice_track_diag, Bm = B rows of maps, Bx = 1) against the walk item by item, ms per track and GB/s of the
(B, N, 7) result."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
from cheetah_amd.accelerator.segment import Segment
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def lattice(B, cells, monitors):
    els = []
    for i in range(cells):
        els += [ca.Quadrupole(t(0.2), k1=(torch.randn(B, **kw) if i == 0 else t(4.2 if i % 2 == 0 else -4.2)), **kw),
                ca.HorizontalCorrector(t(0.05), angle=1e-5 * torch.randn(B, **kw), **kw), ca.Drift(t(0.8), **kw)]
        if i % (cells // monitors) == cells // monitors - 1:
            els += [ca.BPM(is_active=True, **kw)]
    return ca.Segment(els)


if __name__ == "__main__":
    orig = Segment._lattice_stretch
    for B, N, cells, monitors in ((64, 10_000, 25, 25), (512, 100_000, 6, 1), (512, 100_000, 6, 6), (4096, 100_000, 6, 1), (4096, 100_000, 6, 6)):
        seg = lattice(B, cells, monitors)
        beam = ca.ParticleBeam.from_parameters(num_particles=N, **kw)
        with torch.no_grad():
            a = timeit(lambda: seg.track(beam))
            Segment._lattice_stretch = lambda self, plan, i, incoming: None
            try:
                b = timeit(lambda: seg.track(beam), reps=2, warm=1)
            finally:
                Segment._lattice_stretch = orig
        gb = B * N * 28 / 1e9
        print(f"B = {B:5d} x N = {N:7d}, {cells} cells, {monitors} monitors: stretch {a:8.3f} ms ({gb / a * 1e3:7.0f} GB/s of result)   walk {b:8.3f} ms", flush=True)
