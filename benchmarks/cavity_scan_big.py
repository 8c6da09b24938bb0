#!/usr/bin/env python3
"""A phase scan of every cavity of a 16-cell linac over 64 values with a shared beam of 2e5 / 1e6 particles (1.3e7 / 6.4e7 particle
rows: the row-chunk scan kernel with the cavity epilogue): ms per Segment.track, CHX_TUNE_SCAN_WAVE=0 (one (tile, row) per workgroup)
against the default."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.cavity_scan_probe import linac, timeit, kw, t
import cheetah_amd as ca
seg = linac(tuple(range(16)))
for n in (200_000, 1_000_000):
    beam = ca.ParticleBeam.from_parameters(num_particles=n, energy=t(1e8), **kw)
    res = []
    for mode in ("0", "1"):
        os.environ["CHX_TUNE_SCAN_WAVE"] = mode
        with torch.no_grad():
            res.append(timeit(lambda: seg.track(beam)) / 1e3)
    print(f"64 phases of every cavity x {n} particles: tile-row {res[0]:.3f} ms, row chunks {res[1]:.3f} ms", flush=True)
