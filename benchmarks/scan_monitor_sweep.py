#!/usr/bin/env python3
"""4096 settings x 1e5 shared particles with 1 / 6 monitors: ms per stretch call (scan_particles_probe.py's last two shapes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.scan_particles_probe import lattice, timeit, kw
import cheetah_amd as ca
torch.manual_seed(0)
for B, N, cells, monitors in ((4096, 100_000, 6, 1), (4096, 100_000, 6, 6), (512, 100_000, 6, 6)):
    seg = lattice(B, cells, monitors)
    beam = ca.ParticleBeam.from_parameters(num_particles=N, **kw)
    with torch.no_grad():
        print(f"B = {B} x N = {N}, {monitors} monitors: {timeit(lambda: seg.track(beam)):.3f} ms", flush=True)
