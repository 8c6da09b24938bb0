#!/usr/bin/env python3
"""The stretch call of benchmarks/scan_particles_probe.py alone (B = 4096 rows x 1e5 shared particles, 6 cells, 1 or 6 monitors:
argv[1]), for `rocprofv3 --kernel-trace --stats`."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
from benchmarks.scan_particles_probe import lattice, kw

monitors = int(sys.argv[1]) if len(sys.argv) > 1 else 6
seg = lattice(4096, 6, monitors)
beam = ca.ParticleBeam.from_parameters(num_particles=100_000, **kw)
with torch.no_grad():
    for _ in range(6):
        seg.track(beam)
torch.cuda.synchronize()
