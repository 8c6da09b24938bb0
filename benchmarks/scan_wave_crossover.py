#!/usr/bin/env python3
"""Where the row-chunk scan kernel (lattice_scan_wave_kernel) overtakes one (tile, row) per workgroup: ms per stretch call for a few
shapes with CHX_TUNE_SCAN_WAVE=0 (never) and =2 (whenever the layout allows); one monitor, two maps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks.scan_particles_probe import lattice, timeit, kw
import cheetah_amd as ca
torch.manual_seed(0)
for B, N in ((64, 10_000), (64, 100_000), (128, 100_000), (256, 100_000), (1024, 10_000), (4096, 10_000), (512, 100_000), (64, 1_000_000)):
    seg = lattice(B, 6, 1)
    beam = ca.ParticleBeam.from_parameters(num_particles=N, **kw)
    res = []
    for mode in ("0", "2"):
        os.environ["CHX_TUNE_SCAN_WAVE"] = mode
        with torch.no_grad():
            res.append(timeit(lambda: seg.track(beam)))
    print(f"B = {B} x N = {N} ({B * N:.1e} rows): tile-row {res[0]:.3f} ms, row chunks {res[1]:.3f} ms", flush=True)
