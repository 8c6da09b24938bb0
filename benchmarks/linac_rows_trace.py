#!/usr/bin/env python3
"""The 16-cell cavity linac at 64 beam energies (ParameterBeam and 1e4 shared particles), 20 tracks each, for
`rocprofv3 --kernel-trace --stats`: what the 64-row preparation launch and the two passes cost on the device."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)
els = []
for i in range(16):
    els += [ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw),
            ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(-10.0), frequency=t(1.3e9), **kw)]
seg = ca.Segment(els)
energies = torch.linspace(8e7, 1.2e8, 64, **kw)
pb = ca.ParameterBeam.from_parameters(energy=energies, **kw)
beam = ca.ParticleBeam.from_parameters(num_particles=10_000, energy=energies, **kw)
with torch.no_grad():
    for _ in range(20):
        seg.track(pb)
    for _ in range(20):
        seg.track(beam)
torch.cuda.synchronize()
