#!/usr/bin/env python3
"""A linac of 16 cells [Drift, Quadrupole, active Cavity], 1e4 particles, fp32: wall time per `Segment.track` and the host profile
(the run in front of every cavity takes the persistent plan, the cavity chx_cavity_track_scalars)."""
import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cheetah_amd as ca
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)
els = []
for i in range(16):
    els += [ca.Drift(t(0.3), **kw), ca.Quadrupole(t(0.2), k1=t(3.0 if i % 2 else -3.0), **kw),
            ca.Cavity(t(1.0377), voltage=t(18e6), phase=t(-10.0), frequency=t(1.3e9), **kw)]
seg = ca.Segment(els)
beam = ca.ParticleBeam.from_parameters(num_particles=10_000, energy=t(1e8), **kw)
for _ in range(20): seg.track(beam)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): seg.track(beam)
torch.cuda.synchronize(); print("linac of 16 cells (drift, quad, active cavity): us/track", (time.perf_counter() - t0) / 200 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): seg.track(beam)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
