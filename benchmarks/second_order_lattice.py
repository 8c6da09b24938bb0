#!/usr/bin/env python3
"""100-element FODO tracked with the second-order maps (tracking_method="second_order"), 1e6 and 1e4 float32 particles: wall time per
Segment.track and the kernel time of one element (HIP events around 50 back-to-back applications)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)
so = {"tracking_method": "second_order"}
els = []
for _ in range(25):
    els += [ca.Quadrupole(t(0.2), k1=t(4.2), **so, **kw), ca.Drift(t(0.8), **so, **kw), ca.Quadrupole(t(0.2), k1=t(-4.2), **so, **kw),
            ca.Drift(t(0.8), **so, **kw)]
seg = ca.Segment(els)
for n in (1_000_000, 10_000):
    beam = ca.ParticleBeam.from_parameters(num_particles=n, **kw)
    with torch.no_grad():
        for _ in range(3):
            seg.track(beam)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            seg.track(beam)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        T = els[0].second_order_transfer_map(beam.energy, beam.species)
        x = beam.particles
        from cheetah_amd import _ops
        for _ in range(5):
            _ops.apply_second_order(x, T)
        e0.record()
        for _ in range(50):
            _ops.apply_second_order(x, T)
        e1.record(); torch.cuda.synchronize()
    print(f"N={n}: Segment.track {ms:.3f} ms per track ({ms * 10:.1f} us per element); apply_second_order kernel {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
