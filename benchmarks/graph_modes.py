#!/usr/bin/env python3
"""Small-beam workloads replayed from a device graph (torch.cuda.CUDAGraph = hipGraph): the host side of a step is one graph launch.
  c1       README segment (13 elements, 1e4 particles, fp64): track + cloud-in-cell screen reading
  control  the control loop on that segment (fp32): five magnet settings written IN PLACE from an action tensor, track, screen reading
  control_parameter_beam  the same with a ParameterBeam (moments only: the reading is the bivariate normal image)
  control_assigned / control_assigned_parameter_beam  the README's style (README.md:73-77 of the reference): the five settings ASSIGNED as
           new tensors every step (`seg.AREAMQZM1.k1 = scaled[0]`) — eager only: a recording holds addresses, new tensors have new ones
  linac    16 cells [Drift, Quadrupole, active Cavity] (1e4 particles, fp32): every cavity is a map of the energy it receives
  c4       50-element linac with 10 space-charge kicks (128^3, 1e6 particles): the chain with its side stream as graph edges
usage: python benchmarks/graph_modes.py c1|control|control_parameter_beam|linac|c4   -> one JSON line {"graph_mode": {...}}"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import run_configs as rc  # noqa: E402
import cheetah_amd as ca  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c1"


def timed(fn, reps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


if which in ("control_assigned", "control_assigned_parameter_beam"):
    dt = torch.float32
    seg = rc.ares_subcell(dt, rc.t(8.2, dt))
    seg.AREABSCR1.is_active = True
    if which.endswith("parameter_beam"):
        beam = ca.ParameterBeam.from_twiss(beta_x=rc.t(3.14, dt), beta_y=rc.t(42.0, dt), dtype=dt, device="cuda")
    else:
        beam = ca.ParticleBeam.from_twiss(beta_x=rc.t(3.14, dt), beta_y=rc.t(42.0, dt), num_particles=10_000, dtype=dt, device="cuda")
    action = torch.randn(5, device="cuda", dtype=dt)
    scale = torch.tensor([10.0, 10.0, 1e-4, 10.0, 1e-4], device="cuda", dtype=dt)
    settings = [seg.AREAMQZM1.k1, seg.AREAMQZM2.k1, seg.AREAMCVM1.angle, seg.AREAMQZM3.k1, seg.AREAMCHM1.angle]

    def assigned():
        scaled = (action * scale).unbind(0)                   # the caller's ops: one multiply, five views
        seg.AREAMQZM1.k1, seg.AREAMQZM2.k1, seg.AREAMCVM1.angle, seg.AREAMQZM3.k1, seg.AREAMCHM1.angle = scaled
        seg.track(beam)
        return seg.AREABSCR1.reading

    def in_place():
        scaled = action * scale
        for i, target in enumerate(settings):
            target.copy_(scaled[i])
        seg.track(beam)
        return seg.AREABSCR1.reading

    with torch.no_grad():
        a_us = timed(assigned, 2000, 50)
        img_a = assigned().clone()
        seg.AREAMQZM1.k1, seg.AREAMQZM2.k1, seg.AREAMCVM1.angle, seg.AREAMQZM3.k1, seg.AREAMCHM1.angle = [t.clone() for t in (action * scale).unbind(0)]
        settings = [seg.AREAMQZM1.k1, seg.AREAMQZM2.k1, seg.AREAMCVM1.angle, seg.AREAMQZM3.k1, seg.AREAMCHM1.angle]
        p_us = timed(in_place, 2000, 50)
        img_p = in_place()
        same = bool(torch.allclose(img_a, img_p, rtol=1e-5, atol=1e-6 * float(img_p.max())))
    print(json.dumps({"graph_mode": {"workload": which, "eager_assigned_us": a_us, "eager_in_place_us": p_us, "readings_equal": same}}))
    sys.exit(0)

if which in ("c1", "control", "control_parameter_beam"):
    dt = torch.float64 if which == "c1" else torch.float32
    seg = rc.ares_subcell(dt, rc.t(8.2, dt))
    seg.AREABSCR1.is_active = True
    if which == "control_parameter_beam":
        beam = ca.ParameterBeam.from_twiss(beta_x=rc.t(3.14, dt), beta_y=rc.t(42.0, dt), dtype=dt, device="cuda")
    else:
        beam = ca.ParticleBeam.from_twiss(beta_x=rc.t(3.14, dt), beta_y=rc.t(42.0, dt), num_particles=10_000, dtype=dt, device="cuda")
    action = torch.randn(5, device="cuda", dtype=dt)
    scale = torch.tensor([10.0, 10.0, 1e-4, 10.0, 1e-4], device="cuda", dtype=dt)
    settings = [seg.AREAMQZM1.k1, seg.AREAMQZM2.k1, seg.AREAMCVM1.angle, seg.AREAMQZM3.k1, seg.AREAMCHM1.angle]

    def step():
        if which.startswith("control"):
            scaled = action * scale
            for i, target in enumerate(settings):
                target.copy_(scaled[i])                       # in place: the lattice's tensors (and their addresses) stay
        seg.track(beam)
        return seg.AREABSCR1.reading
    reps = 2000
elif which == "linac":
    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    els = []
    for i in range(16):
        els += [ca.Drift(rc.t(0.3, dt), **kw), ca.Quadrupole(rc.t(0.2, dt), k1=rc.t(3.0 if i % 2 else -3.0, dt), **kw),
                ca.Cavity(rc.t(1.0377, dt), voltage=rc.t(18e6, dt), phase=rc.t(-10.0, dt), frequency=rc.t(1.3e9, dt), **kw)]
    seg = ca.Segment(els)
    beam = ca.ParticleBeam.from_parameters(num_particles=10_000, energy=rc.t(1e8, dt), **kw)

    def step():
        return seg.track(beam).particles
    reps = 300
else:
    dt = torch.float32
    kw = {"dtype": dt, "device": "cuda"}
    els = []
    for i in range(10):
        els += [ca.Drift(rc.t(0.1, dt)), ca.SpaceChargeKick(rc.t(0.2, dt), grid_shape=(128, 128, 128), **kw), ca.Drift(rc.t(0.1, dt)),
                ca.Quadrupole(rc.t(0.1, dt), k1=rc.t(4.2 if i % 2 == 0 else -4.2, dt), **kw), ca.Drift(rc.t(0.1, dt))]
    seg = ca.Segment(els)
    beam = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=1_000_000, total_charge=rc.t(1e-9, dt), energy=rc.t(2.5e8, dt),
                                                radius_x=rc.t(1e-3, dt), radius_y=rc.t(1e-3, dt), radius_tau=rc.t(1e-3, dt),
                                                sigma_px=rc.t(1e-6, dt), sigma_py=rc.t(1e-6, dt), sigma_p=rc.t(1e-6, dt), **kw)

    def step():
        return seg.track(beam).particles
    reps = 50

with torch.no_grad():
    eager_us = timed(step, reps, 20 if which != "c4" else 6)       # (also takes the chain guard past its sampling tracks)
    ref = step().clone()
    captured = ca.graph.capture(step)
    graph, out = captured.graph, captured.outputs
    captured()
    torch.cuda.synchronize()
    same = bool(torch.allclose(out, ref, rtol=1e-4 if which == "c4" else 1e-6, atol=0.0 if which != "c4" else 1e-9))
    follows = None
    if which.startswith("control"):
        action.copy_(torch.tensor([0.3, -0.2, 0.5, 0.1, -0.4], device="cuda", dtype=dt))
        graph.replay()
        torch.cuda.synchronize()
        replayed = out.clone()
        now = step()                 # (the image is a sum of float atomics: equal to rounding, not bit for bit)
        follows = bool(torch.allclose(replayed, now, rtol=1e-5, atol=1e-6 * float(now.max()))) and \
            float((replayed - ref).abs().max()) > 1e-3 * float(ref.max())
    replay_us = timed(graph.replay, reps, 20 if which != "c4" else 5)
print(json.dumps({"graph_mode": {"workload": which, "eager_us": eager_us, "graph_replay_us": replay_us, "replay_equals_eager": same,
                                 "replay_follows_in_place_settings": follows}}))
