#!/usr/bin/env python3
"""Timings of the five BASELINE.json configs on one MI355X (secondary to bench.py; numbers go to DESIGN.md).
usage: python benchmarks/run_configs.py [c1 rl c3 c4 c5 screen]   -> one JSON line per config"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402

DEV = "cuda"


def t(v, dt):
    return torch.tensor(v, dtype=dt, device=DEV)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def ares_subcell(dt, k1):
    kw = {"dtype": dt, "device": DEV}
    return ca.Segment([
        ca.Marker(name="AREASOLA1", **kw), ca.Drift(t(0.17504, dt)),
        ca.Quadrupole(t(0.122, dt), k1=k1, name="AREAMQZM1", **kw), ca.Drift(t(0.428, dt)),
        ca.Quadrupole(t(0.122, dt), k1=t(-14.3, dt), name="AREAMQZM2", **kw), ca.Drift(t(0.204, dt)),
        ca.VerticalCorrector(t(0.02, dt), angle=t(9e-5, dt), name="AREAMCVM1", **kw), ca.Drift(t(0.204, dt)),
        ca.Quadrupole(t(0.122, dt), k1=t(3.142, dt), name="AREAMQZM3", **kw), ca.Drift(t(0.179, dt)),
        ca.HorizontalCorrector(t(0.02, dt), angle=t(-1e-4, dt), name="AREAMCHM1", **kw), ca.Drift(t(0.45, dt)),
        ca.Screen(resolution=(2448, 2040), pixel_size=t([3.5488e-6, 2.5003e-6], dt), name="AREABSCR1",
                  method="cloud-in-cell", **kw),
    ])


def c1():
    dt = torch.float64
    seg = ares_subcell(dt, t(8.2, dt))
    beam = ca.ParticleBeam.from_twiss(beta_x=t(3.14, dt), beta_y=t(42.0, dt), num_particles=10_000, dtype=dt, device=DEV)
    ms = timeit(lambda: seg.track(beam), 200)
    seg.AREABSCR1.is_active = True

    def f():
        seg.track(beam)
        return seg.AREABSCR1.reading

    ms2 = timeit(f, 100)
    return {"config": "C1 ARES 13-element segment, 1e4 particles, fp64", "track_ms": ms, "track_plus_cic_reading_ms": ms2,
            "steps_per_s": 1e4 * 13 / (ms * 1e-3)}


def rl():
    """The control loop the reference's README is about: change five magnet settings, track, read the screen."""
    dt = torch.float32
    seg = ares_subcell(dt, t(8.2, dt))
    seg.AREABSCR1.is_active = True
    beams = {"ParticleBeam": ca.ParticleBeam.from_twiss(beta_x=t(3.14, dt), beta_y=t(42.0, dt), num_particles=10_000, dtype=dt,
                                                        device=DEV),
             "ParameterBeam": ca.ParameterBeam.from_twiss(beta_x=t(3.14, dt), beta_y=t(42.0, dt), dtype=dt, device=DEV)}
    actions = torch.randn(300, 5, device=DEV, dtype=dt)
    res = {"config": "control loop on the ARES section: 5 settings + track + screen reading, 1e4 particles, fp32"}
    for name, beam in beams.items():
        counter = [0]

        def step():
            a = actions[counter[0] % 300]
            counter[0] += 1
            seg.AREAMQZM1.k1, seg.AREAMQZM2.k1, seg.AREAMCVM1.angle = a[0] * 10, a[1] * 10, a[2] * 1e-4
            seg.AREAMQZM3.k1, seg.AREAMCHM1.angle = a[3] * 10, a[4] * 1e-4
            seg.track(beam)
            return seg.AREABSCR1.reading

        res[f"{name}_ms_per_step"] = timeit(step, 500, 20)
    return res


def c3(B=4096, N=100_000):
    dt = torch.float32
    seg = ares_subcell(dt, torch.linspace(-30, 30, B, dtype=dt, device=DEV))
    beam = ca.ParticleBeam.from_parameters(num_particles=N, dtype=dt, device=DEV)
    out = {}
    ms = timeit(lambda: out.__setitem__("b", seg.track(beam)), 5, 2)
    o = out["b"]
    ms_sig = timeit(lambda: ca._ops.moments(o.particles, o.survival_probabilities), 5, 2)
    gbytes = B * N * 28 / 1e9
    del o, out["b"]
    ms_fused = timeit(lambda: seg.track_moments(beam), 5, 2)  # chx_track_moments: no (B, N, 7) output at all
    ms_alg = timeit(lambda: seg.track_moments(beam, exact=False), 20, 3)   # one pass over the shared beam + 7x7 algebra per setting
    return {"config": f"C3 k1 scan B={B} x N={N}, fp32, shared beam", "track_ms": ms, "all_moments_ms": ms_sig,
            "fused_track_moments_ms": ms_fused, "algebraic_track_moments_ms": ms_alg, "output_GB": gbytes, "write_GBps": gbytes / (ms * 1e-3),
            "steps_per_s": B * N * 13 / (ms * 1e-3), "fused_steps_per_s": B * N * 13 / (ms_fused * 1e-3)}


def c4(N=1_000_000, g=128):
    dt = torch.float32
    kw = {"dtype": dt, "device": DEV}
    els = []
    for i in range(10):
        els += [ca.Drift(t(0.1, dt)), ca.SpaceChargeKick(t(0.2, dt), grid_shape=(g, g, g), **kw), ca.Drift(t(0.1, dt)),
                ca.Quadrupole(t(0.1, dt), k1=t(4.2 if i % 2 == 0 else -4.2, dt), **kw), ca.Drift(t(0.1, dt))]
    seg = ca.Segment(els)
    beam = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=N, total_charge=t(1e-9, dt), energy=t(2.5e8, dt),
                                                radius_x=t(1e-3, dt), radius_y=t(1e-3, dt), radius_tau=t(1e-3, dt),
                                                sigma_px=t(1e-6, dt), sigma_py=t(1e-6, dt), sigma_p=t(1e-6, dt), **kw)
    ms = timeit(lambda: seg.track(beam), 10, 4)
    sc = els[1]
    ms_kick = timeit(lambda: sc.track(beam), 5, 2)
    return {"config": f"C4 50-element linac, 10 SpaceChargeKicks {g}^3, N={N}, fp32", "track_ms": ms,
            "single_kick_ms": ms_kick, "steps_per_s": N * 50 / (ms * 1e-3)}


def c5(N=1_000_000):
    dt = torch.float32
    kw = {"dtype": dt, "device": DEV}
    k1 = torch.nn.Parameter(t(3.142, dt))
    seg = ca.Segment([ca.Drift(t(1.0, dt)), ca.Quadrupole(t(0.2, dt), k1=k1, **kw), ca.Drift(t(1.0, dt)),
                      ca.Screen(is_active=True, name="scr", **kw)])
    beam = ca.ParticleBeam.from_parameters(num_particles=N, dtype=dt, device=DEV)
    res = {}

    def f():
        k1.grad = None
        with torch.no_grad():
            k1.add_(0.0)  # bump the version like an optimiser step would
        seg.track(beam)
        loss = seg.scr.get_read_beam().sigma_x
        loss.backward()
        res["sigma_x"], res["dk1"] = loss.detach(), k1.grad    # read back after the timed region (no per-step host sync)

    # the step is bound by the host (~0.14 ms of it, 60 us of kernels): a loop of 50 steps times a process whose interpreter and
    # allocator are still warming up (0.138 ms where the same loop a second later takes 0.114, benchmarks/c5_loop_profile.py); an
    # optimisation loop runs thousands of steps. Both are reported; `fwd_bwd_ms` is the sustained one.
    first = timeit(f, 50, 5)
    ms = timeit(f, 1000, 100)
    res = {k: float(v) for k, v in res.items()}
    return {"config": f"C5 d sigma_x(screen)/d k1, N={N}, fp32, forward+backward", "fwd_bwd_ms": ms, "fwd_bwd_ms_first_50_steps": first,
            "steps_timed": 1000, **res}


def screen(N=1_000_000):
    dt = torch.float32
    beam = ca.ParticleBeam.from_parameters(num_particles=N, dtype=dt, device=DEV)
    out = {}
    for method in ("cloud-in-cell", "histogram"):
        scr = ca.Screen(resolution=(2448, 2040), pixel_size=t([3.5488e-6, 2.5003e-6], dt), method=method,
                        is_active=True, dtype=dt, device=DEV)

        def f():
            scr.track(beam)
            return scr.reading

        out[method + "_ms"] = timeit(f, 10, 2)
    return {"config": f"Screen.reading 2448x2040, N={N}, fp32", **out}


if __name__ == "__main__":
    which = sys.argv[1:] or ["c1", "rl", "c3", "c4", "c5", "screen"]
    for w in which:
        print(json.dumps(globals()[w]()), flush=True)
