#!/usr/bin/env python3
"""Every element class inside a 100-element FODO (10 instances spread over it), 1e5 particles, fp32: us per Segment.track; how
far each is from the plain lattice."""
import os, sys, time, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
warnings.simplefilter("ignore")
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)


def timeit(fn, reps=20, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def cell(i):
    return [ca.Quadrupole(t(0.2), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.8), **kw)]


makers = {
    "(plain)": None,
    "Dipole": lambda: ca.Dipole(t(0.2), angle=t(1e-3), **kw),
    "RBend": lambda: ca.RBend(t(0.2), angle=t(1e-3), **kw),
    "Solenoid": lambda: ca.Solenoid(t(0.1), k=t(0.2), **kw),
    "Undulator": lambda: ca.Undulator(t(0.3), **kw),
    "HorizontalCorrector": lambda: ca.HorizontalCorrector(t(0.05), angle=t(1e-5), **kw),
    "CombinedCorrector": lambda: ca.CombinedCorrector(t(0.05), horizontal_angle=t(1e-5), vertical_angle=t(2e-5), **kw),
    "Sextupole": lambda: ca.Sextupole(t(0.1), k2=t(1.0), **kw),
    "TransverseDeflectingCavity": lambda: ca.TransverseDeflectingCavity(t(0.2), voltage=t(1e5), phase=t(0.0), frequency=t(3e9), **kw),
    "Cavity": lambda: ca.Cavity(t(1.0), voltage=t(1e7), phase=t(0.0), frequency=t(1.3e9), **kw),
    "Quadrupole drift_kick_drift": lambda: ca.Quadrupole(t(0.2), k1=t(1.0), tracking_method="drift_kick_drift", **kw),
    "Quadrupole second_order": lambda: ca.Quadrupole(t(0.2), k1=t(1.0), tracking_method="second_order", **kw),
    "Marker": lambda: ca.Marker(**kw),
    "BPM active": lambda: ca.BPM(is_active=True, **kw),
    "Aperture active": lambda: ca.Aperture(x_max=t(5e-3), y_max=t(5e-3), **kw),
    "Screen active": lambda: ca.Screen(resolution=(64, 64), is_active=True, **kw),
    "SpaceChargeKick 32^3": lambda: ca.SpaceChargeKick(t(0.1), **kw),
    "CustomTransferMap": lambda: ca.CustomTransferMap(torch.eye(7, **kw), length=t(0.1), **kw),
}
beam = ca.ParticleBeam.from_parameters(num_particles=100_000, energy=t(1e8), total_charge=t(1e-10), **kw)
pbeam = ca.ParameterBeam.from_parameters(energy=t(1e8), **kw)
for name, mk in makers.items():
    els = []
    for i in range(50):
        els += cell(i)
        if mk is not None and i % 5 == 2:
            els.append(mk())
    seg = ca.Segment(els)
    with torch.no_grad():
        try:
            a = timeit(lambda: seg.track(beam), reps=10)
        except Exception as exc:  # noqa: BLE001
            a = float("nan"); print("   ", name, "ParticleBeam:", type(exc).__name__, str(exc)[:100])
        try:
            b = timeit(lambda: seg.track(pbeam), reps=10)
        except Exception as exc:  # noqa: BLE001
            b = float("nan")
    print(f"10 x {name:30s}: ParticleBeam {a:9.1f} us   ParameterBeam {b:9.1f} us", flush=True)

# ---- the same plain lattice with its SETTINGS in other states ---------------------------------------------------------------
print("settings states (100-element FODO, 1e5 particles / ParameterBeam, us per track under no_grad):")


def fodo(k1_of):
    els = []
    for i in range(50):
        els += [ca.Quadrupole(t(0.2), k1=k1_of(i), **kw), ca.Drift(t(0.8), **kw)]
    return ca.Segment(els)


states = {
    "plain scalars": lambda i: t(4.2 if i % 2 == 0 else -4.2),
    "one nn.Parameter strength": lambda i: torch.nn.Parameter(t(4.2)) if i == 7 else t(4.2 if i % 2 == 0 else -4.2),
    "every strength an nn.Parameter": lambda i: torch.nn.Parameter(t(4.2 if i % 2 == 0 else -4.2)),
    "one strength requires_grad (buffer)": lambda i: t(4.2).requires_grad_() if i == 7 else t(4.2 if i % 2 == 0 else -4.2),
    "one strength of shape (1,)": lambda i: t([4.2]) if i == 7 else t(4.2 if i % 2 == 0 else -4.2),
    "strengths views of one settings tensor": (lambda base: (lambda i: base[i]))(torch.randn(50, **kw)),
}
for name, k1_of in states.items():
    seg = fodo(k1_of)
    with torch.no_grad():
        a = timeit(lambda: seg.track(beam), reps=10)
        b = timeit(lambda: seg.track(pbeam), reps=10)
    print(f"   {name:42s}: ParticleBeam {a:9.1f} us   ParameterBeam {b:9.1f} us", flush=True)
