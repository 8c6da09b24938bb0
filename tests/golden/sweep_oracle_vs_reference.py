#!/usr/bin/env python3
"""Live sweep (build container only: imports the reference from /root/reference) of the oracle's first-order map builders against
the reference's `first_order_transfer_map` on thousands of drawn settings — log-uniform magnitudes over many decades, exact zeros,
and settings placed ON the regime boundaries of the restatement (focusing strengths around k L^2 = +-0.05 where the oracle switches
between series and closed forms, vanishing strengths and angles, energies close to the rest mass). Not a test (the reference does
not travel); the committed fixtures of generate_golden_random_maps.py are a 340-case sample of the same distribution.
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/sweep_oracle_vs_reference.py [cases per kind]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cheetah  # noqa: E402
from oracle import chx_oracle as oracle  # noqa: E402

f64 = {"dtype": torch.float64}
t = lambda v: torch.tensor(v, **f64)  # noqa: E731
rng = np.random.default_rng(7)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000


def mag(lo, hi, zero=0.1, signed=True):
    if rng.random() < zero:
        return 0.0
    v = float(np.exp(rng.uniform(np.log(lo), np.log(hi))))
    return -v if signed and rng.random() < 0.5 else v


def strength(L):
    """k1 such that k1 L^2 sits anywhere from 1e-14 to 30, often right at the series / closed-form switch"""
    r = rng.random()
    if r < 0.25:
        u = 0.05 * (1.0 + rng.uniform(-1e-6, 1e-6) * rng.choice([0, 1, 1e3, 1e5]))
        return float(rng.choice([-1, 1]) * u / (L * L))
    return mag(1e-14, 30.0) / (L * L)


elec = cheetah.Species("electron", **f64)
worst = {}


def check(kind, elem, params, energy):
    want = elem.first_order_transfer_map(t(energy), elec).detach().numpy()
    got = oracle.build_rmatrix(kind, np.asarray([params], dtype=np.float64), np.asarray([energy]))[0]
    if not np.isfinite(want).all():
        return
    denom = np.maximum(np.abs(want), 1e-3 * np.abs(want).max())
    err = float(np.max(np.abs(got - want) / denom))
    if err > worst.get(kind, (0.0,))[0]:
        worst[kind] = (err, params, energy)


for _ in range(N):
    E = float(np.exp(rng.uniform(np.log(6e5 if rng.random() < 0.1 else 5e6), np.log(2e10))))
    L = mag(1e-4, 5.0, zero=0.0, signed=False)
    check("drift", cheetah.Drift(length=t(L), **f64), [L], E)
    p = [L, strength(L), mag(1e-6, 1.5, zero=0.4), mag(1e-7, 5e-3, zero=0.4), mag(1e-7, 5e-3, zero=0.4)]
    check("quadrupole", cheetah.Quadrupole(length=t(p[0]), k1=t(p[1]), tilt=t(p[2]), misalignment=t([p[3], p[4]]), **f64), p, E)
    ang = mag(1e-9, 1.5, zero=0.05)
    p = [L, ang, strength(L) if rng.random() < 0.6 else 0.0, mag(1e-4, 0.4, zero=0.4), mag(1e-4, 0.4, zero=0.4), mag(1e-4, 1.5, zero=0.5),
         mag(0.1, 0.9, zero=0.4, signed=False), mag(0.1, 0.9, zero=0.4, signed=False), mag(1e-3, 0.05, zero=0.3, signed=False)]
    check("dipole", cheetah.Dipole(length=t(p[0]), angle=t(p[1]), k1=t(p[2]), dipole_e1=t(p[3]), dipole_e2=t(p[4]), tilt=t(p[5]),
                                   fringe_integral=t(p[6]), fringe_integral_exit=t(p[7]), gap=t(p[8]), **f64), p, E)
    p = [L, mag(1e-9, 1e-2)]
    check("hcor", cheetah.HorizontalCorrector(length=t(p[0]), angle=t(p[1]), **f64), p, E)
    check("vcor", cheetah.VerticalCorrector(length=t(p[0]), angle=t(p[1]), **f64), p, E)
    p = [L, mag(1e-9, 1e-2), mag(1e-9, 1e-2)]
    check("ccor", cheetah.CombinedCorrector(length=t(p[0]), horizontal_angle=t(p[1]), vertical_angle=t(p[2]), **f64), p, E)
    for ctype, kind in (("standing_wave", "cavity_sw"), ("traveling_wave", "cavity_tw")):
        V = mag(1.0, 0.4 * E if E < 1e9 else 5e7, zero=0.1)
        p = [L, V, float(rng.uniform(-180.0, 180.0)) if rng.random() < 0.8 else float(rng.choice([0.0, 90.0, -90.0, 180.0])),
             mag(1e8, 1.2e10, zero=0.0, signed=False)]
        check(kind, cheetah.Cavity(length=t(p[0]), voltage=t(p[1]), phase=t(p[2]), frequency=t(p[3]), cavity_type=ctype, **f64), p, E)
    p = [L, mag(1e-9, 5.0, zero=0.1), mag(1e-6, 5e-3, zero=0.5), mag(1e-6, 5e-3, zero=0.5)]
    check("solenoid", cheetah.Solenoid(length=t(p[0]), k=t(p[1]), misalignment=t([p[2], p[3]]), **f64), p, E)

# the second-order tensors (track_methods.py:80-281: seven special functions with removable singularities, utils/autograd.py) through
# the elements that use them (quadrupole.py:113-144, sextupole.py:91-122, dipole.py:397-430, drift.py:68-84)
def check_t(kind, elem, params, energy):
    want = elem.second_order_transfer_map(t(energy), elec).detach().numpy()
    if not np.isfinite(want).all():
        return
    got = oracle.build_ttensor(kind, [params], [energy])[0]
    denom = np.maximum(np.abs(want), 1e-6 * np.abs(want).max())
    err = float(np.max(np.abs(got - want) / denom))
    key = "T_" + kind
    if err > worst.get(key, (0.0,))[0]:
        worst[key] = (err, params, energy)


for _ in range(N):
    E = float(np.exp(rng.uniform(np.log(5e6), np.log(2e10))))
    L = mag(1e-3, 3.0, zero=0.0, signed=False)
    check_t("drift", cheetah.Drift(length=t(L), **f64), [L], E)
    p = [L, strength(L) if rng.random() < 0.9 else 0.0, mag(1e-6, 1.0, zero=0.4), mag(1e-7, 5e-3, zero=0.4), mag(1e-7, 5e-3, zero=0.4)]
    check_t("quadrupole", cheetah.Quadrupole(length=t(p[0]), k1=t(p[1]), tilt=t(p[2]), misalignment=t([p[3], p[4]]), **f64), p, E)
    p = [L, mag(1e-6, 500.0, zero=0.1), mag(1e-6, 1.0, zero=0.4), mag(1e-7, 5e-3, zero=0.4), mag(1e-7, 5e-3, zero=0.4)]
    check_t("sextupole", cheetah.Sextupole(length=t(p[0]), k2=t(p[1]), tilt=t(p[2]), misalignment=t([p[3], p[4]]), **f64), p, E)
    ang = mag(1e-9, 1.0, zero=0.05)
    k1 = strength(L) if rng.random() < 0.6 else 0.0
    if rng.random() < 0.1:
        k1 = -(ang / L) ** 2                                                          # kx^2 = k1 + hx^2 = 0 exactly (or one rounding off it)
    # (track_methods.py:129-142: j3 is evaluated from its raw formula, "no proper limit exists" — for |kx^2| L^2 below ~1e-2 its
    # 1 / kx2^3 amplifies the rounding of sx, cx beyond any tolerance in the reference itself: T566 of such a dipole is noise there,
    # and the restatement's noise is another; those draws are left out)
    kx2 = k1 + (ang / L) ** 2
    if kx2 != 0.0 and abs(kx2) * L * L < 1e-2:
        continue
    p = [L, ang, k1, mag(1e-4, 0.4, zero=0.4), mag(1e-4, 0.4, zero=0.4), mag(1e-4, 1.0, zero=0.5), mag(0.1, 0.9, zero=0.4, signed=False),
         mag(0.1, 0.9, zero=0.4, signed=False), mag(1e-3, 0.05, zero=0.3, signed=False)]
    check_t("dipole", cheetah.Dipole(length=t(p[0]), angle=t(p[1]), k1=t(p[2]), dipole_e1=t(p[3]), dipole_e2=t(p[4]), tilt=t(p[5]),
                                     fringe_integral=t(p[6]), fringe_integral_exit=t(p[7]), gap=t(p[8]), **f64), p, E)

# per-particle paths: an active cavity (cavity.py:100-251) and the Bmad-X drift-kick-drift maps of drifts and quadrupoles
# (utils/bmadx.py, quadrupole.py:146-215) on a small drawn beam
def beam_of(n, E):
    return cheetah.ParticleBeam.from_parameters(num_particles=n, sigma_x=t(mag(1e-5, 2e-3, 0, False)), sigma_y=t(mag(1e-5, 2e-3, 0, False)),
                                                sigma_px=t(mag(1e-6, 2e-4, 0, False)), sigma_py=t(mag(1e-6, 2e-4, 0, False)),
                                                sigma_tau=t(mag(1e-6, 1e-3, 0, False)), sigma_p=t(mag(1e-5, 1e-2, 0, False)),
                                                mu_x=t(mag(1e-6, 1e-3)), mu_y=t(mag(1e-6, 1e-3)), energy=t(E), **f64)


def note(key, got, want, params, energy):
    scale = np.abs(want).max(axis=0) + 1e-300
    err = float((np.abs(got - want) / scale).max())
    if err > worst.get(key, (0.0,))[0]:
        worst[key] = (err, params, energy)


torch.manual_seed(3)
for _ in range(max(N // 5, 50)):
    E = float(np.exp(rng.uniform(np.log(5e6), np.log(5e9))))
    L = mag(1e-2, 2.0, zero=0.0, signed=False)
    b = beam_of(64, E)
    x = b.particles.numpy()[None]
    for ctype, kind in (("standing_wave", "cavity_sw"), ("traveling_wave", "cavity_tw")):
        p = [L, mag(1e3, 0.4 * E if E < 1e9 else 5e7, zero=0.1), float(rng.uniform(-180.0, 180.0)), mag(1e8, 1.2e10, zero=0.0, signed=False)]
        cav = cheetah.Cavity(length=t(p[0]), voltage=t(p[1]), phase=t(p[2]), frequency=t(p[3]), cavity_type=ctype, **f64)
        out = cav.track(b)
        if not torch.isfinite(out.particles).all():
            continue
        R = oracle.build_rmatrix(kind, [p], [E])
        coeffs, e_out = oracle.cavity_coeffs([p], [E])
        note("track_" + kind, oracle.cavity_track(x, R, coeffs)[0], out.particles.numpy(), p, E)
        note("energy_" + kind, np.asarray(e_out).reshape(1, 1), out.energy.numpy().reshape(1, 1), p, E)
    steps = int(rng.integers(1, 6))
    p = [L, strength(L) if rng.random() < 0.9 else 0.0, mag(1e-6, 0.5, zero=0.4), mag(1e-7, 2e-3, zero=0.4), mag(1e-7, 2e-3, zero=0.4)]
    q = cheetah.Quadrupole(length=t(p[0]), k1=t(p[1]), tilt=t(p[2]), misalignment=t([p[3], p[4]]), num_steps=steps,
                           tracking_method="drift_kick_drift", **f64)
    want = q.track(b).particles.numpy()
    if np.isfinite(want).all():
        note("dkd_quadrupole", oracle.dkd_track("quadrupole", x, [p], [E], num_steps=steps)[0][0], want, p + [steps], E)
    d = cheetah.Drift(length=t(L), tracking_method="drift_kick_drift", **f64)
    note("dkd_drift", oracle.dkd_track("drift", x, [[L]], [E])[0][0], d.track(b).particles.numpy(), [L], E)

for kind, (err, params, energy) in sorted(worst.items()):
    print(f"{kind:12s} worst relative difference {err:.2e}  at params {params} energy {energy:.6g}")
