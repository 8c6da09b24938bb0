#!/usr/bin/env python3
"""Random sweep of the reference's first-order map builders -> tests/golden/maps_random.npz (per element kind: params_<kind> (n, P), energy_<kind> (n,), species_<kind> (n, 2) = [mass in eV, elementary charges], R_<kind> (n, 7, 7)).
Where maps.npz holds hand-picked settings, this one draws them: magnitudes log-uniform over several decades, random signs,
exact zeros mixed in, electron and proton beams from 5 MeV to 20 GeV. Run in the build container:
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_random_maps.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import cheetah  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
f64 = {"dtype": torch.float64}
t = lambda v: torch.tensor(v, **f64)  # noqa: E731
rng = np.random.default_rng(20260929)


def mag(lo, hi, zero=0.15, signed=True):
    """log-uniform magnitude in [lo, hi], random sign, exactly 0 with probability `zero`"""
    if rng.random() < zero:
        return 0.0
    v = float(np.exp(rng.uniform(np.log(lo), np.log(hi))))
    return -v if signed and rng.random() < 0.5 else v


def length():
    return mag(1e-3, 5.0, zero=0.0, signed=False)


cases = []


def add(kind, elem, params, energy, species):
    R = elem.first_order_transfer_map(t(energy), species)
    assert torch.isfinite(R).all(), (kind, params, energy)
    cases.append({"kind": kind, "params": np.asarray(params, dtype=np.float64), "energy": np.asarray(energy, dtype=np.float64),
                  "species": np.asarray([float(species.mass_eV), float(species.num_elementary_charges)]),
                  "R": R.detach().numpy().astype(np.float64)})


elec, prot = cheetah.Species("electron", **f64), cheetah.Species("proton", **f64)
N = 40
for _ in range(N):
    sp = prot if rng.random() < 0.25 else elec
    E = float(np.exp(rng.uniform(np.log(3e9 if sp is prot else 5e6), np.log(2e10))))
    L = length()
    add("drift", cheetah.Drift(length=t(L), **f64), [L], E, sp)
    p = [length(), mag(1e-6, 50.0), mag(1e-3, 1.5, zero=0.4), mag(1e-6, 5e-3, zero=0.4), mag(1e-6, 5e-3, zero=0.4)]
    add("quadrupole", cheetah.Quadrupole(length=t(p[0]), k1=t(p[1]), tilt=t(p[2]), misalignment=t([p[3], p[4]]), **f64), p, E, sp)
    p = [length(), mag(1e-6, 1.5), mag(1e-3, 20.0, zero=0.5), mag(1e-3, 0.4, zero=0.4), mag(1e-3, 0.4, zero=0.4),
         mag(1e-3, 1.5, zero=0.5), mag(0.1, 0.9, zero=0.4, signed=False), mag(0.1, 0.9, zero=0.4, signed=False),
         mag(1e-3, 0.05, zero=0.3, signed=False)]
    add("dipole", cheetah.Dipole(length=t(p[0]), angle=t(p[1]), k1=t(p[2]), dipole_e1=t(p[3]), dipole_e2=t(p[4]), tilt=t(p[5]),
                                 fringe_integral=t(p[6]), fringe_integral_exit=t(p[7]), gap=t(p[8]), **f64), p, E, sp)
    p = [length(), mag(1e-7, 1e-2)]
    add("hcor", cheetah.HorizontalCorrector(length=t(p[0]), angle=t(p[1]), **f64), p, E, sp)
    p = [length(), mag(1e-7, 1e-2)]
    add("vcor", cheetah.VerticalCorrector(length=t(p[0]), angle=t(p[1]), **f64), p, E, sp)
    p = [length(), mag(1e-7, 1e-2), mag(1e-7, 1e-2)]
    add("ccor", cheetah.CombinedCorrector(length=t(p[0]), horizontal_angle=t(p[1]), vertical_angle=t(p[2]), **f64), p, E, sp)
    for ctype, kind in (("standing_wave", "cavity_sw"), ("traveling_wave", "cavity_tw")):
        # accelerating or mildly decelerating: the outgoing energy stays well above the rest mass
        V = mag(1e4, 0.4 * E if E < 1e9 else 5e7, zero=0.15)
        p = [length(), V, float(rng.uniform(-180.0, 180.0)), mag(1e8, 1.2e10, zero=0.0, signed=False)]
        if sp is prot:
            continue            # the reference's cavity is exercised with electrons only
        add(kind, cheetah.Cavity(length=t(p[0]), voltage=t(p[1]), phase=t(p[2]), frequency=t(p[3]), cavity_type=ctype, **f64), p, E, sp)
    p = [length(), mag(1e-3, 5.0), mag(1e-6, 5e-3, zero=0.5), mag(1e-6, 5e-3, zero=0.5)]
    add("solenoid", cheetah.Solenoid(length=t(p[0]), k=t(p[1]), misalignment=t([p[2], p[3]]), **f64), p, E, sp)

arrays = {}
for kind in sorted({c["kind"] for c in cases}):
    rows = [c for c in cases if c["kind"] == kind]
    arrays[f"params_{kind}"] = np.stack([c["params"] for c in rows])
    arrays[f"energy_{kind}"] = np.stack([c["energy"] for c in rows])
    arrays[f"species_{kind}"] = np.stack([c["species"] for c in rows])
    arrays[f"R_{kind}"] = np.stack([c["R"] for c in rows])
path = os.path.join(OUT, "maps_random.npz")
np.savez_compressed(path, **arrays)
print(len(cases), "cases ->", path, f"{os.path.getsize(path) / 1024:.1f} KiB")
