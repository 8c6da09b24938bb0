#!/usr/bin/env python3
"""ParameterBeams through drawn lattices with diagnostics -> tests/golden/parameter_beams_random.npz: ten beamlines of 5-9 linear
elements (the draws of generate_golden_random_structured.py), some with a VECTORISED quadrupole strength, corrector angle or
cavity voltage (shapes (3,) and (2, 1)), an active BPM in the middle and — for the unvectorised ones — an active Screen at the
end (drawn resolution, pixel size, binning, misalignment): the outgoing mu / cov / energy / s / total charge, the BPM reading,
the Screen's read beam and its image (the bivariate-normal density, screen.py:252-291), in float64. For the vectorised lattices
the fixture records that the reference's Screen raises NotImplementedError for a vectorised ParameterBeam.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_random_parameter_beams.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cheetah  # noqa: E402
import generate_golden_random_structured as S  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
f64 = {"dtype": torch.float64}
S.rng = np.random.default_rng(86420)
rng = S.rng


def build(module, spec, kw):
    kind, args = spec
    targs = {k: (tuple(v) if k == "resolution" else torch.tensor(v, **kw) if isinstance(v, (float, list)) else v) for k, v in args.items()}
    return getattr(module, kind)(**targs, **kw)


if __name__ == "__main__":
    arrays = {"n_cases": np.asarray(10)}
    for i in range(10):
        vectorised = i % 3 == 2
        spec = []
        while len(spec) < int(rng.integers(5, 10)):
            s = S.draw_simple()
            if s[0] in ("Marker", "BPM"):
                continue
            spec.append(s)
        if vectorised:
            shape = [(3,), (2, 1)][int(rng.integers(0, 2))]
            hit = False
            for kind, args in spec:
                key = {"Quadrupole": "k1", "HorizontalCorrector": "angle", "VerticalCorrector": "angle", "Cavity": "voltage",
                       "Solenoid": "k"}.get(kind)
                if key is not None and (not hit or rng.random() < 0.4):
                    args[key] = (args[key] * rng.uniform(0.5, 1.5, shape)).tolist()
                    hit = True
            if not hit:
                spec.append(["Quadrupole", {"length": 0.2, "k1": rng.uniform(-10, 10, shape).tolist()}])
        spec.insert(len(spec) // 2, ["BPM", {"is_active": True, "name": "bpm"}])
        if not vectorised:
            res = [int(rng.integers(40, 90)), int(rng.integers(30, 70))]
            spec.append(["Screen", {"resolution": res, "pixel_size": [S.u(2e-5, 2e-4), S.u(2e-5, 2e-4)],
                                    "binning": int(rng.choice([1, 2])), "misalignment": [S.u(-3e-4, 3e-4), S.u(-3e-4, 3e-4)],
                                    "is_active": True, "name": "screen"}])
        seg = cheetah.Segment([build(cheetah, s, f64) for s in spec])
        energy = float(np.exp(rng.uniform(np.log(3e7), np.log(3e9))))
        e_t = torch.tensor(energy * rng.uniform(0.8, 1.2, (3,)), **f64) if (vectorised and rng.random() < 0.5 and "(2, 1)" not in str(spec)) \
            else torch.tensor(energy, **f64)
        beam = cheetah.ParameterBeam.from_twiss(
            beta_x=torch.tensor(S.u(1.0, 20.0), **f64), alpha_x=torch.tensor(S.u(-2.0, 2.0), **f64),
            emittance_x=torch.tensor(S.u(1e-10, 1e-8), **f64), beta_y=torch.tensor(S.u(1.0, 20.0), **f64),
            alpha_y=torch.tensor(S.u(-2.0, 2.0), **f64), emittance_y=torch.tensor(S.u(1e-10, 1e-8), **f64),
            sigma_tau=torch.tensor(S.u(1e-5, 1e-3), **f64), sigma_p=torch.tensor(S.u(1e-4, 3e-3), **f64),
            cov_taup=torch.tensor(S.u(-1e-9, 1e-9), **f64), energy=e_t, total_charge=torch.tensor(S.u(1e-11, 1e-9), **f64), **f64)
        beam.mu[..., 0] += S.u(-2e-4, 2e-4)       # off axis, so that correctors and misaligned screens matter
        beam.mu[..., 2] += S.u(-2e-4, 2e-4)
        out = seg.track(beam)
        arrays[f"spec_{i}"] = np.asarray(json.dumps(spec))
        arrays[f"mu_in_{i}"], arrays[f"cov_in_{i}"] = beam.mu.numpy(), beam.cov.numpy()
        arrays[f"energy_in_{i}"], arrays[f"charge_in_{i}"] = beam.energy.numpy(), beam.total_charge.numpy()
        arrays[f"mu_out_{i}"], arrays[f"cov_out_{i}"] = out.mu.numpy(), out.cov.numpy()
        arrays[f"energy_out_{i}"], arrays[f"s_out_{i}"], arrays[f"charge_out_{i}"] = out.energy.numpy(), out.s.numpy(), out.total_charge.numpy()
        arrays[f"bpm_{i}"] = seg.bpm.reading.numpy()
        if not vectorised:
            arrays[f"image_{i}"] = seg.screen.reading.numpy()
            rb = seg.screen.get_read_beam()
            arrays[f"read_mu_{i}"], arrays[f"read_cov_{i}"] = rb.mu.numpy(), rb.cov.numpy()
            print(i, "image", arrays[f"image_{i}"].shape, "sum", float(arrays[f"image_{i}"].sum()), "max", float(arrays[f"image_{i}"].max()))
        else:
            scr = cheetah.Screen(is_active=True, **f64)
            scr.track(out)
            try:
                scr.reading
                arrays[f"screen_raises_{i}"] = np.asarray("")
            except NotImplementedError as err:
                arrays[f"screen_raises_{i}"] = np.asarray(type(err).__name__)
            print(i, "vectorised: mu", out.mu.shape, "energy", out.energy.shape, "screen:", str(arrays[f"screen_raises_{i}"]))
    np.savez_compressed(os.path.join(OUT, "parameter_beams_random.npz"), **arrays)
    print("wrote", len(arrays), "arrays")
