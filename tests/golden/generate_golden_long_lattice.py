#!/usr/bin/env python3
"""A beamline of 700 mergeable elements (more than one persistent device plan of this engine takes) with a CustomTransferMap in
the middle, tracked by the reference -> tests/golden/long_lattice.npz: element list as JSON, 400 incoming particles and the
reference's `Segment.track` result in float64 (particles, s), a ParameterBeam's mu / cov, and three beams in one ParticleBeam.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_long_lattice.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import cheetah  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
f64 = {"dtype": torch.float64}
rng = np.random.default_rng(20261001)
u = lambda lo, hi: float(rng.uniform(lo, hi))  # noqa: E731


def build(module, spec, fk):
    kind, kw = spec
    args = {k: (torch.tensor(v, **fk) if isinstance(v, (float, list)) else v) for k, v in kw.items()}
    return getattr(module, kind)(**args, **fk)


if __name__ == "__main__":
    specs = []
    for i in range(175):
        k1 = u(0.8, 1.6) * (1 if i % 2 == 0 else -1)
        q = {"length": 0.2, "k1": k1}
        if i % 7 == 0:
            q["misalignment"] = [u(-1e-4, 1e-4), u(-1e-4, 1e-4)]
        if i % 11 == 0:
            q["tilt"] = u(-0.05, 0.05)
        specs += [["Quadrupole", q], ["Drift", {"length": u(0.6, 1.0)}],
                  [str(rng.choice(["HorizontalCorrector", "VerticalCorrector"])), {"length": 0.05, "angle": u(-2e-5, 2e-5)}],
                  (["Dipole", {"length": 0.2, "angle": u(-2e-3, 2e-3), "dipole_e1": u(-1e-3, 1e-3)}] if i % 9 == 0 else
                   ["Marker", {}] if i % 5 == 0 else ["Drift", {"length": u(0.05, 0.2)}])]
    m = np.eye(7)
    m[0, 1], m[2, 3], m[0, 6], m[3, 2] = 0.4, 0.3, 2e-5, -0.1
    specs[350] = ["CustomTransferMap", {"predefined_transfer_map": m.tolist(), "length": 0.25}]
    seg = cheetah.Segment([build(cheetah, s, f64) for s in specs])
    energy = torch.tensor(1.3e8, **f64)
    beam = cheetah.ParticleBeam.from_twiss(beta_x=torch.tensor(6.0, **f64), beta_y=torch.tensor(4.0, **f64), alpha_x=torch.tensor(0.3, **f64),
                                           emittance_x=torch.tensor(2e-9, **f64), emittance_y=torch.tensor(2e-9, **f64),
                                           sigma_p=torch.tensor(5e-4, **f64), sigma_tau=torch.tensor(1e-4, **f64), energy=energy,
                                           num_particles=400, **f64)
    out = seg.track(beam)
    pb = beam.as_parameter_beam()
    pout = seg.track(pb)
    parts = beam.particles.unsqueeze(0) * torch.tensor([0.5, 1.0, 1.5], **f64).reshape(3, 1, 1)
    parts[..., 6] = 1.0
    many = seg.track(cheetah.ParticleBeam(parts, energy, **f64))
    arrays = {"spec": np.asarray(json.dumps(specs)), "energy": energy.numpy(), "in": beam.particles.numpy(), "out": out.particles.numpy(),
              "s_out": out.s.numpy(), "pb_mu_in": pb.mu.numpy(), "pb_cov_in": pb.cov.numpy(), "pb_mu": pout.mu.numpy(),
              "pb_cov": pout.cov.numpy(), "many_out": many.particles.numpy()}
    print(len(specs), "elements; sigma_x out", float(out.sigma_x), "s", float(out.s), "finite", bool(torch.isfinite(out.particles).all()))
    np.savez_compressed(os.path.join(OUT, "long_lattice.npz"), **arrays)
