#!/usr/bin/env python3
"""Drawn lattices with ACTIVE beam position monitors, ACTIVE apertures and cavities between the magnets, tracked by the reference
-> tests/golden/diagnostics_stretch.npz. Eight lattices of 18-40 elements; per lattice the element list as JSON, 1500 incoming
particles with drawn survival probabilities, and what the reference's `Segment.track` leaves in float64: outgoing particles,
survival probabilities, energy, s and every monitor's reading; for a ParameterBeam (lattices without apertures... an aperture only
warns there) mu, cov, energy and the readings; and, for two lattices, a vectorised ParticleBeam of three beams.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_diagnostics_stretch.py
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import cheetah  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
f64 = {"dtype": torch.float64}
rng = np.random.default_rng(20260930)


def u(lo, hi):
    return float(rng.uniform(lo, hi))


def draw(with_cavities, with_apertures):
    kinds = ["Drift", "Drift", "Quadrupole", "Quadrupole", "HorizontalCorrector", "VerticalCorrector", "Dipole", "BPM", "BPM", "Marker"]
    if with_cavities:
        kinds += ["Cavity"]
    if with_apertures:
        kinds += ["Aperture", "Aperture"]
    kind = str(rng.choice(kinds))
    if kind == "Drift":
        return [kind, {"length": u(0.05, 1.0)}]
    if kind == "Quadrupole":
        kw = {"length": u(0.05, 0.4), "k1": u(-12.0, 12.0)}
        if rng.random() < 0.3:
            kw["misalignment"] = [u(-2e-4, 2e-4), u(-2e-4, 2e-4)]
        if rng.random() < 0.2:
            kw["tilt"] = u(-0.3, 0.3)
        return [kind, kw]
    if kind in ("HorizontalCorrector", "VerticalCorrector"):
        return [kind, {"length": u(0.01, 0.2), "angle": u(-3e-4, 3e-4)}]
    if kind == "Dipole":
        return [kind, {"length": u(0.2, 0.8), "angle": u(-0.05, 0.05), "dipole_e1": u(-0.05, 0.05)}]
    if kind == "Cavity":
        return [kind, {"length": u(0.3, 1.1), "voltage": u(2e6, 1.5e7), "phase": u(-40.0, 40.0), "frequency": 1.3e9,
                       "cavity_type": str(rng.choice(["standing_wave", "traveling_wave"]))}]
    if kind == "BPM":
        return [kind, {"is_active": True, "misalignment": [u(-3e-4, 3e-4), u(-3e-4, 3e-4)]}]
    if kind == "Aperture":
        return [kind, {"x_max": u(4e-4, 2e-3), "y_max": u(4e-4, 2e-3), "shape": str(rng.choice(["rectangular", "elliptical"])),
                       "is_active": True}]
    return [kind, {}]


def build(module, spec, fk):
    kind, kw = spec
    args = {k: (torch.tensor(v, **fk) if isinstance(v, (float, list)) else v) for k, v in kw.items()}
    return getattr(module, kind)(**args, **fk)


if __name__ == "__main__":
    arrays = {"n_lattices": np.asarray(8)}
    torch.manual_seed(99)            # (lattice 0's beam is drawn before the first per-lattice seed below)
    for i in range(8):
        with_cav, with_ap = i % 2 == 1, i >= 4
        specs = [draw(with_cav, with_ap) for _ in range(int(rng.integers(18, 41)))]
        if not any(s[0] == "BPM" for s in specs):
            specs.insert(len(specs) // 2, ["BPM", {"is_active": True, "misalignment": [1e-4, -1e-4]}])
        specs.append(["BPM", {"is_active": True, "misalignment": [0.0, 0.0]}])     # one monitor reads the outgoing beam
        seg = cheetah.Segment([build(cheetah, s, f64) for s in specs])
        energy = torch.tensor(u(2e7, 2e8), **f64)
        beam = cheetah.ParticleBeam.from_parameters(num_particles=1500, mu_x=torch.tensor(u(-2e-4, 2e-4), **f64),
                                                    mu_y=torch.tensor(u(-2e-4, 2e-4), **f64), sigma_x=torch.tensor(u(1e-4, 4e-4), **f64),
                                                    sigma_y=torch.tensor(u(1e-4, 4e-4), **f64), sigma_px=torch.tensor(2e-5, **f64),
                                                    sigma_py=torch.tensor(2e-5, **f64), sigma_p=torch.tensor(1e-3, **f64),
                                                    sigma_tau=torch.tensor(1e-4, **f64), energy=energy, **f64)
        torch.manual_seed(100 + i)
        w = torch.rand(1500, **f64)
        w[torch.rand(1500) < 0.05] = 0.0
        beam = cheetah.ParticleBeam(beam.particles, energy, particle_charges=beam.particle_charges, survival_probabilities=w, **f64)
        out = seg.track(beam)
        bpms = [e for e in seg.elements if isinstance(e, cheetah.BPM)]
        arrays[f"lat{i}_spec"] = np.asarray(json.dumps(specs))
        arrays[f"lat{i}_energy"] = energy.numpy()
        arrays[f"lat{i}_in"] = beam.particles.numpy()
        arrays[f"lat{i}_w"] = w.numpy()
        arrays[f"lat{i}_q"] = beam.particle_charges.numpy()
        arrays[f"lat{i}_out"] = out.particles.numpy()
        arrays[f"lat{i}_w_out"] = out.survival_probabilities.numpy()
        arrays[f"lat{i}_energy_out"] = out.energy.numpy()
        arrays[f"lat{i}_s_out"] = out.s.numpy()
        arrays[f"lat{i}_readings"] = torch.stack([b.reading for b in bpms]).numpy()
        # ParameterBeam (the reference's apertures only warn for it)
        pb = cheetah.ParameterBeam.from_parameters(mu_x=torch.tensor(1e-4, **f64), mu_py=torch.tensor(3e-6, **f64),
                                                   sigma_x=torch.tensor(2e-4, **f64), sigma_y=torch.tensor(3e-4, **f64),
                                                   sigma_p=torch.tensor(1e-3, **f64), sigma_tau=torch.tensor(1e-4, **f64), energy=energy, **f64)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            pout = seg.track(pb)
        arrays[f"lat{i}_pb_mu_in"] = pb.mu.numpy()
        arrays[f"lat{i}_pb_cov_in"] = pb.cov.numpy()
        arrays[f"lat{i}_pb_mu"] = pout.mu.numpy()
        arrays[f"lat{i}_pb_cov"] = pout.cov.numpy()
        arrays[f"lat{i}_pb_energy"] = pout.energy.numpy()
        arrays[f"lat{i}_pb_readings"] = torch.stack([b.reading for b in bpms]).numpy()
        if i in (1, 6):
            # three beams in one ParticleBeam (vectorised over the beam axis only: lattice settings and energy are scalars)
            parts = beam.particles.unsqueeze(0) * torch.tensor([0.6, 1.0, 1.7], **f64).reshape(3, 1, 1)
            parts[..., 6] = 1.0
            many = cheetah.ParticleBeam(parts, energy, particle_charges=beam.particle_charges, survival_probabilities=w, **f64)
            mout = seg.track(many)
            arrays[f"lat{i}_many_out"] = mout.particles.numpy()
            arrays[f"lat{i}_many_w_out"] = torch.broadcast_to(mout.survival_probabilities, (3, 1500)).numpy()
            arrays[f"lat{i}_many_readings"] = torch.stack([b.reading for b in bpms]).numpy()
        print(i, len(specs), "elements,", len(bpms), "monitors, lost", int((out.survival_probabilities == 0).sum()), "energy", float(out.energy))
    np.savez_compressed(os.path.join(OUT, "diagnostics_stretch.npz"), **arrays)
