#!/usr/bin/env python3
"""SCANS of lattice settings tracked with particles by the reference -> tests/golden/scan_stretch.npz. Six drawn lattices of 18-36
elements with ACTIVE beam position monitors, ACTIVE apertures (lattices 3-5) and cavities (odd lattices); about a third of the
quadrupole strengths and corrector angles are (4,) tensors — four lattice settings in one `Segment.track`. Per lattice: the element
list as JSON (a vectorised setting is a list of four numbers), 1200 incoming particles with drawn survival probabilities and what
the reference leaves in float64: the (4, 1200, 7) outgoing particles, the survival probabilities (whatever shape they have),
energy, s and EVERY monitor's reading with its own shape ((2,) in front of the first vectorised element, (4, 2) behind); the same
scan for a ParameterBeam; and both once more with a (4,) BEAM ENERGY on top (`lat*_escan_*`: a scan of energies through the cavities); for the lattices with cavities a PHASE scan of every cavity plus a
voltage scan of the first (`lat*_cscan_*`), some rows losing energy; and a 2-D GRID scan written by broadcasting — strengths of shape
(3, 1) and (1, 2) — on the lattice with otherwise scalar settings (`lat*_gscan_*`).
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_scan_stretch.py
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
rng = np.random.default_rng(20261001)
ROWS = 4
SUB = 300       # particles whose outgoing coordinates are kept for the energy / cavity / grid scans (the readings see all 1200)


def u(lo, hi):
    return float(rng.uniform(lo, hi))


def scan(lo, hi):
    """A setting: one number, or (every third time) one number per row of the scan."""
    return [u(lo, hi) for _ in range(ROWS)] if rng.random() < 0.35 else u(lo, hi)


def draw(with_cavities, with_apertures):
    kinds = ["Drift", "Drift", "Quadrupole", "Quadrupole", "HorizontalCorrector", "VerticalCorrector", "BPM", "BPM", "Marker"]
    kinds += ["Cavity"] if with_cavities else []
    kinds += ["Aperture", "Aperture"] if with_apertures else []
    kind = str(rng.choice(kinds))
    if kind == "Drift":
        return [kind, {"length": u(0.05, 1.0)}]
    if kind == "Quadrupole":
        return [kind, {"length": u(0.05, 0.3), "k1": scan(-10.0, 10.0)}]
    if kind in ("HorizontalCorrector", "VerticalCorrector"):
        return [kind, {"length": u(0.01, 0.2), "angle": scan(-3e-4, 3e-4)}]
    if kind == "Cavity":
        return [kind, {"length": u(0.3, 1.1), "voltage": u(2e6, 1.5e7), "phase": u(-40.0, 40.0), "frequency": 1.3e9,
                       "cavity_type": str(rng.choice(["standing_wave", "traveling_wave"]))}]
    if kind == "BPM":
        return [kind, {"is_active": True, "misalignment": [u(-3e-4, 3e-4), u(-3e-4, 3e-4)]}]
    if kind == "Aperture":
        return [kind, {"x_max": u(5e-4, 2e-3), "y_max": u(5e-4, 2e-3), "shape": str(rng.choice(["rectangular", "elliptical"])),
                       "is_active": True}]
    return [kind, {}]


def build(module, spec, fk):
    kind, kw = spec
    args = {k: (torch.tensor(v, **fk) if isinstance(v, (float, list)) else v) for k, v in kw.items()}
    return getattr(module, kind)(**args, **fk)


if __name__ == "__main__":
    arrays = {"n_lattices": np.asarray(6), "rows": np.asarray(ROWS)}
    torch.manual_seed(299)          # (from_parameters draws from torch's generator: lattice 0 too starts from a fixed state)
    for i in range(6):
        with_cav, with_ap = i % 2 == 1, i >= 3
        specs = [["Drift", {"length": u(0.1, 0.5)}], ["BPM", {"is_active": True, "misalignment": [1e-4, -1e-4]}]]   # a monitor in front of the scan
        if with_ap:
            specs.insert(1, ["Aperture", {"x_max": 1.5e-3, "y_max": 1.2e-3, "shape": "elliptical", "is_active": True}])
        specs += [draw(with_cav, with_ap) for _ in range(int(rng.integers(16, 34)))]
        if not any(isinstance(v, list) and len(v) == ROWS for s in specs for v in s[1].values()):
            specs.append(["HorizontalCorrector", {"length": 0.1, "angle": [u(-3e-4, 3e-4) for _ in range(ROWS)]}])
        specs.append(["BPM", {"is_active": True, "misalignment": [0.0, 0.0]}])
        seg = cheetah.Segment([build(cheetah, s, f64) for s in specs])
        energy = torch.tensor(u(2e7, 2e8), **f64)
        beam = cheetah.ParticleBeam.from_parameters(num_particles=1200, mu_x=torch.tensor(u(-2e-4, 2e-4), **f64),
                                                    mu_y=torch.tensor(u(-2e-4, 2e-4), **f64), sigma_x=torch.tensor(u(1e-4, 4e-4), **f64),
                                                    sigma_y=torch.tensor(u(1e-4, 4e-4), **f64), sigma_px=torch.tensor(2e-5, **f64),
                                                    sigma_py=torch.tensor(2e-5, **f64), sigma_p=torch.tensor(1e-3, **f64),
                                                    sigma_tau=torch.tensor(1e-4, **f64), energy=energy, **f64)
        torch.manual_seed(300 + i)
        w = torch.rand(1200, **f64)
        w[torch.rand(1200) < 0.05] = 0.0
        beam = cheetah.ParticleBeam(beam.particles, energy, particle_charges=beam.particle_charges, survival_probabilities=w, **f64)
        out = seg.track(beam)
        bpms = [e for e in seg.elements if isinstance(e, cheetah.BPM)]
        assert out.particles.shape == (ROWS, 1200, 7)
        arrays[f"lat{i}_spec"] = np.asarray(json.dumps(specs))
        arrays[f"lat{i}_energy"] = energy.numpy()
        arrays[f"lat{i}_in"] = beam.particles.numpy()
        arrays[f"lat{i}_w"] = w.numpy()
        arrays[f"lat{i}_q"] = beam.particle_charges.numpy()
        arrays[f"lat{i}_out"] = out.particles.numpy()
        arrays[f"lat{i}_w_out"] = out.survival_probabilities.numpy()
        arrays[f"lat{i}_energy_out"] = out.energy.numpy()
        arrays[f"lat{i}_s_out"] = out.s.numpy()
        arrays[f"lat{i}_n_bpms"] = np.asarray(len(bpms))
        for k, b in enumerate(bpms):
            arrays[f"lat{i}_reading{k}"] = b.reading.numpy()
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
        for k, b in enumerate(bpms):
            arrays[f"lat{i}_pb_reading{k}"] = b.reading.numpy()
        # the same lattice under a scan of BEAM ENERGIES (a (4,) energy on top of the vectorised settings), particles and moments
        escan = torch.tensor([float(energy) * f for f in (0.7, 0.9, 1.15, 1.4)], **f64)
        ebeam = cheetah.ParticleBeam(beam.particles, escan, particle_charges=beam.particle_charges, survival_probabilities=w, **f64)
        eout = seg.track(ebeam)
        arrays[f"lat{i}_escan_energy"] = escan.numpy()
        arrays[f"lat{i}_escan_out"] = eout.particles[..., :SUB, :].numpy()
        arrays[f"lat{i}_escan_w_out"] = eout.survival_probabilities[..., :SUB].numpy()
        arrays[f"lat{i}_escan_energy_out"] = eout.energy.numpy()
        for k, b in enumerate(bpms):
            arrays[f"lat{i}_escan_reading{k}"] = b.reading.numpy()
        epb = cheetah.ParameterBeam(pb.mu, pb.cov, escan, **f64)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            epout = seg.track(epb)
        arrays[f"lat{i}_escan_pb_mu"] = epout.mu.numpy()
        arrays[f"lat{i}_escan_pb_cov"] = epout.cov.numpy()
        arrays[f"lat{i}_escan_pb_energy"] = epout.energy.numpy()
        for k, b in enumerate(bpms):
            arrays[f"lat{i}_escan_pb_reading{k}"] = b.reading.numpy()
        if with_cav:
            # a PHASE scan of every cavity (and a voltage scan of the first) on top: (4,) cavity settings, some rows LOSE energy in
            # some cavities (cavity.py:157 looks at the whole batch)
            crng = np.random.default_rng(5000 + i)
            cspecs, first = json.loads(json.dumps(specs)), True
            for sp in cspecs:
                if sp[0] == "Cavity":
                    sp[1]["phase"] = [float(crng.uniform(-60.0, 60.0)) for _ in range(ROWS - 1)] + [float(crng.uniform(120.0, 170.0))]
                    if first:
                        sp[1]["voltage"] = [float(crng.uniform(2e6, 1.5e7)) for _ in range(ROWS)]
                        first = False
            cseg = cheetah.Segment([build(cheetah, sp, f64) for sp in cspecs])
            cbpms = [e for e in cseg.elements if isinstance(e, cheetah.BPM)]
            cout = cseg.track(beam)
            arrays[f"lat{i}_cscan_spec"] = np.asarray(json.dumps(cspecs))
            arrays[f"lat{i}_cscan_out"] = cout.particles[..., :SUB, :].numpy()
            arrays[f"lat{i}_cscan_w_out"] = cout.survival_probabilities[..., :SUB].numpy()
            arrays[f"lat{i}_cscan_energy_out"] = cout.energy.numpy()
            for k, b in enumerate(cbpms):
                arrays[f"lat{i}_cscan_reading{k}"] = b.reading.numpy()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                cpout = cseg.track(pb)
            arrays[f"lat{i}_cscan_pb_mu"] = cpout.mu.numpy()
            arrays[f"lat{i}_cscan_pb_cov"] = cpout.cov.numpy()
            arrays[f"lat{i}_cscan_pb_energy"] = cpout.energy.numpy()
            print("   cavity scan: energies out", cout.energy.numpy(), "NaNs", int(np.isnan(cout.particles.numpy()).sum()))
        # a 2-D GRID scan by broadcasting: the lattice with scalar settings, then one quadrupole strength of shape (3, 1) and a later
        # one of shape (1, 2) (and, with cavities, the last cavity's phase (1, 2))
        grng = np.random.default_rng(7000 + i)
        gspecs = json.loads(json.dumps(specs))
        for sp in gspecs:
            for q, v in sp[1].items():
                if isinstance(v, list) and len(v) == ROWS and q in ("k1", "angle"):
                    sp[1][q] = v[0]
        quads = [sp for sp in gspecs if sp[0] == "Quadrupole"]
        if len(quads) >= 2:
            quads[0][1]["k1"] = [[float(grng.uniform(-8.0, 8.0))] for _ in range(3)]
            quads[-1][1]["k1"] = [[float(grng.uniform(-8.0, 8.0)) for _ in range(2)]]
            cavs = [sp for sp in gspecs if sp[0] == "Cavity"]
            if cavs:
                cavs[-1][1]["phase"] = [[float(grng.uniform(-50.0, 50.0)) for _ in range(2)]]
            gseg = cheetah.Segment([build(cheetah, sp, f64) for sp in gspecs])
            gbpms = [e for e in gseg.elements if isinstance(e, cheetah.BPM)]
            gout = gseg.track(beam)
            arrays[f"lat{i}_gscan_spec"] = np.asarray(json.dumps(gspecs))
            arrays[f"lat{i}_gscan_out"] = gout.particles[..., :SUB, :].numpy()
            arrays[f"lat{i}_gscan_w_out"] = gout.survival_probabilities[..., :SUB].numpy()
            arrays[f"lat{i}_gscan_energy_out"] = gout.energy.numpy()
            for k, b in enumerate(gbpms):
                arrays[f"lat{i}_gscan_reading{k}"] = b.reading.numpy()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                gpout = gseg.track(pb)
            arrays[f"lat{i}_gscan_pb_mu"] = gpout.mu.numpy()
            arrays[f"lat{i}_gscan_pb_cov"] = gpout.cov.numpy()
            arrays[f"lat{i}_gscan_pb_energy"] = gpout.energy.numpy()
            for k, b in enumerate(gbpms):
                arrays[f"lat{i}_gscan_pb_reading{k}"] = b.reading.numpy()
            print("   grid scan: out", tuple(gout.particles.shape), "w", tuple(gout.survival_probabilities.shape), "energy", tuple(gout.energy.shape),
                  "readings", sorted({tuple(b.reading.shape) for b in gbpms}))
        print(i, len(specs), "elements,", len(bpms), "monitors, w_out", tuple(out.survival_probabilities.shape),
              "lost per row", (out.survival_probabilities == 0).sum(dim=-1).tolist(), "readings", [tuple(b.reading.shape) for b in bpms])
    np.savez_compressed(os.path.join(OUT, "scan_stretch.npz"), **arrays)
