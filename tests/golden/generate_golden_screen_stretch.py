#!/usr/bin/env python3
"""Drawn lattices with ACTIVE SCREENS between the magnets (and at the end), next to active monitors, apertures and cavities, tracked
by the reference -> tests/golden/screen_stretch.npz. Six lattices of 14-30 elements with two or three screens each (cloud-in-cell
and histogram images, misaligned screens, binning 2); per lattice the element list as JSON, 1200 incoming particles with drawn
charges and survival probabilities, and what the reference's `Segment.track` leaves in float64: the outgoing beam, every monitor's
reading and per screen the READ BEAM (particles shifted by the misalignment, charges, survival probabilities, energy, s —
screen.py:187-214) and its IMAGE (screen.py:241-344); for a ParameterBeam (lattices without apertures) the outgoing moments, every
screen's read moments and its image. Plus the RL control step the README describes (README.md:43-88): the ARES subcell at three
sets of its five magnet settings, a 2000-particle beam, the final screen's image.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_screen_stretch.py
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


def u(lo, hi):
    return float(rng.uniform(lo, hi))


def draw_screen(k):
    res = [(48, 40), (64, 32), (40, 56), (96, 64)][int(rng.integers(0, 4))]
    kw = {"resolution": list(res), "pixel_size": [u(4e-5, 1.5e-4), u(4e-5, 1.5e-4)], "is_active": True,
          "method": "histogram" if rng.random() < 0.25 else "cloud-in-cell", "name": f"scr{k}"}
    if rng.random() < 0.5:
        kw["misalignment"] = [u(-3e-4, 3e-4), u(-3e-4, 3e-4)]
    if rng.random() < 0.3:
        kw["binning"] = 2
    return ["Screen", kw]


def draw(with_cavities, with_apertures):
    kinds = ["Drift", "Drift", "Quadrupole", "Quadrupole", "HorizontalCorrector", "VerticalCorrector", "Dipole", "BPM", "Marker"]
    if with_cavities:
        kinds += ["Cavity"]
    if with_apertures:
        kinds += ["Aperture"]
    kind = str(rng.choice(kinds))
    if kind == "Drift":
        return [kind, {"length": u(0.05, 0.6)}]
    if kind == "Quadrupole":
        kw = {"length": u(0.05, 0.3), "k1": u(-10.0, 10.0)}
        if rng.random() < 0.3:
            kw["misalignment"] = [u(-2e-4, 2e-4), u(-2e-4, 2e-4)]
        return [kind, kw]
    if kind in ("HorizontalCorrector", "VerticalCorrector"):
        return [kind, {"length": u(0.01, 0.2), "angle": u(-2e-4, 2e-4)}]
    if kind == "Dipole":
        return [kind, {"length": u(0.2, 0.6), "angle": u(-0.03, 0.03)}]
    if kind == "Cavity":
        return [kind, {"length": u(0.3, 1.1), "voltage": u(2e6, 1.5e7), "phase": u(-40.0, 40.0), "frequency": 1.3e9,
                       "cavity_type": str(rng.choice(["standing_wave", "traveling_wave"]))}]
    if kind == "BPM":
        return [kind, {"is_active": True, "misalignment": [u(-3e-4, 3e-4), u(-3e-4, 3e-4)]}]
    if kind == "Aperture":
        return [kind, {"x_max": u(6e-4, 2e-3), "y_max": u(6e-4, 2e-3), "shape": str(rng.choice(["rectangular", "elliptical"])),
                       "is_active": True}]
    return [kind, {}]


def build(module, spec, fk):
    kind, kw = spec
    args = {}
    for k, v in kw.items():
        if k == "resolution":
            args[k] = tuple(v)
        elif isinstance(v, (float, list)):
            args[k] = torch.tensor(v, **fk)
        else:
            args[k] = v
    return getattr(module, kind)(**args, **fk)


def ares_subcell(fk):
    """ARES EA subcell AREASOLA1 -> AREABSCR1 (docs/examples/ARESlatticeStage3v1_9.json; README.md:43-58), a reduced screen."""
    t = lambda v: torch.tensor(v, **fk)  # noqa: E731
    return cheetah.Segment([
        cheetah.Marker(name="AREASOLA1"), cheetah.Drift(t(0.17504), **fk),
        cheetah.Quadrupole(t(0.122), k1=t(8.2), name="AREAMQZM1", **fk), cheetah.Drift(t(0.428), **fk),
        cheetah.Quadrupole(t(0.122), k1=t(-14.3), name="AREAMQZM2", **fk), cheetah.Drift(t(0.204), **fk),
        cheetah.VerticalCorrector(t(0.02), angle=t(9e-5), name="AREAMCVM1", **fk), cheetah.Drift(t(0.204), **fk),
        cheetah.Quadrupole(t(0.122), k1=t(3.142), name="AREAMQZM3", **fk), cheetah.Drift(t(0.179), **fk),
        cheetah.HorizontalCorrector(t(0.02), angle=t(-1e-4), name="AREAMCHM1", **fk), cheetah.Drift(t(0.45), **fk),
        cheetah.Screen(resolution=(306, 255), pixel_size=t([2.8390e-5, 2.0002e-5]), is_active=True, method="cloud-in-cell",
                       name="AREABSCR1", **fk)])


if __name__ == "__main__":
    arrays = {"n_lattices": np.asarray(6)}
    for i in range(6):
        with_cav, with_ap = i % 2 == 1, i >= 3
        specs = [draw(with_cav, with_ap) for _ in range(int(rng.integers(14, 31)))]
        n_scr = int(rng.integers(2, 4))
        cuts = sorted(int(c) for c in rng.choice(np.arange(2, len(specs)), size=n_scr - 1, replace=False))
        for k, c in enumerate(reversed(cuts)):
            specs.insert(c, draw_screen(n_scr - 2 - k))
        specs.append(draw_screen(n_scr - 1))                         # the last element is a screen (the RL layout)
        if i == 2:
            specs.append(["Drift", {"length": 0.3}])                 # ... except here: a run behind the last screen
        seg = cheetah.Segment([build(cheetah, s, f64) for s in specs])
        energy = torch.tensor(u(2e7, 2e8), **f64)
        torch.manual_seed(300 + i)
        beam = cheetah.ParticleBeam.from_parameters(num_particles=1200, mu_x=torch.tensor(u(-1e-4, 1e-4), **f64),
                                                    mu_y=torch.tensor(u(-1e-4, 1e-4), **f64), sigma_x=torch.tensor(u(1e-4, 3e-4), **f64),
                                                    sigma_y=torch.tensor(u(1e-4, 3e-4), **f64), sigma_px=torch.tensor(2e-5, **f64),
                                                    sigma_py=torch.tensor(2e-5, **f64), sigma_p=torch.tensor(1e-3, **f64),
                                                    sigma_tau=torch.tensor(1e-4, **f64), energy=energy, **f64)
        w = torch.rand(1200, **f64)
        w[torch.rand(1200) < 0.05] = 0.0
        q = beam.particle_charges * (0.5 + torch.rand(1200, **f64))
        q[torch.rand(1200) < 0.1] *= -1.0                           # (the image takes |q|: screen.py:327-331)
        beam = cheetah.ParticleBeam(beam.particles, energy, particle_charges=q, survival_probabilities=w, **f64)
        out = seg.track(beam)
        bpms = [e for e in seg.elements if isinstance(e, cheetah.BPM)]
        screens = [e for e in seg.elements if isinstance(e, cheetah.Screen)]
        arrays[f"lat{i}_spec"] = np.asarray(json.dumps(specs))
        arrays[f"lat{i}_energy"] = energy.numpy()
        arrays[f"lat{i}_in"] = beam.particles.numpy()
        arrays[f"lat{i}_w"] = w.numpy()
        arrays[f"lat{i}_q"] = q.numpy()
        arrays[f"lat{i}_out"] = out.particles.numpy()
        arrays[f"lat{i}_w_out"] = out.survival_probabilities.numpy()
        arrays[f"lat{i}_energy_out"] = out.energy.numpy()
        arrays[f"lat{i}_s_out"] = out.s.numpy()
        if bpms:
            arrays[f"lat{i}_readings"] = torch.stack([b.reading for b in bpms]).numpy()
        for k, scr in enumerate(screens):
            rb = scr.get_read_beam()
            arrays[f"lat{i}_scr{k}_rows"] = rb.particles.numpy()
            arrays[f"lat{i}_scr{k}_q"] = rb.particle_charges.numpy()
            arrays[f"lat{i}_scr{k}_w"] = rb.survival_probabilities.numpy()
            arrays[f"lat{i}_scr{k}_energy"] = rb.energy.numpy()
            arrays[f"lat{i}_scr{k}_s"] = rb.s.numpy()
            arrays[f"lat{i}_scr{k}_image"] = scr.reading.numpy()
        if not with_ap:
            pb = cheetah.ParameterBeam.from_parameters(mu_x=torch.tensor(1e-4, **f64), mu_py=torch.tensor(3e-6, **f64),
                                                       sigma_x=torch.tensor(2e-4, **f64), sigma_y=torch.tensor(3e-4, **f64),
                                                       sigma_p=torch.tensor(1e-3, **f64), sigma_tau=torch.tensor(1e-4, **f64),
                                                       energy=energy, total_charge=torch.tensor(2e-10, **f64), **f64)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                pout = seg.track(pb)
            arrays[f"lat{i}_pb_mu_in"] = pb.mu.numpy()
            arrays[f"lat{i}_pb_cov_in"] = pb.cov.numpy()
            arrays[f"lat{i}_pb_mu"] = pout.mu.numpy()
            arrays[f"lat{i}_pb_cov"] = pout.cov.numpy()
            arrays[f"lat{i}_pb_energy"] = pout.energy.numpy()
            arrays[f"lat{i}_pb_s"] = pout.s.numpy()
            for k, scr in enumerate(screens):
                rb = scr.get_read_beam()
                arrays[f"lat{i}_pb_scr{k}_mu"] = rb.mu.numpy()
                arrays[f"lat{i}_pb_scr{k}_cov"] = rb.cov.numpy()
                arrays[f"lat{i}_pb_scr{k}_energy"] = rb.energy.numpy()
                arrays[f"lat{i}_pb_scr{k}_s"] = rb.s.numpy()
                arrays[f"lat{i}_pb_scr{k}_image"] = scr.reading.numpy()
        print(i, len(specs), "elements,", len(screens), "screens,", len(bpms), "monitors, lost",
              int((out.survival_probabilities == 0).sum()), "image sums", [float(s.reading.sum()) for s in screens])

    # the control step: five magnet settings of the ARES subcell, the beam, the final screen's image
    seg = ares_subcell(f64)
    torch.manual_seed(400)
    beam = cheetah.ParticleBeam.from_twiss(beta_x=torch.tensor(3.14, **f64), beta_y=torch.tensor(42.0, **f64), num_particles=2000, **f64)
    pb = cheetah.ParameterBeam.from_twiss(beta_x=torch.tensor(3.14, **f64), beta_y=torch.tensor(42.0, **f64), **f64)
    actions = np.array([[8.2, -14.3, 9e-5, 3.142, -1e-4], [3.0, -2.0, 5e-5, 1.0, -4e-5], [-6.5, 11.0, -2e-4, -4.0, 1.5e-4]])
    arrays["control_in"] = beam.particles.numpy()
    arrays["control_q"] = beam.particle_charges.numpy()
    arrays["control_energy"] = beam.energy.numpy()
    arrays["control_pb_mu_in"] = pb.mu.numpy()
    arrays["control_pb_cov_in"] = pb.cov.numpy()
    arrays["control_actions"] = actions
    for k, a in enumerate(actions):
        seg.AREAMQZM1.k1, seg.AREAMQZM2.k1 = torch.tensor(a[0], **f64), torch.tensor(a[1], **f64)
        seg.AREAMCVM1.angle, seg.AREAMQZM3.k1 = torch.tensor(a[2], **f64), torch.tensor(a[3], **f64)
        seg.AREAMCHM1.angle = torch.tensor(a[4], **f64)
        out = seg.track(beam)
        arrays[f"control{k}_out"] = out.particles.numpy()
        arrays[f"control{k}_image"] = seg.AREABSCR1.reading.numpy()
        pout = seg.track(pb)
        arrays[f"control{k}_pb_mu"] = pout.mu.numpy()
        arrays[f"control{k}_pb_cov"] = pout.cov.numpy()
        arrays[f"control{k}_pb_image"] = seg.AREABSCR1.reading.numpy()
        print("control", k, "image sum", float(arrays[f"control{k}_image"].sum()), "pb image max", float(arrays[f"control{k}_pb_image"].max()))
    np.savez_compressed(os.path.join(OUT, "screen_stretch.npz"), **arrays)
    print("bytes", os.path.getsize(os.path.join(OUT, "screen_stretch.npz")))
