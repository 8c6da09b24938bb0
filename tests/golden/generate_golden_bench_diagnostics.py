#!/usr/bin/env python3
"""What bench.py's `DIAGNOSTICS_LATTICES` entries must leave behind, from the REFERENCE's own run of the same lattices and beams
(benchmarks/diagnostics_inputs.py: data drawn on the host from fixed seeds) in float64 -> tests/golden/bench_diagnostics.json.
Per entry: the outgoing beam's sigma_x, sigma_y, energy, the surviving weight and the last monitor's reading — for the scans the
rows 0 and 63. bench.py compares its float32 run with these numbers before it reports a time.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_bench_diagnostics.py
"""
import json
import os
import sys
import warnings

import torch

sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/repo")
import cheetah  # noqa: E402
from benchmarks import diagnostics_inputs as di  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
f64 = {"dtype": torch.float64}


def pick(t, b=None):
    t = t if (b is None or t.dim() == 0) else t[b]
    return float(t)


def observe(seg, out, rows=None):
    """The numbers bench.py checks; `rows`: the batch rows of a scan to look at (None: scalar results)."""
    bpms = [e for e in seg.elements if isinstance(e, cheetah.BPM)]
    res = {}
    for tag, b in ([("", None)] if rows is None else [(f"_row{r}", r) for r in rows]):
        res["sigma_x" + tag] = pick(out.sigma_x, b)
        res["sigma_y" + tag] = pick(out.sigma_y, b)
        res["energy" + tag] = pick(out.energy, b)
        if bpms:
            r = bpms[-1].reading
            res["last_reading_x" + tag] = float(r[..., 0] if b is None or r.dim() == 1 else r[b, 0])
    if isinstance(out, cheetah.ParticleBeam):
        w = out.survival_probabilities
        res["survived"] = float(w.sum() if w.dim() == 1 else w[0].sum())
    return res


if __name__ == "__main__":
    x = di.particles().double()
    small = x[:di.N_SMALL]
    E0 = torch.tensor(1e8, **f64)
    beam = cheetah.ParticleBeam(x, E0, **f64)
    small_beam = cheetah.ParticleBeam(small, E0, **f64)
    mu, cov = di.parameter_beam_moments()
    pbeam = cheetah.ParameterBeam(mu, cov, E0, **f64)
    energies = torch.tensor(di.energies(), **f64)
    expected = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, specs in (("bpm_lattice", di.bpm_lattice()), ("aperture_lattice", di.aperture_lattice()), ("cavity_linac", di.cavity_linac())):
            seg = di.segment(cheetah, specs, f64)
            expected[name] = {"particle_beam": observe(seg, seg.track(beam))}
            if name != "aperture_lattice":
                expected[name]["parameter_beam"] = observe(seg, seg.track(pbeam))
        seg = di.segment(cheetah, di.orbit_response(), f64)
        expected["orbit_response_64_settings"] = {"parameter_beam": observe(seg, seg.track(pbeam), rows=(0, 63)),
                                                  "particle_beam_1e4": observe(seg, seg.track(small_beam), rows=(0, 63))}
        seg = di.segment(cheetah, di.cavity_linac(), f64)
        e_pb = cheetah.ParameterBeam(mu, cov, energies, **f64)
        e_beam = cheetah.ParticleBeam(small, energies, **f64)
        expected["linac_64_energies"] = {"parameter_beam": observe(seg, seg.track(e_pb), rows=(0, 63)),
                                         "particle_beam_1e4": observe(seg, seg.track(e_beam), rows=(0, 63))}
        seg = di.segment(cheetah, di.cavity_linac(phase=di.phases()), f64)
        expected["linac_64_phases_of_every_cavity"] = {"parameter_beam": observe(seg, seg.track(pbeam), rows=(0, 63)),
                                                       "particle_beam_1e4": observe(seg, seg.track(small_beam), rows=(0, 63))}
        seg = di.segment(cheetah, di.cavity_linac(off=(5, 9)), f64)
        expected["linac_two_cavities_off"] = {"parameter_beam": observe(seg, seg.track(pbeam)), "particle_beam": observe(seg, seg.track(beam))}
    with open(os.path.join(OUT, "bench_diagnostics.json"), "w") as fh:
        json.dump(expected, fh, indent=1, sort_keys=True)
    for k, v in expected.items():
        print(k, {kk: {q: f"{vv:.6g}" for q, vv in list(vvv.items())[:4]} for kk, vvv in v.items()})
