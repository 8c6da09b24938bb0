#!/usr/bin/env python3
"""The reference's own speed guard (tests/test_speed.py:21-35: the ARES experimental-area section AREASOLA1 -> AREABSCR1 with its
screen switched on, 1e5 particles, `segment.track` + `AREABSCR1.reading`) as numbers bench.py can hold its timed step to:
the section from the lattice file the reference itself wrote (tests/golden/ares_lattice.json), the beam from the host generator both
sides share (benchmarks/diagnostics_inputs.py `particles()`: sigma_x = sigma_y = 175 um like the reference's test), tracked by the
REFERENCE in float64 -> tests/golden/ares_speed.json: the outgoing beam's sigma_x / sigma_y, the reading's sum, its centre of mass
and the number of lit pixels (method 'histogram', the file's default).
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_ares_speed.py"""
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

if __name__ == "__main__":
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        full = cheetah.Segment.from_lattice_json(os.path.join(OUT, "ares_lattice.json"))
        seg = full.subcell("AREASOLA1", "AREABSCR1").to(torch.float64)
        seg.AREABSCR1.is_active = True
        x = di.particles().double()
        beam = cheetah.ParticleBeam(x, torch.tensor(1e8, dtype=torch.float64), dtype=torch.float64)
        out = seg.track(beam)
        img = seg.AREABSCR1.reading
    h, w = img.shape
    total = float(img.sum())
    cx = float((img.sum(0) * torch.arange(w, dtype=img.dtype)).sum() / img.sum())
    cy = float((img.sum(1) * torch.arange(h, dtype=img.dtype)).sum() / img.sum())
    res = {"elements": [type(e).__name__ for e in seg.elements], "method": seg.AREABSCR1.method, "image_shape": [h, w],
           "sigma_x": float(out.sigma_x), "sigma_y": float(out.sigma_y), "image_sum": total, "image_centre_x": cx, "image_centre_y": cy,
           "lit_pixels": int((img != 0).sum()), "image_max": float(img.max()), "torch": torch.__version__}
    json.dump(res, open(os.path.join(OUT, "ares_speed.json"), "w"), indent=1)
    print(res)
