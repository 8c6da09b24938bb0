#!/usr/bin/env python3
"""Broadcasting semantics of the reference -> tests/golden/shapes.json: for every combination of vector shapes on a quadrupole
strength, on the incoming particles (or mu / cov) and on the beam energy — (), (1,), (3,), (1, 3), (2, 1) — through
[Drift, Quadrupole(k1), Cavity, Drift, BPM] and [Drift, Quadrupole(k1), Drift]: the shapes of the outgoing particles / mu / cov,
energy, survival probabilities (or total charge) and s, two checksums of the values — or the name of the exception raised.
ParticleBeam and ParameterBeam, float64. Data only (no arrays: inputs are generated from the shapes by the same formula on both sides).
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_shapes.py
"""
import itertools
import json
import os
import sys
import warnings

import torch

sys.path.insert(0, "/root/reference")
import cheetah  # noqa: E402

warnings.simplefilter("ignore")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(OUT))
from shape_cases import SHAPES, describe, inputs  # noqa: E402


if __name__ == "__main__":
    table = {}
    for lattice, kind in itertools.product(("plain", "cavity"), ("particle", "parameter")):
        for k1s, bs, es in itertools.product(SHAPES, SHAPES, SHAPES):
            key = f"{lattice}|{kind}|{k1s}|{bs}|{es}"
            try:
                seg, beam = inputs(cheetah, k1s, bs, es, kind, lattice)
                table[key] = describe(seg.track(beam), kind, seg)
            except Exception as err:  # noqa: BLE001
                table[key] = {"raises": type(err).__name__}
    ok = sum(1 for v in table.values() if "raises" not in v)
    print(len(table), "combinations,", ok, "tracked,", len(table) - ok, "raise")
    with open(os.path.join(OUT, "shapes.json"), "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
