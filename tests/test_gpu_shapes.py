"""Broadcasting semantics against the reference (tests/golden/shapes.json): 500 combinations of vector shapes — (), (1,), (3,),
(1, 3), (2, 1) — on a quadrupole strength, on the incoming particles (or mu / cov) and on the beam energy, through a plain and a
cavity lattice, ParticleBeam and ParameterBeam: the SHAPES of every outgoing tensor (particles / mu / cov, energy, survival /
total charge, s, BPM reading) and two checksums of the values."""
import itertools
import json
import os

import pytest
import torch

from tests.shape_cases import SHAPES, describe, inputs

pytestmark = pytest.mark.gpu


def test_broadcast_shapes_and_values_vs_reference():
    import cheetah_amd as ca

    table = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shapes.json")))
    wrong = []
    for lattice, kind in itertools.product(("plain", "cavity"), ("particle", "parameter")):
        for k1s, bs, es in itertools.product(SHAPES, SHAPES, SHAPES):
            key = f"{lattice}|{kind}|{k1s}|{bs}|{es}"
            ref = table[key]
            assert "raises" not in ref
            seg, beam = inputs(ca, k1s, bs, es, kind, lattice, dev="cuda")
            got = describe(seg.track(beam), kind, seg)
            for name, value in ref.items():
                if isinstance(value, list):
                    if got[name] != value:
                        wrong.append((key, name, got[name], value))
                elif got[name] != pytest.approx(value, rel=1e-9, abs=1e-9 * ref["abs"] if name == "sum" else 1e-18):
                    # (`sum` of symmetric ramps cancels to rounding noise: its tolerance comes from the sum of magnitudes)
                    wrong.append((key, name, got[name], value))
    assert not wrong, (len(wrong), wrong[:12])
