"""Autograd semantics against the reference (tests/golden/grad_flags.json, cases in tests/grad_cases.py): every tensor setting of
every element kind made trainable in turn, the beam energy and the incoming coordinates as well, ParticleBeam and ParameterBeam.
Which outgoing tensors carry a graph (coordinates, covariance, energy, path length, survival probabilities), the loss, and
d loss / d (trainable tensor) to 2e-7; where the reference raises, the same exception class."""
import json
import os
import warnings

import pytest
import torch

from tests.grad_cases import cases, run

pytestmark = pytest.mark.gpu


def test_graph_flags_and_gradients_vs_reference():
    import cheetah_amd as ca

    table = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grad_flags.json")))
    wrong = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for kind, trainable in cases():
            for beam_kind in ("particle", "parameter"):
                key = f"{kind}|{trainable}|{beam_kind}"
                ref = table[key]
                try:
                    got = run(ca, kind, trainable, beam_kind, "cuda")
                except Exception as err:  # noqa: BLE001
                    got = {"raises": type(err).__name__, "message": str(err)[:80]}
                if "raises" in ref or "raises" in got:
                    if got.get("raises") != ref.get("raises"):
                        wrong.append((key, got, ref))
                    continue
                for flag in ("coords", "energy", "s", "survival", "cov"):
                    if flag in ref and got[flag] != ref[flag]:
                        wrong.append((key, flag, got[flag], ref[flag]))
                if got["loss"] != pytest.approx(ref["loss"], rel=1e-9):
                    wrong.append((key, "loss", got["loss"], ref["loss"]))
                if isinstance(ref["grad"], list):
                    if not isinstance(got["grad"], list):
                        wrong.append((key, "grad", got["grad"], ref["grad"][:3]))
                    else:
                        # (the floor: a gradient that vanishes by the symmetry of the ramp beam is rounding noise, 1e-15 of a
                        # loss of order one, in both implementations)
                        scale = max(max(abs(v) for v in ref["grad"]), 1e-300)
                        if max(abs(a - b) for a, b in zip(got["grad"], ref["grad"])) > 2e-7 * scale + 1e-12 * abs(ref["loss"]):
                            wrong.append((key, "grad", got["grad"][:3], ref["grad"][:3]))
                elif got["grad"] != ref["grad"]:
                    wrong.append((key, "grad", got["grad"], ref["grad"]))
    assert not wrong, (len(wrong), wrong)
