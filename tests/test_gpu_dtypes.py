"""Dtype semantics against the reference (tests/golden/dtypes.json, cases in tests/dtype_cases.py): every element kind built in
float32 / float64 with a float32 / float64 ParticleBeam or ParameterBeam. Where the reference tracks, the dtypes of the outgoing
coordinates, the energy and the diagnostic reading are the same; where it raises (mismatched dtypes in the map product,
ParameterBeams on the non-linear methods and the space-charge kick), this engine raises the same exception class."""
import json
import os
import warnings

import pytest
import torch

from tests.dtype_cases import DT, elements, outcome

pytestmark = pytest.mark.gpu


def test_dtype_outcomes_vs_reference():
    import cheetah_amd as ca

    table = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dtypes.json")))
    wrong = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name in elements(ca, torch.float64, "cuda"):
            for e_dt in DT:
                for b_dt in DT:
                    for kind in ("particle", "parameter"):
                        key = f"{name}|{e_dt}|{b_dt}|{kind}"
                        ref = table[key]
                        try:
                            got = outcome(ca, name, e_dt, b_dt, kind, "cuda")
                        except Exception as err:  # noqa: BLE001
                            got = {"raises": type(err).__name__}
                        if "raises" in ref:
                            if got.get("raises") != ref["raises"]:
                                wrong.append((key, got, ref["raises"]))
                        elif got != ref:
                            wrong.append((key, got, ref))
    assert not wrong, (len(wrong), wrong)
