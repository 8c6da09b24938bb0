#!/usr/bin/env python3
"""Dtype semantics of the reference -> tests/golden/dtypes.json (cases in tests/dtype_cases.py): for every element kind built in
float32 / float64 and a float32 / float64 ParticleBeam or ParameterBeam — the dtype of the outgoing coordinates, energy and
diagnostic reading, or the name of the exception raised.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_dtypes.py
"""
import json
import os
import sys
import warnings

sys.path.insert(0, "/root/reference")
import cheetah  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(OUT))
from dtype_cases import DT, elements, outcome  # noqa: E402
import torch  # noqa: E402

warnings.simplefilter("ignore")
table = {}
for name in elements(cheetah, torch.float64, None):
    for e_dt in DT:
        for b_dt in DT:
            for kind in ("particle", "parameter"):
                key = f"{name}|{e_dt}|{b_dt}|{kind}"
                try:
                    table[key] = outcome(cheetah, name, e_dt, b_dt, kind, None)
                except Exception as err:  # noqa: BLE001
                    table[key] = {"raises": type(err).__name__, "message": str(err)[:100]}
n_raise = sum(1 for v in table.values() if "raises" in v)
print(len(table), "cases,", n_raise, "raise")
for k, v in table.items():
    if "raises" in v:
        print("  ", k, v["raises"], v["message"][:70])
json.dump(table, open(os.path.join(OUT, "dtypes.json"), "w"), indent=0, sort_keys=True)
