#!/usr/bin/env python3
"""Autograd semantics of the reference -> tests/golden/grad_flags.json (cases in tests/grad_cases.py): every tensor setting of
every element kind made trainable in turn, plus the beam energy and the incoming coordinates, for ParticleBeam and ParameterBeam —
which outgoing tensors carry a graph, the loss, and d loss / d (trainable tensor), or the exception raised. float64.
Run in the build container:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_grad_flags.py
"""
import json
import os
import sys
import warnings

sys.path.insert(0, "/root/reference")
import cheetah  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(OUT))
from grad_cases import cases, run  # noqa: E402

warnings.simplefilter("ignore")
table = {}
for kind, trainable in cases():
    for beam_kind in ("particle", "parameter"):
        key = f"{kind}|{trainable}|{beam_kind}"
        try:
            table[key] = run(cheetah, kind, trainable, beam_kind, None)
        except Exception as err:  # noqa: BLE001
            table[key] = {"raises": type(err).__name__, "message": str(err)[:90]}
print(len(table), "cases;", sum("raises" in v for v in table.values()), "raise;", sum(v.get("grad") == "no graph" for v in table.values()), "without graph;",
      sum(v.get("grad") is None for v in table.values() if "raises" not in v), "unused")
json.dump(table, open(os.path.join(OUT, "grad_flags.json"), "w"), indent=0, sort_keys=True)
