#!/usr/bin/env python3
"""Public attribute names of every class the reference exports at top level (`import cheetah`), written to
tests/golden/api_surface.json. Run in the build container (the reference is importable there):
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/generate_golden_api_surface.py
Names only — what `tests/test_utils_api.py::test_public_attribute_names_of_the_reference` checks this package against."""
import inspect
import json
import os
import sys

import torch

sys.path.insert(0, "/root/reference")
import cheetah  # noqa: E402

module_names = set(dir(torch.nn.Module))
surface = {}
for name in sorted(dir(cheetah)):
    obj = getattr(cheetah, name)
    if name.startswith("_") or not inspect.isclass(obj):
        continue
    surface[name] = sorted(a for a in dir(obj) if not a.startswith("_") and a not in module_names)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "api_surface.json")
with open(out, "w") as f:
    json.dump(surface, f, indent=0, sort_keys=True)
print(len(surface), "classes,", sum(len(v) for v in surface.values()), "names ->", out)
