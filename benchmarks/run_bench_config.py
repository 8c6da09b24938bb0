#!/usr/bin/env python3
"""One (or a few) of bench.py's side configs on their own: python benchmarks/run_bench_config.py DIAGNOSTICS_LATTICES [C5 ...] -> JSON."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cheetah_amd as ca  # noqa: E402

print(json.dumps(bench.other_configs(ca, torch, "cuda:0", only=set(sys.argv[1:]) or None), indent=1))
