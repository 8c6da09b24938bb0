#!/usr/bin/env python3
"""C4 kick with the Green-function chain on the MAIN stream (no overlap): per-kernel durations in isolation.
Run under rocprofv3 --kernel-trace --stats to compare with the overlapped default (profiles/run_profile_c4.sh)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402
from benchmarks import run_configs as rc  # noqa: E402

ca.SpaceChargeKick._side_stream = classmethod(lambda cls, device: None)
print(json.dumps(rc.c4()))
