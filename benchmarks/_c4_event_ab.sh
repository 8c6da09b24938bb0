#!/bin/bash
# C4 per track with the fork / join events of the chain kicks recorded with a system-scope fence (0), a device-scope release (1), no fence (2)
for rep in 1 2 3; do
  for v in 0 1 2; do
    echo -n "event flags $v: "
    CHX_TUNE_EVENT_FLAGS=$v python benchmarks/run_configs.py c4 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['track_ms'])"
  done
done
