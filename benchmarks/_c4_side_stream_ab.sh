#!/bin/bash
# C4 per track with the chain kicks' Green-function kernels on a side stream (default) and on the caller's stream, alternating processes
for rep in 1 2 3; do
  for v in 1 0; do
    echo -n "side stream $v: "
    CHX_SC_CHAIN_SIDE_STREAM=$v python benchmarks/run_configs.py c4 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['track_ms'])"
  done
done
