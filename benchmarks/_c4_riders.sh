#!/bin/bash
# C4 with and without the riders (geometry inside the deposit / corner-table launches, bookkeeping inside the first FFT pass)
for mode in 0 1 0 1; do
  CHX_SC_RIDERS=$mode python benchmarks/run_configs.py c4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('riders=$mode C4 track %.3f ms, single kick %.3f ms' % (d['track_ms'], d['single_kick_ms']))"
done
