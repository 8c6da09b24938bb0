#!/bin/bash
# C4 track time, three processes (run-to-run spread of this config is ~0.05 ms)
for i in 1 2 3; do python benchmarks/run_configs.py c4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 track %.3f ms, single kick %.3f ms' % (d['track_ms'], d['single_kick_ms']))"; done
