#!/bin/bash
# Round 6: what bounds the tile deposit and the tile gather when they run alone (benchmarks/tile_gather_bench.py, 1e6 rows on 128^3)
for g in 0 1 2 3; do
  echo "CHX_TUNE_GATHER_DIAG=$g"; CHX_TUNE_GATHER_DIAG=$g python benchmarks/tile_gather_bench.py 2>&1 | grep "tile gather, tile order"
done
for d in 0 1 2 4 7; do
  echo "CHX_TUNE_DEPOSIT_DIAG=$d"; CHX_TUNE_DEPOSIT_DIAG=$d python benchmarks/tile_gather_bench.py 2>&1 | grep "tile deposit"
done
