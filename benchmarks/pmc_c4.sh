#!/bin/bash
# SQ counters of config C4's kernels (one pass; --pmc goes with --kernel-trace only)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_c4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VALU \
    --kernel-trace --output-format csv -d $OUT/trace -o c4 -- python $REPO/benchmarks/run_configs.py c4 > $OUT/bench.log 2> $OUT/trace.log
ls $OUT/trace
