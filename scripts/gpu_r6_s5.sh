#!/bin/bash
OUT=gpurun_out/r6_s5
mkdir -p $OUT
T="tests/test_hip_f16x2.py -k chain_training_step_vs_reference"
python -m pytest $T -m gpu -q -x -p no:cacheprovider > $OUT/cur.log 2>&1; echo "current rc=$?"; grep -E "passed|failed|nstat|Assertion" $OUT/cur.log | head -n 4
REFNERF_LIB=ab/sq_sync.so python -m pytest $T -m gpu -q -x -p no:cacheprovider > $OUT/sync.log 2>&1; echo "TQ_SYNC rc=$?"; grep -E "passed|failed|nstat|Assertion" $OUT/sync.log | head -n 4
