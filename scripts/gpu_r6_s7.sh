#!/bin/bash
OUT=gpurun_out/r6_s7
mkdir -p $OUT
T="tests/test_hip_f16x2.py -k chain_training_step_vs_reference"
for v in ab/sq_nog1.so ab/sq_nog4.so; do
  REFNERF_LIB=$v python -m pytest $T -m gpu -q -x -p no:cacheprovider > $OUT/$(basename $v).log 2>&1; echo "$v rc=$?"; grep -E "passed|failed|AssertionError" $OUT/$(basename $v).log | head -n 3
done
