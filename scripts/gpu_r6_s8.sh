#!/bin/bash
OUT=gpurun_out/r6_s8
mkdir -p $OUT
T="tests/test_hip_f16x2.py -k chain_training_step_vs_reference and model_trained_train and f16x2-f16"
for v in ab/sq_nog1.so ab/sq_025fa43.so; do
 for i in 1 2 3; do
  REFNERF_LIB=$v python -m pytest "tests/test_hip_f16x2.py" -k "chain_training_step_vs_reference and model_trained_train and f16x2-f16" -m gpu -q -s -p no:cacheprovider 2>&1 | grep -E "f16x2 chains vs reference|passed|failed" | cut -c1-260
 done
done | tee $OUT/repeat.log
