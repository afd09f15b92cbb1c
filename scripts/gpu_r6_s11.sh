#!/bin/bash
OUT=gpurun_out/r6_s11
mkdir -p $OUT
export REFNERF_LIB=ab/sq_fwdfresh.so
for i in 1 2; do
python -m pytest tests/test_hip_f16x2.py -k chain_training_step_vs_reference -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^FAILED|passed|failed" | tail -n 3
done
python -m pytest tests -m gpu -q -p no:cacheprovider -k "train or grad or backward or shard or trajectory or optimiser or loss or specular or basis" > $OUT/pytest_train.log 2>&1; echo "pytest train rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_train.log | tail -n 8
python scripts/time_train.py f16x2 2>&1 | tail -n 1
