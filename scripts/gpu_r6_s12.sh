#!/bin/bash
export REFNERF_LIB=ab/sq_fwdfresh_sync.so
for i in 1 2; do
python -m pytest tests/test_hip_f16x2.py -k chain_training_step_vs_reference -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^FAILED|passed|failed" | tail -n 3
done
python -m pytest tests/test_hip_f16x2.py tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -k "trajectory or optimiser or c_abi_training or trained_long" 2>&1 | grep -E "^FAILED|passed|failed" | tail -n 6
