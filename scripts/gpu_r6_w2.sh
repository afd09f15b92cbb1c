#!/bin/bash
# round 6: the weight-gradient GEMM without its conversion pass (ab/sq_raw.so): gradient tests + timing
mkdir -p gpurun_out/r6_w2
export REFNERF_NO_FINITE_CHECK=1
timeout 600 python scripts/ab_train_modes.py ab/sq_base.so ab/sq_raw.so 2>&1 | grep -v Warning | tee gpurun_out/r6_w2/ab.log
REFNERF_LIB=ab/sq_raw.so timeout 1500 python -m pytest tests/test_hip_f16x2.py tests/test_geometry_losses.py -x -q -m gpu -k "train or chain or grad or loss or trajectory or optimiser" 2>&1 | tail -n 15 | tee gpurun_out/r6_w2/pytest.log
