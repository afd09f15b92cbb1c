#!/bin/bash
mkdir -p gpurun_out/r6_w6
export REFNERF_NO_FINITE_CHECK=1
timeout 900 python scripts/ab_train_modes.py ab/sq_raw2.so ab/sq_raw2_none.so ab/sq_raw2_nodma_nomfma.so ab/sq_raw2_nomfma.so ab/sq_raw2_nodma.so 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_w6/ab.log
