#!/bin/bash
mkdir -p gpurun_out/r6_w7
timeout 600 python scripts/ab_wgrad_check.py ab/sq_base.so ab/sq_pipe.so 2>&1 | grep -v "Warning\|amdgpu.ids" | grep "==\|<--" | tee gpurun_out/r6_w7/check.log
export REFNERF_NO_FINITE_CHECK=1
timeout 900 python scripts/ab_train_modes.py ab/sq_raw2.so ab/sq_pipe.so ab/sq_pipe_nodma.so ab/sq_pipe_nodma_nomfma.so ab/sq_pipe.so 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_w7/ab.log
