#!/bin/bash
mkdir -p gpurun_out/r6_w4
timeout 600 python scripts/ab_wgrad_check.py ab/sq_base.so ab/sq_raw.so 2>&1 | grep -v "Warning\|amdgpu.ids" | grep "==\|<--" | tee gpurun_out/r6_w4/check.log
export REFNERF_NO_FINITE_CHECK=1
timeout 900 python scripts/ab_train_modes.py ab/sq_base.so ab/sq_raw.so ab/sq_raw_nodma.so ab/sq_raw_nocomp.so ab/sq_raw.so 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_w4/ab.log
