#!/bin/bash
# the 22-bit mode's spatial jobs on the raw-fragment body (ab/sq_lo.so) against the committed build (ab/sq_head.so)
mkdir -p gpurun_out/r6_w16
timeout 600 python scripts/ab_wgrad_check.py ab/sq_head.so ab/sq_lo.so 2>&1 | grep -v "Warning\|amdgpu.ids" | grep "==\|<--" | tee gpurun_out/r6_w16/check.log
export REFNERF_WGRAD_MODE=bf16x3
timeout 900 python scripts/ab_train_modes.py ab/sq_head.so ab/sq_lo.so ab/sq_head.so ab/sq_lo.so 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_w16/ab.log
