#!/bin/bash
mkdir -p gpurun_out/r6_w14
timeout 600 python scripts/ab_wgrad_check.py ab/sq_head.so ab/sq_kmin.so ab/sq_swz.so 2>&1 | grep -v "Warning\|amdgpu.ids" | grep "==\|<--" | tee gpurun_out/r6_w14/check.log
timeout 900 python scripts/ab_train_modes.py ab/sq_head.so ab/sq_kmin.so ab/sq_swz.so ab/sq_head.so ab/sq_kmin.so ab/sq_swz.so 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_w14/ab.log
