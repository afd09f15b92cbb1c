#!/bin/bash
mkdir -p gpurun_out/r6_w12
timeout 600 python scripts/ab_wgrad_check.py ab/sq_cur.so ab/sq_cv1.so 2>&1 | grep -v "Warning\|amdgpu.ids" | grep "==\|<--" | tee gpurun_out/r6_w12/check.log
timeout 900 python scripts/ab_train_modes.py ab/sq_cur.so ab/sq_cv1.so ab/sq_cur.so ab/sq_cv1.so 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_w12/ab.log
