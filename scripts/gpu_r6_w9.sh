#!/bin/bash
mkdir -p gpurun_out/r6_w9
timeout 600 python scripts/ab_wgrad_check.py ab/sq_base.so ab/sq_pipe3.so 2>&1 | grep -v "Warning\|amdgpu.ids" | grep "==\|<--" | tee gpurun_out/r6_w9/check.log
