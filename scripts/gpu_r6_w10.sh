#!/bin/bash
mkdir -p gpurun_out/r6_w10
timeout 900 python scripts/ab_wgrad_check.py ab/sq_base.so ab/sq_pipe3.so ab/sq_pipe3_split8.so 2>&1 | grep -v "Warning\|amdgpu.ids" | grep "==\|<--\|big\|Error\|error" | tee gpurun_out/r6_w10/check.log
