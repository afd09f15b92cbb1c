#!/bin/bash
mkdir -p gpurun_out/r6_w3
timeout 900 python scripts/ab_wgrad_check.py ab/sq_base.so ab/sq_raw.so 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_w3/check.log
