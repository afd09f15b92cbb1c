#!/bin/bash
# round 6: GPU suite + training timing on the in-tree build (weight-gradient GEMM without its conversion pass)
mkdir -p gpurun_out/r6_w11
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 15 | tee gpurun_out/r6_w11/pytest.log
timeout 300 python scripts/time_train.py f16x2 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_w11/train.log
REFNERF_WGRAD_MODE=bf16x3 timeout 300 python scripts/time_train.py f16x2 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a gpurun_out/r6_w11/train.log
