#!/bin/bash
mkdir -p gpurun_out/r6_w17
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 6 | tee gpurun_out/r6_w17/pytest.log
timeout 300 python scripts/time_train.py f16x2 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_w17/train.log
REFNERF_WGRAD_MODE=bf16x3 timeout 300 python scripts/time_train.py f16x2 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a gpurun_out/r6_w17/train.log
