#!/bin/bash
# the training forward with its lane constants formed per section (ab/sq_V6.so: 152 -> 80 B of scratch): training tests twice, whole suite, timing
mkdir -p gpurun_out/r6_w21
BIS_VARIANTS="V6" bash scripts/gpu_r6_bis.sh
REFNERF_LIB=ab/sq_V6.so python -m pytest tests -m gpu -q 2>&1 | tail -n 4 | tee gpurun_out/r6_w21/pytest.log
python scripts/ab_train_modes.py - ab/sq_V6.so - ab/sq_V6.so 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_w21/ab.log
