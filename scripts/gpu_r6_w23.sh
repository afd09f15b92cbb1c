#!/bin/bash
# training forward without scratch (ab/sq_bnl2.so: bottleneck fragments re-read inside the skip layer, shuffles on a fresh lane index):
# training tests twice, whole suite, timing against the in-tree build
mkdir -p gpurun_out/r6_w23
BIS_VARIANTS="bnl2" bash scripts/gpu_r6_bis.sh
REFNERF_LIB=ab/sq_bnl2.so python -m pytest tests -m gpu -q 2>&1 | tail -n 4 | tee gpurun_out/r6_w23/pytest.log
python scripts/ab_train_modes.py - ab/sq_bnl2.so - ab/sq_bnl2.so 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_w23/ab.log
