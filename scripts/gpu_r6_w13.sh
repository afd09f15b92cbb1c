#!/bin/bash
mkdir -p gpurun_out/r6_w13
export REFNERF_NO_FINITE_CHECK=1
timeout 900 python scripts/ab_train_modes.py ab/sq_nomfma.so ab/sq_nomfma_contig.so ab/sq_nomfma.so ab/sq_nomfma_contig.so 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r6_w13/ab.log
