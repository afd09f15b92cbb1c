#!/bin/bash
# cache policy of the weight-gradient GEMM's operand stream (read-once data): default / nt / sc0 nt / sc1 / sc1 nt
mkdir -p gpurun_out/r6_w18
for m in f16 bf16x3; do
  export REFNERF_WGRAD_MODE=$m; echo "== wgrad mode $m"
  timeout 900 python scripts/ab_train_modes.py ab/sq_aux0.so ab/sq_aux2.so ab/sq_aux3.so ab/sq_aux16.so ab/sq_aux18.so ab/sq_aux0.so ab/sq_aux2.so 2>&1 | grep -v "Warning\|amdgpu.ids"
done | tee gpurun_out/r6_w18/ab.log
