#!/bin/bash
# fresh_lane() from builtins (no vector instruction inside inline asm): the rewrites of the training forward's P6 that failed from
# run to run, rebuilt on it; then the whole GPU suite on the in-tree build
mkdir -p gpurun_out/r6_w20
BIS_VARIANTS="V1fix V5fix" bash scripts/gpu_r6_bis.sh
python -m pytest tests -m gpu -q -x 2>&1 | tail -n 4 | tee gpurun_out/r6_w20/pytest.log
python scripts/time_train.py f16x2 2>&1 | tail -n 1
python scripts/time_modes.py 8192 192 f16x2 bf16 2>&1 | tail -n 2
