#!/bin/bash
OUT=gpurun_out/r6_s10
mkdir -p $OUT
for v in "" ab/main_ring4.so; do
  echo "== lib ${v:-in-tree}"
  REFNERF_LIB=$v python scripts/time_modes.py 4096 128 f16x2 bf16 f16 2>&1 | tail -n 3
  REFNERF_LIB=$v python scripts/time_modes.py 4096 128 f16x2 2>&1 | tail -n 1
  REFNERF_LIB=$v python scripts/time_modes.py 8192 192 f16x2 bf16 2>&1 | tail -n 2
done 2>&1 | tee $OUT/ab_ring4.log
