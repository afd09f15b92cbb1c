#!/bin/bash
# round 6, session 2: rendezvous without the fence's lgkmcnt drain (wavefront-scope fences keep the allocator at 0 spills), L2 prefetch waves
OUT=gpurun_out/r6_s2
mkdir -p $OUT
for v in "" ab/main_bare5.so ab/main_bare6.so; do
  echo "== lib ${v:-in-tree}"
  REFNERF_LIB=$v python scripts/time_modes.py 4096 128 f16x2 bf16 f16 2>&1 | tail -n 3
  REFNERF_LIB=$v python scripts/time_modes.py 4096 128 f16x2 2>&1 | tail -n 1
done 2>&1 | tee $OUT/ab_bare.log
for v in ab/main_l2pf4.so ab/main_l2pf8.so; do
  echo "== lib $v (REFNERF_LDS_PAD=6400)"
  REFNERF_LDS_PAD=6400 REFNERF_LIB=$v python scripts/time_modes.py 4096 128 f16x2 2>&1 | tail -n 1
  REFNERF_LDS_PAD=6400 REFNERF_LIB=$v python scripts/time_modes.py 4096 128 f16x2 2>&1 | tail -n 1
done 2>&1 | tee $OUT/ab_l2pf.log
echo "== in-tree with REFNERF_LDS_PAD=6400"; REFNERF_LDS_PAD=6400 python scripts/time_modes.py 4096 128 f16x2 2>&1 | tail -n 1
