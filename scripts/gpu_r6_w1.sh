#!/bin/bash
# round 6: the weight-gradient GEMM taken apart (timing-only builds of scripts/build_sq_variant.sh)
mkdir -p gpurun_out/r6_w1
export REFNERF_NO_FINITE_CHECK=1
timeout 1500 python scripts/ab_train_modes.py ab/sq_base.so ab/sq_slmaj.so ab/sq_nostage.so ab/sq_nocomp.so ab/sq_dmaonly.so ab/sq_nodma.so ab/sq_base.so 2>&1 | grep -v Warning | tee gpurun_out/r6_w1/ab.log
