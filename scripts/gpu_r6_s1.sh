#!/bin/bash
# round 6, session 1: float64 evidence for the RGB gate + index-exactness on the oracle's step function, A/B of the weight stream's cache policy
OUT=gpurun_out/r6_s1
mkdir -p $OUT
make -C oracle all > $OUT/oracle_build.log 2>&1
python -m pytest tests/test_hip_parity.py -m gpu -q -x -p no:cacheprovider -k "sampler or level_forward_vs_oracle" > $OUT/pytest_sampler.log 2>&1; echo "pytest sampler rc=$?"; tail -n 3 $OUT/pytest_sampler.log
python scripts/parity_f64_collect.py > $OUT/f64_collect.log 2>&1; echo "collect rc=$?"; cat $OUT/f64_collect.log | tail -n 12
for v in "" ab/main_aux2.so ab/main_aux16.so; do
  echo "== lib ${v:-in-tree}"
  REFNERF_LIB=$v python scripts/time_modes.py 4096 128 f16x2 bf16 2>&1 | tail -n 2
  REFNERF_LIB=$v python scripts/time_modes.py 4096 128 f16x2 2>&1 | tail -n 1
done | tee $OUT/ab_stream_policy.log
