#!/bin/bash
# bisect of the training forward's lane-constant rewrite (ab/sq_*.so): which change breaks which test
mkdir -p gpurun_out/r6_bis
for v in ${BIS_VARIANTS:-W1 W2 W3 W4 W5 V1}; do
  for rep in 1 2; do
    REFNERF_LIB=ab/sq_$v.so timeout 900 python -m pytest tests/test_hip_f16x2.py -m gpu -q -p no:cacheprovider -k "chain_training or twenty or split_chain or trained_long_split" 2>&1 | grep -E "passed|failed" | sed "s/^/[$v run $rep] /"
  done
done 2>&1 | tee gpurun_out/r6_bis/bis2.log
