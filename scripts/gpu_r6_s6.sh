#!/bin/bash
OUT=gpurun_out/r6_s6
mkdir -p $OUT
T="tests/test_hip_f16x2.py -k chain_training_step_vs_reference"
for v in ab/sq_23d6ce2.so ab/sq_025fa43.so; do
  REFNERF_LIB=$v python -m pytest $T -m gpu -q -x -p no:cacheprovider > $OUT/$(basename $v).log 2>&1; echo "$v rc=$?"; grep -E "passed|failed|AssertionError" $OUT/$(basename $v).log | head -n 3
done
T2="tests/test_basis.py -k general_basis_training_step"
python -m pytest $T2 -m gpu -q -p no:cacheprovider > $OUT/basis_cur.log 2>&1; echo "basis current rc=$?"; grep -E "passed|failed" $OUT/basis_cur.log | tail -n 2
REFNERF_LIB=ab/main_4c676af.so python -m pytest $T2 -m gpu -q -p no:cacheprovider > $OUT/basis_oldmain.log 2>&1; echo "basis old-main rc=$?"; grep -E "passed|failed" $OUT/basis_oldmain.log | tail -n 2
