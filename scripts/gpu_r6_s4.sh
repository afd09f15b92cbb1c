#!/bin/bash
# round 6, session 4: training kernels after the spill removal (tests + timing), A/B of the logits' log on the one gradient test that moved
OUT=gpurun_out/r6_s4
mkdir -p $OUT
python -m pytest tests/test_geometry_losses.py -m gpu -q -s -p no:cacheprovider -k "full_loss_set" > $OUT/geom_detlogf.log 2>&1; echo "geometry (det logf) rc=$?"; grep -E "gradient rel-L2|passed|failed" $OUT/geom_detlogf.log
REFNERF_LIB=ab/both_logfocml.so python -m pytest tests/test_geometry_losses.py -m gpu -q -s -p no:cacheprovider -k "full_loss_set" > $OUT/geom_ocml.log 2>&1; echo "geometry (ocml logf) rc=$?"; grep -E "gradient rel-L2|passed|failed" $OUT/geom_ocml.log
python -m pytest tests -m gpu -q -p no:cacheprovider -k "train or grad or backward or shard or trajectory or optimiser or loss or specular or basis or mismatched" > $OUT/pytest_train.log 2>&1; echo "pytest train rc=$?"; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest_train.log | tail -n 12
python scripts/time_train.py f16x2 bf16 2>&1 | tail -n 3 | tee $OUT/time_train.log
python scripts/time_train.py f16x2 2>&1 | tail -n 1 | tee -a $OUT/time_train.log
