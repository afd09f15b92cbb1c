#!/bin/bash
OUT=gpurun_out/r6_s9
mkdir -p $OUT
for v in "" ab/both_logfocml.so; do
  REFNERF_LIB=$v python -m pytest tests/test_geometry_losses.py -m gpu -q -s -p no:cacheprovider -k "full_loss_set" 2>&1 | grep -E "gradient rel-L2|passed|failed" | sed "s#^#[${v:-det logf}] #"
done | tee $OUT/geom_ab.log
python -m pytest tests/test_hip_f16x2.py tests/test_geometry_losses.py tests/test_hip_shards.py -m gpu -q -p no:cacheprovider -k "train or chain or trajectory or optimiser or loss or shard or c5" 2>&1 | grep -E "^FAILED|passed|failed" | tail -n 5
python scripts/time_train.py f16x2 2>&1 | tail -n 1 | tee $OUT/time_train.log
python scripts/time_train.py f16x2 2>&1 | tail -n 1 | tee -a $OUT/time_train.log
