#!/bin/bash
# in-tree build with the training forward's lane constants formed per section (152 -> 48 B of scratch): whole suite twice the training tests, timing
mkdir -p gpurun_out/r6_w22
python -m pytest tests -m gpu -q 2>&1 | tail -n 4 | tee gpurun_out/r6_w22/pytest.log
python -m pytest tests/test_hip_f16x2.py tests/test_geometry_losses.py tests/test_hip_shards.py -m gpu -q -p no:cacheprovider -k "train or chain or trajectory or optimiser or loss or shard or c5 or normals" 2>&1 | tail -n 2
python scripts/time_train.py f16x2 2>&1 | tail -n 1
python scripts/time_train.py f16x2 2>&1 | tail -n 1
