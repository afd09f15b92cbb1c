#!/bin/bash
# A/B of bench.py's training-step kernels for several library builds: scripts/ab_bench_train.sh a.so b.so ...
# (each build is copied over the in-tree library for its run; the original is restored at the end)
LIB=refnerf-pl_amd/csrc/librefnerf_hip.so
cp $LIB /tmp/lib_keep.so
for rep in 1 2; do
for so in "$@"; do
  cp $so $LIB
  REFNERF_BENCH_PROBE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-image > /tmp/ab_bt.json 2>/dev/null
  python - "$so" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_bt.json"))
for k in ("train_step", "train_step_bf16"):
    print(sys.argv[1], k, round(d[k]["ms_per_step"], 3), "loss", d[k]["loss"], {kk: round(v["avg_launch_ms"], 3) for kk, v in d[k]["kernels"].items()})
PY
done; done
cp /tmp/lib_keep.so $LIB
