#!/bin/bash
# GPU tests (all) + one bench.py line.  usage: scripts/gpu_quick.sh <tag> [bench args]
TAG=${1:-q}; shift
OUT=gpurun_out/quick_$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|passed|failed" $OUT/pytest.log | tail -15
python bench.py --steps 20 --warmup 5 "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
for k in ("value", "ms_per_step"): print(k, d[k])
print("roofline", {k: d["roofline"][k] for k in ("frac", "avg_launch_ms", "kernel")})
for m in ("f32_mode", "bf16_mode", "f16_mode"):
    if m in d: print(m, d[m]["value"], d[m]["roofline"]["frac"], d[m]["roofline"]["avg_launch_ms"])
for k in ("mode_agreement", "parity", "cpu_baseline", "cpu_baseline_torch", "full_image_render_ms", "llff_image_render_ms"):
    if k in d: print(k, json.dumps(d[k]))
for k in ("train_step", "train_step_bf16"):
    if k in d: print(k, d[k]["ms_per_step"], d[k]["value"], {kk: (v["avg_launch_ms"], round(v["frac"], 3)) for kk, v in d[k]["kernels"].items()})
PY
