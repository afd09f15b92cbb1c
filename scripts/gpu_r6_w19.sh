#!/bin/bash
# ring variants without scratch: GPU suite, C3 bench, C3 profile passes
mkdir -p gpurun_out/session_r6f
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/session_r6f/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/session_r6f/pytest.log; tail -n 3 gpurun_out/session_r6f/pytest.log
python bench.py --config C3 --steps 20 --warmup 5 > gpurun_out/session_r6f/bench_C3.json 2> gpurun_out/session_r6f/bench_C3.err; echo "bench C3 rc=$?"; cp gpurun_out/bench_full.json gpurun_out/session_r6f/bench_C3.full.json; cut -c1-400 gpurun_out/session_r6f/bench_C3.json
python bench.py > gpurun_out/session_r6f/bench_C2.json 2> gpurun_out/session_r6f/bench_C2.err; echo "bench C2 rc=$?"; cp gpurun_out/bench_full.json gpurun_out/session_r6f/bench_C2.full.json; cut -c1-400 gpurun_out/session_r6f/bench_C2.json
bash scripts/gpu_r6_prof.sh
