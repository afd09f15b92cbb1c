#!/bin/bash
# round 6, final build (2): GPU suite, smoke, the default bench line, the r06 profiles
mkdir -p gpurun_out/session_r6d
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/session_r6d/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/session_r6d/pytest.log; tail -n 3 gpurun_out/session_r6d/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/session_r6d/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 5 gpurun_out/session_r6d/smoke.log
python bench.py > gpurun_out/session_r6d/bench_C2.json 2> gpurun_out/session_r6d/bench_C2.err; echo "bench rc=$?"; cp gpurun_out/bench_full.json gpurun_out/session_r6d/bench_C2.full.json; cut -c1-900 gpurun_out/session_r6d/bench_C2.json
bash scripts/gpu_r6_prof.sh
