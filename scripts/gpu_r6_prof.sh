#!/bin/bash
# round 6: rocprofv3 kernel trace + PMC passes of bench.py at C2 and C3 -> gpurun_out/prof_r06, gpurun_out/prof_r06_C3
bash scripts/profile_gpu.sh r06 > gpurun_out/prof_r06.log 2>&1; echo "C2 prof rc=$?"
BENCH_ARGS="--config C3" bash scripts/profile_gpu.sh r06_C3 > gpurun_out/prof_r06_C3.log 2>&1; echo "C3 prof rc=$?"
du -sh gpurun_out/prof_r06 gpurun_out/prof_r06_C3
