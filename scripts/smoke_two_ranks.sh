#!/bin/bash
# Smoke test of bench.py's N > 1 code path on a ONE-GPU box, through bench.py's own launcher (`--gpus 2` starts the
# two rank processes itself): the ranks share device 0 and use gloo instead of RCCL (REFNERF_BENCH_SHARE_GPU /
# REFNERF_BENCH_BACKEND).  usage: scripts/smoke_two_ranks.sh [bench args, e.g. --config C4]
export REFNERF_BENCH_BACKEND=gloo REFNERF_BENCH_SHARE_GPU=1
mkdir -p gpurun_out
timeout -s ABRT 300 python -X faulthandler bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-image "$@" \
  > gpurun_out/n2.log 2> gpurun_out/n2.err
rc=$?
tail -5 gpurun_out/n2.log
tail -5 gpurun_out/n2.err
echo "smoke_two_ranks rc=$rc"
exit $rc
