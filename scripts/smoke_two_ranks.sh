#!/bin/bash
# Smoke test of bench.py's N > 1 code path on a ONE-GPU box: two ranks share the GPU, gloo instead of RCCL
# (REFNERF_BENCH_BACKEND / REFNERF_BENCH_SHARE_GPU); each rank runs under a 90 s watchdog that dumps its Python stack.
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 WORLD_SIZE=2 REFNERF_BENCH_BACKEND=gloo REFNERF_BENCH_SHARE_GPU=1
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT 90 python -X faulthandler bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/n2_rank$r.log 2>&1 &
done
wait
tail -25 gpurun_out/n2_rank0.log
echo ---- rank1
tail -25 gpurun_out/n2_rank1.log
