#!/bin/bash
# kernel trace of a few training steps with two builds: the small kernels' own durations
mkdir -p gpurun_out/r6_w15
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in head swz; do
  export REFNERF_LIB=ab/sq_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6_w15/$v -o t -- python3 scripts/time_train.py f16x2 > gpurun_out/r6_w15/$v.log 2>&1
  grep -h "delta_kappa_min\|wgrad_sq256\|wgrad_reduce\|level_bwd_sq\|level_fwd_train_sq_h" $(find gpurun_out/r6_w15/$v -name "*kernel_stats.csv") | cut -d, -f1-4,6-7 | sed "s/^/[$v] /"
  find gpurun_out/r6_w15/$v -name "*.db" -delete; find gpurun_out/r6_w15/$v -size +4M -delete
done
