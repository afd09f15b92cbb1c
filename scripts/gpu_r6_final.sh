#!/bin/bash
# round 6, final build: the whole GPU session (suite, smoke, bench in every configuration, two-rank smokes) + the r06 profiles
bash scripts/gpu_session.sh r6c > gpurun_out/session_r6c.log 2>&1; tail -n 30 gpurun_out/session_r6c.log
bash scripts/gpu_r6_prof.sh
