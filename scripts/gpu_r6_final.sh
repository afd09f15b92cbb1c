#!/bin/bash
# round 6, final build: the whole GPU session (suite, smoke, bench in every configuration, two-rank smokes) + the r06 profiles
bash scripts/gpu_session.sh r6g > gpurun_out/session_r6g.log 2>&1; tail -n 12 gpurun_out/session_r6g.log | cut -c1-300
bash scripts/gpu_r6_prof.sh
