#!/bin/bash
cd "$(dirname "$0")"
for b in bin/a0_rand bin/b7_rand bin/hp6 bin/hb_check bin/hb bin/hb_noprio bin/hb_lockstep bin/hb_epi60 bin/a3_rand_epi60; do timeout 60 ./$b 2048; done
