#!/bin/bash
cd "$(dirname "$0")"
for b in bin/m0_nolds bin/m1_deadlds bin/m2_oneacc_nolds; do timeout 60 ./$b 2048; done
