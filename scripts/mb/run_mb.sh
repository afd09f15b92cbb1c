#!/bin/bash
# run every built variant (scripts/mb/build_mb.sh) on the GPU of this box: 2048 workgroups, 272 chunks each
cd "$(dirname "$0")"
for b in bin/*; do ./$b 2048; done
