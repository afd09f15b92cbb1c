#!/bin/bash
# builds happen in the container (hipcc cross-compiles); this just runs the variants on the GPU box
cd $(dirname $0)
for b in bin5/*; do ./$b 2048; done
