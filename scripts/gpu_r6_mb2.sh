#!/bin/bash
# trunk-loop probe: compiler-placed lgkmcnt(0) waits against hand-counted waits on inline-asm fragment reads
mkdir -p gpurun_out/r6_mb2; cd scripts/mb
for v in "" "-DASMFRAG" "-DNFR=8" "-DASMFRAG -DNFR=8" "-DREORDER" "-DASMFRAG -DREORDER" "-DASMFRAG -DREORDER -DNFR=8" "-DASMFRAG -DREORDER -DBAREBAR" "-DASMFRAG -DREORDER -DNODMA" "-DNODMA"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $v mb_sq32.hip -o /tmp/mb_x 2>/dev/null && echo "[$v] $(/tmp/mb_x)"
done 2>&1 | tee ../../gpurun_out/r6_mb2/mb.log
