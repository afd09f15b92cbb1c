#!/bin/bash
# A/B builds of the round-5 training kernels: scripts/build_sq_variant.sh <tag> [-D... flags]
# -> ab/sq_<tag>.so = refnerf_sq_train.hip compiled with the flags, linked with the in-tree refnerf_hip.o (never touches the
# in-tree library).  Time them with: python scripts/ab_train_modes.py ab/sq_a.so ab/sq_b.so
set -e
cd "$(dirname "$0")/../refnerf-pl_amd/csrc"
tag=$1; shift
mkdir -p ../../ab
[ -f refnerf_hip.o ] || make refnerf_hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -I../../include -I. "$@" -c refnerf_sq_train.hip -o ../../ab/sq_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared refnerf_hip.o ../../ab/sq_$tag.o -o ../../ab/sq_$tag.so
rm -f ../../ab/sq_$tag.o
echo built ab/sq_$tag.so "$@"
