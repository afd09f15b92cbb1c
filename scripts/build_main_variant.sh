#!/bin/bash
# A/B builds of the first translation unit: scripts/build_main_variant.sh <tag> [-D... flags] -> ab/main_<tag>.so
set -e
cd "$(dirname "$0")/../refnerf-pl_amd/csrc"
tag=$1; shift
mkdir -p ../../ab
[ -f refnerf_sq_train.o ] || make refnerf_sq_train.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -I../../include -I. "$@" -c refnerf_hip.hip -o ../../ab/main_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared ../../ab/main_$tag.o refnerf_sq_train.o -o ../../ab/main_$tag.so
rm -f ../../ab/main_$tag.o
echo built ab/main_$tag.so "$@"
