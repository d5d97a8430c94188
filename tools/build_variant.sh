#!/bin/bash
# build_variant.sh NAME [extra hipcc flags]: a second build of the library for A/B runs (select with RRT_HIP_LIBRARY)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p build_variants
cd rustraytracer_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -munsafe-fp-atomics -fno-gpu-rdc -fno-slp-vectorize -ffp-contract=on -Wall -Wno-unused-function "$@" -x hip rrt_api.cpp rrt_kernels.hip rrt_lbvh.hip rrt_ploc.hip rrt_gsah.hip rrt_wavefront.hip -o ../../build_variants/librrt_$NAME.so
