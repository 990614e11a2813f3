#!/bin/bash
# Build an experimental copy of the product library with extra compile flags:
#   tools/build_variant.sh <suffix> [flags...]   -> ray_tracer_2_amd/librt2_mi355x_<suffix>.so
set -e
cd "$(dirname "$0")/.."
SUF=$1; shift
C=ray_tracer_2_amd/csrc
/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -shared \
  -I include "$@" $C/rt_kernel.hip $C/rt_api.hip $C/rt_bvh_search.hip $C/host/obj_loader.cpp $C/host/bvh.cpp $C/host/scene.cpp \
  $C/host/png_decode.cpp $C/host/scene_capi.cpp $C/host/ray_tracer.cpp -lz -o ray_tracer_2_amd/librt2_mi355x_$SUF.so
