#!/bin/bash
# Kernel tuning builds: scripts/tune_build.sh NAME [extra hipcc flags...]  ->  build/tune/liblk_NAME.so
# (affine + reference bicubic solve kernels only; select with LK_ENGINE_LIB=build/tune/liblk_NAME.so)
set -e
cd "$(dirname "$0")/../correlation_amd/csrc"
name=$1; shift
mkdir -p ../../build/tune
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -DLK_TUNE_ONLY_AFFINE_BICUBIC "$@" \
  -o ../../build/tune/liblk_$name.so lk_engine.cpp lk_tracker.cpp lk_group.cpp lk_image_io.cpp lk_kernels.hip -lrocprofiler-sdk-roctx -lrccl -lz
echo build/tune/liblk_$name.so
