#!/bin/bash
# cycles per step of the 512-thread instance on config 3's annulus (one workgroup per sector), without / with the software pipeline
mkdir -p gpurun_out/c3trace
for lib in trace512n trace512; do
  echo "== $lib"
  LK_C3_ONLY=annulus LK_TRACE_OUT=gpurun_out/c3trace/$lib.npz LK_ENGINE_LIB=$PWD/build/tune/liblk_$lib.so timeout -k 10 200 python3 scripts/quick_c3.py 3 2>&1 | tail -1
  python3 scripts/trace_fine.py gpurun_out/c3trace/$lib.npz
done
