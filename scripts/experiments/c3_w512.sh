#!/bin/bash
# config 3 with the 512-thread instances held to 128 VGPRs (two workgroups per CU; build/tune/liblk_w512.so) against the shipped library
for lib in "" $PWD/build/tune/liblk_w512.so; do for only in "" annulus blob; do
  echo "== lib ${lib:-shipped} only ${only:-both}: $(LK_ENGINE_LIB=$lib LK_C3_ONLY=$only timeout -k 10 200 python3 scripts/quick_c3.py 8 2>&1 | grep solve_ms | tail -1 | cut -c1-110)"
done; done
