#!/bin/bash
# A/B of tuning libs on the workgroup-wide classes: config 3 (whole, annulus, blob), config 1, mixed sizes
libs=${1:-base}; reps=${2:-2}
for rep in $(seq $reps); do
for lib in $libs; do
  export LK_ENGINE_LIB=$PWD/build/tune/liblk_$lib.so
  for only in "" annulus blob; do echo "== $lib C3 $only"; LK_C3_ONLY=$only timeout -k 10 200 python3 scripts/quick_c3.py 6 2>&1 | tail -1; done
  echo "== $lib config1"; timeout -k 10 200 python3 scripts/config1.py 2>&1 | tail -3
  echo "== $lib mixed"; timeout -k 10 200 python3 scripts/mixed_classes.py 2>&1 | tail -2
done
done
