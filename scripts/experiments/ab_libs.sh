#!/bin/bash
# A/B of tuning libs: scripts/experiments/ab_libs.sh "libA libB" "C2 C4 C5" [reps]   (names as in build/tune/liblk_NAME.so)
libs=${1:-base}; wls=${2:-C2}; reps=${3:-2}
for rep in $(seq $reps); do
for lib in $libs; do
  for wl in $wls; do
    echo "== $lib $wl default"; LK_ENGINE_LIB=$PWD/build/tune/liblk_$lib.so timeout -k 10 200 python3 scripts/quick_solve.py $wl 10 2>&1 | tail -1
  done
done
done
