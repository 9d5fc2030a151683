#!/bin/bash
# A/B of tuning libs in reference-order mode: ab_ref.sh "libA libB" "C2 C4 C5" [reps]
libs=${1:-base}; wls=${2:-C2}; reps=${3:-2}
for rep in $(seq $reps); do
for lib in $libs; do
  for wl in $wls; do
    echo "== $lib $wl reference order"; LK_REF_ORDER=1 LK_ENGINE_LIB=$PWD/build/tune/liblk_$lib.so timeout -k 10 300 python3 scripts/quick_solve.py $wl 10 2>&1 | tail -1
  done
done
done
