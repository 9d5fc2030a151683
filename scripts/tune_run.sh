#!/bin/bash
# Times the C2 solve of every tuning build (scripts/tune_build.sh): scripts/tune_run.sh [workload] [names...]
cd "$(dirname "$0")/.."
wl=${1:-C2}; shift
names=${@:-$(ls build/tune/ | sed 's/liblk_\(.*\)\.so/\1/')}
for n in $names; do
  for g in "" 16; do
    echo "== $n group=${g:-default}"
    LK_ENGINE_LIB=$PWD/build/tune/liblk_$n.so LK_FORCE_GROUP=$g timeout -k 5 120 python scripts/quick_solve.py $wl 30 2>&1 | tail -1
  done
done
