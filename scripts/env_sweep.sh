#!/bin/bash
# C2 solve time under tuning hooks of the engine
cd "$(dirname "$0")/.."
for env in "" "LK_FORCE_PERSISTENT=1" "LK_FORCE_PERSISTENT=1 LK_ALIGN=0" "LK_ALIGN=0" "LK_FORCE_GROUP=16 LK_FORCE_PERSISTENT=1" "LK_FORCE_GROUP=16 LK_FORCE_PERSISTENT=1 LK_ALIGN=0" "LK_FORCE_GROUP=64" "LK_FORCE_GROUP=64 LK_FORCE_PERSISTENT=1"; do
  echo "== ${env:-default}"
  env $env timeout -k 5 120 python scripts/quick_solve.py ${1:-C2} 30 2>&1 | tail -1
done
