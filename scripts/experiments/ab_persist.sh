#!/bin/bash
# C2 launch shape experiments: persistent queue / lane-group width (tuning lib given as $1, default product lib)
lib=${1:-}
run() { echo "== $*"; env "$@" ${lib:+LK_ENGINE_LIB=$PWD/$lib} timeout -k 10 200 python3 scripts/quick_solve.py C2 10 2>&1 | tail -1; }
run LK_X=0
run LK_FORCE_PERSISTENT=1
run LK_FORCE_GROUP=16
run LK_FORCE_GROUP=16 LK_FORCE_PERSISTENT=1
run LK_FORCE_GROUP=64
run LK_X=0
