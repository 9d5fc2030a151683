#!/bin/bash
run() { echo "== $W $*"; env "$@" timeout -k 10 250 python3 scripts/experiments/shard_forecast.py $W 2>&1 | grep "1/" | cut -c1-110; }
W=C4
run LK_SEQ_SMALL=0 LK_WORLDS=2,4,8,16
run LK_X=0 LK_WORLDS=16
W=C2
run LK_SEQ_FILL=1000 LK_WORLDS=2,4,8
run LK_SEQ_FILL=820 LK_WORLDS=2,4
