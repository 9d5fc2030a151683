#!/bin/bash
# one rank's block of an 8-rank sharded sequence (latency regime: fewer sectors than resident lane groups): which instance is fastest?
run() { echo "== $W $*: $(env "$@" LK_WORLDS=8 timeout -k 10 200 python3 scripts/experiments/shard_forecast.py $W 2>&1 | grep "1/8" | cut -c1-120)"; }
W=C4
run LK_X=0
run LK_SEQ_SMALL=0
run LK_MODE=batch_invariant
run LK_MODE=reference_order
run LK_SEQ_FILL=2000
run LK_SEQ_FILL=800
W=C2
run LK_X=0
run LK_SEQ_FILL=1000
run LK_SEQ_FILL=2000
run LK_FORCE_GROUP=64
run LK_FORCE_GROUP=16
