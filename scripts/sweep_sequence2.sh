#!/bin/bash
run() { echo "== $W $*"; env "$@" LK_SEQ_LOOP=0 timeout -k 10 200 python scripts/quick_sequence.py $W $N 3 2>&1 | grep -v amdgpu.ids | tail -1; }
N=64
W=C2
run LK_MODE=default
run LK_MODE=default LK_SEQ_FILL=560
run LK_MODE=default LK_SEQ_FILL=720
run LK_MODE=batch_invariant
run LK_MODE=reference_order
W=C4
run LK_MODE=default
run LK_MODE=batch_invariant
run LK_MODE=reference_order
W=C4B
run LK_MODE=default
run LK_MODE=default LK_SEQ_SMALL=100
run LK_MODE=batch_invariant
run LK_MODE=batch_invariant LK_SEQ_SMALL=100
run LK_MODE=reference_order
