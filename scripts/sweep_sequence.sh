#!/bin/bash
# tuning sweep of the frame-pipelined windows (run on the GPU box): scripts/sweep_sequence.sh > gpurun_out/.../sweep.log
run() { echo "== $*"; env "$@" LK_SEQ_LOOP=0 timeout -k 10 200 python scripts/quick_sequence.py $W 64 3 2>&1 | grep -v amdgpu.ids | tail -1; }
W=C2
run LK_MODE=default
run LK_MODE=default LK_FORCE_GROUP=16
run LK_MODE=default LK_ALIGN=0
run LK_MODE=default LK_FORCE_GROUP=16 LK_ALIGN=0
run LK_MODE=default LK_SEQ_GRID=750
run LK_MODE=default LK_SEQ_GRID=500
run LK_MODE=default LK_FORCE_GROUP=16 LK_SEQ_GRID=750
run LK_MODE=batch_invariant LK_ALIGN=0
run LK_MODE=reference_order LK_SEQ_GRID=750
run LK_MODE=reference_order LK_SEQ_GRID=1400
W=C4
run LK_MODE=batch_invariant LK_ALIGN=0
run LK_MODE=batch_invariant LK_SEQ_GRID=750
run LK_MODE=reference_order LK_SEQ_GRID=750
run LK_MODE=reference_order LK_SEQ_GRID=1400
