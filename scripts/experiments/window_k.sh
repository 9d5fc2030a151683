#!/bin/bash
# lk_sequence_run on config 4 (64 pairs, host frames): pairs per window (LK_SEQ_WINDOW)
for k in 8 16 24 32 64; do
  echo "== window $k: $(LK_SEQ_WINDOW=$k LK_ONLY_FIRST=1 timeout -k 10 300 python3 scripts/run_tracking_modes.py 65 2>&1 | grep eulerian/first | head -1 | cut -c1-140)"
done
