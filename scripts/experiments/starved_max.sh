#!/bin/bash
# the starved rule (a level with at most LK_STARVED_MAX samples - default 2P = 12 - is solved with the reference's sequential sums and QR):
# what do configs 5 and 4 cost, and how far from the CPU engine do they land, when 9-sample levels are left to the lane groups?
for m in 12 8; do
  for wl in C5 C4 C4B; do
    echo "== max $m $wl: $(LK_STARVED_MAX=$m timeout -k 10 200 python3 scripts/quick_solve.py $wl 6 2>&1 | grep solve_ms | tail -1 | cut -c1-120)"
  done
  LK_STARVED_MAX=$m timeout -k 10 600 python -m pytest tests/test_reference_order_gpu.py -q -s -k "config5_sectors or config4_sectors" 2>&1 | grep -E "engine vs oracle|passed|failed|assert" | cut -c1-400
done
