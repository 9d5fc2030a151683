#!/bin/bash
# are the lane-group widths of the size classes still the best after four rounds of kernel changes?  (LK_FORCE_GROUP)
for wl in C5 C4B C4 C2; do for g in 0 16 32 64; do
  echo "== $wl group $g: $(LK_FORCE_GROUP=$g timeout -k 10 200 python3 scripts/quick_solve.py $wl 6 2>&1 | grep solve_ms | tail -1 | cut -c1-110)"
done; done
