#!/bin/bash
# config 5 / config 4: the starved-level kernel's evaluation budget per sector (LK_EVAL_CAP) against the chain's time
for wl in C5 C4; do for cap in 6 8 10 12 16 20 28; do
  echo "== $wl cap $cap: $(LK_EVAL_CAP=$cap timeout -k 10 200 python3 scripts/quick_solve.py $wl 6 2>&1 | grep solve_ms | tail -1)"
done; done
