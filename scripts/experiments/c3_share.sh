#!/bin/bash
# config 3: the team class's share of the workgroup slots (LK_TEAM_SHARE, permille; default 1250 x its share of the samples = 398)
for s in ${SHARES:-250 280 310 340 370 398 430}; do
  echo "== share $s: $(LK_TEAM_SHARE=$s timeout -k 10 200 python3 scripts/quick_c3.py 8 2>&1 | grep solve_ms | tail -1 | cut -c1-140)"
done
