#!/bin/bash
# A/B of the C2 one-pair launch: build/ab/<sha> (git worktrees with their own tuning build) against tuning builds of the working tree
#   for sha in 83d2d7b d73d63d; do git worktree add --detach build/ab/$sha $sha && (cd build/ab/$sha && scripts/tune_build.sh ab); done
#   scripts/tune_build.sh v3;  gpurun -- 'scripts/ab_c2.sh v3'      (results of round 4: profiles/r04_ab_c2.txt)
for rep in 1 2 3; do
  for sha in 83d2d7b d73d63d; do
    (cd build/ab/$sha && echo -n "$sha rep $rep: " && LK_ENGINE_LIB=$PWD/build/tune/liblk_ab.so timeout -k 5 100 python scripts/quick_solve.py C2 60 2>&1 | grep solve_ms | cut -c1-60)
  done
  for v in "$@"; do
    echo -n "$v    rep $rep: "; LK_ENGINE_LIB=$PWD/build/tune/liblk_$v.so timeout -k 5 100 python scripts/quick_solve.py C2 60 2>&1 | grep solve_ms | cut -c1-60
  done
done
