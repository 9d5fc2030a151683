#!/bin/bash
# does config 5's one-lane kernel (3123 wavefronts for 3072 slots) pay a second round?  the same sectors on a 443 x 443 grid (3067 wavefronts) beside it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in 447 443; do
  out=gpurun_out/c5rounds/g$g
  mkdir -p $out
  LK_GRID=$g timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 scripts/quick_solve.py C5 5 > $out/log.txt 2>&1
  echo "== grid $g x $g"; grep solve_ms $out/log.txt | tail -2
  grep "lk_solve_kernel" $(find $out -name "*kernel_stats.csv") | cut -d, -f1-4,6,7 | cut -c30-160
  find $out -name "*kernel_trace.csv" -delete
done
