#!/bin/bash
# Collects the evidence kept under profiles/ on the GPU box (run through gpurun from the repo root):
#   scripts/profile_round.sh r01
# 1. bench.py under rocprofv3 --kernel-trace --stats -> <tag>_bench.json, <tag>_bench_kernel_stats.csv
# 2. separate --pmc passes on scripts/quick_solve.py (HBM traffic, L2, SQ activity) -> <tag>_pmc_summary.txt
# PMC passes never share a run with tracing (pool rule) and the program follows `--` directly.
set -uo pipefail
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
# The default command keeps 3 pairs in flight on 16-lane-group engines and measures the single launch
# on a 32-lane-group engine: the two template instances show up as separate rows of the stats, and the
# row of the 32-lane instance (launched one at a time only) is what roofline.kernel_ms has to agree with.
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python3 bench.py --steps 100 --warmup 10 > "$out/${tag}_bench.json" 2> "$out/bench.err"
find "$out/trace" -name '*kernel_stats.csv' -exec cp {} "$out/${tag}_bench_kernel_stats.csv" \;
find "$out/trace" -name '*kernel_trace.csv' -delete   # tens of MB; the stats are what is kept
echo "bench traced" >> "$out/progress.log"
# one pair at a time throughout (the 'sequential' block of the default output as its own run)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace1" -o bench -- python3 bench.py --inflight 1 --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs > "$out/${tag}_bench_inflight1.json" 2> "$out/bench1.err"
find "$out/trace1" -name '*kernel_stats.csv' -exec cp {} "$out/${tag}_bench_inflight1_kernel_stats.csv" \;
find "$out/trace1" -name '*kernel_trace.csv' -delete
echo "bench --inflight 1 traced" >> "$out/progress.log"
# derived counters take a whole pass each on gfx950 ("exceeds the capabilities of the hardware" otherwise)
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum" "SQ_INSTS_VALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i + 1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d "$out/pmc$i" -o pmc -- python3 scripts/quick_solve.py C2 5 > "$out/pmc$i.log" 2>&1 || echo "pass $i ($set) failed" >> "$out/progress.log"
  echo "pass $i done" >> "$out/progress.log"
done
python3 scripts/summarize_pmc.py "$out" "$tag" > "$out/${tag}_pmc_summary.txt"
tail -1 "$out/${tag}_bench.json" | cut -c1-300
cat "$out/${tag}_pmc_summary.txt"
