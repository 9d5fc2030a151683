#!/bin/bash
# Config 3 evidence for profiles/ (run through gpurun from the repo root): scripts/profile_c3.sh r03
#   <tag>_c3_steps.txt  solve times of config 3 and of its two sector kinds alone, with the round-3 mechanisms switched off one by one
#                       (LK_EVAL_LISTS=0: the reference's list order; LK_TEAM_SHARE=0: team and one-workgroup class take turns),
#                       the launches of one solve (rocprofv3 --kernel-trace) and the reference-order mode
set -uo pipefail
tag=${1:-r03}
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
f="$out/${tag}_c3_steps.txt"
: > "$f"
run() { echo "== $*" >> "$f"; env "$@" timeout -k 10 250 python3 scripts/quick_c3.py 8 2>&1 | tail -1 >> "$f"; }
run LK_X=0
run LK_TEAM_SHARE=0
run LK_EVAL_LISTS=0
run LK_EVAL_LISTS=0 LK_TEAM_SHARE=0
run LK_C3_ONLY=annulus
run LK_C3_ONLY=annulus LK_EVAL_LISTS=0
run LK_C3_ONLY=blob
run LK_REF_ORDER=1 LK_C3_ONLY=annulus
echo "== launches of two solves (default)" >> "$f"
timeout -k 10 250 rocprofv3 --kernel-trace -d "$out/c3" -o t -- python3 scripts/quick_c3.py 3 > "$out/c3.log" 2>&1
python3 scripts/c4_chain_timeline.py "$out/c3/t_results.db" 6 >> "$f" 2>&1
rm -rf "$out/c3"
cat "$f"
# counters of the 512-thread instance on the annulus, lists in the reference's order against the row-major copy
# (separate --pmc passes, no tracing): L2 requests (TCC_HIT + TCC_MISS), VALU instructions, wait and wave cycles
for ev in 0 1; do
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE"; do
    i=$((i + 1))
    LK_C3_ONLY=annulus LK_EVAL_LISTS=$ev timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d "$out/pmc$i" -o pmc -- python3 scripts/quick_c3.py 2 > "$out/c3pmc_${ev}_$i.log" 2>&1 || echo "c3 pass $ev/$i failed" >> "$out/progress.log"
  done
  python3 scripts/summarize_pmc.py "$out" "$tag config 3 annulus, LK_EVAL_LISTS=$ev" 2>/dev/null | grep -E "units|passes|lk_solve_kernel<3, 2, 512" > "$out/${tag}_c3_pmc_eval$ev.txt"
  rm -rf "$out"/pmc*/
done
cat "$out/${tag}_c3_pmc_eval0.txt" "$out/${tag}_c3_pmc_eval1.txt"
