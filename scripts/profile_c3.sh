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
