#!/bin/bash
# Reference-order mode (lk_set_reference_order) under the profiler - run through gpurun from the repo root:
#   scripts/profile_reforder.sh r03 [lib]
# 1. solve time of C2 / C4 / C5 in that mode (scripts/quick_solve.py, LK_REF_ORDER=1)
# 2. separate --pmc passes on `quick_solve.py C2 5` (no tracing in those runs) -> <tag>_reforder_pmc.txt
# 3. build/tune/liblk_trace_ord.so (scripts/tune_build.sh trace_ord -DLK_TRACE -D'LK_TRACE_PICK(G,S)=((S)&&(G)<=64&&(G)>1)'):
#    per-wavefront timeline of one C2 launch -> <tag>_reforder_wave_timeline.txt
set -uo pipefail
tag=${1:-r03}
lib=${2:-}
out=gpurun_out/prof_${tag}_reforder
mkdir -p "$out"
export TMPDIR=/tmp
export LK_REF_ORDER=1
[ -n "$lib" ] && export LK_ENGINE_LIB=$PWD/$lib
{
  for wl in C2 C4 C5; do timeout -k 10 200 python3 scripts/quick_solve.py $wl 10 2>&1 | tail -2; done
  echo "== C2, one wavefront per sector (LK_FORCE_GROUP=64)"
  LK_FORCE_GROUP=64 timeout -k 10 200 python3 scripts/quick_solve.py C2 10 2>&1 | tail -1
} > "$out/${tag}_reforder_times.txt"
echo "times done" >> "$out/progress.log"
i=0
for set in "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d "$out/pmc$i" -o pmc -- python3 scripts/quick_solve.py C2 5 > "$out/pmc$i.log" 2>&1 || echo "pass $i ($set) failed" >> "$out/progress.log"
  echo "pass $i done" >> "$out/progress.log"
done
python3 scripts/summarize_pmc.py "$out" "$tag (LK_REF_ORDER=1)" > "$out/${tag}_reforder_pmc.txt"
if [ -f build/tune/liblk_trace_ord.so ]; then
  LK_ENGINE_LIB=$PWD/build/tune/liblk_trace_ord.so timeout -k 10 120 python3 scripts/trace_solve.py "$out/trace_c2_ord.npz" > "$out/trace.log" 2>&1 &&
    { python3 scripts/trace_brief.py "$out/trace_c2_ord.npz"; python3 scripts/trace_report.py "$out/trace_c2_ord.npz"; } > "$out/${tag}_reforder_wave_timeline.txt" 2>&1
fi
rm -rf "$out"/pmc*/
cat "$out/${tag}_reforder_times.txt" "$out/${tag}_reforder_pmc.txt" "$out/${tag}_reforder_wave_timeline.txt"
