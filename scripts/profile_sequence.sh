#!/bin/bash
# Evidence of the frame-pipelined windows for profiles/ (run through gpurun from the repo root): scripts/profile_sequence.sh r04
#  1. rocprofv3 --kernel-trace --stats of scripts/quick_sequence.py (window + the one-pair loop of the same sequence)
#     -> <tag>_seq_<WL>_kernel_stats.csv
#  2. separate --pmc passes over the window alone -> <tag>_seq_pmc.txt, <tag>_seq_traffic.json (VALU instructions and HBM
#     bytes per pair of a window: bench.py's sequence.*.window.valu_issue_frac)
#  3. build/tune/liblk_trace_seq*.so present: the per-wavefront account of one window -> <tag>_seq_wave_timeline.txt
set -uo pipefail
tag=${1:-r04}
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
for wl in C2 C4; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/seqtrace_$wl" -o t -- python3 scripts/quick_sequence.py $wl 64 3 > "$out/${tag}_seq_${wl}.log" 2>&1
  find "$out/seqtrace_$wl" -name '*kernel_stats.csv' -exec cp {} "$out/${tag}_seq_${wl}_kernel_stats.csv" \;
  rm -rf "$out/seqtrace_$wl"
  echo "trace $wl done" >> "$out/progress.log"
done
: > "$out/${tag}_seq_pmc.txt"
for cfg in "C2 default" "C2 reference_order" "C4 default" "C4 reference_order" "C4 batch_invariant"; do
  set -- $cfg
  i=0
  for set in "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    LK_MODE=$2 LK_SEQ_LOOP=0 timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d "$out/pmc$i" -o pmc -- python3 scripts/quick_sequence.py $1 64 1 > "$out/spmc.log" 2>&1 || echo "pass $i ($set) of $cfg failed" >> "$out/progress.log"
  done
  python3 scripts/summarize_pmc.py "$out" "$tag: window of 64 pairs, $1, LK_MODE=$2 (scripts/quick_sequence.py $1 64 1: three launches of the window kernel)" | sed 's/ on `python3 scripts\/quick_solve.py C2 5`, MI355X,//' | grep -v "^units\|lk_pyramid\|lk_guess\|lk_set_views\|lk_stale" >> "$out/${tag}_seq_pmc.txt"
  rm -rf "$out"/pmc*/
  echo "pmc $cfg done" >> "$out/progress.log"
done
: > "$out/${tag}_seq_wave_timeline.txt.new"
for lib in build/tune/liblk_trace_seq32.so:C2:default build/tune/liblk_trace_seq16.so:C4:default build/tune/liblk_trace_seq16.so:C2:reference_order; do
  IFS=: read -r so wl mode <<< "$lib"
  [ -f "$so" ] && LK_MODE=$mode LK_ENGINE_LIB=$PWD/$so timeout -k 10 200 python3 scripts/trace_sequence.py $wl 64 2>&1 | grep -v amdgpu.ids >> "$out/${tag}_seq_wave_timeline.txt.new" && echo >> "$out/${tag}_seq_wave_timeline.txt.new"
done
# (a timeline is kept only if the traced builds ran - stale or missing tuning libraries leave the committed one alone)
if grep -q "clock" "$out/${tag}_seq_wave_timeline.txt.new" 2>/dev/null; then mv "$out/${tag}_seq_wave_timeline.txt.new" "$out/${tag}_seq_wave_timeline.txt"; else rm -f "$out/${tag}_seq_wave_timeline.txt.new"; fi
# the windows' per-pair constants into <tag>_traffic.json (beside the one-pair constants of scripts/profile_round.sh, if that ran first)
python3 - "$out/${tag}_seq_pmc.txt" "$out/${tag}_traffic.json" "$out/${tag}_seq_wave_timeline.txt" "$tag" <<'PY'
import json, os, re, sys
pmc, dst, timeline, tag = sys.argv[1:5]
base = dst if os.path.exists(dst) else f"profiles/{tag}_traffic.json"   # (the committed file's other constants stay)
d = json.load(open(base)) if os.path.exists(base) else {}
valu, hbm = {}, {}
for b in re.split(r"rocprofv3 --pmc passes", open(pmc).read())[1:]:
    m = re.search(r"window of 64 pairs, (\w+), LK_MODE=(\w+)", b)
    vals = {mm.group(1): float(mm.group(2)) for mm in re.finditer(r"> (\w+)\s+dispatches=\d+ mean=([0-9.e+]+)", b)}
    if m and "SQ_INSTS_VALU" in vals:
        key = f"{m.group(1)}_{m.group(2)}"
        valu[key] = vals["SQ_INSTS_VALU"] / 64
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            hbm[key] = (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 / 64
valu = dict({k: v for k, v in d.get("valu_insts_per_window_pair", {}).items() if k != "source"}, **valu)
hbm = dict({k: v for k, v in d.get("hbm_bytes_per_window_pair", {}).items() if k != "source"}, **hbm)
d["valu_insts_per_window_pair"] = dict(valu, source=f"profiles/{tag}_seq_pmc.txt: SQ_INSTS_VALU of one 64-pair window launch / 64")
d["hbm_bytes_per_window_pair"] = dict(hbm, source=f"profiles/{tag}_seq_pmc.txt: (FETCH_SIZE + WRITE_SIZE) KiB of one 64-pair window launch / 64")
clk = d.get("measured_clock_GHz", {"one_pair_C2": 2.07})
if os.path.exists(timeline):
    for blk in open(timeline).read().split("\n\n"):
        m = re.match(r"(C\w+):", blk.strip())
        mm = re.search(r"mode (\w+):", blk)
        c = re.search(r"clock ([0-9.]+) GHz", blk)
        if m and mm and c:
            clk[f"window_{m.group(1)}_{mm.group(1)}"] = float(c.group(1))
clk["source"] = f"profiles/{tag}_seq_wave_timeline.txt, profiles/*_wave_timeline.txt: shader cycles / device time of the traced wavefronts"
d["measured_clock_GHz"] = clk
json.dump(d, open(dst, "w"), indent=1)
PY
cat "$out/${tag}_seq_pmc.txt"; cat "$out/${tag}_seq_wave_timeline.txt" 2>/dev/null
