#!/bin/bash
# Collects the evidence kept under profiles/ on the GPU box (run through gpurun from the repo root):
#   scripts/profile_round.sh r03
# 1. bench.py (default command: one pair at a time) under rocprofv3 --kernel-trace --stats
#    -> <tag>_bench.json, <tag>_bench_kernel_stats.csv
# 2. separate --pmc passes on scripts/quick_solve.py (HBM traffic, L2, SQ activity) -> <tag>_pmc_summary.txt,
#    <tag>_traffic.json (the per-launch HBM bytes bench.py quotes as roofline.traffic)
# 3. config 4's launch chain (start / end of every launch of one solve) -> <tag>_c4_chain.txt
# 4. if build/tune/liblk_trace.so exists (scripts/tune_build.sh trace -DLK_TRACE): the per-wavefront timeline of
#    one C2 launch -> <tag>_wave_timeline.txt
# PMC passes never share a run with tracing (pool rule) and the program follows `--` directly.
set -uo pipefail
tag=${1:-r03}
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python3 bench.py --steps 100 --warmup 10 > "$out/${tag}_bench.json" 2> "$out/bench.err"
find "$out/trace" -name '*kernel_stats.csv' -exec cp {} "$out/${tag}_bench_kernel_stats.csv" \;
find "$out/trace" -name '*kernel_trace.csv' -delete   # tens of MB; the stats are what is kept
echo "bench traced" >> "$out/progress.log"
# derived counters take a whole pass each on gfx950 ("exceeds the capabilities of the hardware" otherwise)
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum" "SQ_INSTS_VALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i + 1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d "$out/pmc$i" -o pmc -- python3 scripts/quick_solve.py C2 5 > "$out/pmc$i.log" 2>&1 || echo "pass $i ($set) failed" >> "$out/progress.log"
  echo "pass $i done" >> "$out/progress.log"
done
python3 scripts/summarize_pmc.py "$out" "$tag" > "$out/${tag}_pmc_summary.txt"
rm -rf "$out"/pmc*/
# the same passes in reference-order mode (lk_set_reference_order(1): the 16-lane reference-order instance)
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i + 1))
  LK_REF_ORDER=1 timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d "$out/pmc$i" -o pmc -- python3 scripts/quick_solve.py C2 5 > "$out/rpmc$i.log" 2>&1 || echo "reference-order pass $i ($set) failed" >> "$out/progress.log"
  echo "reference-order pass $i done" >> "$out/progress.log"
done
python3 scripts/summarize_pmc.py "$out" "$tag (LK_REF_ORDER=1)" > "$out/${tag}_reforder_pmc.txt"
python3 - "$out/${tag}_pmc_summary.txt" "$out/${tag}_reforder_pmc.txt" "$tag" > "$out/${tag}_traffic.json" <<'PY'
import json, re, sys


def means(path, kernel):
    out = {}
    for line in open(path):
        m = re.match(re.escape(kernel) + r"\s+(\w+)\s+dispatches=\d+ mean=([0-9.e+]+)", line)
        if m:
            out[m.group(1)] = float(m.group(2))
    return out


d = means(sys.argv[1], "lk_solve_kernel<3, 2, 32, 64, false, false, false>")
r = means(sys.argv[2], "lk_solve_kernel<3, 2, 16, 64, true, true, false>")
tag = sys.argv[3]
keep = {}
try:   # (what scripts/profile_sequence.sh put into the committed file - the windows' constants and clocks - stays)
    keep = json.load(open(f"profiles/{tag}_traffic.json"))
except Exception:
    pass
keep.update({
    "solve_kernel_hbm_bytes_per_launch_C2": (d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0,
    "solve_kernel_valu_insts_per_launch_C2": d["SQ_INSTS_VALU"],
    "reference_order_hbm_bytes_per_launch_C2": (r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024.0,
    "reference_order_valu_insts_per_launch_C2": r["SQ_INSTS_VALU"],
    "source": f"profiles/{tag}_pmc_summary.txt (lk_solve_kernel<3,2,32,64,false,false>) and profiles/{tag}_reforder_pmc.txt "
              "(lk_solve_kernel<3,2,16,64,true,true>): FETCH_SIZE + WRITE_SIZE in KiB per dispatch (these byte / dword loads read 1:1 on "
              "the counter, calibrated on the pyramid kernel in profiles/r01_pmc_traffic.txt), SQ_INSTS_VALU summed over the device"})
print(json.dumps(keep, indent=1))
PY
timeout -k 10 200 rocprofv3 --kernel-trace -d "$out/c4" -o t -- python3 scripts/quick_solve.py C4 10 > "$out/c4.log" 2>&1
{ grep solve_ms "$out/c4.log"; python3 scripts/c4_chain_timeline.py "$out/c4/t_results.db" 8; } > "$out/${tag}_c4_chain.txt" 2>&1
if [ -f build/tune/liblk_trace.so ]; then
  LK_ENGINE_LIB=$PWD/build/tune/liblk_trace.so timeout -k 10 120 python3 scripts/trace_solve.py "$out/trace_c2.npz" > /dev/null 2>&1 &&
    { python3 scripts/trace_brief.py "$out/trace_c2.npz"; python3 scripts/trace_report.py "$out/trace_c2.npz"; } > "$out/${tag}_wave_timeline.txt" 2>&1
fi
if [ -f build/tune/liblk_trace_ord.so ]; then   # scripts/tune_build.sh trace_ord -DLK_TRACE '-DLK_TRACE_PICK(G,S)=((S)&&(G)==16)'
  LK_REF_ORDER=1 LK_ENGINE_LIB=$PWD/build/tune/liblk_trace_ord.so timeout -k 10 120 python3 scripts/trace_solve.py "$out/trace_c2_ord.npz" > /dev/null 2>&1 &&
    { python3 scripts/trace_brief.py "$out/trace_c2_ord.npz"; python3 scripts/trace_report.py "$out/trace_c2_ord.npz"; python3 scripts/trace_top.py "$out/trace_c2_ord.npz" 8; } > "$out/${tag}_reforder_wave_timeline.txt" 2>&1
fi
if [ -f build/tune/liblk_trace_ord_fine.so ]; then   # ... -DLK_TRACE_FINE: cycles per step split into fetch / evaluation / solve / rest
  LK_REF_ORDER=1 LK_ENGINE_LIB=$PWD/build/tune/liblk_trace_ord_fine.so timeout -k 10 120 python3 scripts/trace_solve.py "$out/trace_c2_ord_fine.npz" > /dev/null 2>&1 &&
    { python3 scripts/trace_fine.py "$out/trace_c2_ord_fine.npz"; python3 scripts/trace_top.py "$out/trace_c2_ord_fine.npz" 8; } >> "$out/${tag}_reforder_wave_timeline.txt" 2>&1
fi
python3 - "$out/${tag}_traffic.json" "$out/${tag}_wave_timeline.txt" <<'PY'
import json, os, re, sys
d = json.load(open(sys.argv[1]))
if os.path.exists(sys.argv[2]):
    c = re.search(r"clock ([0-9.]+) GHz", open(sys.argv[2]).read())
    if c:
        d.setdefault("measured_clock_GHz", {})["one_pair_C2"] = float(c.group(1))
json.dump(d, open(sys.argv[1], "w"), indent=1)
PY
# the one-rank rehearsal of the multi-process bench (RCCL in the loop, every block of the N > 1 line)
LK_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 400 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-other-configs > "$out/${tag}_bench_dist_rehearsal.json" 2> "$out/bench_dist.err"
rm -rf "$out/c4" "$out"/pmc*/ "$out/trace" "$out"/*.npz
scripts/profile_c3.sh "$tag" > /dev/null 2>&1   # config 3: <tag>_c3_steps.txt
# VALU instructions of one solve of configs 3, 4 and 5 (all launches of the chain) -> <tag>_traffic.json (bench.py: other_configs.*.valu_issue_frac)
declare -A vps
for wl in C4 C5; do   # (quick_solve.py WL 3: 1 + 3 solves)
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d "$out/vps_$wl" -o pmc -- python3 scripts/quick_solve.py $wl 3 > "$out/vps_$wl.log" 2>&1
  vps[$wl]=$(python3 scripts/valu_per_solve.py "$out/vps_$wl" 4 2>> "$out/vps_kernels.txt")
done
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d "$out/vps_C3" -o pmc -- python3 scripts/quick_c3.py 3 > "$out/vps_C3.log" 2>&1   # (1 + 3 solves)
vps[C3]=$(python3 scripts/valu_per_solve.py "$out/vps_C3" 4 2>> "$out/vps_kernels.txt")
python3 - "$out/${tag}_traffic.json" "${vps[C3]}" "${vps[C4]}" "${vps[C5]}" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
d["valu_insts_per_solve"] = {"C3": float(sys.argv[2]), "C4": float(sys.argv[3]), "C5": float(sys.argv[4]),
                             "source": "rocprofv3 --pmc SQ_INSTS_VALU over scripts/quick_c3.py / quick_solve.py C4 / C5: all lk_solve_kernel launches of one solve"}
json.dump(d, open(sys.argv[1], "w"), indent=1)
PY
rm -rf "$out"/vps_C*/
tail -1 "$out/${tag}_bench.json" | cut -c1-400
cat "$out/${tag}_pmc_summary.txt"
