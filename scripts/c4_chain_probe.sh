#!/bin/bash
# Start/end of the launches of the config-4 solve chain for several LK_EVAL_CAP values (rocprofv3 kernel trace).
# usage (on the GPU box): scripts/c4_chain_probe.sh OUTDIR [LIB]
out=${1:-gpurun_out/c4chain}; lib=${2:-}
mkdir -p "$out"; export TMPDIR=/tmp
[ -n "$lib" ] && export LK_ENGINE_LIB=$lib
for cap in ${CAPS:-8 16 32 64}; do
  export LK_EVAL_CAP=$cap
  rocprofv3 --kernel-trace -d "$out/cap$cap" -o t -- python3 scripts/quick_solve.py C4 10 > "$out/cap$cap.log" 2>&1 || exit 1
  echo "== LK_EVAL_CAP=$cap"; grep solve_ms "$out/cap$cap.log"
  python3 scripts/c4_chain_timeline.py "$out/cap$cap/t_results.db" 4
done
