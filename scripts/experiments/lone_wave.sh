#!/bin/bash
# cycles per step of wavefronts that are alone on their SIMDs (LK_GRID=32: 1024 sectors, 512 wavefronts) against the full C2 grid
mkdir -p gpurun_out/lone
for g in 32 0; do
  echo "== LK_GRID=$g"
  LK_GRID=$g LK_ENGINE_LIB=$PWD/build/tune/liblk_trace_fine.so timeout -k 10 200 python3 scripts/trace_solve.py gpurun_out/lone/trace_$g.npz 2>&1 | tail -1
  python3 scripts/trace_fine.py gpurun_out/lone/trace_$g.npz
  python3 scripts/trace_report.py gpurun_out/lone/trace_$g.npz 2>&1 | sed -n 1,4p
done
