#!/usr/bin/env python3
"""Step-granular model of a C2 solve launch (32-lane groups, two sectors per wavefront) on 1024 SIMDs x 4 wavefront
slots, fed with per-sector evaluation traces (/tmp/traces_C2.npz: the pyramid level of every evaluation of every
sector, written by tests/tools/sector_traces.py).  Calibration (profiles/r02_wave_timeline.txt): a wavefront needs
~9 cycles per instruction whether it shares its SIMD or not (4 x 1/9 saturates the SIMD's 1 / 2.2), a step is
~700 instructions + 390 per trip.  Used to estimate scheduling variants before building them:
    python scripts/sim/launch_sim.py [park_at] [persistent] [consumer_lanes]"""
import heapq
import sys

import numpy as np

z = np.load("/tmp/traces_C2.npz")
levels, n_ev = z["levels"], z["n_ev"]
S = len(n_ev)
N_LEVEL = {0: 361, 1: 90, 2: 25}
CYC_PER_INSTR, CLOCK = 9.0, 2.07e9
park_at = int(sys.argv[1]) if len(sys.argv) > 1 else 0          # evaluations after which a sector is parked (0: never)
persistent = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cons_lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 64
SLOTS = 4096


def step_cost(ns, lanes):
    trips = max(-(-n // lanes) for n in ns)
    return (700 + 390 * trips) * CYC_PER_INSTR / CLOCK * 1e6     # microseconds


def pair_time(a, b, parked):
    """two sectors of a wavefront, level-aligned (the one ahead waits at a level change), solo once one of them is done
    for good; a sector that reaches park_at evaluations leaves (-> `parked` with its progress)"""
    na, nb = int(n_ev[a]), (int(n_ev[b]) if b >= 0 else 0)
    la = park_at if park_at and na > park_at else na
    lb = park_at if park_at and nb > park_at else nb
    ta, tb = [int(x) for x in levels[a, :la]], ([int(x) for x in levels[b, :lb]] if b >= 0 else [])
    t = 0.0
    for L in (2, 1, 0):
        ca, cb = ta.count(L), tb.count(L)
        last = L == 0 or (park_at and (la < na or lb < nb) and False)
        both, extra = min(ca, cb), abs(ca - cb)
        t += both * step_cost([N_LEVEL[L]], 32)
        # the rest of the level: one sector alone; its partner is done for good only at the last level it has
        partner_done = (L == min(ta) if ca < cb else L == min(tb)) if (ta and tb) else True
        t += extra * step_cost([N_LEVEL[L]], 64 if partner_done else 32)
    if la < na:
        parked.append((a, la))
    if b >= 0 and lb < nb:
        parked.append((b, lb))
    return t


# workgroup b solves the sector pair (b & 7) * chunk + (b >> 3) (each XCD a contiguous run of sectors, lk_solve_kernel)
n_wg = (S + 1) // 2
chunk = (n_wg + 7) // 8
order = [(b & 7) * chunk + (b >> 3) for b in range(8 * chunk)]
pairs = [(2 * i, 2 * i + 1 if 2 * i + 1 < S else -1) for i in order if 2 * i < S]
free = [(0.0, i) for i in range(SLOTS)]      # (time the slot frees, slot)
heapq.heapify(free)
events = []                                  # consumer work: (ready time, sector, from eval)
end = 0.0
parked_all = []
if not persistent:
    for a, b in pairs:
        t0, slot = heapq.heappop(free)
        parked = []
        d = pair_time(a, b, parked)
        heapq.heappush(free, (t0 + d, slot))
        end = max(end, t0 + d)
        parked_all += [(t0 + d, s, k) for s, k in parked]
else:
    for a, b in pairs:                       # a persistent wave takes the next pair when both sectors are done: same model
        t0, slot = heapq.heappop(free)
        parked = []
        d = pair_time(a, b, parked)
        heapq.heappush(free, (t0 + d, slot))
        end = max(end, t0 + d)
        parked_all += [(t0 + d, s, k) for s, k in parked]
main_end = end
# consumer: one parked sector per wavefront, cons_lanes lanes, takes a free slot as soon as one exists after the park
cons_end = 0.0
for ready, s, k in sorted(parked_all):
    t0, slot = heapq.heappop(free)
    t0 = max(t0, ready)
    d = sum(step_cost([N_LEVEL[int(levels[s, i])]], cons_lanes) for i in range(k, n_ev[s]))
    heapq.heappush(free, (t0 + d, slot))
    cons_end = max(cons_end, t0 + d)
print(f"park_at {park_at} persistent {persistent} consumer lanes {cons_lanes}: main launch ends {main_end:.1f} us, "
      f"{len(parked_all)} sectors parked, consumer ends {cons_end:.1f} us -> {max(main_end, cons_end):.1f} us")
