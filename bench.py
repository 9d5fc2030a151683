#!/usr/bin/env python3
"""Benchmark of the hot path (BASELINE.json metric): correlation-point-iterations/s and
achieved algorithmic HBM GB/s on the 2048^2 speckle pair of config C2.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one image pair whose level-0 pixels are already
resident in HBM: build both image pyramids (und, def), solve every sector of the ROI grid
coarse-to-fine on the device, leave the 48-byte result records in HBM.
`value` / `ms_per_step` are measured ONE PAIR AT A TIME (--inflight 1, the default): the rate a
tracked sequence gets, and the run whose dominant kernel `roofline` describes - same engine, same
template instance, same launches.  Independent pairs can overlap (--inflight P: step k on engine
k % P, each engine on its own HIP stream, the straggler tail of one solve filled by the next
pair's); that throughput is reported in the `overlapped` block at N = 1, never as `value`.
N > 1 ("weak" scaling): one process per GPU; rank 0's deformed frames are broadcast over
RCCL/xGMI a round of --round steps at a time (double-buffered: round m+1 travels while round m
is solved, the way the reference prefetches the next frame, manager_class.cpp:1438-1447), each
rank correlates its own 10 000-sector grid (the C2 grid shifted by `rank` pixels - a denser
measurement grid on the same pair) and the records of a round are all-gathered in one collective
(asynchronously, double-buffered).  value = point-iterations of ALL ranks / max-over-ranks time.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
N_SIMDS, PEAK_CLOCK_HZ, VALU_ISSUE_CYCLES = 1024, 2.4e9, 2.0   # 256 CUs x 4 SIMDs; one wave64 f32 VALU instruction per 2 cycles per SIMD


def profile_constants():
    """Per-launch constants of the C2 solve kernels from the committed rocprofv3 --pmc passes (profiles/): HBM bytes
    (FETCH_SIZE + WRITE_SIZE) and VALU wavefront-instructions (SQ_INSTS_VALU).  NOT counters read in this run."""
    for name in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            try:
                with open(path) as f:
                    d = json.load(f)
                d["file"] = f"profiles/{name}"
                return d
            except Exception:
                pass
    return {}


def valu_issue_frac(valu_insts, kernel_ms, clock_hz=PEAK_CLOCK_HZ):
    """Share of the chip's VALU issue slots the launch used: instructions x 2 cycles / (1024 SIMDs x clock x time).
    The 2 cycles per wave64 f32 instruction are an assumption the frame-pipelined windows bound from above: config 2's
    window sustains 0.65 of that rate over 10 ms (profiles/r04_seq_pmc.txt), so an instruction costs a SIMD < 3.1 cycles."""
    if not valu_insts or not kernel_ms:
        return None
    return valu_insts * VALU_ISSUE_CYCLES / (N_SIMDS * clock_hz * kernel_ms * 1e-3)


def measured_clock_hz(key):
    """shader clock the traced wavefronts of that launch actually ran at (profiles/*_wave_timeline.txt), or None"""
    g = profile_constants().get("measured_clock_GHz", {}).get(key)
    return g * 1e9 if g else None


def cpu_baseline(wl, und, dfm, budget_sectors):
    """The CPU oracle (a port of the reference's CPU engine, oracle/lk_oracle.c) timed on
    this box's host cores on a bounded sample of the same workload."""
    from oracle import lk_oracle as lo  # checker/baseline only - never the measured GPU path
    o = lo.Oracle(interp=lo.IM_BICUBIC, model=wl.model, py_stop=wl.py_stop)
    o.set_image(0, und)
    o.set_image(1, dfm)
    xdim, ydim, cen = lo.rect_sector_geometry(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
    S = len(cen)
    pick = np.linspace(0, S - 1, min(budget_sectors, S)).astype(np.int64)
    n = (2 * xdim + 1) * (2 * ydim + 1)
    cat = np.concatenate([lo.rect_points(cx - xdim, cy - ydim, cx + xdim, cy + ydim) for cx, cy in cen[pick]])
    off = np.arange(len(pick), dtype=np.int64) * n
    cnt = np.full(len(pick), n, np.int32)
    centers = cen[pick].astype(np.float32)
    cores = os.cpu_count() or 1
    out = {}
    for label, threads in (("1_thread", 1), ("all_cores", cores)):
        t0 = time.perf_counter()
        res = o.correlate_packed(cat, off, cnt, centers=centers, nthreads=threads)
        dt = time.perf_counter() - t0
        out[label] = (len(pick) / dt, dt, threads, res)
    # the reference's own yardstick: the same sectors with its default split of a sector's samples over
    # NUMBER_OF_THREADS = 20 chunks (defines.hpp:10, correlation_class.cpp:169-189, :253-275) - another summation order
    o20 = lo.Oracle(interp=lo.IM_BICUBIC, model=wl.model, py_stop=wl.py_stop, n_threads=20)
    o20.set_image(0, und)
    o20.set_image(1, dfm)
    out["split_20"] = o20.correlate_packed(cat, off, cnt, centers=centers, nthreads=cores)
    return out, len(pick)


def other_configs(ca):
    """BASELINE configs 3, 4 (one pair of its sector geometry) and 5 on this GPU: solve time of
    one launch sequence, counters, and the distance to the synthetic ground truth.  Extra
    evidence next to the headline line - never part of `value`."""
    from correlation_amd.workload import C4, C4B, C5
    out = {}

    valu_per_solve = profile_constants().get("valu_insts_per_solve", {})   # (profile constants: profiles/*_traffic.json)

    def timed(e, n=3, config=None):
        g = np.zeros(6, np.float32)
        r = e.correlate_all(g)
        ms = []
        for _ in range(n):
            e.correlate_all(g)
            ms.append(e.stats()["solve_ms"])
        st = e.stats()
        ms = float(np.median(ms))
        m = {"solve_ms": ms, "sectors": int(len(r)), "point_iterations_per_s": st["point_iterations"] / (ms * 1e-3),
             "algorithmic_GBps": st["algorithmic_bytes"] / (ms * 1e-3) / 1e9,
             "evaluations_per_sector": st["evaluations"] / st["sectors"],
             "error_free_fraction": float((r["error_code"] == 0).mean())}
        if config in valu_per_solve:   # share of the chip's VALU issue slots over the whole launch chain of the solve
            m["valu_issue_frac"] = valu_issue_frac(valu_per_solve[config], ms)
        return r, m

    def parity_block(wl, und, dfm, r):
        """default mode against the CPU oracle (1 thread order) on every sector of a starved config: error codes,
        NaN set, iteration counts, |dp01| (correlation_class.cpp:441-499, :552-587 on starved levels)"""
        from oracle import lk_oracle as lo   # the checker - never the measured path
        o = lo.Oracle(interp=lo.IM_BICUBIC, model=wl.model, py_stop=wl.py_stop)
        o.set_image(0, np.asarray(und))
        o.set_image(1, np.asarray(dfm))
        xd, yd, cen = lo.rect_sector_geometry(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
        n = (2 * xd + 1) * (2 * yd + 1)
        cat = np.concatenate([lo.rect_points(cx - xd, cy - yd, cx + xd, cy + yd) for cx, cy in cen])
        t0 = time.perf_counter()
        w = o.correlate_packed(cat, np.arange(len(cen), dtype=np.int64) * n, np.full(len(cen), n, np.int32),
                               centers=cen.astype(np.float32), nthreads=os.cpu_count() or 1)
        dt = time.perf_counter() - t0
        nan_g, nan_w = np.isnan(r["p"]).any(1), np.isnan(w["p"]).any(1)
        both = ~nan_g & ~nan_w & (r["error_code"] == 0) & (w["error_code"] == 0)
        d = np.abs(r["p"][both][:, :2] - w["p"][both][:, :2]).max(1)
        return {"sectors": int(len(w)), "oracle_seconds": dt,
                "error_codes_differ": int((r["error_code"] != w["error_code"]).sum()),
                "oracle_error_free": int((w["error_code"] == 0).sum()), "engine_error_free": int((r["error_code"] == 0).sum()),
                "nan_records_engine": int(nan_g.sum()), "nan_records_oracle": int(nan_w.sum()),
                "nan_set_differs": int((nan_g != nan_w).sum()),
                "iterations_equal_fraction": float((r["iterations"] == w["iterations"]).mean()),
                "abs_dp01_p50": float(np.percentile(d, 50)), "abs_dp01_p99": float(np.percentile(d, 99)),
                "abs_dp01_max": float(d.max()),
                "note": "default mode vs oracle(T=1); starved levels are chaotic in the reference itself "
                        "(tests/test_parity_gpu.py: the same comparison with bounds on 3000-sector subsets)"}

    def rect(wl, truth, seed, parity=False):
        und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=truth, seed=seed, device="cuda")
        e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop)
        e.set_undeformed_image(und)
        e.set_deformed_image(dfm)
        e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
        e.commit_sectors()
        r, m = timed(e, config=wl.name.split(":")[0])
        c = wl.size / 2.0
        ok = r["error_code"] == 0
        u_true = truth[0] + truth[2] * (r["und_cx"] - c) + truth[3] * (r["und_cy"] - c)
        m["median_abs_u_minus_truth"] = float(np.nanmedian(np.abs(r["p"][:, 0] - u_true)[ok]))
        m["nan_records"] = int(np.isnan(r["p"]).any(1).sum())   # singular starved levels: the reference's QR returns NaN too
        m["workload"] = wl.name
        e.set_reference_order(1)
        _, mr = timed(e)
        m["reference_order_mode"] = {"solve_ms": mr["solve_ms"], "frac_of_hbm_peak": mr["algorithmic_GBps"] / HBM_PEAK_GBS}
        e.close()
        if parity:
            try:
                m["parity_vs_cpu"] = parity_block(wl, und, dfm, r)
            except Exception as ex:
                m["parity_vs_cpu"] = {"error": repr(ex)}
        return m

    try:  # the opt-in separable evaluation of the same bicubic surface (include/lk_engine.h)
        from correlation_amd.workload import C2, C4, C4B, C5, shard_range
        und, dfm = ca.speckle.speckle_pair(C2.size, C2.size, p=C2.truth, seed=7)
        res = {}
        for label, interp in (("reference_bicubic", ca.IM_BICUBIC), ("separable", ca.IM_BICUBIC_SEPARABLE)):
            e = ca.HipCorrelationEngine(fitting_model=C2.model, py_stop=C2.py_stop, interpolation=interp)
            e.set_undeformed_image(und)
            e.set_deformed_image(dfm)
            e.set_rect_grid(C2.x_begin, C2.x_begin, C2.x_end, C2.x_end, C2.hs, C2.vs)
            e.commit_sectors()
            r, m = timed(e, 10)
            res[label] = (r, m)
            e.close()
        a, b = res["separable"][0], res["reference_bicubic"][0]
        m = res["separable"][1]
        m["solve_ms_reference_bicubic"] = res["reference_bicubic"][1]["solve_ms"]
        m["max_abs_dp01_vs_reference_bicubic"] = float(np.abs(a["p"] - b["p"])[:, :2].max())
        m["max_rel_dchi_vs_reference_bicubic"] = float((np.abs(a["chi"] - b["chi"]) / b["chi"]).max())
        m["workload"] = "C2 with LK_IM_BICUBIC_SEPARABLE (extension, not the headline)"
        out["C2_separable_bicubic"] = m
    except Exception as ex:
        out["C2_separable_bicubic"] = {"error": repr(ex)}
    try:
        out["C4_one_pair"] = rect(C4, C4.truth, 7, parity=True)
    except Exception as ex:  # extra evidence must never take the headline line down
        out["C4_one_pair"] = {"error": repr(ex)}
    try:
        out["C4B_one_pair"] = rect(C4B, C4B.truth, 7)
    except Exception as ex:
        out["C4B_one_pair"] = {"error": repr(ex)}
    try:
        out["C5"] = rect(C5, (1.3, -0.7, 0.0005, 0.0, 0.0, -0.00025), 13)
    except Exception as ex:
        out["C5"] = {"error": repr(ex)}
    try:
        truth = (1.1, 0.6, 0.0008, 0.0004, -0.0004, 0.0012)
        und, dfm = ca.speckle.speckle_pair(4096, 4096, p=truth, seed=11, device="cuda")
        e = ca.HipCorrelationEngine(fitting_model=ca.FM_UVUXUYVXVY)
        e.set_undeformed_image(und)
        e.set_deformed_image(dfm)
        rs, as_, ri, ro = 8, 32, 600.0, 1800.0
        dr, da = np.float32((ro - ri) / rs), np.float32(2 * np.pi) / np.float32(as_)
        e.set_sectors_annular(0, np.float32([[np.float32(ri + i * dr), dr, np.float32(j) * da, da, 2048.0, 2048.0]
                                             for i in range(rs) for j in range(as_)]), as_)
        s = rs * as_
        t = 2 * np.pi * np.arange(64) / 64
        rad = np.where(np.arange(64) % 2 == 0, 1500.0, 900.0)
        e.resetPolygon_blob(s, np.stack([2048 + rad * np.cos(t), 2048 + rad * np.sin(t)], 1).astype(np.float32))
        e.commit_sectors()
        r, m = timed(e, config="C3")
        u_true = truth[0] + truth[2] * (r["und_cx"] - 2048) + truth[3] * (r["und_cy"] - 2048)
        m["max_abs_u_minus_truth"] = float(np.nanmax(np.abs(r["p"][:, 0] - u_true)))
        m["nan_records"] = int(np.isnan(r["p"]).any(1).sum())
        m["largest_sector_samples"] = int(r["n_points"].max())
        m["workload"] = "C3: 4096x4096, 8x32 annular sectors + one 64-vertex blob, affine, pyramid 0/1/2"
        e.set_reference_order(1)
        _, mr = timed(e, n=2)
        e.set_reference_order(20)
        _, mr20 = timed(e, n=2)
        m["reference_order_mode"] = {"solve_ms": mr["solve_ms"], "frac_of_hbm_peak": mr["algorithmic_GBps"] / HBM_PEAK_GBS,
                                     "solve_ms_number_of_threads_20": mr20["solve_ms"],
                                     "note": "a 512-thread workgroup per big sector (seven wavefronts form the products, one adds them in "
                                             "sample order); with number_of_threads = 1 the blob's 4.2 M samples are ONE chain of dependent "
                                             "additions per sum and evaluation - that chain is this time; with the reference's default 20 "
                                             "threads a team of 20 workgroups solves the 20 thread chunks side by side"}
        out["C3"] = m
        e.close()
    except Exception as ex:
        out["C3"] = {"error": repr(ex)}
    return out


def sequence_frames(ca, torch, dev, size, n_frames):
    """the 64-pair synthetic sequence of BASELINE config 4 (constant velocity (0.8, -0.4) px / frame + 1e-4 / frame dilation),
    rendered on the GPU, resident in HBM as one [n_frames + 1, H, W] u8 tensor"""
    frames = ca.speckle.speckle_sequence(size, size, n_frames + 1, velocity=(0.8, -0.4), dilation=1e-4, seed=7, device=str(dev))
    return torch.from_numpy(np.stack(frames)).to(dev)


def sequence_block(ca, torch, dev, wl, d_frames, mode, repeats=3, check_loop=True):
    """A tracked SEQUENCE of `wl`'s sector grid (BASELINE config 4 is one; Eulerian description, first image as the
    reference, constant-velocity guesses, manager_class.cpp:1380-1496, :2677-2699), frames resident in HBM.
    `window`: all pairs in ONE frame-pipelined window (lk_correlate_sequence_async: every sector advances through the frames
    on its own).  `one_pair_at_a_time`: the same sequence as a loop of guess kernel + pyramid + solve launches.
    ms_per_pair of both includes the pyramid of the pair's new frame and the guesses; kernel_ms_per_pair is the window's
    solve alone (HIP events on the engine's stream)."""
    n = int(d_frames.shape[0]) - 1
    size = int(d_frames.shape[1])
    c = (size / 2 - 0.5, size / 2 - 0.5)
    zero = np.zeros(6, np.float32)

    def engine():
        e = ca.HipCorrelationEngine(fitting_model=wl.model, py_stop=wl.py_stop, device=dev.index or 0)
        if mode == "reference_order":
            e.set_reference_order(1)
        elif mode == "batch_invariant":
            e.set_batch_invariant(True)
        e.set_image_device(ca.IMG_UND, d_frames[0].data_ptr(), size, size)
        e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
        e.commit_sectors()
        return e

    e = engine()
    e.sequence_reserve(n)
    wall, kern = [], []
    for r in range(repeats + 1):
        e.synchronize()
        t0 = time.perf_counter()
        for i in range(n):                                       # upload copy + pyramid of every new frame (next-frame stream)
            e.sequence_set_frame_device(i, d_frames[i + 1].data_ptr(), size, size)
        e.adjust_initial_guess(0, True, zero, c)
        e.correlate_sequence_async(n, constant_velocity=True, host_records=False)
        e.wait_sequence(False)
        e.synchronize()
        t1 = time.perf_counter()
        if r:
            wall.append((t1 - t0) * 1e3 / n)
            kern.append(e.stats()["solve_ms"] / n)
    st = e.stats()
    pipelined = e.sequence_is_pipelined
    e.adjust_initial_guess(0, True, zero, c)
    rec = e.correlate_sequence(n, constant_velocity=True)
    ms, kms = float(np.median(wall)), float(np.median(kern))
    out = {"workload": wl.name, "mode": mode, "pairs": n, "sectors": int(rec.shape[1]), "frames_resident": n,
           "pipelined_instances": bool(pipelined),
           "window": {"ms_per_pair": ms, "kernel_ms_per_pair": kms,
                      "point_iterations_per_s": st["point_iterations"] / n / (ms * 1e-3),
                      "algorithmic_GBps": st["algorithmic_bytes"] / n / (kms * 1e-3) / 1e9,
                      "frac": st["algorithmic_bytes"] / n / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "algorithmic_bytes_per_pair": st["algorithmic_bytes"] / n,
                      "evaluations_per_sector_and_pair": st["evaluations"] / st["sectors"],
                      "error_free_fraction_first_pair": float((rec[0]["error_code"] == 0).mean()),
                      "error_free_fraction_last_pair": float((rec[-1]["error_code"] == 0).mean())}}
    key = f"{wl.name.split(':')[0]}_{mode}"
    vk = profile_constants().get("valu_insts_per_window_pair", {}).get(key)
    if vk:
        out["window"]["valu_issue_frac"] = valu_issue_frac(vk, kms)
        clk = measured_clock_hz(f"window_{key}")
        if clk:
            out["window"]["valu_issue_frac_at_measured_clock"] = valu_issue_frac(vk, kms, clk)
            out["window"]["measured_clock_GHz"] = clk / 1e9
        out["window"]["traffic_hbm_bytes_per_pair"] = profile_constants().get("hbm_bytes_per_window_pair", {}).get(key)
    if check_loop:
        a = engine()
        a.set_timing(False)
        same, t_loop = 0, 0.0
        for r in range(2):
            a.synchronize()
            t0 = time.perf_counter()
            got = []
            for k in range(n):
                a.set_image_device(ca.IMG_DEF, d_frames[k + 1].data_ptr(), size, size)
                a.adjust_initial_guess(k, True, zero, c)
                if r:   # (second pass: the records, for the comparison - outside the timing)
                    a.correlate_all_async()
                    got.append(a.wait_results())
                else:
                    a.correlate_all_device(0, 0)
            a.synchronize()
            if not r:
                t_loop = (time.perf_counter() - t0) * 1e3 / n
        same = sum(int(g.tobytes() == rec[k].tobytes()) for k, g in enumerate(got))
        out["one_pair_at_a_time"] = {"ms_per_pair": t_loop, "frames_with_identical_records": same, "of": n,
                                     "note": "records of the window against the loop's: identical bytes are required in "
                                             "reference-order and batch-invariant mode (tests/test_sequence_window_gpu.py); the default "
                                             "mode's loop widens lane groups by batch composition, its window uses fixed groups"}
        out["window"]["speedup_vs_one_pair_at_a_time"] = t_loop / ms
        a.close()
    e.close()
    return out


def sharded_config(ca, torch, dist, wl, rank, world, local_rank, steps, warmup):
    """One pair of `wl` with its sector grid split over the ranks in contiguous blocks (the code path of
    --workload): per step the deformed frame is broadcast from rank 0 (RCCL over xGMI), every rank builds both
    pyramids and solves its shard, the 48-byte records are all-gathered.  Also times the WHOLE grid on one GPU
    (every rank on its own device, no collectives) so that the ratio is in the same line.  Every rank must call
    this; local failures are agreed on before any collective is entered."""
    from correlation_amd.workload import shard_range
    use_dist = dist is not None
    dev = torch.device("cuda", local_rank)
    info = {"workload": wl.name, "n_ranks": dist.get_world_size() if use_dist else 1, "sectors_total": wl.hs * wl.vs}
    ok, err = 1, None
    full = shard = None
    try:
        truth = wl.truth if wl.size <= 2048 else (1.3, -0.7, 0.0005, 0.0, 0.0, -0.00025)
        und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=truth, seed=7 if wl.size <= 2048 else 13,
                                           device=f"cuda:{local_rank}" if wl.size > 2048 else None)
        d_und = torch.from_numpy(und).to(dev)
        d_def = torch.from_numpy(dfm).to(dev)
        st_ = torch.cuda.Stream(dev)

        def engine(first, count):
            e = ca.HipCorrelationEngine(interpolation=ca.IM_BICUBIC, fitting_model=wl.model, py_stop=wl.py_stop, device=local_rank)
            e.set_stream(st_.cuda_stream)
            e.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs, first, count)
            e.commit_sectors()
            e.set_timing(False)
            return e
        full = engine(0, wl.hs * wl.vs)
        first, count = shard_range(wl.hs * wl.vs, rank, world)
        shard = engine(first, count) if world > 1 else full
        cap = (wl.hs * wl.vs + world - 1) // world
        d_guess = torch.zeros((wl.hs * wl.vs, 6), dtype=torch.float32, device=dev)
        d_rec_full = torch.zeros((wl.hs * wl.vs, 48), dtype=torch.uint8, device=dev)
        d_rec = torch.zeros((cap, 48), dtype=torch.uint8, device=dev)
        d_all = torch.empty((world, cap, 48), dtype=torch.uint8, device=dev)
    except Exception as ex:   # noqa: BLE001 - reported in the block, never fatal for the headline
        ok, err = 0, repr(ex)
    if use_dist:
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok_all = int(flag.item())
    else:
        ok_all = ok
    if not ok_all:
        for e in {id(x): x for x in (full, shard) if x is not None}.values():
            e.close()
        info["error"] = err or "set-up failed on another rank"
        return info
    torch.cuda.set_stream(st_)

    def timed(fn, n):
        for _ in range(max(1, warmup)):
            fn()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / n * 1e3

    def one_gpu():
        full.set_image_pair_device(d_und.data_ptr(), d_def.data_ptr(), wl.size, wl.size)
        full.correlate_all_device(d_guess.data_ptr(), d_rec_full.data_ptr())

    def sharded():
        if use_dist:
            dist.broadcast(d_def, src=0)
        shard.set_image_pair_device(d_und.data_ptr(), d_def.data_ptr(), wl.size, wl.size)
        shard.correlate_all_device(d_guess.data_ptr(), d_rec.data_ptr())
        if use_dist:
            dist.all_gather_into_tensor(d_all.view(-1, 48), d_rec)

    ok, err = 1, None
    ms1 = ms = None
    st1 = {"point_iterations": 0}
    try:
        ms1 = timed(one_gpu, steps)
        full.set_timing(True)
        one_gpu()
        torch.cuda.synchronize(dev)
        st1 = full.stats()
    except Exception as ex:   # noqa: BLE001
        ok, err = 0, repr(ex)
    if use_dist:   # (a rank that failed alone must not leave the others in the collectives of the sharded run)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok_all = int(flag.item())
    else:
        ok_all = ok
    if ok_all:
        try:
            ms = timed(sharded, steps)
        except Exception as ex:   # noqa: BLE001
            ok, err = 0, repr(ex)
        if use_dist:
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok_all = int(flag.item())
        else:
            ok_all = ok
    if not ok_all:
        for e in {id(x): x for x in (full, shard)}.values():
            e.close()
        info["error"] = err or "the timed region failed on another rank"
        return info
    info.update({"ms_per_step": ms, "ms_per_step_1gpu": ms1, "speedup_vs_1gpu": ms1 / ms,
                 "point_iterations_per_s": st1["point_iterations"] / (ms * 1e-3),
                 "point_iterations_per_s_1gpu": st1["point_iterations"] / (ms1 * 1e-3),
                 "sectors_per_rank": int(shard.n_sectors), "scaling": "strong",
                 "step": "broadcast of the deformed frame + both pyramids + solve of the rank's block + all-gather of records"
                         if use_dist else "both pyramids + solve of the whole grid (one rank: no collectives)"})
    for e in {id(x): x for x in (full, shard)}.values():
        e.close()
    return info


def sharded_sequence(ca, torch, dist, wl, d_frames, rank, world, local_rank, K=32):
    """A tracked sequence with the sector grid split over the ranks in contiguous blocks (BASELINE config 4's
    shape) - correlation_amd/distributed.py: ShardedWindowSequence (covered by a 2-rank gloo test on the CPU): every rank keeps
    the undeformed pyramid (built ONCE) and its block's guess history; per window of K pairs ONE broadcast of the K new frames
    from rank 0 - issued before the window it overlaps is launched, into the other half of a double buffer - every rank solves
    the window for its block (frame-pipelined instances), ONE all-gather of K x block records (behind the next window).
    Beside it the whole grid on one GPU (every rank on its own device, no collectives).  Every rank must call this; failures
    are agreed on before and after the timed regions.  Windows of 32 pairs: a rank's block of the grid is solved in the latency regime
    (a window lasts as long as its slowest sector's chain), where longer windows average the slow frames of different sectors out -
    an eighth of config 4's grid: 8 / 16 / 32 / 64 pairs per window -> 0.60 / 0.55 / 0.50 / 0.46 ms per pair."""
    from correlation_amd.distributed import ShardedWindowSequence
    use_dist = dist is not None
    dev = torch.device("cuda", local_rank)
    n = int(d_frames.shape[0]) - 1
    size = int(d_frames.shape[1])
    K = min(K, n)
    c = (size / 2 - 0.5, size / 2 - 0.5)
    info = {"workload": wl.name, "n_ranks": dist.get_world_size() if use_dist else 1, "sectors_total": wl.hs * wl.vs, "pairs": n,
            "window_pairs": K, "scaling": "strong"}

    def agree(ok):
        if not use_dist:
            return ok
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    engines, shard, full, err = [], None, None, None
    try:
        def sequence(d):
            e = ca.HipCorrelationEngine(interpolation=ca.IM_BICUBIC, fitting_model=wl.model, py_stop=wl.py_stop, device=local_rank)
            engines.append(e)
            sq = ShardedWindowSequence(e, d, dev, window=K)
            sq.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
            return sq
        shard = sequence(dist if use_dist else None)
        full = sequence(None) if world > 1 else shard
    except Exception as ex:   # noqa: BLE001
        err = repr(ex)
    if not agree(err is None):
        for e in engines:
            e.close()
        info["error"] = err or "set-up failed on another rank"
        return info

    def run(sq, barrier):
        torch.cuda.synchronize(dev)
        if use_dist and barrier:
            dist.barrier()
        t0 = time.perf_counter()
        sq.run(d_frames, constant_velocity=True, center=c, fetch=False)
        if use_dist and barrier:
            dist.barrier()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) * 1e3

    ok, err = True, None
    ms_sharded = ms_one = None
    try:
        run(shard, True)                     # warm-up (allocations, first launches)
        ms_sharded = run(shard, True)
        if world > 1:
            run(full, False)
            ms_one = run(full, False)
        else:
            ms_one = ms_sharded
    except Exception as ex:   # noqa: BLE001
        ok, err = False, repr(ex)
    if not agree(ok):
        info["error"] = err or "the timed region failed on another rank"
    else:
        t = torch.tensor([ms_sharded, ms_one], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_sharded, ms_one = float(t[0].item()), float(t[1].item())
        info.update({"ms_per_pair": ms_sharded / n, "ms_per_pair_1gpu": ms_one / n, "speedup_vs_1gpu": ms_one / ms_sharded,
                     "sectors_per_rank": int(shard.count),
                     "step": "per window: one broadcast of its frames (behind the window before), pyramids, the frame-pipelined solve of "
                             "the rank's block, one all-gather of its records (behind the next window); the undeformed pyramid once per "
                             "sequence"})
    for e in engines:
        e.close()
    return info


def native_group_child(args, workload):
    """`bench.py --native` (include/lk_group.h: one process, ncclCommInitAll, one thread + stream per device) as a
    CHILD process with a time limit, so that nothing it does can take this process - or the headline line - down."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--native", "--gpus", str(args.gpus), "--workload", workload,
           "--steps", str(max(10, args.steps // 2)), "--warmup", str(max(3, args.warmup // 2))]
    # (the child is its own single-process job: nothing of this process's torch.distributed.run environment may reach it -
    # `--native` refuses WORLD_SIZE > 1, and a second process with the same RANK / MASTER_PORT has no business near the store)
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK",
                        "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "LK_BENCH_FORCE_DIST")
           and not k.startswith(("TORCHELASTIC_", "TORCH_NCCL_", "NCCL_ASYNC"))}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
        for ln in reversed(r.stdout.strip().splitlines()):
            if ln.startswith("{"):
                d = json.loads(ln)
                return {"value": d["value"], "ms_per_step": d["ms_per_step"], "n_ranks": d["config"].get("n_ranks"),
                        "scaling": d["scaling"], "workload": d["config"]["workload"], "parallelism": d["config"]["parallelism"],
                        "solve_ms_slowest_member": d["roofline"]["kernel_ms"], "error_free_fraction": d["per_pair"]["error_free_fraction"]}
        return {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
    except Exception as ex:   # noqa: BLE001
        return {"error": repr(ex)}


def native_group_bench(args, ca, wl):
    """`bench.py --native --gpus N` (ONE process): the same step through include/lk_group.h - frames enter on
    device 0, travel by ncclBroadcast, every device solves its contiguous block of the sector grid (strong
    scaling: one grid shared by all devices), records come back by ncclAllGather.  Same JSON contract."""
    import torch
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("--native is the single-process path: run it without torch.distributed.run")
    n = args.gpus
    und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=wl.truth if wl.size <= 2048 else (1.3, -0.7, 0.0005, 0.0, 0.0, -0.00025),
                                       seed=7 if wl.size <= 2048 else 13, device="cuda:0" if wl.size > 2048 else None)
    d_und, d_def = torch.from_numpy(und).to("cuda:0"), torch.from_numpy(dfm).to("cuda:0")
    torch.cuda.synchronize()
    g = ca.HipCorrelationGroup(n, fitting_model=wl.model, py_stop=wl.py_stop)
    g.for_each_engine("lk_set_timing", 0)
    g.set_image_device(ca.IMG_UND, d_und.data_ptr(), wl.size, wl.size)
    g.set_image_device(ca.IMG_DEF, d_def.data_ptr(), wl.size, wl.size)
    g.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
    g.commit_sectors()

    def step():
        g.set_image_device(ca.IMG_UND, d_und.data_ptr(), wl.size, wl.size)   # broadcast + pyramid on every device
        g.set_image_device(ca.IMG_DEF, d_def.data_ptr(), wl.size, wl.size)
        g.correlate_all(None, fetch=False)                                   # solve + all-gather, records stay in HBM

    for _ in range(args.warmup):
        step()
    g.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    g.synchronize()
    dt = time.perf_counter() - t0
    g.for_each_engine("lk_set_timing", 1)
    res = g.correlate_all(None)
    st = g.stats()
    S = g.n_sectors
    line = {"metric": "correlation-point-iterations/sec", "value": st["point_iterations"] * args.steps / dt,
            "unit": "point-iterations/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl.name, "sectors_total": S, "n_ranks": g.comm_ranks, "parallelism": f"lk_group: one process, {n} device(s), "
                       "one host thread + HIP stream per device, ncclBroadcast of frames, ncclAllGather of records",
                       "step": "broadcast + pyramid(und) + pyramid(def) on every device, sharded solve, record all-gather"},
            "roofline": {"bound": "hbm", "achieved": st["algorithmic_bytes"] / (st["solve_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS * n,
                         "unit": "GB/s", "frac": st["algorithmic_bytes"] / (st["solve_ms"] * 1e-3) / 1e9 / (HBM_PEAK_GBS * n),
                         "traffic": None, "kernel": "lk_solve_kernel, slowest member's engine-timed solve of one step",
                         "kernel_ms": st["solve_ms"], "algorithmic_bytes_per_launch": st["algorithmic_bytes"]},
            "per_pair": {"point_iterations": st["point_iterations"], "error_free_fraction": float((res["error_code"] == 0).mean())}}
    print(json.dumps(line))
    g.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sectors", type=int, default=10000)
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-sharded-configs", action="store_true", help="skip the C2-strong / C4 / C5 sharded blocks")
    ap.add_argument("--no-native-group", action="store_true", help="skip the lk_group (C-ABI, single process) block")
    ap.add_argument("--inflight", type=int, default=1,
                    help="image pairs in flight per GPU in the TIMED region (default 1: one pair at a time, the rate "
                         "of a tracked sequence and the run `roofline` describes).  P > 1: step k runs on engine "
                         "k %% P, each on its own HIP stream")
    ap.add_argument("--round", type=int, default=3,
                    help="N > 1: steps whose frames travel in one broadcast and whose records leave in one "
                         "all-gather (the host cost of a collective exceeds a step's GPU time)")
    ap.add_argument("--sequence-frames", type=int, default=64,
                    help="N = 1: pairs of the `sequence` blocks (configs 2 and 4 as tracked sequences; 0: skip them)")
    ap.add_argument("--overlap", type=int, default=3,
                    help="N = 1: pairs in flight of the extra `overlapped` block (0: skip it)")
    ap.add_argument("--native", action="store_true",
                    help="single process, no torch.distributed: the C-ABI group (include/lk_group.h) drives --gpus devices "
                         "with one host thread + stream each, ncclBroadcast of frames, ncclAllGather of records")
    ap.add_argument("--workload", default="C2", choices=["C2", "C4", "C4B", "C5"],
                    help="C2 (default, the headline: weak scaling, every rank its own 10k-sector grid); "
                         "C4 / C4B / C5: ONE pair of that config with its sector grid sharded over the ranks "
                         "(strong scaling; frame broadcast + record all-gather per round)")
    args = ap.parse_args()

    import torch
    import correlation_amd as ca
    from correlation_amd.workload import C2, C4, C4B, C5, shard_range

    if args.native:
        return native_group_bench(args, ca, {"C2": C2, "C4": C4, "C4B": C4B, "C5": C5}[args.workload])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP engine is the product, there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    use_dist = world > 1 or os.environ.get("LK_BENCH_FORCE_DIST") == "1"  # the latter: 1-rank rehearsal
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            os.environ.pop("NCCL_DEBUG")   # (RCCL's version banner goes to stdout, where the ONE JSON line is expected)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL

    wl = {"C2": C2, "C4": C4, "C4B": C4B, "C5": C5}[args.workload]
    strong = args.workload != "C2"
    if wl.size > 2048:   # 8192^2: rendered on the GPU (every rank renders; only rank 0's def frame is used)
        und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=(1.3, -0.7, 0.0005, 0.0, 0.0, -0.00025), seed=13,
                                           device=f"cuda:{local_rank}")
    else:
        und, dfm = ca.speckle.speckle_pair(wl.size, wl.size, p=wl.truth, seed=7)
    dev = torch.device("cuda", local_rank)
    d_und = torch.from_numpy(und).to(dev)
    d_def = torch.from_numpy(dfm).to(dev) if rank == 0 else torch.empty_like(d_und)

    P = max(1, args.inflight)                     # engines (pairs in flight) of the timed region
    R = max(1, args.round) if use_dist else P     # steps per round of collectives
    R = ((R + P - 1) // P) * P

    def make_engine(pairs_in_flight, reference_order=0):
        # a dedicated (non-null) HIP stream shared by torch, RCCL and the engine: torch events
        # recorded on it bracket exactly the engine's launches
        eng = ca.HipCorrelationEngine(interpolation=ca.IM_BICUBIC, fitting_model=wl.model, py_stop=wl.py_stop,
                                      device=local_rank)
        st_ = torch.cuda.Stream(dev)
        assert st_.cuda_stream != 0
        eng.set_stream(st_.cuda_stream)
        eng.set_pairs_in_flight(pairs_in_flight)
        eng.set_reference_order(reference_order)
        if strong:   # one grid, contiguous blocks of the sector index per rank
            first, count = shard_range(wl.hs * wl.vs, rank, world)
            eng.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs, first, count)
        else:        # weak scaling: rank r correlates the C2 grid shifted by r px (same sector size, distinct ROIs)
            eng.set_rect_grid(wl.x_begin + rank, wl.x_begin + rank, wl.x_end + rank - 8 * (world > 1),
                              wl.x_end + rank - 8 * (world > 1), wl.hs, wl.vs)
        eng.commit_sectors()
        eng.set_timing(False)   # the engine's own per-call HIP events stay out of the timed frames
        return eng, st_

    engines = [make_engine(P) for _ in range(P)]
    lanes = [engines[j % P] for j in range(R)]    # lane j of a round -> (engine, stream)
    e, stream = engines[0]
    S = e.n_sectors
    S_cap = (wl.hs * wl.vs + world - 1) // world if strong else S   # equal all-gather blocks
    n0 = e.sector_info(0)[0]
    d_guess = torch.zeros((S, 6), dtype=torch.float32, device=dev)
    # Steps are dealt to the lanes in rounds of R.  The R frames of round m+1 travel in ONE
    # broadcast while round m is being solved, and the R record blocks of round m leave in ONE
    # all-gather once its solves are done (the reference prefetches the next frame the same
    # way, manager_class.cpp:1438-1447): two collectives per round keep the host out of the way
    # (one broadcast and one gather per step cost ~0.2 ms of host time per step - more than a
    # step takes on the GPU).  Two sets of buffers alternate between rounds.
    d_defs = [torch.stack([d_def] * R) for _ in range(2)]                       # [2][R, H, W]
    d_ress = [torch.zeros((R, S_cap, 48), dtype=torch.uint8, device=dev) for _ in range(2)]
    d_alls = [torch.empty((world, R, S_cap, 48), dtype=torch.uint8, device=dev) for _ in range(2)] if use_dist else None
    solved = [torch.cuda.Event() for _ in range(R)]     # lane j's latest solve
    pending = {"bcast": [None, None], "gather": [None, None], "k": 0}

    def after_all_lanes(st_):   # st_ continues after the latest solve of every lane
        for ev in solved:
            st_.wait_event(ev)

    def prefetch(m):   # frames of round m -> d_defs[m % 2], over RCCL / xGMI
        return dist.broadcast(d_defs[m % 2], src=0, async_op=True) if use_dist else None

    def gather(m):     # warp parameters of round m
        return dist.all_gather_into_tensor(d_alls[m % 2].view(-1, 48), d_ress[m % 2].view(-1, 48), async_op=True)

    torch.cuda.set_stream(lanes[0][1])
    pending["bcast"][0] = prefetch(0)

    def step():
        k = pending["k"]
        m, j = divmod(k, R)
        eng, st_ = lanes[j]
        torch.cuda.set_stream(st_)   # RCCL orders its work against torch's current stream: the lane's own
        if use_dist:
            pending["bcast"][m % 2].wait()               # the frames of round m have arrived
            if j == 0:
                after_all_lanes(st_)                     # round m-1 no longer reads the other frame set
                pending["bcast"][(m + 1) % 2] = prefetch(m + 1)   # round m+1 travels during this one
            if pending["gather"][m % 2] is not None:
                pending["gather"][m % 2].wait()          # the records of round m-2 have left d_ress[m % 2]
        # upload + pyramid build of both frames of the pair (one launch, CudaClass::resetImagePyramids)
        eng.set_image_pair_device(d_und.data_ptr(), d_defs[m % 2][j].data_ptr(), wl.size, wl.size)
        eng.correlate_all_device(d_guess.data_ptr(), d_ress[m % 2][j].data_ptr())     # the solve
        if use_dist:
            solved[j].record(st_)
            if j == R - 1:
                after_all_lanes(st_)
                pending["gather"][m % 2] = gather(m)
        pending["k"] = k + 1

    def fence():
        if use_dist:
            m, j = divmod(pending["k"], R)
            if j != 0:                                   # a round cut short by the end of the timed region
                st_ = lanes[j - 1][1]
                torch.cuda.set_stream(st_)
                after_all_lanes(st_)
                pending["gather"][m % 2] = gather(m)
            for w in pending["gather"]:
                if w is not None:
                    w.wait()
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    m_last, j_last = divmod(pending["k"] - 1, R)
    res = d_ress[m_last % 2][j_last][:S].cpu().numpy().view(ca.RESULT_DTYPE).reshape(-1)   # the last timed step's records

    # The dominant kernel of the SAME engine on the SAME stream: per-launch duration measured live
    # with HIP events around K back-to-back solve launches (no pyramid launches in between), then
    # one untimed solve with the engine's own events on to fill lk_stats.
    torch.cuda.set_stream(stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(args.steps):
        e.correlate_all_device(d_guess.data_ptr(), d_ress[0][0].data_ptr())
    e1.record(stream)
    torch.cuda.synchronize(dev)
    solve_avg_ms = e0.elapsed_time(e1) / args.steps
    e.set_timing(True)
    e.set_image_device(ca.IMG_DEF, d_defs[0][0].data_ptr(), wl.size, wl.size)
    e.correlate_all_device(d_guess.data_ptr(), d_ress[0][0].data_ptr())
    torch.cuda.synchronize(dev)
    st = e.stats()   # counters + the engine's own HIP-event time of the LAST solve
    e.set_timing(False)

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    pit = torch.tensor([float(st["point_iterations"])], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(pit, op=dist.ReduceOp.SUM)
    dt_max = float(tmax.item())
    total_pit = float(pit.item()) * args.steps

    # ---- extra blocks at N = 1 (never part of `value`) ----------------------------------------
    overlapped = None
    if world == 1 and not use_dist and args.overlap > 1 and P == 1:
        Q = args.overlap
        extra = [make_engine(Q) for _ in range(Q)]
        d_r = [torch.zeros((S, 48), dtype=torch.uint8, device=dev) for _ in range(Q)]

        def ostep(k):
            eng, st_ = extra[k % Q]
            eng.set_image_pair_device(d_und.data_ptr(), d_def.data_ptr(), wl.size, wl.size)
            eng.correlate_all_device(d_guess.data_ptr(), d_r[k % Q].data_ptr())

        for k in range(min(args.warmup, 3 * Q)):
            ostep(k)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(args.steps):
            ostep(k)
        torch.cuda.synchronize(dev)
        dt_o = time.perf_counter() - t0
        t0 = time.perf_counter()
        for k in range(args.steps):   # solves only
            extra[k % Q][0].correlate_all_device(d_guess.data_ptr(), d_r[k % Q].data_ptr())
        torch.cuda.synchronize(dev)
        ms_o = (time.perf_counter() - t0) / args.steps * 1e3
        extra[0][0].set_timing(True)
        extra[0][0].correlate_all_device(d_guess.data_ptr(), d_r[0].data_ptr())
        torch.cuda.synchronize(dev)
        st_o = extra[0][0].stats()
        overlapped = {"pairs_in_flight": Q, "value": float(st_o["point_iterations"]) * args.steps / dt_o,
                      "ms_per_step": 1e3 * dt_o / args.steps, "solve_ms_per_launch": ms_o,
                      "algorithmic_GBps": st_o["algorithmic_bytes"] / (ms_o * 1e-3) / 1e9,
                      "frac_of_hbm_peak": st_o["algorithmic_bytes"] / (ms_o * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "note": f"{Q} independent pairs in flight, one engine and HIP stream each (16-lane groups); a tracked "
                              "sequence cannot do this (pair k+1 starts from pair k's result)"}
        for eng, _ in extra:
            eng.close()

    ref_order = None
    res_ref = None
    if world == 1 and not use_dist and not strong:
        er, st_r = make_engine(1, reference_order=1)
        torch.cuda.set_stream(st_r)
        d_rr = torch.zeros((S, 48), dtype=torch.uint8, device=dev)
        er.set_image_pair_device(d_und.data_ptr(), d_def.data_ptr(), wl.size, wl.size)
        for _ in range(3):
            er.correlate_all_device(d_guess.data_ptr(), d_rr.data_ptr())
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_r = max(10, args.steps // 4)
        r0.record(st_r)
        for _ in range(n_r):
            er.correlate_all_device(d_guess.data_ptr(), d_rr.data_ptr())
        r1.record(st_r)
        torch.cuda.synchronize(dev)
        ms_r = r0.elapsed_time(r1) / n_r
        res_ref = d_rr.cpu().numpy().view(ca.RESULT_DTYPE).reshape(-1)
        st_rr = er.stats()
        pc_r = profile_constants()
        ref_order = {"solve_ms": ms_r, "kernel_ms": ms_r, "point_iterations_per_s": st_rr["point_iterations"] / (ms_r * 1e-3),
                     "algorithmic_GBps": st_rr["algorithmic_bytes"] / (ms_r * 1e-3) / 1e9,
                     "frac": st_rr["algorithmic_bytes"] / (ms_r * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "algorithmic_bytes_per_launch": st_rr["algorithmic_bytes"],
                     "valu_issue_frac": valu_issue_frac(pc_r.get("reference_order_valu_insts_per_launch_C2"), ms_r),
                     "traffic": pc_r.get("reference_order_hbm_bytes_per_launch_C2"),
                     "kernel": "lk_solve_kernel<fm_UVUxUyVxVy, im_bicubic, 16 lanes, reference order>: K back-to-back launches "
                               "between two HIP events on the engine's stream",
                     "note": "lk_set_reference_order(1): ordered sums + the restated QR at every level; records "
                             "bit-identical to the CPU oracle (see parity_vs_cpu)"}
        er.close()
        torch.cuda.set_stream(stream)

    end_to_end = None
    if world == 1 and not use_dist and not strong:
        try:
            ee = ca.HipCorrelationEngine(interpolation=ca.IM_BICUBIC, fitting_model=wl.model, py_stop=wl.py_stop, device=local_rank)
            ee.set_rect_grid(wl.x_begin, wl.x_begin, wl.x_end, wl.x_end, wl.hs, wl.vs)
            ee.commit_sectors()
            g0 = np.zeros(6, np.float32)

            def pair(two):
                if two:
                    ee.set_undeformed_image(und)
                ee.set_deformed_image(dfm)
                return ee.correlate_all(g0)      # guesses up, solve, 48-byte records down, synchronised

            n_e = max(10, args.steps // 4)
            out_ms = {}
            for label, two in (("two_new_frames", True), ("one_new_frame", False)):
                for _ in range(3):
                    pair(two)
                t0 = time.perf_counter()
                for _ in range(n_e):
                    pair(two)
                out_ms[label] = (time.perf_counter() - t0) / n_e * 1e3
            # the reference's own loop shape (manager_class.cpp:1438-1447): while pair k is solved the next frame is loaded into the
            # next-image slot, then def <- nxt by pointer rotation.  The frame sits in PINNED host memory (lk_pin_host_memory), so
            # lk_set_image(LK_IMG_NXT) returns once the copy is enqueued on the next-frame stream: no helper thread - same host
            # frames, same records copied back, synchronous per pair
            import ctypes as C_
            nxt = np.ascontiguousarray(dfm).copy()
            pinned = ee.lib.lk_pin_host_memory(C_.c_void_p(nxt.ctypes.data), C_.c_size_t(nxt.nbytes)) == 0
            ee.set_deformed_image(dfm)

            ee.adjust_initial_guess(0, False, g0, (wl.size / 2.0, wl.size / 2.0))   # the engine-held (zero) guesses, once

            def prefetched(n):
                ee.set_next_image(nxt)
                for _ in range(n):
                    ee.correlate_all_async()          # the engine-held guesses; records follow the solve into pinned memory
                    ee.wait_results()                 # ... and are copied out
                    ee.makeDefPyramidFromNxt()
                    ee.set_next_image(nxt)
                ee.synchronize()

            prefetched(3)
            t0 = time.perf_counter()
            prefetched(n_e)
            out_ms["one_new_frame_prefetched"] = (time.perf_counter() - t0) / n_e * 1e3
            ee.synchronize()
            if pinned:
                ee.lib.lk_unpin_host_memory(C_.c_void_p(nxt.ctypes.data))
            st_e = ee.stats()
            end_to_end = {"ms_per_pair": out_ms["two_new_frames"], "ms_per_pair_one_new_frame": out_ms["one_new_frame"],
                          "ms_per_pair_one_new_frame_prefetched": out_ms["one_new_frame_prefetched"],
                          "point_iterations_per_s": st_e["point_iterations"] / (out_ms["two_new_frames"] * 1e-3),
                          "what": "host frames (pageable memory) -> lk_set_image upload + pyramids -> solve -> records copied back to "
                                  "the host, one pair at a time, synchronous (BASELINE.md 4.3); `one_new_frame`: the undeformed frame "
                                  "stays (a sequence with a fixed reference); `_prefetched`: the same with the next frame - in pinned host memory "
                                  "(lk_pin_host_memory) - uploaded into the next-image slot beside the running solve, def <- nxt by rotation: the "
                                  "reference's frame loop (manager_class.cpp:1438-1447).  Never the reported `value`."}
            ee.close()
        except Exception as ex:   # noqa: BLE001
            end_to_end = {"error": repr(ex)}

    # ---- tracked sequences: configs 2 and 4 as 64-pair sequences, frame-pipelined window against the one-pair loop ----
    sequence = None
    d_frames = None
    if not strong and args.sequence_frames > 1:
        try:
            d_frames = sequence_frames(ca, torch, dev, C2.size, args.sequence_frames)
        except Exception:   # noqa: BLE001 (the blocks that need them say so)
            d_frames = None
    if world == 1 and not use_dist and not strong and args.sequence_frames > 1:
        sequence = {}
        try:
            if d_frames is None:
                raise RuntimeError("the synthetic sequence could not be rendered")
            for key, w_ in (("C2", C2), ("C4", C4)):
                sequence[key] = {}
                for mode in ("default", "reference_order"):
                    try:
                        sequence[key][mode] = sequence_block(ca, torch, dev, w_, d_frames, mode)
                    except Exception as ex:   # noqa: BLE001 - extra evidence must never take the headline line down
                        sequence[key][mode] = {"error": repr(ex)}
        except Exception as ex:   # noqa: BLE001
            sequence["error"] = repr(ex)
        torch.cuda.set_stream(stream)

    # ---- multi-GPU configs: one pair with the sector grid split over the ranks (every rank takes part) ----
    sharded = {}
    if not strong and not args.no_sharded_configs:
        for key, w_ in (("C2_strong", C2), ("C4_sharded", C4), ("C5_sharded", C5)):
            try:
                sharded[key] = sharded_config(ca, torch, dist if use_dist else None, w_, rank, world, local_rank,
                                              max(5, args.steps // 10), max(2, args.warmup // 10))
            except Exception as ex:   # noqa: BLE001 (a failure after the ranks agreed: reported, the headline stands)
                sharded[key] = {"error": repr(ex)}
        # ... and tracked SEQUENCES sharded the same way (frame-pipelined windows; the case in which one node's GPUs scale)
        ok_frames = 1 if d_frames is not None else 0
        if use_dist:
            flag = torch.tensor([ok_frames], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok_frames = int(flag.item())
        for key, w_ in (("C4_sequence", C4), ("C2_sequence", C2)):
            if not ok_frames:
                sharded[key] = {"error": "the synthetic sequence could not be rendered on every rank"}
                continue
            try:
                sharded[key] = sharded_sequence(ca, torch, dist if use_dist else None, w_, d_frames, rank, world, local_rank)
            except Exception as ex:   # noqa: BLE001
                sharded[key] = {"error": repr(ex)}
        torch.cuda.set_stream(stream)
    d_frames = None
    native = None
    if not strong and not args.no_native_group:
        # the C-ABI group (one process, all devices) runs as a child of rank 0 while the other ranks keep off the GPUs:
        # a host-only rendezvous through a file, bounded waits on both sides
        flag_path = None
        if use_dist:
            tok = torch.tensor([int.from_bytes(os.urandom(6), "little") if rank == 0 else 0], dtype=torch.int64, device=dev)   # (not guessable)
            dist.broadcast(tok, src=0)
            torch.cuda.synchronize(dev)
            flag_path = os.path.join("/tmp", f"lk_bench_native_{int(tok.item())}.done")
        if rank == 0:
            native = native_group_child(args, "C2")
            if flag_path:
                with open(flag_path, "w") as f:
                    f.write("done")
        elif flag_path:
            t_wait = time.time()
            while not os.path.exists(flag_path) and time.time() - t_wait < 300:
                time.sleep(0.2)
            if not os.path.exists(flag_path):
                sys.stderr.write(f"rank {rank}: the native-group child of rank 0 did not finish within 300 s\n")

    if rank == 0:
        value = total_pit / dt_max
        achieved = st["algorithmic_bytes"] / (solve_avg_ms * 1e-3) / 1e9
        # HBM bytes per solve launch: a PROFILE CONSTANT from the PMC passes of this workload
        # (FETCH_SIZE + WRITE_SIZE, corrected; profiles/*_pmc_traffic.txt), not a counter read in this run
        traffic, traffic_src, valu_frac = None, None, None
        if not strong:   # the PMC profile is of the C2 launch
            pc = profile_constants()
            traffic = pc.get("solve_kernel_hbm_bytes_per_launch_C2")
            if traffic is not None:
                traffic_src = f"profile constant: {pc['file']} (rocprofv3 --pmc passes), bytes per launch"
            valu_frac = valu_issue_frac(pc.get("solve_kernel_valu_insts_per_launch_C2"), solve_avg_ms)
        line = {
            "metric": "correlation-point-iterations/sec",
            "value": value,
            "unit": "point-iterations/s",
            "n_gpus": world,
            "n_ranks": dist.get_world_size() if use_dist else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl.name, "sectors_per_gpu": S, "sectors_total": wl.hs * wl.vs if strong else S * world,
                       "samples_per_sector": n0,
                       "interpolation": "bicubic",
                       "parallelism": (f"sector grid sharded x{world} (strong scaling)" if strong else
                                       (f"WEAK scaling x{world}: every rank correlates its own {S}-sector grid on the broadcast pair (the work grows "
                                        "with N); the strong-scaling runs - ONE grid / ONE sequence split over the ranks - are "
                                        "`sharded_configs`" if world > 1 else "one GPU")),
                       "pairs_in_flight": P,
                       "step": "pyramid(und)+pyramid(def)+solve of one pair, inputs resident in HBM, "
                               + ("one pair at a time" if P == 1 else f"up to {P} independent pairs overlap on the GPU")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": ("lk_solve_kernel<fm_UVUxUyVxVy, im_bicubic> of the engine that produced `value` (the events "
                                    "also bracket the SAFE pass for ill-conditioned sectors, an empty launch here)"
                                    if not strong else
                                    "lk_solve_kernel<fm_UVUxUyVxVy, im_bicubic>: all launches of one solve"),
                         "kernel_ms": solve_avg_ms,
                         "valu_issue_frac": valu_frac,
                         "valu_issue_frac_at_measured_clock": (valu_issue_frac(profile_constants().get("solve_kernel_valu_insts_per_launch_C2"), solve_avg_ms,
                                                                               measured_clock_hz("one_pair_C2"))
                                                               if (not strong and measured_clock_hz("one_pair_C2")) else None),
                         "measured_clock_GHz": (measured_clock_hz("one_pair_C2") or 0) / 1e9 or None,
                         "valu_issue_frac_is": "SQ_INSTS_VALU per launch (profile constant) x 2 cycles / (1024 SIMDs x 2.4 GHz x kernel_ms) - beside it the same at the clock the traced "
                                               "wavefronts ran at (`measured_clock_GHz`, from profiles/r0N_wave_timeline.txt through the traffic file named above): "
                                               "the binding limit of this kernel is VALU issue and the critical path of its slowest sectors, not HBM",
                         "measured": "K back-to-back solve launches of the engine and stream the timed region ran on, "
                                     "between two HIP events on that stream",
                         "algorithmic_bytes_per_launch": st["algorithmic_bytes"]},
            "per_pair": {"sectors_per_s": (wl.hs * wl.vs if strong else S * world) * args.steps / dt_max,
                         "evaluations": st["evaluations"], "sample_evaluations": st["sample_evaluations"],
                         "point_iterations": st["point_iterations"],
                         "mean_point_iterations_per_sector": st["point_iterations"] / S,
                         "error_free_fraction": float((res["error_code"] == 0).mean()),
                         "nan_records": int(np.isnan(res["p"]).any(1).sum()),
                         "last_solve_ms": st["solve_ms"], "last_pyramid_ms": st["pyramid_ms"]},
        }
        if overlapped:
            line["overlapped"] = overlapped
        if ref_order:
            line["reference_order_mode"] = ref_order
        if end_to_end:
            line["end_to_end"] = end_to_end
        if sequence:
            line["sequence"] = sequence
        if sharded:
            line["sharded_configs"] = sharded
        if native is not None:
            line["native_group"] = native
        if world == 1 and not args.no_cpu_baseline and not strong:
            base, nsec = cpu_baseline(wl, und, dfm, args.cpu_sectors)
            pit_per_sector = st["point_iterations"] / S
            rate_mt, dt_mt, thr_mt, res_mt = base["all_cores"]
            rate_1, dt_1, _, res_1 = base["1_thread"]
            line["cpu_baseline"] = {
                "value": rate_mt * pit_per_sector, "unit": "point-iterations/s", "cores": thr_mt, "kind": "port",
                "sample": f"{nsec} evenly spaced sectors of the same C2 pair, oracle/lk_oracle.c, OpenMP across "
                          f"sectors, {dt_mt:.2f} s; point-iterations counted with the GPU run's mean per sector",
                "single_thread_value": rate_1 * pit_per_sector, "single_thread_seconds": dt_1,
                "sectors_per_s": rate_mt, "single_thread_sectors_per_s": rate_1,
            }
            # the baseline doubles as a full-size parity sample
            xs = np.linspace(0, S - 1, min(args.cpu_sectors, S)).astype(np.int64)
            def distance(r, w):   # one set of records against another: north_star's "chi to 1e-5" and what is around it
                rel_dchi = np.abs(r["chi"] - w["chi"]) / np.abs(w["chi"])
                same_it = r["iterations"] == w["iterations"]
                return {"max_abs_dp01": float(np.abs(r["p"][:, :2] - w["p"][:, :2]).max()),
                        "max_rel_dchi": float(rel_dchi.max()),
                        "iterations_equal_fraction": float(same_it.mean()),
                        "rel_dchi_p50": float(np.percentile(rel_dchi, 50)), "rel_dchi_p99": float(np.percentile(rel_dchi, 99)),
                        "rel_dchi_le_1e-5_fraction": float((rel_dchi <= 1e-5).mean()),
                        "max_rel_dchi_where_iterations_equal": float(rel_dchi[same_it].max()) if same_it.any() else None}

            line["parity_vs_cpu"] = {
                "sectors": int(len(xs)),
                "fast_mode": distance(res[xs], res_1),
                # the yardstick: the CPU engine against ITSELF with its default 20 sample chunks per sector instead of 1
                # (the sums of a sector in another order - what any parallel evaluation, the reference's own included, does)
                "cpu_split_20_vs_split_1": distance(base["split_20"], res_1),
            }
            if res_ref is not None:
                same = np.array([res_ref[i].tobytes() == res_1[j].tobytes() for j, i in enumerate(xs)])
                line["parity_vs_cpu"]["reference_order_mode"] = {
                    "records_bit_identical": int(same.sum()), "of": int(len(xs)),
                    "max_rel_dchi": float((np.abs(res_ref["chi"][xs] - res_1["chi"]) / np.abs(res_1["chi"])).max())}
        if world == 1 and not args.no_other_configs and not use_dist and not strong:
            for eng in {id(x[0]): x[0] for x in engines}.values():
                eng.close()
            engines = []
            line["other_configs"] = other_configs(ca)
        print(json.dumps(line))
    for eng in {id(x[0]): x[0] for x in engines}.values():
        eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
