"""bench.py's contract, checked without a GPU: the command line the driver uses, and the JSON lines committed under
profiles/ by scripts/profile_round.sh (the default N = 1 run and the one-rank rehearsal of the multi-process run,
LK_BENCH_FORCE_DIST=1, which goes through RCCL and prints every block of the N > 1 line)."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NUM = (int, float)


def check_sequence_blocks(d, n1):
    """round 4: tracked sequences - frame-pipelined windows against the one-pair loop (N = 1) and sharded over the ranks"""
    for key in ("C4_sequence", "C2_sequence"):
        s = d["sharded_configs"][key]
        assert "error" not in s, s
        assert s["n_ranks"] == d["n_ranks"] and s["scaling"] == "strong" and s["pairs"] >= 2 and s["window_pairs"] >= 1
        assert s["ms_per_pair"] > 0 and abs(s["speedup_vs_1gpu"] - s["ms_per_pair_1gpu"] / s["ms_per_pair"]) < 1e-9
    if n1:
        for cfg in ("C2", "C4"):
            for mode in ("default", "reference_order"):
                b = d["sequence"][cfg][mode]
                assert "error" not in b, b
                w = b["window"]
                assert b["pipelined_instances"] is True and w["kernel_ms_per_pair"] > 0 and w["ms_per_pair"] >= w["kernel_ms_per_pair"]
                assert abs(w["frac"] - w["algorithmic_bytes_per_pair"] / (w["kernel_ms_per_pair"] * 1e-3) / 1e9 / 8000.0) < 1e-9
                assert w["speedup_vs_one_pair_at_a_time"] > 1.0
                # (the windows' profile constants - profiles/r04_traffic.json, scripts/profile_sequence.sh - must reach the line:
                # a refresh of the one-pair constants once dropped them and the line carried nulls)
                if (cfg, mode) != ("C4", "reference_order"):   # (no counter pass of its own: config 4's default mode runs the same instance)
                    assert 0.05 < (w.get("valu_issue_frac") or 0) < 1.0 and (w.get("traffic_hbm_bytes_per_pair") or 0) > 0, (cfg, mode)
                    assert (w.get("measured_clock_GHz") or 0) > 1.0, (cfg, mode)
            ro = d["sequence"][cfg]["reference_order"]["one_pair_at_a_time"]
            assert ro["frames_with_identical_records"] == ro["of"] == d["sequence"][cfg]["reference_order"]["pairs"]
        assert d["end_to_end"]["ms_per_pair_one_new_frame_prefetched"] > 0


def check_line(d, n1):
    for k, t in (("metric", str), ("value", NUM), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", NUM), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], t), k
    assert d["metric"] == "correlation-point-iterations/sec" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["scaling"] in ("weak", "strong")
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["kernel_ms"] > 0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert "valu_issue_frac" in r and (r["traffic"] is None or r["traffic"] > 0)
    # value is consistent with the step time and the per-pair counters
    per_pair = d["per_pair"]["point_iterations"] * d["n_gpus"]
    assert abs(d["value"] - per_pair / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    # one pair of each multi-GPU config with its sector grid split over the ranks, and the ratio to one GPU
    for key in ("C2_strong", "C4_sharded", "C5_sharded"):
        s = d["sharded_configs"][key]
        assert "error" not in s, s
        assert s["n_ranks"] == d["n_ranks"] and s["scaling"] == "strong"
        assert s["ms_per_step"] > 0 and s["ms_per_step_1gpu"] > 0
        assert abs(s["speedup_vs_1gpu"] - s["ms_per_step_1gpu"] / s["ms_per_step"]) < 1e-9
    g = d["native_group"]
    assert "error" not in g, g
    assert g["n_ranks"] == d["n_gpus"] and g["ms_per_step"] > 0 and g["scaling"] == "strong"
    if n1:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
        ro = d["reference_order_mode"]
        assert abs(ro["frac"] - ro["algorithmic_bytes_per_launch"] / (ro["kernel_ms"] * 1e-3) / 1e9 / 8000.0) < 1e-9
        assert d["parity_vs_cpu"]["reference_order_mode"]["records_bit_identical"] == d["parity_vs_cpu"]["reference_order_mode"]["of"]
        assert d["end_to_end"]["ms_per_pair"] > d["ms_per_step"]
        p4 = d["other_configs"]["C4_one_pair"]["parity_vs_cpu"]
        assert p4["sectors"] == 50176 and p4["nan_set_differs"] <= 8 and p4["error_codes_differ"] <= 40
        for key in ("C3", "C4_one_pair", "C5"):
            assert d["other_configs"][key]["solve_ms"] > 0


def test_command_line_of_the_driver():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in r.stdout


@pytest.mark.parametrize("pattern,n1", [("profiles/r0[3-9]_bench.json", True), ("profiles/r0[3-9]_bench_dist_rehearsal.json", False)])
def test_committed_bench_lines_follow_the_contract(pattern, n1):
    files = sorted(glob.glob(os.path.join(ROOT, pattern)))
    if not files:
        pytest.skip("no committed line of this kind yet")
    for f in files:
        lines = [ln for ln in open(f).read().splitlines() if ln.startswith("{")]
        assert len(lines) == 1, f
        check_line(json.loads(lines[0]), n1)
        if os.path.basename(f) >= "r04":
            check_sequence_blocks(json.loads(lines[0]), n1)


def test_native_group_child_leaves_the_launcher_environment_behind(monkeypatch):
    """`bench.py --gpus N` under torch.distributed.run starts `bench.py --native` (one process, all devices) as a child of
    rank 0.  The child must not inherit the launcher's rendezvous variables: `--native` refuses WORLD_SIZE > 1, so with
    them the block would read {"error": ...} in every run that has more than one rank - the runs it exists for."""
    import argparse
    import subprocess

    import bench
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw.get("env")
        line = {"value": 1.0, "ms_per_step": 1.0, "scaling": "strong", "config": {"workload": "C2", "parallelism": "lk_group", "n_ranks": 8},
                "roofline": {"kernel_ms": 0.1}, "per_pair": {"error_free_fraction": 1.0}}
        return subprocess.CompletedProcess(cmd, 0, stdout="noise\n" + json.dumps(line) + "\n", stderr="")

    monkeypatch.setattr(subprocess, "run", fake_run)
    for k, v in (("WORLD_SIZE", "8"), ("RANK", "0"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29500"),
                 ("TORCHELASTIC_RUN_ID", "x"), ("LK_BENCH_FORCE_DIST", "1"), ("HSA_ENABLE_IPC_MODE_LEGACY", "0")):
        monkeypatch.setenv(k, v)
    out = bench.native_group_child(argparse.Namespace(gpus=8, steps=20, warmup=5), "C2")
    assert out["n_ranks"] == 8 and "error" not in out
    env = seen["env"]
    assert env is not None and env.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"      # (what RCCL needs stays)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "LK_BENCH_FORCE_DIST"):
        assert k not in env, k
    assert "--native" in seen["cmd"] and seen["cmd"][seen["cmd"].index("--gpus") + 1] == "8"
