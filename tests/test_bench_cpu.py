"""bench.py's host-side pieces that need no GPU: argument presets, the layout traffic floor of a
move list, the cpu_baseline leg (oracle on the host cores) and the self-launch command."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_config_presets_and_overrides():
    a = bench.parse_args([])
    assert (a.layers, a.block_size, a.seq_len, a.batch, a.kv_dtype, a.steady_cap) == (32, 16, 32768, 1, "fp16", 0)
    a = bench.parse_args(["--config", "c4"])
    assert (a.layers, a.seq_len, a.batch) == (80, 16384, 32)
    a = bench.parse_args(["--config", "c3"])
    assert (a.batch, a.steady_cap) == (256, 4096)
    a = bench.parse_args(["--config", "c5"])
    assert (a.kv_dtype, a.block_size, a.seq_len) == ("fp8", 32, 65536)
    a = bench.parse_args(["--config", "c4", "--batch", "4"])       # explicit flags win
    assert (a.layers, a.batch) == (80, 4)
    a = bench.parse_args(["--config", "c3i"])                       # config 3, initial phase
    assert (a.batch, a.keep, a.seq_len, a.steady_cap) == (16, 0.125, 32768, 0)
    assert bench.parse_args([]).keep == 0.5 and bench.parse_args(["--config", "c3i", "--keep", "0.25"]).keep == 0.25


def test_traffic_floor_of_a_small_move_list():
    bs, bb = 4, 256
    # head 0: two moves into block 5 (kept partly) from blocks 9 and 10; head 1: a full overwrite
    # of block 2 (4 moves) from block 7
    cmi = torch.tensor([[20, 36], [22, 40], [0, 0], [8, 28], [9, 29], [10, 30], [11, 31]], dtype=torch.int32)
    cmc = torch.tensor([[[2, 4]]], dtype=torch.int32)
    offs = torch.tensor([[[0, 3]]], dtype=torch.int32)
    f = bench.traffic_floor(cmi, cmc, offs, bs, bb)
    assert (f["dst_blocks"], f["dst_blocks_fully_overwritten"], f["src_blocks"]) == (2, 1, 3)
    img = 2 * bb + 8 * bs
    assert f["bytes"] == 1 * img + 2 * img + 3 * img + 6 * 8
    z = bench.traffic_floor(cmi, torch.zeros_like(cmc), offs, bs, bb)
    assert z["bytes"] == 0


def test_cpu_baseline_leg_runs_on_all_cores():
    a = bench.parse_args(["--layers", "2", "--kv-heads", "2", "--seq-len", "512"])
    c = bench.cpu_baseline(a)
    assert c["kind"] == "port" and c["host_cpus"] == os.cpu_count() and c["value"] > 0
    assert c["cores"] == c["torch_threads"] and str(os.cpu_count()) in c["S1_seconds_by_torch_threads"]
    assert set(c["stage_seconds"]) == {"S1_schedule", "S2_moves", "S3_compact"}
    assert c["single_core_port"]["cores"] == 1 and c["single_core_port"]["value"] > 0


def test_self_launch_command(monkeypatch):
    seen = {}

    class R:
        returncode = 0

    def fake_run(cmd, env=None, cwd=None):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench.self_launch(bench.parse_args(["--gpus", "4", "--steps", "3"])) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_compact_line_of_a_full_result_stays_under_4k():
    """round 5's driver record had `parsed: null`: the line had grown to 22 KB.  The full result of that run
    (profiles/r5_bench.json) through compact_line: the contract's keys, roofline and cpu_baseline survive, the
    line stays under 4 KB, the rest is what the detail file is for"""
    import json
    res = json.load(open(os.path.join(REPO, "profiles", "r5_bench.json")))
    assert len(json.dumps(res)) > 20000
    txt = bench.compact_line(res, os.path.join(REPO, "bench_detail.json"))
    assert len(txt) < bench.LINE_MAX_BYTES and "\n" not in txt
    d = json.loads(txt)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_checked", "stages_ms"):
        assert k in d, k
    assert d["detail"] == "bench_detail.json" and d["config"]["workload"].startswith("c2:")
    r = d["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
              "algorithmic_bytes_per_launch", "avg_launch_ms", "layout_amplification", "alg_frac_ceiling"):
        assert k in r, k
    assert abs(r["frac"] - res["roofline"]["frac"]) < 1e-5 and abs(r["traffic"] / res["roofline"]["traffic"] - 1) < 1e-5
    c = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "cpu_model", "threads_by_stage", "stage_seconds", "sample"} <= set(c)
    assert d["parity_checked"] == {"bit_exact": True, "mismatched": [], "kv_bytes": res["parity_checked"]["kv_bytes"]}
    assert set(d["other_configs"]) == {"c5", "c3"} and d["other_configs"]["c3"]["value"] > 0
