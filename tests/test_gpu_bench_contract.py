"""bench.py's output contract on the GPU box: the single-rank line carries `roofline` and
`cpu_baseline`; the N > 1 path (barrier, per-rank shards, all_gather of the two scalars,
whole-job value) is exercised with two ranks sharing the box's one GPU through the gloo test
hook (on a multi-GPU node the same code runs over RCCL, one rank per GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
        "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}


def _last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_single_rank_line(tmp_path):
    """the LAST stdout line is the compact one the driver parses (< 4 KB: round 5's 22 KB line was not parsed); the
    legs that do not fit it are in the detail file the line names"""
    detail = tmp_path / "detail.json"
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "4", "--warmup", "1",
                          "--seq-len", "4096", "--no-adjacent", "--engine-cache-frac", "0.02", "--no-live-traffic",
                          "--detail-json", str(detail)],
                         capture_output=True, text=True, timeout=600, cwd=REPO,
                         env={k: v for k, v in os.environ.items() if k != "KVC_SCHEDULE_PATH"})   # (the automatic choice is asserted below)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.rstrip("\n").splitlines()[-1]
    assert last.startswith("{") and len(last) < 4096, len(last)
    d = _last_json(out.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1
    assert d["unit"] == "KV slots/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["call_form"] == "fork"          # the headline is the fork's own call (scheduler.py:245-260, 491-529)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 and 0 < r["frac"] < 1
    assert r["algorithmic_bytes_per_launch"] > 0 and r["avg_launch_ms"] > 0 and r["layout_amplification"] >= 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["host_cpus"] == os.cpu_count() and c["cores"] >= 1 and c["value"] > 0
    assert set(c["stage_seconds"]) == {"S1_schedule", "S2_moves", "S3_compact"}
    assert d["parity_checked"]["bit_exact"] is True and d["parity_checked"]["mismatched"] == []
    assert set(d["stages_ms"]) == {"S1_schedule_evictions", "S2_schedule_moves", "S3_execute_moves"}
    h = d["stages_ms_hinted"]                # the same step, host list + total_slots= + tracked table
    assert h["same_counts"] is True and 0 < h["S1_schedule_evictions"] < 10 * d["stages_ms"]["S1_schedule_evictions"]
    assert "workload" in d["config"] and d["config"]["freed_blocks"] > 0
    # value == units / time
    units = d["config"]["evicted_slots"] + d["config"]["moved_slots"]
    assert abs(d["value"] - units / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-4
    # ---- the detail file: everything, unrounded
    assert d["detail"] and detail.exists()
    full = json.loads(detail.read_text())
    assert KEYS <= set(full) and abs(full["value"] / d["value"] - 1) < 1e-5
    r = full["roofline"]
    assert r["traffic_floor_bytes"] >= r["algorithmic_bytes_per_launch"] > 0
    assert abs(r["frac_of_floor"] - r["floor_GBps"] / r["peak"]) < 1e-12
    assert r["pattern_ceiling_GBps"] is None or r["pattern_ceiling_GBps"]["rmw_2R1W"] > 1000
    assert full["cpu_baseline"]["single_core_port"]["cores"] == 1
    ec = full["engine_sized_cache"]      # the same step in a (here: 2 % of HBM) larger cache: identical work
    assert ec["evicted_slots"] == d["config"]["evicted_slots"]
    assert abs(ec["moved_slots"] / d["config"]["moved_slots"] - 1) < 0.02      # another seed's metrics
    assert ec["cache_blocks"] > 8 * 32 * 8 * 257 and 0 < ec["roofline"]["frac"] < 1
    bt = ec["S1_with_block_tables"]    # the optional argument, measured next to the drop-in figure
    assert bt["used"] is True and bt["same_counts"] is True and 0 < bt["ms"] < 10 * ec["stages_ms"]["S1_schedule_evictions"]
    s0 = full["stages_ms_S0"]
    assert {"S0_aggregate_decode", "S0_aggregate_decode_fused_clear", "S0_aggregate_prefill",
            "S0_prefill_epilogue"} <= set(s0) and all(v["ms"] > 0 for v in s0.values())
    assert full["S1_call_forms"]["reference_call_form"]["same_counts"] is True


def test_headline_only_line():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--seq-len", "2048", "--headline-only", "--no-live-traffic", "--detail-json", ""],
                         capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert KEYS <= set(d) and "cpu_baseline" in d and d["parity_checked"]["bit_exact"] is True
    assert "stages_ms_hinted" not in d and "other_configs" not in d and d["detail"] is None


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_two_ranks_share_the_gpu(launcher):
    """`python bench.py --gpus 2` starts its own ranks (the command shape the driver uses); the
    torch.distributed.run form is what it re-executes itself under"""
    env = dict(os.environ, KVC_BENCH_BACKEND="gloo")
    tail = [os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--seq-len", "4096"]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", "29517"] + tail
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    assert len(d["per_rank"]) == 2
    units = sum(r["units"] for r in d["per_rank"])
    worst = max(r["seconds"] for r in d["per_rank"])
    assert abs(d["value"] - units / worst) / d["value"] < 1e-5      # (the line carries 6 significant digits)
    # both ranks processed a full shard of their own
    assert all(r["units"] > 0 for r in d["per_rank"])


def test_strong_scaling_splits_a_fixed_batch():
    env = dict(os.environ, KVC_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2",
                          "--warmup", "1", "--seq-len", "2048", "--batch", "3", "--scaling", "strong"],
                         capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and len(d["per_rank"]) == 2
    u = sorted(r["units"] for r in d["per_rank"])
    assert abs(u[1] / u[0] - 2.0) < 0.05          # 3 equal sequences -> shards of 1 and 2


def test_nccl_with_more_ranks_than_gpus_is_refused():
    import torch
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "1",
                          "--warmup", "0", "--seq-len", "1024"], capture_output=True, text=True,
                         timeout=600, cwd=REPO)
    assert out.returncode != 0 and "GPU(s)" in (out.stderr + out.stdout)


def test_c4_preset_with_two_ranks_at_toy_size():
    """BASELINE configs[3] (`--config c4`: the 70B shape as one of 8 GPUs sees it) goes through the
    N > 1 path too, at a size two ranks sharing one GPU can hold: the preset's 80 layers and 32
    sequences overridden by explicit flags, everything else -- rank-private caches, per-rank
    shards, the all_gather of the two scalars -- as the driver will run it over RCCL"""
    env = dict(os.environ, KVC_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--config", "c4",
                          "--layers", "4", "--seq-len", "512", "--batch", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and len(d["per_rank"]) == 2
    assert d["config"]["workload"].startswith("c4:") and "batch 2/GPU" in d["config"]["workload"]
    assert d["config"]["freed_blocks"] > 0 and all(r["units"] > 0 for r in d["per_rank"])


def test_eight_ranks_share_the_gpu_weak_and_strong():
    """N = 8 without 8 GPUs: `--gpus 8 --config c4` (BASELINE configs[3]'s shape at toy size) and a fixed batch of 256
    sequences split over 8 ranks (`--scaling strong --batch 256`), eight gloo ranks sharing the box's one GPU --
    rank-private caches, per-rank shards, the barrier-bracketed timed region, the all_gather of two scalars and the
    whole-job value, as the driver will run them over RCCL with one rank per GPU"""
    env = dict(os.environ, KVC_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--config", "c4",
                          "--layers", "2", "--seq-len", "256", "--batch", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=1200, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and len(d["per_rank"]) == 8
    assert d["config"]["workload"].startswith("c4:") and all(r["units"] > 0 for r in d["per_rank"])
    units = sum(r["units"] for r in d["per_rank"])
    assert abs(d["value"] - units / max(r["seconds"] for r in d["per_rank"])) / d["value"] < 1e-5      # (the line carries 6 significant digits)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--layers", "1", "--seq-len", "128",
                          "--batch", "256", "--scaling", "strong", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=1200, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and len(d["per_rank"]) == 8
    u = [r["units"] for r in d["per_rank"]]
    assert min(u) > 0 and max(u) / min(u) < 1.2          # 256 equal sequences -> 32 per rank (other seeds' metrics)
    assert "256 sequences split over 8 GPUs" in d["config"]["workload"]


def test_default_line_carries_the_other_configurations(tmp_path):
    """the default workload also reports short runs of BASELINE configs[4] and configs[2] (here
    with tiny step counts; bench.py shrinks a configuration that does not fit): one number each in the
    line, the legs in the detail file"""
    import bench
    old = bench.OTHER_CONFIGS
    detail = tmp_path / "detail.json"
    out = subprocess.run([sys.executable, "-c",
                          "import bench, sys; bench.OTHER_CONFIGS = (('c5', 2, 1), ('c3', 2, 1)); "
                          "sys.argv = ['bench.py', '--steps', '3', '--warmup', '1', '--no-adjacent', '--no-s0', "
                          "'--no-cpu-baseline', '--no-engine-cache', '--no-probe', '--no-live-traffic', "
                          f"'--detail-json', {str(detail)!r}]; bench.main()"],
                         capture_output=True, text=True, timeout=900, cwd=REPO,
                         env={k: v for k, v in os.environ.items() if k != "KVC_SCHEDULE_PATH"})   # (the automatic choice is asserted below)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert len(out.stdout.rstrip("\n").splitlines()[-1]) < 4096
    assert set(d["other_configs"]) == {"c5", "c3"} and all(v["value"] > 0 and v["bit_exact"] in (True, None)
                                                          for v in d["other_configs"].values())
    d = json.loads(detail.read_text())
    oc = {o["config"]: o for o in d["other_configs"]}
    assert set(oc) == {"c5", "c3"} and old
    for name, o in oc.items():
        assert "skipped" not in o, o
        assert o["stages_ms"]["S1_schedule_evictions"] > 0 and 0 < o["roofline"]["frac"] < 1
        assert 0 < o["roofline"]["frac_of_floor"] < 1.2 and o["S1_lower_bound_GBps"] > 0
    assert oc["c3"]["S1_schedule"] == "small_eviction" and oc["c5"]["S1_schedule"] == "bracket"
    # ... and configs[2] as a whole decode step (S0 + S1 + S2 + S3): two sweeps of the store against harvest-ahead
    ds = oc["c3"]["decode_step"]
    assert ds["parity_checked"]["bit_exact"] is True and ds["parity_checked"]["variants_agree"] is True
    # ... and in the fork's own flow (aggregate_decode() predicting the next call): on lists, oracle-checked
    ff = ds["fork_flow"]
    assert ff["harvested_steps"] == 2 and ff["parity_checked"]["bit_exact"] is True and ff["S1_schedule"] == "small_eviction"
    # ... and as the step without a sweep of the store: the attention's epilogues make the lists, no aggregate_decode
    fa = ds["fused_attention"]
    assert fa["parity_checked"]["bit_exact"] is True and fa["steps_on_the_epilogues_lists"] == 2, fa
    assert fa["stages_ms"]["S0_aggregate_decode"] == 0.0 and fa["S1_schedule_reason"].endswith("[lists: the attention's epilogue]")
    # ... and S1 through the call the fork makes (device tensor of counts, no total_slots) next to the list form
    for o in oc.values():
        cf = o["S1_call_forms"]
        # (the same schedule; the fork's form may run it with N on the device -- "bracket (...) [N on the device]")
        assert cf["reference_call_form"]["same_counts"] is True
        assert cf["reference_call_form"]["schedule"].split(" ")[0] == cf["list_form"]["schedule"].split(" ")[0]
        assert o["S1_reference_call_form_ms"] == cf["reference_call_form"]["ms"] > 0
    assert ds["two_sweeps"]["harvested_steps"] == 0 and ds["harvest_ahead"]["harvested_steps"] == 2
    assert ds["harvest_ahead"]["stages_ms"]["S1_schedule_evictions"] < ds["two_sweeps"]["stages_ms"]["S1_schedule_evictions"]
    assert "decode_step" not in oc["c5"]
    # ... whose S1 ran on the pivots of the call before; the line carries the sampling variant next to it
    assert oc["c3"]["S1_schedule_reason"].endswith("[pivots: the call before]")
    assert oc["c3"]["S1_sampled_pivots"]["same_counts"] is True and oc["c3"]["S1_sampled_pivots"]["ms"] > 0


def test_engine_leg_traffic_is_measured_live(tmp_path):
    """roofline.traffic of the engine-sized leg comes from two short rocprofv3 --pmc passes the
    bench runs itself (FETCH_SIZE / WRITE_SIZE, separate passes); here at a small size (where
    part of the images stays in the caches: only the order of magnitude is checked)."""
    import shutil
    if shutil.which("rocprofv3") is None and not os.path.exists("/opt/rocm/bin/rocprofv3"):
        pytest.skip("no rocprofv3 on this box")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--seq-len", "4096", "--no-adjacent", "--no-s0", "--no-cpu-baseline", "--no-probe",
                          "--engine-cache-frac", "0.02", "--detail-json", str(tmp_path / "detail.json")],
                         capture_output=True, text=True, timeout=900, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    _last_json(out.stdout)
    d = json.loads((tmp_path / "detail.json").read_text())
    r = d["engine_sized_cache"]["roofline"]
    assert r["traffic_source"].startswith("measured by this run"), r.get("traffic_source")
    floor = r["floor_GBps"] * 1e9 * r["avg_launch_ms"] * 1e-3
    assert 0.5 * floor < r["traffic"] < 4.0 * floor, (r["traffic"], floor)
