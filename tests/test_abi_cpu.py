"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/kvc_mi355x.h declares; the Python surface mirrors the reference names;
the ops refuse to run without a HIP device (no silent fallback)."""
import inspect
import os

import re

import numpy as np

import pytest
import torch

import vllm_kvcompress_amd as kvc
from vllm_kvcompress_amd import _custom_ops as ops
from vllm_kvcompress_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "kvc_mi355x.h")).read()
    declared = set(re.findall(r"\b(kvc_[a-z0-9_]+)\s*\(", header))
    declared -= {"kvc_schedule_params", "kvc_attention_params"}
    assert declared, "no declarations found"
    lib = kvc.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    assert lib.kvc_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define KVC_ABI_VERSION (\d+)", header).group(1))


def test_python_surface_matches_reference_signatures():
    """positional order of the reference wrappers (vllm/_custom_ops.py:1065,1158,1220,641)"""
    want = {
        "count_block_evictions": ["evicted_block_count", "evicted_logical_indices",
                                  "evicted_kv_offsets", "hanging_token_count", "block_size",
                                  "null_value", "evicted_blocks_per_seq"],
        "schedule_cache_moves": ["out_cache_moves_indices", "out_cache_moves_count",
                                 "evicted_logical_indices", "evicted_kv_count",
                                 "evicted_kv_offsets", "block_tables", "context_lens",
                                 "block_size"],
        "execute_cache_moves": ["k_cache", "v_cache", "kv_metrics", "kv_position",
                                "cache_moves_indices", "cache_moves_count", "evicted_kv_offsets",
                                "blocks_per_head", "threads_per_head"],
        "reshape_and_cache_kvc": ["key", "value", "key_cache", "value_cache", "kv_metrics",
                                  "slot_mapping", "kv_metric_head_bias", "kv_cache_dtype",
                                  "k_scale", "v_scale"],
    }
    # vllm/_custom_ops.py:135-155 and :167-191
    attn_tail = ["query", "key_cache", "value_cache", "num_kv_heads", "scale", "block_tables",
                 "context_lens", "kv_position", "last_position", "kv_metric_buffer_len",
                 "block_size", "max_context_len", "alibi_slopes", "kv_cache_dtype", "k_scale",
                 "v_scale", "record_kv_metrics"]
    want["paged_attention_kvc_v1"] = ["out", "kv_metric_out"] + attn_tail
    want["paged_attention_kvc_v2"] = ["out", "kv_metric_out", "exp_sum", "max_logits", "tmp_out",
                                      "tmp_kv_metric_out"] + attn_tail
    for fn, params in want.items():
        assert list(inspect.signature(getattr(ops, fn)).parameters) == params
    from vllm_kvcompress_amd.kvcompress.metrics import CompressionMetrics
    init = list(inspect.signature(CompressionMetrics.__init__).parameters)[1:]
    assert init == ["block_size", "num_layers", "num_kv_heads", "num_queries_per_kv",
                    "max_kv_per_sort", "kv_head_bias_file", "kv_head_bias_weight", "device",
                    "random", "even_layer_evict", "use_l2", "use_average",
                    "record_decoding_metrics", "num_attention_sinks"]
    sched = list(inspect.signature(CompressionMetrics.schedule_evictions).parameters)[1:8]
    assert sched == ["seq_indices", "seq_positions", "evicted_blocks_per_seq", "context_lens",
                     "hanging_token_count", "evicted_kv_offsets", "num_protected"]
    for m in ("init_kv_metadata", "clear_temp_metrics", "insert_metadata", "remove_metadata",
              "aggregate_prefill", "aggregate_decode", "profile_schedule_evictions"):
        assert callable(getattr(CompressionMetrics, m))


def test_dispatcher_names_registered():
    from vllm_kvcompress_amd import torch_ops
    torch_ops.register()
    torch_ops.register()          # idempotent
    for name in ("count_block_evictions", "schedule_t1_cache_moves", "execute_cache_moves"):
        assert hasattr(torch.ops._C_kvc_ops, name)
    assert hasattr(torch.ops._C_cache_ops, "kvcompress_reshape_and_cache")
    assert hasattr(torch.ops._C, "kvcompress_paged_attention_v1")
    assert hasattr(torch.ops._C, "kvcompress_paged_attention_v2")


def test_no_cpu_fallback():
    t = torch.zeros(4, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="HIP device"):
        ops.count_block_evictions(t, t, t, t, 4, 99)
    from vllm_kvcompress_amd.kvcompress.metrics import CompressionMetrics
    with pytest.raises(RuntimeError, match="HIP device"):
        CompressionMetrics(16, 2, 2, 1, 1000, None, 0.0, device="cpu")


def test_product_never_imports_oracle():
    """the oracle is test infrastructure; the package must not reference it"""
    pkg = os.path.join(REPO, "vllm_kvcompress_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "kvc_oracle" not in src.replace("oracle/kvc_oracle.py", ""), f


def test_sources_are_gfx950_only():
    """no compatibility layers in the product: no dual CUDA/HIP paths, no hipify residue, no
    Triton, no 32-lane idioms; build script targets gfx950 and nothing else"""
    pkg = os.path.join(REPO, "vllm_kvcompress_amd")
    banned = ("__HIP_PLATFORM", "__CUDA_ARCH__", "cuda_runtime", "__shfl_sync", "__ballot_sync",
              "import triton", "warpSize", "hipify")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".hip", ".h", ".py", ".sh")):
                src = open(os.path.join(root, f)).read()
                for b in banned:
                    assert b not in src, (f, b)
    build = open(os.path.join(pkg, "csrc", "build.sh")).read()
    assert build.count("--offload-arch=gfx950") == build.count("--offload-arch") >= 1


def test_schedule_plan_is_a_host_decision():
    """kvc_schedule_evictions_plan: which schedule a call enqueues follows from the parameters alone
    (no device needed): 1 = small-eviction (the hint admits it), 2 = bracket (bulk, >= 64 Ki slots per
    sequence and 64 blocks per head, <= 1024 heads per sequence, the reference's batch > 1 rule up to
    256 sequences), 0 = the digit rounds"""
    import ctypes
    lib = kvc.load()

    def plan(B=1, L=32, H=8, bs=16, slots_per_head=32768, mode=0, hint=-1, path=0):
        p = _lib.KvcScheduleParams()
        p.num_seqs, p.num_layers, p.num_kv_heads, p.block_size = B, L, H, bs
        p.total_slots = B * L * H * slots_per_head
        p.num_blocks = p.total_slots // bs
        p.mode, p.max_evicted_blocks_hint, p.schedule_path = mode, hint, path
        return int(lib.kvc_schedule_evictions_plan(ctypes.byref(p)))

    assert plan() == 2                                            # config 2
    assert plan(bs=32, slots_per_head=65536) == 2                 # config 5
    assert plan(hint=200) == 1                                    # <= 2 blocks per head on average: small-eviction
    assert plan(hint=200, path=1) == 0 and plan(path=1) == 0      # digit rounds forced
    assert plan(hint=200, path=4) == 2                            # bracket forced
    assert plan(slots_per_head=128) == 0                          # 32 Ki slots in the sequence
    assert plan(L=80, slots_per_head=512) == 0                    # heads of 32 blocks
    assert plan(slots_per_head=128, path=4) == 2                  # ... unless forced
    assert plan(B=16, mode=0, slots_per_head=4096) == 2           # the reference's batch > 1 rule
    assert plan(B=300, mode=0, slots_per_head=4096, L=4) == 0     # more coupled sequences than the gated launch has tables for
    assert plan(B=300, mode=1, slots_per_head=4096, L=4) == 2     # sequences that do not couple: any number
    assert plan(L=160, H=8, slots_per_head=2048) == 0             # 1280 heads per sequence
    assert plan(L=160, H=8, slots_per_head=2048, path=4) == 0


def test_schedule_plan_says_why():
    """kvc_schedule_evictions_plan_reason: one case per KVC_WHY_* code (include/kvc_mi355x.h) -- which
    static limit sent a call to which schedule, and the form of the fallback behind a taken one"""
    import ctypes
    from vllm_kvcompress_amd.kvcompress.metrics import CompressionMetrics
    lib = kvc.load()
    header = open(os.path.join(REPO, "include", "kvc_mi355x.h")).read()
    codes = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"KVC_WHY_([A-Z_]+) = (\d+)", header)}
    assert set(codes.values()) == set(_lib.WHY), "the Python names must cover the header's codes"

    def why(B=1, L=32, H=8, bs=16, slots_per_head=32768, mode=0, hint=-1, path=0, total=None, num_blocks=None):
        p = _lib.KvcScheduleParams()
        p.num_seqs, p.num_layers, p.num_kv_heads, p.block_size = B, L, H, bs
        p.total_slots = B * L * H * slots_per_head if total is None else total
        p.num_blocks = p.total_slots // max(bs, 1) if num_blocks is None else num_blocks
        p.mode, p.max_evicted_blocks_hint, p.schedule_path = mode, hint, path
        plan = int(lib.kvc_schedule_evictions_plan(ctypes.byref(p)))
        r = int(lib.kvc_schedule_evictions_plan_reason(ctypes.byref(p)))
        return plan, _lib.WHY[r & 0xFF], _lib.WHY[(r >> 8) & 0xFF], _lib.WHY[(r >> 16) & 0xFF]

    assert why(hint=200) == (1, "taken", "taken", "taken")
    assert why(hint=8, B=300, L=4, slots_per_head=4096) == (1, "taken", "taken", "coupled_batch")
    assert why(hint=8, B=300, L=4, slots_per_head=4096, mode=1) == (1, "taken", "taken", "taken")
    assert why() == (2, "hint_unknown", "taken", "taken")                       # a device tensor of counts: bracket
    assert why(hint=100000) == (2, "bulk_eviction", "taken", "taken")          # compress_once
    assert why(hint=200, path=1) == (0, "forced_path", "forced_path", "taken")
    assert why(hint=200, path=4) == (2, "forced_path", "taken", "taken")
    assert why(hint=1, bs=4, slots_per_head=64) == (0, "block_size", "small_batch", "taken")
    assert why(hint=1, L=160, slots_per_head=2048) == (0, "heads_per_seq", "heads_per_seq", "taken")
    assert why(hint=1, L=128, H=8, bs=8, slots_per_head=256) == (0, "thresholds_lds", "small_batch", "taken")
    assert why(hint=1, B=70000, L=1, H=1, slots_per_head=16, mode=1)[1] == "index_range"
    assert why(hint=1, num_blocks=1 << 29, slots_per_head=64) == (0, "index_range", "small_batch", "taken")
    assert why(hint=100000, B=300, L=4, slots_per_head=4096) == (0, "bulk_eviction", "coupled_batch", "taken")
    assert why(slots_per_head=128) == (0, "hint_unknown", "small_batch", "taken")
    assert why(total=0)[1:3] == ("empty", "empty")
    # ... and the sentence CompressionMetrics.last_schedule_reason makes of it
    d = CompressionMetrics._describe_plan
    assert d(1, 0, False) == "small_eviction"
    assert d(1, 7 << 16, False) == "small_eviction (fallback: gated launch chain, coupled_batch)"
    assert d(2, 3, False) == "bracket (small_eviction: hint_unknown)"
    assert d(0, 3 | (9 << 8), False) == "general (small_eviction: hint_unknown; bracket: small_batch)"
    assert d(0, 1 | (1 << 8), True) == "general (small_eviction: backoff_after_fallback; bracket: forced_path)"


def test_harvest_eligibility_and_buffer_sizes():
    """ABI version 5, the host-side questions (no device needed): which calls may run on lists the aggregation
    pass made / take the pivots of the call before (any call that takes the small-eviction schedule, whatever its
    keys depend on), and how large the buffer between them is"""
    import ctypes
    lib = kvc.load()
    keep = []

    def params(B=1, L=32, H=8, bs=16, slots_per_head=4096, mode=1, hint=8, path=0, use_average=0, bias=False, uniform=0):
        p = _lib.KvcScheduleParams()
        p.num_seqs, p.num_layers, p.num_kv_heads, p.block_size = B, L, H, bs
        p.total_slots = B * L * H * slots_per_head
        p.num_blocks = p.total_slots // bs
        p.mode, p.max_evicted_blocks_hint, p.schedule_path = mode, hint, path
        p.use_average, p.uniform_evict = use_average, uniform
        if bias:
            keep.append(ctypes.create_string_buffer(16))
            p.bias = ctypes.cast(keep[-1], ctypes.c_void_p)
        return p

    def both(qpk=4, **kw):
        p = params(**kw)
        return int(lib.kvc_harvest_eligible(ctypes.byref(p), qpk)), int(lib.kvc_pivot_memory_eligible(ctypes.byref(p)))

    assert both() == (1, 1)
    assert both(qpk=8) == (1, 1) and both(qpk=7) == (1, 1) and both(qpk=1) == (1, 1) and both(qpk=0) == (0, 1)
    assert both(B=16) == (1, 1)                                   # per_sequence: sequences do not need each other
    assert both(B=16, mode=0) == (1, 1)                           # the reference's batch > 1 rule: the position rows are streamed too
    assert both(B=1, mode=0) == (1, 1)
    assert both(use_average=1) == (1, 1) and both(bias=True) == (1, 1)     # keys that depend on the position: the same
    assert both(path=3) == (1, 1)                                 # the full pass forced (tests)
    assert both(hint=100000) == (0, 0) and both(hint=-1) == (0, 0)         # bulk / unknown: not the small-eviction schedule
    assert both(path=1) == (0, 0) and both(bs=4) == (0, 0) and both(uniform=1) == (0, 0)
    assert int(lib.kvc_harvest_eligible(None, 4)) == 0 and int(lib.kvc_pivot_memory_eligible(None)) == 0
    # the attention epilogue's harvest: keys that are the sum alone (the reference's batch > 1 rule included: the epilogue
    # counts the masked slots per head); not averaged / biased metrics
    att = lambda **kw: int(lib.kvc_attention_harvest_eligible(ctypes.byref(params(**kw))))
    assert att() == 1 and att(B=16) == 1 and att(B=1, mode=0) == 1 and att(B=16, mode=0) == 1
    assert att(use_average=1) == 0 and att(bias=True) == 0 and att(path=3) == 0
    assert att(hint=100000) == 0 and att(path=1) == 0 and int(lib.kvc_attention_harvest_eligible(None)) == 0
    # the buffer: 256 B of header, 4 B per sequence, then (lists) 64 counters a cache line apart, 2 x 4 B + 256 x 8 B per head,
    # then what lists made by the attention's epilogue were made with: 4 B per head, 8 B per sequence
    for G, B in ((256, 1), (65536, 256), (4096, 16)):
        piv, full = int(lib.kvc_harvest_pivot_bytes(B)), int(lib.kvc_harvest_buffer_bytes(G, B))
        assert piv % 256 == 0 and 256 + 4 * B <= piv < 256 + 4 * B + 256
        body = 64 * 128 + G * 8 + G * 256 * 8 + G * 4 + B * 8
        assert full % 256 == 0 and full >= piv + body and full < piv + body + 6 * 256
    assert int(lib.kvc_harvest_buffer_bytes(0, 1)) == 0 and int(lib.kvc_harvest_pivot_bytes(0)) == 0


def test_host_policy_around_predicted_pivots():
    """CompressionMetrics' bookkeeping for calls that ran on predicted pivots (harvested lists / the call before's
    pivots), without a device: a raised flag doubles the allowance, drops the pivots and pauses predictions for
    2, 4 ... 256 calls; 64 clean predicted calls in a row halve the allowance again; flags of calls that sampled
    keep their own penalty (the digit rounds for 1, 2, 4 ... 64 calls)"""
    from vllm_kvcompress_amd.kvcompress.metrics import CompressionMetrics
    cm = object.__new__(CompressionMetrics)
    cm.harvest_misses, cm.harvest_widen, cm._hv_widen0 = 0, 0.25, 0.25
    cm._hv, cm._hv_lists = {"k": [1]}, {"k": [1]}
    cm._hv_streak = cm._hv_pause = cm._hv_pause_len = 0
    cm.schedule_path, cm._fb_penalty, cm._fb_backoff = 0, 0, 0
    pauses, widens = [], []
    for _ in range(10):
        cm._hv = {"k": [1]}
        cm._note_flag(1, True)
        assert cm._hv is None and cm._hv_lists is None
        pauses.append(cm._hv_pause)
        widens.append(cm.harvest_widen)
    assert pauses == [2, 4, 8, 16, 32, 64, 128, 256, 256, 256] and cm.harvest_misses == 10
    assert widens[:6] == [0.5, 1.0, 2.0, 4.0, 8.0, 8.0] and cm._fb_penalty == 0
    for n in range(64 * 5):
        cm._note_flag(0, True)
    assert cm._hv_pause_len == 0 and cm.harvest_widen == 0.25          # 8 -> 4 -> 2 -> 1 -> 0.5 -> 0.25
    cm._note_flag(1, True)
    assert cm._hv_pause == 2 and cm.harvest_widen == 0.5
    # the allowance in evictions: a quarter of it may go to a sequence that frees more than it did
    cm.harvest_widen = 0.25
    a = lambda *v: np.asarray(v, dtype=np.int64)
    assert cm._k_within(a(8, 8), a(8, 8)) and cm._k_within(a(9, 1), a(8, 8)) and not cm._k_within(a(10, 8), a(8, 8))
    assert not cm._k_within(a(8, 8, 8), a(8, 8))
    assert not cm._k_within(a(8), a(8, 8)) and cm._k_within(a(0, 0), a(0, 0)) and not cm._k_within(a(1, 0), a(0, 0))
    # a call that sampled: the usual penalty
    cm._note_flag(1, False)
    assert (cm._fb_penalty, cm._fb_backoff) == (1, 1)
    cm._note_flag(1, False)
    assert cm._fb_penalty == 2
    cm._note_flag(0, False)
    assert cm._fb_penalty == 0 and cm.harvest_misses == 11


def test_c_host_compiles_against_the_header(tmp_path):
    """tests/cabi/cabi_host.cpp -- the host without Python that tests/test_gpu_cabi.py runs on the GPU -- compiles
    against include/kvc_mi355x.h and links against the library here as well (no device needed for that): the
    header, the struct layouts it uses and the exported symbols stay in step"""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    libdir = os.path.join(REPO, "vllm_kvcompress_amd")
    host_o, orc_o, exe = str(tmp_path / "cabi_host.o"), str(tmp_path / "kvc_oracle.o"), str(tmp_path / "cabi_host")
    subprocess.check_call(["gcc", "-O1", "-fopenmp", "-c", os.path.join(REPO, "oracle", "kvc_oracle.c"), "-o", orc_o])
    gomp = subprocess.check_output(["gcc", "-print-file-name=libgomp.so"], text=True).strip()
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-c",
                           os.path.join(REPO, "tests", "cabi", "cabi_host.cpp"), "-I", os.path.join(REPO, "include"),
                           "-o", host_o])
    subprocess.check_call([hipcc, host_o, orc_o, gomp, "-L", libdir, "-lkvc_mi355x", f"-Wl,-rpath,{libdir}", "-o", exe])
    assert os.path.getsize(exe) > 0
