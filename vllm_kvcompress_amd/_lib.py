"""ctypes binding of libkvc_mi355x.so (the C ABI declared in include/kvc_mi355x.h).

There is no CPU fallback: if the shared library is missing, or a tensor is not on a
HIP device, the ops raise.  The library is built in-tree by
``vllm_kvcompress_amd/csrc/build.sh`` (``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int32, c_int64, c_size_t, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# KVC_MI355X_LIB: alternative build of the same library (kernel experiments)
LIB_PATH = os.environ.get("KVC_MI355X_LIB", os.path.join(_HERE, "libkvc_mi355x.so"))

MAX_INT = 2147483000  # reference vllm/kvcompress/metrics.py:12
ABI_VERSION = 8       # KVC_ABI_VERSION of include/kvc_mi355x.h: the struct layouts mirrored below

# KVC_LAYOUT_* of include/kvc_mi355x.h: how the bytes INSIDE a cache block are laid out.  The fork hands the ops
# views of an opaque [2, NB, bs * hd] tensor (reference vllm/attention/ops/paged_attn.py:262-284); the three ops that
# interpret a block -- reshape_and_cache_kvc, paged_attention_kvc_*, execute_cache_moves -- are all this package's,
# so the layout is a switch of the package: "reference" (default: K [hd/x][bs][x], V [hd][bs], byte for byte the
# fork's) or "slot_major" (K [bs][hd], V [bs][hd]: a slot is two contiguous runs, a move two contiguous copies).
# One layout per process, chosen before the first cache write: KVC_BLOCK_LAYOUT=slot_major or set_block_layout().
LAYOUTS = {"reference": 0, "slot_major": 1}
_block_layout = os.environ.get("KVC_BLOCK_LAYOUT", "reference") or "reference"
if _block_layout not in LAYOUTS:
    raise ImportError(f"KVC_BLOCK_LAYOUT={_block_layout!r}: expected one of {sorted(LAYOUTS)}")


def set_block_layout(name: str) -> None:
    """``"reference"`` or ``"slot_major"`` for every cache op of this process from now on (a cache written in one
    layout must be read and compacted in the same one: switch before the first write, or convert with
    ``vllm_kvcompress_amd.layout.convert_block_layout``)."""
    global _block_layout
    if name not in LAYOUTS:
        raise ValueError(f"block layout {name!r}: expected one of {sorted(LAYOUTS)}")
    _block_layout = name
    try:                                   # the compiled dispatcher binding keeps its own copy
        import torch
        if hasattr(torch.ops, "_kvc_mi355x") and hasattr(torch.ops._kvc_mi355x, "set_block_layout"):
            torch.ops._kvc_mi355x.set_block_layout(LAYOUTS[name])
    except (ImportError, RuntimeError):
        pass


def block_layout() -> str:
    return _block_layout


def block_layout_id() -> int:
    return LAYOUTS[_block_layout]


# KVC_WHY_* of include/kvc_mi355x.h (kvc_schedule_evictions_plan_reason)
WHY = {0: "taken", 1: "forced_path", 2: "block_size", 3: "hint_unknown", 4: "bulk_eviction",
       5: "heads_per_seq", 6: "thresholds_lds", 7: "coupled_batch", 8: "index_range",
       9: "small_batch", 10: "empty", 11: "uniform_evict"}


class KvcScheduleParams(ctypes.Structure):
    """mirror of ``kvc_schedule_params`` (include/kvc_mi355x.h)"""
    _fields_ = [
        ("metrics", c_void_p), ("token_positions", c_void_p),
        ("seq_index_by_block", c_void_p), ("layer_index_by_block", c_void_p),
        ("head_index_by_block", c_void_p), ("logical_block_num_by_block", c_void_p),
        ("num_blocks", c_int64),
        ("block_size", c_int32), ("num_layers", c_int32), ("num_kv_heads", c_int32),
        ("num_seqs", c_int32),
        ("seq_slot_of_seq", c_void_p), ("seq_slot_len", c_int32),
        ("seq_positions", c_void_p), ("num_protected", c_void_p),
        ("evicted_blocks_per_seq", c_void_p), ("context_lens", c_void_p),
        ("hanging_token_count", c_void_p), ("evicted_kv_offsets", c_void_p),
        ("total_slots", c_int64),
        ("use_average", c_int32), ("num_sinks", c_int32),
        ("bias", c_void_p), ("position_bins", c_void_p), ("num_bins", c_int32),
        ("bias_weight", c_float), ("mode", c_int32), ("null_value", c_int32), ("lean", c_int32),
        ("max_evicted_blocks_hint", c_int32),
        ("block_tables", c_void_p), ("seq_index_of_slot", c_void_p),
        ("max_num_seqs", c_int32), ("block_tables_width", c_int32),
        ("schedule_path", c_int32), ("sample_stride", c_int32), ("fallback_grid", c_int32),
        ("uniform_evict", c_int32),
        ("eli_dirty_map", c_void_p),
        ("harvest_buf", c_void_p), ("harvest", c_int32), ("harvest_widen", c_float),
        ("harvest_position_delta", c_int32),
        ("evicted_logical_indices", c_void_p), ("evicted_kv_count", c_void_p),
        ("evicted_block_count", c_void_p),
        ("total_slots_dev", c_void_p), ("flag_mirror", c_void_p), ("flag_ticket", c_uint32),
    ]


class KvcAttentionParams(ctypes.Structure):
    """mirror of ``kvc_attention_params`` (include/kvc_mi355x.h)"""
    _fields_ = [
        ("out", c_void_p), ("kv_metric_out", c_void_p), ("exp_sums", c_void_p),
        ("max_logits", c_void_p), ("tmp_out", c_void_p), ("tmp_kv_metric_out", c_void_p),
        ("fused_metrics", c_void_p),
        ("query", c_void_p), ("key_cache", c_void_p), ("value_cache", c_void_p),
        ("block_tables", c_void_p), ("context_lens", c_void_p), ("kv_position", c_void_p),
        ("last_position", c_void_p), ("kv_metric_buffer_len", c_void_p),
        ("alibi_slopes", c_void_p),
        ("q_stride", c_int64), ("kv_block_stride", c_int64),
        ("scale", c_float), ("k_scale", c_float), ("v_scale", c_float),
        ("num_seqs", c_int32), ("num_heads", c_int32), ("num_kv_heads", c_int32),
        ("head_size", c_int32), ("block_size", c_int32),
        ("max_num_blocks_per_seq", c_int32), ("max_context_len", c_int32),
        ("dtype", c_int32), ("kv_cache_dtype", c_int32), ("record_kv_metrics", c_int32),
        ("fused_use_l2", c_int32), ("schedule", c_int32),
        ("harvest_buf", c_void_p), ("harvest_seq_slot", c_void_p), ("harvest_seq_positions", c_void_p),
        ("harvest_num_protected", c_void_p), ("harvest_num_seqs", c_int32), ("harvest_layer", c_int32),
        ("harvest_num_layers", c_int32), ("harvest_num_sinks", c_int32),
        ("block_layout", c_int32),
    ]


# name -> (restype, argtypes); every symbol declared in include/kvc_mi355x.h
SYMBOLS = {
    "kvc_abi_version": (c_int32, []),
    "kvc_last_error": (c_char_p, []),
    "kvc_count_block_evictions": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                            c_int64, c_int32, c_int32, c_void_p]),
    "kvc_schedule_t1_cache_moves": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                              c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "kvc_cache_moves_dirty_map_bytes": (c_size_t, [c_int64, c_int32]),
    "kvc_cache_moves_plan_bytes": (c_size_t, []),
    "kvc_schedule_t1_cache_moves_ex": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                                 c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                                 c_int32, c_int32, c_int32, c_int32, c_void_p, c_size_t,
                                                 c_void_p, c_void_p]),
    "kvc_execute_cache_moves_planned": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                  c_void_p, c_void_p, c_int32, c_int64, c_int32,
                                                  c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "kvc_execute_cache_moves_workspace_bytes": (c_size_t, [c_int32, c_int64]),
    "kvc_execute_cache_moves": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32,
                                          c_int32, c_int32, c_void_p, c_size_t, c_void_p]),
    "kvc_execute_cache_moves_plan": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int64,
                                               c_int32, c_int32, c_int32, c_int32, c_void_p,
                                               c_size_t, c_void_p]),
    "kvc_execute_cache_moves_apply": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_void_p, c_int32, c_int64, c_int32,
                                                c_int32, c_int32, c_int32, c_void_p, c_size_t,
                                                c_void_p]),
    "kvc_execute_cache_moves_slot_major_plan": (c_int32, [c_void_p, c_int32, c_void_p, c_size_t, c_void_p]),
    "kvc_execute_cache_moves_slot_major": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                     c_void_p, c_void_p, c_int32, c_int64, c_int32,
                                                     c_int32, c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "kvc_schedule_evictions_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32, c_int32]),
    "kvc_schedule_evictions": (c_int32, [ctypes.POINTER(KvcScheduleParams), c_void_p, c_size_t,
                                         c_void_p]),
    "kvc_schedule_batch_summary": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32,
                                             c_void_p, c_size_t, c_void_p]),
    "kvc_schedule_batch_summary_wait": (c_int32, [c_void_p]),
    "kvc_schedule_batch_summary_deferred": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int64,
                                                      c_void_p, c_int64, c_void_p]),
    "kvc_schedule_batch_summary_ticket": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
    "kvc_schedule_evictions_uses_small_eviction_schedule": (c_int32, [ctypes.POINTER(KvcScheduleParams)]),
    "kvc_schedule_evictions_plan": (c_int32, [ctypes.POINTER(KvcScheduleParams)]),
    "kvc_schedule_evictions_plan_reason": (c_int32, [ctypes.POINTER(KvcScheduleParams)]),
    "kvc_schedule_evictions_uses_block_tables": (c_int32, [ctypes.POINTER(KvcScheduleParams)]),
    "kvc_schedule_evictions_fallback_offset": (c_size_t, [c_int64, c_int32, c_int32, c_int32]),
    "kvc_harvest_buffer_bytes": (c_size_t, [c_int32, c_int32]),
    "kvc_harvest_pivot_bytes": (c_size_t, [c_int32]),
    "kvc_pivot_memory_eligible": (c_int32, [ctypes.POINTER(KvcScheduleParams)]),
    "kvc_harvest_eligible": (c_int32, [ctypes.POINTER(KvcScheduleParams), c_int32]),
    "kvc_attention_harvest_eligible": (c_int32, [ctypes.POINTER(KvcScheduleParams)]),
    "kvc_attention_harvest_begin": (c_int32, [ctypes.POINTER(KvcScheduleParams), c_void_p]),
    "kvc_aggregate_decode_harvest": (c_int32, [ctypes.POINTER(KvcScheduleParams), c_void_p, c_int32, c_int32,
                                               c_int32, c_void_p]),
    "kvc_aggregate_decode": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                       c_void_p]),
    "kvc_aggregate_prefill": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                        c_void_p]),
    "kvc_prefill_metric_epilogue_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "kvc_prefill_metric_epilogue": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                              c_int32, c_int32, c_int32, c_int32, c_int32,
                                              c_void_p, c_size_t, c_void_p]),
    "kvc_reshape_and_cache": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                        c_int32, c_int64, c_int64, c_void_p]),
    "kvc_reshape_and_cache_fp8": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                            c_int32, c_int32, c_int64, c_int64, c_float, c_float,
                                            c_void_p]),
    "kvc_reshape_and_cache_layout": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                               c_int32, c_int64, c_int64, c_int32, c_void_p]),
    "kvc_reshape_and_cache_fp8_layout": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                   c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                                   c_int32, c_int32, c_int64, c_int64, c_float, c_float,
                                                   c_int32, c_void_p]),
    "kvc_free_compressed_blocks_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "kvc_free_compressed_blocks": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                             c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                             c_size_t, c_void_p]),
    "kvc_append_slots_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int64]),
    "kvc_append_slots": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                   c_int32, c_int32, c_int64, c_int32, c_int32, c_void_p, c_size_t, c_void_p]),
    "kvc_add_sequence_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int64]),
    "kvc_add_sequence": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int64, c_int32,
                                   c_int32, c_int32, c_void_p, c_size_t, c_void_p]),
    "kvc_prefill_metric_fused_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "kvc_prefill_metric_fused": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                           c_int32, c_int32, c_int32, c_int32, c_int32, c_int64, c_int64,
                                           c_float, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                           c_size_t, c_void_p]),
    "kvc_paged_attention_decode_uses_partitions": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32,
                                                             c_int32]),
    "kvc_paged_attention_decode_uses_partitions_in": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32,
                                                                c_int32, c_int32]),
    "kvc_paged_attention_decode": (c_int32, [ctypes.POINTER(KvcAttentionParams), c_void_p]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load the HIP library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime; it must be the one already loaded when the kernels'
    # library resolves libamdhip64 (a second runtime instance sees no device)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run vllm_kvcompress_amd/csrc/build.sh or __graft_entry__.build()). "
            "There is no CPU fallback for the KV-Compress ops.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    # kvc_schedule_params / kvc_attention_params are mirrored field by field above: a library of another ABI version
    # (an old build picked up through KVC_MI355X_LIB) would read pointers at the wrong offsets
    if lib.kvc_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {lib.kvc_abi_version()}, this package mirrors version "
                          f"{ABI_VERSION} (include/kvc_mi355x.h): rebuild it (vllm_kvcompress_amd/csrc/build.sh)")
    _lib = lib
    return lib


def check(rc: int) -> None:
    """Error convention of the reference ops: TORCH_CHECK -> RuntimeError."""
    if rc != 0:
        msg = load().kvc_last_error()
        raise RuntimeError(msg.decode() if msg else f"kvc error {rc}")
