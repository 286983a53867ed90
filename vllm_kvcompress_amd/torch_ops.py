"""Register the HIP implementations under the reference's dispatcher names.

The fork's Python calls ``torch.ops._C_kvc_ops.count_block_evictions`` etc.
(``vllm/_custom_ops.py:1074, 1169, 1247, 649``); its C++ registers them with
``TORCH_LIBRARY_EXPAND(_C_kvc_ops)`` / ``(_C_cache_ops)`` (``csrc/torch_bindings.cpp:372-418,
353-362``).  ``register()`` makes the same names resolve here: by loading the compiled binding
``libkvc_torch.so`` (``csrc/kvc_torch_binding.cpp``, C++ kernels registered with
TORCH_LIBRARY_IMPL that call the C ABI of libkvc_mi355x.so directly), or -- when that library
has not been built, or on request -- by defining the same schemas from Python with Python
callables that go through ``_custom_ops`` and ctypes.  The decode attention
with metric output lives in the fork's main library ``_C`` (``csrc/torch_bindings.cpp:52-80``,
called from ``vllm/_custom_ops.py:156, 192``).
"""
from __future__ import annotations

import os

import torch

from . import _custom_ops as ops

_KEEP = []          # Library objects must stay alive
_REGISTERED = ""          # "" | "compiled" | "python"

_KVC_SCHEMAS = {
    # the V1 pair (csrc/torch_bindings.cpp:374-394): dead in the reference (scheduler.py:285
    # ``if False:``), registered for surface completeness, raise when called
    "schedule_cache_evictions":
        "(Tensor(a!) evicted_kv_indices, Tensor(b!) evicted_logical_indices, "
        "Tensor(c!) evicted_kv_count, Tensor(d!) remaining_kv_count, Tensor evicted_kv_offsets, "
        "Tensor sorted_indices, Tensor seq_block_offsets, Tensor layer_by_block, "
        "Tensor head_by_block, Tensor virtual_block_num_by_block, Tensor evicted_blocks_per_seq, "
        "Tensor context_lens, Tensor hanging_token_count, Tensor kv_position, "
        "Tensor last_position, Tensor protected_window_size, int block_size, "
        "bool evict_evenly_per_layer, Tensor? control_layers, int max_evicted_kv, "
        "int null_eviction_index, bool truncate) -> ()",
    "truncate_cache_evictions":
        "(Tensor(a!) evicted_kv_indices, Tensor(b!) evicted_logical_indices, "
        "Tensor(c!) evicted_kv_count, Tensor evicted_kv_offsets, Tensor hanging_token_count, "
        "int block_size, int max_evicted_kv, int null_eviction_index) -> ()",
    "count_block_evictions":
        "(Tensor(a!) evicted_block_count, Tensor(b!) evicted_logical_indices, "
        "Tensor evicted_kv_offsets, Tensor hanging_token_count, int block_size, "
        "int null_value) -> ()",
    "schedule_t1_cache_moves":
        "(Tensor(a!) cache_moves_idx, Tensor(b!) cache_moves_count, "
        "Tensor evicted_logical_indices, Tensor evicted_kv_count, Tensor evicted_kv_offsets, "
        "Tensor block_tables, Tensor context_lens, int block_size) -> ()",
    "execute_cache_moves":
        "(Tensor(a!) k_cache, Tensor(b!) v_cache, Tensor(c!) kv_metrics, Tensor(d!) kv_position, "
        "Tensor cache_moves_idx, Tensor cache_moves_count, Tensor evicted_kv_offsets, "
        "int blocks_per_head, int threads_per_head) -> ()",
}
_CACHE_SCHEMAS = {
    "kvcompress_reshape_and_cache":
        "(Tensor key, Tensor value, Tensor(a!) key_cache, Tensor(b!) value_cache, "
        "Tensor(c!) kv_metrics, Tensor slot_mapping, Tensor kv_metric_head_bias, "
        "str kv_cache_dtype, float k_scale, float v_scale) -> ()",
}


# library `_C` (csrc/torch_bindings.cpp:52-80): the decode attention with metric output
_ATTN_SCHEMAS = {
    "kvcompress_paged_attention_v1":
        "(Tensor(a!) out, Tensor(b!) kv_metric_out, Tensor query, Tensor key_cache, "
        "Tensor value_cache, int num_kv_heads, float scale, Tensor block_tables, "
        "Tensor context_lens, Tensor kv_position, Tensor last_position, "
        "Tensor kv_metric_buffer_len, int block_size, int max_context_len, "
        "Tensor? alibi_slopes, str kv_cache_dtype, float k_scale, float v_scale, "
        "bool record_kv_metrics) -> ()",
    "kvcompress_paged_attention_v2":
        "(Tensor(a!) out, Tensor(b!) kv_metric_out, Tensor exp_sums, Tensor max_logits, "
        "Tensor tmp_out, Tensor tmp_kv_metric_out, Tensor query, Tensor key_cache, "
        "Tensor value_cache, int num_kv_heads, float scale, Tensor block_tables, "
        "Tensor context_lens, Tensor kv_position, Tensor last_position, "
        "Tensor kv_metric_buffer_len, int block_size, int max_context_len, "
        "Tensor? alibi_slopes, str kv_cache_dtype, float k_scale, float v_scale, "
        "bool record_kv_metrics) -> ()",
}


def _v1_dead(*args):
    raise RuntimeError(ops.V1_DEAD_MESSAGE)


def _count_block_evictions(ebc, eli, offs, hang, block_size, null_value):
    ops.count_block_evictions(ebc, eli, offs, hang, block_size, null_value)


def _schedule_t1_cache_moves(cmi, cmc, eli, ekc, offs, bt, ctx, block_size):
    # the bare op does not clear the workspace (the Python wrapper does, _custom_ops.py:1168)
    ops._schedule_t1_cache_moves(cmi, cmc, eli, ekc, offs, bt, ctx, block_size, zero_fill=False)


def _execute_cache_moves(k, v, m, p, cmi, cmc, offs, blocks_per_head, threads_per_head):
    ops.execute_cache_moves(k, v, m, p, cmi, cmc, offs, blocks_per_head, threads_per_head)


def _reshape_and_cache(key, value, kc, vc, met, slots, bias, dtype, k_scale, v_scale):
    ops.reshape_and_cache_kvc(key, value, kc, vc, met, slots, bias, dtype, k_scale, v_scale)


def _bind(ns, schemas, impls):
    # FRAGMENT coexists with a TORCH_LIBRARY of the same namespace loaded before or after (the
    # fork's own extension); a schema is defined only if nobody has defined the op yet
    lib = torch.library.Library(ns, "FRAGMENT")
    for name, schema in schemas.items():
        if not _has_op(ns, name):
            lib.define(name + schema)
    for name, fn in impls.items():
        lib.impl(name, fn, "CUDA")
    _KEEP.append(lib)


def _has_op(ns: str, name: str) -> bool:
    try:
        torch._C._dispatch_find_schema_or_throw(f"{ns}::{name}", "")
        return True
    except RuntimeError:
        return False


COMPILED_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libkvc_torch.so")


def register(binding: str = "auto") -> str:
    """Make ``torch.ops._C_kvc_ops.*``, ``torch.ops._C_cache_ops.kvcompress_reshape_and_cache`` and
    ``torch.ops._C.kvcompress_paged_attention_v1/_v2`` resolve to the MI355X kernels.

    ``binding``: "compiled" = load ``libkvc_torch.so`` (csrc/kvc_torch_binding.cpp: C++ kernels
    registered with TORCH_LIBRARY_IMPL, no Python frame between the dispatcher and the C ABI);
    "python" = Python callables going through ``_custom_ops`` / ctypes; "auto" = compiled when
    the library has been built, else python.  Returns the binding in effect.  Import order
    next to the fork's own extension does not matter (FRAGMENT definitions on both sides); a
    process uses ONE binding -- registering a second kernel for the same op and key is an error
    in the dispatcher."""
    global _REGISTERED
    if _REGISTERED:
        return _REGISTERED
    if binding not in ("auto", "compiled", "python"):
        raise ValueError(binding)
    if binding == "compiled" or (binding == "auto" and os.path.exists(COMPILED_LIB)):
        from . import _lib
        _lib.load()                     # libkvc_mi355x.so first (and torch's HIP runtime before it)
        torch.ops.load_library(COMPILED_LIB)
        _REGISTERED = "compiled"
        # the C++ kernels keep their own copy of the attention schedule
        torch.ops._kvc_mi355x.set_attention_schedule(int(ops._ATTENTION_SCHEDULE))
        return _REGISTERED
    _bind("_C_kvc_ops", _KVC_SCHEMAS, {
        "schedule_cache_evictions": _v1_dead,
        "truncate_cache_evictions": _v1_dead,
        "count_block_evictions": _count_block_evictions,
        "schedule_t1_cache_moves": _schedule_t1_cache_moves,
        "execute_cache_moves": _execute_cache_moves,
    })
    _bind("_C_cache_ops", _CACHE_SCHEMAS, {"kvcompress_reshape_and_cache": _reshape_and_cache})
    _bind("_C", _ATTN_SCHEMAS, {
        "kvcompress_paged_attention_v1": ops.paged_attention_kvc_v1,
        "kvcompress_paged_attention_v2": ops.paged_attention_kvc_v2,
    })
    _REGISTERED = "python"
    return _REGISTERED
