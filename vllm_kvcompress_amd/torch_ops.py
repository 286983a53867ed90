"""Register the HIP implementations under the reference's dispatcher names.

The fork's Python calls ``torch.ops._C_kvc_ops.count_block_evictions`` etc.
(``vllm/_custom_ops.py:1074, 1169, 1247, 649``); its C++ registers them with
``TORCH_LIBRARY_EXPAND(_C_kvc_ops)`` / ``(_C_cache_ops)`` (``csrc/torch_bindings.cpp:372-418,
353-362``).  ``register()`` defines the same schemas (or overrides the CUDA-key kernels if
the namespace already exists) and binds them to libkvc_mi355x.so.  The decode attention
with metric output lives in the fork's main library ``_C`` (``csrc/torch_bindings.cpp:52-80``,
called from ``vllm/_custom_ops.py:156, 192``).
"""
from __future__ import annotations

import torch

from . import _custom_ops as ops

_KEEP = []          # Library objects must stay alive
_REGISTERED = False

_KVC_SCHEMAS = {
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
    try:
        lib = torch.library.Library(ns, "DEF")
        for name, schema in schemas.items():
            lib.define(name + schema)
    except RuntimeError:
        # namespace already defined (e.g. the fork's own extension is loaded): override
        lib = torch.library.Library(ns, "IMPL")
    for name, fn in impls.items():
        lib.impl(name, fn, "CUDA")
    _KEEP.append(lib)


def register() -> None:
    global _REGISTERED
    if _REGISTERED:
        return
    _bind("_C_kvc_ops", _KVC_SCHEMAS, {
        "count_block_evictions": _count_block_evictions,
        "schedule_t1_cache_moves": _schedule_t1_cache_moves,
        "execute_cache_moves": _execute_cache_moves,
    })
    _bind("_C_cache_ops", _CACHE_SCHEMAS, {"kvcompress_reshape_and_cache": _reshape_and_cache})
    _bind("_C", _ATTN_SCHEMAS, {
        "kvcompress_paged_attention_v1": ops.paged_attention_kvc_v1,
        "kvcompress_paged_attention_v2": ops.paged_attention_kvc_v2,
    })
    _REGISTERED = True
