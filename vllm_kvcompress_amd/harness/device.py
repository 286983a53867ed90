"""Put a synthetic :class:`PagedState` on a HIP device and drive the hot path through
the reference-shaped op surface (used by tests, ``bench.py`` and ``smoke()``)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from .. import _custom_ops as ops
from ..kvcompress.metrics import CompressionMetrics
from .synth import PagedState


@dataclass
class DeviceState:
    cm: CompressionMetrics
    context_lens: torch.Tensor          # [L,B,H] i32
    block_tables: torch.Tensor          # [L,B,H,M] i32
    hanging_token_count: torch.Tensor   # [B,L,H] i32
    evicted_kv_offsets: torch.Tensor    # [B,L,H] i32
    seq_positions: torch.Tensor         # [B] i32
    total_slots: int


def upload(st: PagedState, device="cuda:0", num_queries_per_kv: int = 1, *, use_average=False,
           num_sinks=0, bias=None, position_bins=None, bias_weight=0.0,
           mode="reference") -> DeviceState:
    dev = torch.device(device)
    cm = CompressionMetrics(st.block_size, st.num_layers, st.num_kv_heads, num_queries_per_kv,
                            10 ** 9, None, float(bias_weight), device=device,
                            use_average=use_average, num_attention_sinks=num_sinks)
    cm.init_kv_metadata(st.num_blocks)
    cm.metrics.copy_(torch.from_numpy(st.metrics))
    cm.token_positions.copy_(torch.from_numpy(st.token_positions))
    cm.seq_index_by_block.copy_(torch.from_numpy(st.seq_index_by_block))
    cm.layer_index_by_block.copy_(torch.from_numpy(st.layer_index_by_block))
    cm.head_index_by_block.copy_(torch.from_numpy(st.head_index_by_block))
    cm.logical_block_num_by_block.copy_(torch.from_numpy(st.logical_block_num_by_block))
    if bias is not None:
        cm.kv_metric_head_bias.bias = torch.from_numpy(np.ascontiguousarray(bias)).to(dev)
        cm.kv_metric_head_bias.position_bins = torch.from_numpy(
            np.ascontiguousarray(position_bins)).to(dev)
        cm._has_bias = True
    cm.schedule_mode = mode
    return DeviceState(
        cm=cm,
        context_lens=torch.from_numpy(st.context_lens).to(dev),
        block_tables=torch.from_numpy(st.block_tables).to(dev),
        hanging_token_count=torch.from_numpy(st.hanging_token_count).to(dev),
        evicted_kv_offsets=torch.from_numpy(st.evicted_kv_offsets).to(dev),
        seq_positions=torch.from_numpy(st.seq_positions).to(dev),
        total_slots=st.total_slots,
    )


def schedule(ds: DeviceState, st: PagedState, evicted_blocks, move_rows: Optional[int] = None,
             uniform_evict: bool = False):
    """A3 -> A5.  Returns (eli, ekc, ebc, cache_moves_idx, cache_moves_count)."""
    eli, ekc, ebc = ds.cm.schedule_evictions(
        list(st.seq_indices), ds.seq_positions,
        [int(x) for x in np.asarray(evicted_blocks).reshape(-1)],     # a host list, like the reference scheduler's
        ds.context_lens, ds.hanging_token_count, ds.evicted_kv_offsets, list(st.protected),
        uniform_evict=uniform_evict, total_slots=ds.total_slots)
    rows = ds.total_slots if move_rows is None else move_rows
    cmi = torch.full((rows, 2), 77, dtype=torch.int32, device=ds.cm.device)
    cmc = torch.empty_like(ekc)
    ops.schedule_cache_moves(cmi, cmc, eli, ekc, ds.evicted_kv_offsets, ds.block_tables,
                             ds.context_lens, st.block_size)
    return eli, ekc, ebc, cmi, cmc


def split_kv_cache(kv_cache: torch.Tensor, head_size: int):
    """KVCAttention.split_kv_cache (reference vllm/attention/ops/paged_attn.py:272-284)"""
    x = 16 // kv_cache.element_size()
    num_blocks = kv_cache.shape[1]
    key_cache = kv_cache[0].view(num_blocks, head_size // x, -1, x)
    value_cache = kv_cache[1].view(num_blocks, head_size, -1)
    return key_cache, value_cache
