"""``CompressionScheduler``: the host glue of one compression iteration, on MI355X.

Mirror of ``vllm/kvcompress/scheduler.py`` (``_schedule_seq_evictions`` :100-181,
``_schedule_compression`` :184-560, ``schedule_compression`` :565-574) for the hot path
only.  The reference's version is welded to engine objects (``Sequence``,
``BlockSpaceManagerKVC``, ``SamplingParams``); this one takes the same information as plain
data (:class:`SeqCompressionRequest` + the ``BlockState`` tensors) so that a maintainer can
call it from the fork's ``_schedule_compression`` after extracting those fields, or use it
stand-alone as the multi-iteration harness does.

Per call it does exactly what the reference does, in the reference's order:
select sequences (most stale first, stop at ``max_kv_per_compression``), order them by batch
slot, batch views of ``context_lens`` / ``block_tables``, hanging tokens, ``evicted_kv_offsets``
(exclusive cumsum in (b,l,h) order), ``CompressionMetrics.schedule_evictions``,
``schedule_cache_moves`` into the persistent move workspace, then (block-state side)
``free_compressed_blocks``.  The moves themselves are executed by the caller with
``execute_cache_moves`` (the worker does that in the reference, cache_engine.py:139-151).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from .. import _custom_ops as ops
from .block_state import free_compressed_blocks
from .metrics import CompressionMetrics

MAX_INT = 2147483000


@dataclass
class CacheMoves:
    """reference scheduler.py:19-31"""
    index: torch.Tensor       # [max_kv_per_compression, 2] i32 (dst, src) physical slots
    count: torch.Tensor       # [B, L, H] i32
    offsets: torch.Tensor     # [B, L, H] i32


@dataclass
class CompressionOutputs:
    """reference scheduler.py:34-41 (+ the freed physical blocks of the block-state update)"""
    cache_moves: CacheMoves
    freed_block_count: Dict[int, torch.Tensor]    # seq_id -> [L, H] i32
    seq_ids: List[int]
    slot_indices: List[int]
    freed_blocks: Optional[torch.Tensor] = None


@dataclass
class SeqCompressionRequest:
    """what ``_schedule_compression`` reads from (Sequence, SamplingParams, block manager)"""
    seq_id: int
    slot_index: int                    # block_manager.get_slot_index(seq)
    seq_len: int                       # seq.data.get_len()
    block_count: int                   # block_manager.get_sequence_block_count(seq)
    kv_count: int                      # block_manager.get_sequence_kv_count(seq)
    target_compression_rate: float = 1.0
    max_cache_tokens: int = -1
    protected_window_size: int = 50
    compress_once: bool = False
    compressed: bool = False           # seq.compressed (set by the scheduler)


class CompressionScheduler:
    def __init__(self, block_size: int, num_layers: int, num_kv_heads: int,
                 max_kv_per_compression: int, compression_metrics: CompressionMetrics,
                 device: str = "cuda:0", even_layer_evict: bool = False,
                 compression_interval: int = 1, new_token_limit: int = -1,
                 zero_fill_moves: bool = True) -> None:
        self.block_size = block_size
        self.num_layers = num_layers
        self.num_kv_heads = num_kv_heads
        self.max_kv_per_compression = max_kv_per_compression
        self.compression_metrics = compression_metrics
        self.device = torch.device(device)
        self.even_layer_evict = even_layer_evict
        self.compression_interval = compression_interval
        self.new_token_limit = new_token_limit
        # the reference clears the WHOLE move workspace every call (_custom_ops.py:1168); nothing
        # reads rows outside [offset_g, offset_g + count_g), so an engine may opt out (extension)
        self.zero_fill_moves = zero_fill_moves
        self.iteration_count = 0
        self.new_tokens = 0
        self._iters_since_compression: Dict[int, int] = {}
        self._aggregation_due = False      # schedule_compression(aggregate_decode=True): the step's aggregation has not run yet
        # persistent move workspace (reference scheduler.py:74-86).  Registered with the op surface:
        # the wrapper's per-call fill_(0) of the whole table (vllm/_custom_ops.py:1168) then only
        # clears the rows the previous call wrote -- the table's contents are the same
        self.cache_move_indices = ops.track_move_table(
            torch.empty((max_kv_per_compression, 2), dtype=torch.int32, device=self.device))

    # ---- reference scheduler.py:100-181 ---------------------------------------------------
    def schedule_seq_evictions(self, req: SeqCompressionRequest):
        """Returns (evict_kv_count, evict_block_count) for one sequence."""
        bs = self.block_size
        if req.compress_once and req.compressed:
            return 0, 0
        req.compressed = True
        max_cache_tokens = req.max_cache_tokens
        if max_cache_tokens > 0:
            max_cache_tokens = (max_cache_tokens + bs - 1) // bs * bs
        if req.target_compression_rate < 1.0 and max_cache_tokens > 0:
            raise RuntimeError("both compression_rate and max_cache_tokens "
                               "specified during compression")
        total_kv_heads = self.num_layers * self.num_kv_heads
        if max_cache_tokens >= 0:
            max_cache_blocks = (max_cache_tokens * total_kv_heads + bs - 1) // bs
            evict_block_count = max(0, req.block_count - max_cache_blocks)
        else:
            protected_tokens = (req.protected_window_size + bs - 1) // bs * bs
            compressible = req.seq_len - protected_tokens
            if compressible <= 0:
                return 0, 0
            target_kv = (math.ceil(compressible * total_kv_heads * req.target_compression_rate)
                         + protected_tokens * total_kv_heads)
            evict_kv = max(0, req.kv_count - target_kv)
            evict_block_count = (evict_kv + bs - 1) // bs
        if self.even_layer_evict:
            evict_block_count = evict_block_count // self.num_layers * self.num_layers
        assert evict_block_count <= max(
            req.block_count - (req.protected_window_size + bs - 1) // bs * total_kv_heads, 0)
        return evict_block_count * bs, evict_block_count

    def complete_seqs(self, seq_ids: List[int]) -> None:
        for s in seq_ids:
            self._iters_since_compression.pop(s, None)
        if seq_ids:
            # a finished sequence's batch slot goes to another sequence: pivots remembered per slot are not that one's
            self.compression_metrics.forget_pivots()

    def increment_new_tokens(self, n: int) -> None:
        self.new_tokens += n

    # ---- reference scheduler.py:565-574 ---------------------------------------------------
    def schedule_compression(self, requests: List[SeqCompressionRequest],
                             block_tables: torch.Tensor, context_lens: torch.Tensor,
                             force: bool = False, free_mask: Optional[torch.Tensor] = None,
                             aggregate_decode: bool = False) -> Optional[CompressionOutputs]:
        """``aggregate_decode=True`` (not in the reference's signature): the last decode step's
        ``kv_metrics.aggregate_decode()`` (llm_engine.py:1634) has NOT run yet -- the engine left it to
        this call, which is the next reader of the metrics.  It then runs here exactly once: as
        ``aggregate_decode_and_harvest`` with the batch of the ``schedule_evictions`` right behind it
        (one sweep of the metric store instead of two), or as the plain pass when
        nothing is compressed this iteration.  The sums, and everything computed from them, are the
        same as with the reference's order.  "Exactly once" holds on every way out of this call: if anything
        raises before the aggregation has run (the policy's assertion, an out-of-budget batch, a device fault
        reported by an earlier schedule call), the plain pass runs on the way out, so the step's attention is in
        the store and temp_metrics is cleared before the exception reaches the engine."""
        self._aggregation_due = bool(aggregate_decode)
        try:
            self.iteration_count += 1
            out = None
            if force or (self.iteration_count >= self.compression_interval
                         or (self.new_token_limit > -1 and self.new_tokens > self.new_token_limit)):
                self.iteration_count = 0
                self.new_tokens = 0
                out = self._schedule_compression(requests, block_tables, context_lens, free_mask, aggregate_decode)
        except BaseException as first:
            # an aborted step: the PLAIN pass (no prediction for a next call, no lists) so that the step's attention is
            # in the store and temp_metrics is cleared; a second failure on the way out (a device fault the poll
            # reports, a launch error) is chained to the first, never put in its place
            if self._aggregation_due:
                self._aggregation_due = False
                try:
                    self.compression_metrics.aggregate_decode(predict=False)
                except Exception as second:
                    raise first from second
            raise
        if self._aggregation_due:                       # nothing was compressed this iteration: the normal way out
            self._aggregation_due = False
            self.compression_metrics.aggregate_decode()
        return out

    # ---- reference scheduler.py:184-560 ---------------------------------------------------
    def _schedule_compression(self, requests, block_tables, context_lens, free_mask, aggregate_decode=False):
        bs, L, H = self.block_size, self.num_layers, self.num_kv_heads
        total_kv_count = 0
        chosen: List[SeqCompressionRequest] = []
        evicted_blocks: List[int] = []
        # most stale sequences first (:195-198)
        for _, _, req in sorted(((self._iters_since_compression.get(r.seq_id, 0), r.seq_id, r)
                                 for r in requests), key=lambda x: (x[0], x[1]), reverse=True):
            _, n = self.schedule_seq_evictions(req)
            if n == 0:
                continue
            total_kv_count += req.block_count * bs
            if total_kv_count > self.max_kv_per_compression:       # :211-215
                break
            chosen.append(req)
            evicted_blocks.append(n)
        if not chosen:
            return None                # (schedule_compression runs the plain aggregation on the way out)
        order = sorted(range(len(chosen)), key=lambda i: chosen[i].slot_index)   # :235-238
        chosen = [chosen[i] for i in order]
        evicted_blocks = [evicted_blocks[i] for i in order]
        slots = [r.slot_index for r in chosen]
        B = len(chosen)
        last_token_positions = torch.tensor([r.seq_len - 1 for r in chosen], dtype=torch.int32,
                                            device=self.device)                  # :256-260
        ctx = context_lens[:, slots].contiguous()                                # [L,B,H]
        bt = block_tables[:, slots].contiguous()                                 # [L,B,H,M]
        rem = ctx % bs
        hanging = torch.where(rem == 0, torch.full_like(rem, bs), rem).transpose(0, 1).contiguous()
        per_head = ((ctx.transpose(0, 1) + (bs - 1)) // bs * bs).flatten().cumsum(dim=0)   # :274-280
        total_slots = int(per_head[-1].item())
        offsets = (torch.cat([torch.zeros_like(per_head[:1]), per_head[:-1]])
                   .reshape(B, L, H).type(torch.int32).contiguous())
        protected = [r.protected_window_size for r in chosen]
        if aggregate_decode:
            self._aggregation_due = False            # (the call below aggregates on every path, its own failures included)
            self.compression_metrics.aggregate_decode_and_harvest(slots, last_token_positions, protected, ctx,
                                                                  total_slots=total_slots)
        if total_slots > self.max_kv_per_compression:
            raise RuntimeError("compression batch exceeds max_kv_per_compression")
        eli, ekc, ebc = self.compression_metrics.schedule_evictions(
            slots, last_token_positions, evicted_blocks, ctx, hanging, offsets, protected, total_slots=total_slots)
        cache_moves_count = torch.empty((B, L, H), dtype=torch.int32, device=self.device)
        if self.zero_fill_moves:
            ops.schedule_cache_moves(self.cache_move_indices, cache_moves_count, eli, ekc, offsets,
                                     bt, ctx, bs)                                # :505-523
        else:
            ops._schedule_t1_cache_moves(self.cache_move_indices, cache_moves_count, eli, ekc,
                                         offsets, bt, ctx, bs, zero_fill=False)
        cache_moves = CacheMoves(self.cache_move_indices, cache_moves_count, offsets)
        freed_block_count = {r.seq_id: ebc[i] for i, r in enumerate(chosen)}     # :531-534
        for sid in self._iters_since_compression:                                # :92-96
            self._iters_since_compression[sid] += 1
        for r in chosen:
            self._iters_since_compression[r.seq_id] = 0
        # block-state side (:552-556): free tail blocks, shrink context_lens, drop metadata
        freed = free_compressed_blocks(block_tables, context_lens, slots, ebc.contiguous(),
                                       self.compression_metrics.seq_index_by_block, bs, free_mask)
        return CompressionOutputs(cache_moves, freed_block_count, [r.seq_id for r in chosen], slots,
                                  freed)
