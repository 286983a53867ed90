"""An engine loop whose KV-Compress state never leaves the device: prefill -> decode -> compress, every transition a
device op of this package (SURVEY.md 8(f) F2's purpose: a self-contained continual harness).

What the fork keeps on the GPU -- ``BlockState.block_tables`` / ``context_lens`` (vllm/kvcompress/block.py:95-126),
``ParallelBlockAllocator.free_mask`` (block_manager.py:76-118), the ``CompressionMetrics`` stores (metrics.py:220-275)
and the unified KV cache -- lives here as device tensors from the first token on; nothing is mirrored in NumPy.  The
order of one iteration is ``LLMEngine.step``'s (llm_engine.py:1556-1634): compression first (schedule_evictions ->
schedule_cache_moves -> execute_cache_moves -> free_compressed_blocks), then the scheduler's append of one slot per
head, then the model's cache write and attention (here: the caller's K/V rows and softmax weights), then
``aggregate_decode``.  The host sees what the fork's host sees: the small ``context_lens`` table for the eviction
policy (scheduler.py:100-181 reads block counts per sequence) and the allocation counts.

Test / tool infrastructure (tests/test_gpu_engine_from_prefill.py drives it next to the oracle's NumPy restatements
of the same transitions; tools/soak_from_prefill.py soaks it)."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .. import _custom_ops as ops
from ..kvcompress.block_state import add_sequence, append_slots, free_compressed_blocks
from ..kvcompress.metrics import CompressionMetrics
from . import synth
from .device import split_kv_cache


class DeviceEngine:
    def __init__(self, *, num_layers: int, num_kv_heads: int, head_size: int, block_size: int, num_blocks: int,
                 max_num_seqs: int, max_blocks_per_head: int, num_queries_per_kv: int = 1, dtype=torch.float16,
                 device="cuda:0", mode: str = "per_sequence", protected_window: int = 32,
                 max_cache_tokens: int = -1):
        L, H, bs = num_layers, num_kv_heads, block_size
        self.L, self.H, self.hd, self.bs, self.NB = L, H, head_size, bs, num_blocks
        self.dev = torch.device(device)
        self.protected, self.cap = int(protected_window), int(max_cache_tokens)
        self.cm = CompressionMetrics(bs, L, H, num_queries_per_kv, 10 ** 9, None, 0.0, device=device)
        self.cm.schedule_mode = mode
        self.cm.init_kv_metadata(num_blocks)
        self.cm.metrics.zero_()
        self.cm.temp_metrics.zero_()
        self.block_tables = torch.zeros((L, max_num_seqs, H, max_blocks_per_head), dtype=torch.int32, device=self.dev)
        self.context_lens = torch.zeros((L, max_num_seqs, H), dtype=torch.int32, device=self.dev)
        self.free_mask = torch.ones((num_blocks,), dtype=torch.bool, device=self.dev)
        self.kv_cache = torch.zeros((2, num_blocks, bs * head_size), dtype=dtype, device=self.dev)
        self.k_cache, self.v_cache = split_kv_cache(self.kv_cache, head_size)
        self.bias = torch.zeros((H,), dtype=torch.float32, device=self.dev)
        self.slots: List[int] = []                      # batch slots in use, ascending
        self.seq_len = {}                               # slot -> tokens of the sequence (the next one not cached yet)
        # the scheduler's persistent move workspace (reference scheduler.py:74-86), sized on first use
        self._moves: Optional[torch.Tensor] = None
        self.last = {}

    # ---- prefill -----------------------------------------------------------------------------------------------------
    def add_sequence(self, slot: int, key: torch.Tensor, value: torch.Tensor,
                     prefill_metrics: Optional[torch.Tensor] = None) -> int:
        """``key`` / ``value`` [L, T, H, hd]: the prompt's K/V rows per layer; ``prefill_metrics`` [L, T, H * qpk]
        (what the prefill attention hands to ``aggregate_prefill``, metrics.py:396-427).  Returns the blocks allocated."""
        T = int(key.shape[1])
        n, sm = add_sequence(self.block_tables, self.context_lens, slot, T, self.free_mask, self.cm, self.bs)
        for l in range(self.L):
            ops.reshape_and_cache_kvc(key[l], value[l], self.k_cache, self.v_cache, self.cm.metrics, sm[l].reshape(-1),
                                      self.bias, "auto", 1.0, 1.0)
            if prefill_metrics is not None:
                self.cm.aggregate_prefill(prefill_metrics[l], sm[l])
        self.slots = sorted(self.slots + [slot])
        self.seq_len[slot] = T + 1                       # (the token sampled from the prompt, not cached yet)
        self.last["slot_mapping"] = sm
        return n

    # ---- one iteration -------------------------------------------------------------------------------------------------
    def compress(self) -> Optional[dict]:
        """the iteration's compression (llm_engine.py:1556-1570), for every resident sequence over its cap"""
        if self.cap < 0 or not self.slots:
            return None
        L, H, bs = self.L, self.H, self.bs
        slots = list(self.slots)
        B = len(slots)
        ctx_h = self.context_lens[:, slots].cpu().numpy()                 # the policy's view (small)
        evicted = [synth.evict_block_count(context_lens_lh=ctx_h[:, b, :], seq_len=self.seq_len[s], block_size=bs,
                                           protected_window_size=self.protected, max_cache_tokens=self.cap)
                   for b, s in enumerate(slots)]
        if not any(evicted):
            return None
        ctx = self.context_lens[:, slots].contiguous()                    # scheduler.py:262-280, as torch ops on device
        bt = self.block_tables[:, slots].contiguous()
        rem = ctx % bs
        hang = torch.where(rem == 0, torch.full_like(rem, bs), rem).transpose(0, 1).contiguous()
        padded = ((ctx.transpose(0, 1) + bs - 1) // bs * bs).flatten().cumsum(0)
        offs = torch.cat([torch.zeros_like(padded[:1]), padded[:-1]]).reshape(B, L, H).to(torch.int32)
        pos_t = torch.tensor([self.seq_len[s] for s in slots], dtype=torch.int, device=self.dev) - 1
        k_t = torch.tensor(evicted, dtype=torch.int, device=self.dev)
        eli, ekc, ebc = self.cm.schedule_evictions(slots, pos_t, k_t, ctx, hang, offs, tuple([self.protected] * B))
        N = int(eli.numel())
        if self._moves is None or self._moves.shape[0] < N:
            self._moves = torch.empty((max(N, 1) * 2, 2), dtype=torch.int32, device=self.dev)
        cmc = torch.empty((B, L, H), dtype=torch.int32, device=self.dev)
        ops.schedule_cache_moves(self._moves, cmc, eli, ekc, offs, bt, ctx, bs)
        ops.execute_cache_moves(self.k_cache, self.v_cache, self.cm.metrics, self.cm.token_positions, self._moves, cmc,
                                offs, 1, 16)
        freed = free_compressed_blocks(self.block_tables, self.context_lens, slots, ebc, self.cm.seq_index_by_block, bs,
                                       self.free_mask, max_freed=sum(evicted))
        out = dict(evicted=evicted, eli=eli, ekc=ekc, ebc=ebc, cmi=self._moves[:N], cmc=cmc, freed=freed, offs=offs,
                   slots=slots, N=N)
        self.last["compress"] = out
        return out

    def decode(self, key: torch.Tensor, value: torch.Tensor, temp_metrics: Optional[torch.Tensor] = None) -> int:
        """one decode step for every resident sequence: ``key`` / ``value`` [L, B, H, hd] rows of the token sampled
        last (batch order = ascending slots); ``temp_metrics`` [NB, bs, qpk]: the step's softmax weights as the
        attention would leave them (written into the store's scratch, then ``aggregate_decode``).  Returns the newly
        allocated blocks."""
        slots = list(self.slots)
        B = len(slots)
        last_pos = [self.seq_len[s] - 1 for s in slots]
        n = append_slots(self.block_tables, self.context_lens, slots, last_pos, self.free_mask, self.cm, self.bs,
                         write_token_position=True)
        # get_decode_slot_mapping (block.py:305-320): torch ops on device, like the reference
        c1 = (self.context_lens[:, slots] - 1).long()                                   # [L, B, H]
        blk = self.block_tables[:, slots].gather(3, (c1 // self.bs).unsqueeze(-1)).squeeze(-1).long()
        slot_map = blk * self.bs + c1 % self.bs
        for l in range(self.L):
            ops.reshape_and_cache_kvc(key[l], value[l], self.k_cache, self.v_cache, self.cm.metrics,
                                      slot_map[l].reshape(-1), self.bias, "auto", 1.0, 1.0)
        if temp_metrics is not None:
            self.cm.clear_temp_metrics()
            self.cm.temp_metrics.copy_(temp_metrics)
            self.cm.aggregate_decode()
        for s in slots:
            self.seq_len[s] += 1
        self.last["decode_slot_mapping"] = slot_map
        return n

    def remove_sequence(self, slot: int) -> None:
        """``_remove_sequence`` (block_manager.py:224-239): the sequence's blocks go back to the free list"""
        ctx = self.context_lens[:, slot]                                                 # [L, H]
        nblk = (ctx + self.bs - 1) // self.bs
        M = self.block_tables.shape[3]
        live = torch.arange(M, device=self.dev)[None, None, :] < nblk[..., None]
        blocks = self.block_tables[:, slot][live].long()
        self.free_mask[blocks] = True
        self.cm.remove_metadata(blocks)
        self.context_lens[:, slot] = 0
        self.slots.remove(slot)
        del self.seq_len[slot]
        self.cm.forget_pivots()
