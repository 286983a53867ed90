"""A minimal host-side simulator of the block-state transitions AROUND the hot path, so
that multi-iteration ("continual compression") scenarios can be driven identically through
the oracle and through the HIP path in tests.

It restates, in NumPy, only what the reference's block manager does to the tensors the hot
path reads:

* after a compression: the last ``evicted_block_count`` blocks of every head are freed
  (``seq_index_by_block = -1``), ``context_lens -= evicted_kv_count``
  (vllm/kvcompress/block_manager.py:466-530, block.py:184-210, metrics.py:366-370);
* a decode step appends one KV per head: ``context_lens += 1``; heads that were exactly at a
  block boundary get one new block whose ``token_positions`` row is
  ``last_token_position + arange(bs)`` (block.py:536-569, block_manager.py:269-294); the new
  slot's metric is the head bias (0) (csrc/kvcompress_cache_kernels.cu:55-58).

This is test/bench infrastructure (state bookkeeping on the host), not part of the product
path; the reference's block manager itself is out of scope.
"""
from __future__ import annotations

import numpy as np

from .synth import PagedState, hanging_tokens, kv_offsets


class EngineSim:
    def __init__(self, st: PagedState, seq_lens):
        self.st = st
        self.seq_lens = np.asarray(seq_lens, dtype=np.int64).copy()
        used = st.seq_index_by_block >= 0
        self.free = list(np.nonzero(~used)[0][::-1])          # pop() takes the lowest index

    # --- after schedule + moves + execute --------------------------------------------------
    def apply_compression(self, evicted_kv_count, evicted_block_count):
        st = self.st
        bs = st.block_size
        L, B, H = st.context_lens.shape
        for b in range(B):
            for l in range(L):
                for h in range(H):
                    nfree = int(evicted_block_count[b, l, h])
                    if nfree == 0:
                        continue
                    ctx = int(st.context_lens[l, b, h])
                    nblk = (ctx + bs - 1) // bs
                    blocks = st.block_tables[l, b, h, nblk - nfree:nblk]
                    st.seq_index_by_block[blocks] = -1
                    self.free.extend(int(x) for x in blocks[::-1])
                    st.context_lens[l, b, h] = ctx - int(evicted_kv_count[b, l, h])
        self._refresh()

    # --- one decode step: every head caches the KV of the previously sampled token ----------
    def append_token(self):
        st = self.st
        bs = st.block_size
        L, B, H = st.context_lens.shape
        need = 0
        for b in range(B):
            last_pos = int(self.seq_lens[b]) - 1               # position of the token being cached
            for l in range(L):
                for h in range(H):
                    ctx = int(st.context_lens[l, b, h])
                    if ctx % bs == 0:                          # needs a fresh block
                        blk = self.free.pop()
                        need += 1
                        j = ctx // bs
                        if j >= st.block_tables.shape[3]:
                            grow = np.zeros(st.block_tables.shape[:3] + (j + 1 - st.block_tables.shape[3],),
                                            dtype=np.int32)
                            st.block_tables = np.concatenate([st.block_tables, grow], axis=3)
                        st.block_tables[l, b, h, j] = blk
                        st.seq_index_by_block[blk] = b
                        st.layer_index_by_block[blk] = l
                        st.head_index_by_block[blk] = h
                        st.logical_block_num_by_block[blk] = j
                        st.token_positions[blk] = last_pos + np.arange(bs, dtype=np.int32)
                    blk = int(st.block_tables[l, b, h, ctx // bs])
                    st.token_positions[blk, ctx % bs] = last_pos
                    st.metrics[blk, ctx % bs] = 0.0            # head bias (zeros in the engine)
                    st.context_lens[l, b, h] = ctx + 1
            self.seq_lens[b] += 1
        st.seq_positions = (self.seq_lens - 1).astype(np.int32)
        self._refresh()
        return need

    def _refresh(self):
        st = self.st
        st.hanging_token_count = hanging_tokens(st.context_lens.transpose(1, 0, 2), st.block_size)
        st.evicted_kv_offsets = kv_offsets(st.context_lens, st.block_size)
