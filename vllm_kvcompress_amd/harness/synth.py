"""Seeded synthetic paged-cache states for the eviction/compaction path.

Everything here is host-side NumPy: it only *describes* a per-head paged KV
cache (block tables, per-block metadata, per-slot metrics and positions) in
the layout the reference keeps on device:

* per-block metadata and per-slot stores of ``CompressionMetrics``
  (reference ``vllm/kvcompress/metrics.py:220-275``),
* ``context_lens [L,B,H]`` / ``block_tables [L,B,H,M]`` of ``BlockState``
  (reference ``vllm/kvcompress/block.py:95-126``),
* ``hanging_token_count`` (``block.py:330-335``) and ``evicted_kv_offsets``
  (``vllm/kvcompress/scheduler.py:274-280``).

The engine invariants honoured are the ones listed in SURVEY.md section 8(d):
``seq_pos = seq_len - 1``, cached positions ``<= seq_pos - 1``, empty tail
slots carry positions ``>= seq_pos``, and metrics are tie-free per sequence
(a permutation cast to float32) unless ties are requested explicitly.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

MAX_INT = 2147483000  # reference vllm/kvcompress/metrics.py:12


@dataclass
class PagedState:
    """One compression batch worth of paged-cache bookkeeping (host copy)."""
    block_size: int
    num_layers: int
    num_kv_heads: int
    num_seqs: int
    num_blocks: int                      # NB: physical blocks in the cache
    # --- CompressionMetrics stores -------------------------------------------------
    metrics: np.ndarray                  # [NB, bs] f32
    token_positions: np.ndarray          # [NB, bs] i32
    seq_index_by_block: np.ndarray       # [NB] i32, -1 = unallocated
    layer_index_by_block: np.ndarray     # [NB] i32
    head_index_by_block: np.ndarray      # [NB] i32
    logical_block_num_by_block: np.ndarray  # [NB] i32
    # --- BlockState view -----------------------------------------------------------
    context_lens: np.ndarray             # [L, B, H] i32
    block_tables: np.ndarray             # [L, B, H, M] i32
    hanging_token_count: np.ndarray      # [B, L, H] i32
    evicted_kv_offsets: np.ndarray       # [B, L, H] i32
    # --- per sequence --------------------------------------------------------------
    seq_indices: list                    # batch-slot ids (ascending)
    seq_positions: np.ndarray            # [B] i32  (= seq_len - 1)
    protected: list                      # [B] protected window sizes
    extra: dict = field(default_factory=dict)

    @property
    def total_slots(self) -> int:
        nblk = (self.context_lens + self.block_size - 1) // self.block_size
        return int(nblk.sum()) * self.block_size

    @property
    def total_heads(self) -> int:
        return self.num_seqs * self.num_layers * self.num_kv_heads


def hanging_tokens(context_lens_blh: np.ndarray, block_size: int) -> np.ndarray:
    """``where(ctx % bs == 0, bs, ctx % bs)`` (reference block.py:330-335)."""
    r = context_lens_blh % block_size
    return np.where(r == 0, block_size, r).astype(np.int32)


def kv_offsets(context_lens_lbh: np.ndarray, block_size: int) -> np.ndarray:
    """Exclusive cumsum of per-head allocated slots in (b,l,h) order
    (reference scheduler.py:274-280)."""
    per_head = ((context_lens_lbh.transpose(1, 0, 2).astype(np.int64) + block_size - 1)
                // block_size) * block_size
    flat = per_head.reshape(-1)
    off = np.zeros_like(flat)
    np.cumsum(flat[:-1], out=off[1:])
    assert off[-1] + flat[-1] < 2 ** 31, "int32 slot offsets overflow (SURVEY Q11)"
    return off.reshape(per_head.shape).astype(np.int32)


def make_state(
    *,
    num_layers: int,
    num_kv_heads: int,
    block_size: int,
    seq_lens: Sequence[int],
    seed: int = 0,
    protected: int | Sequence[int] = 1,
    compressed: bool = False,
    spare_block_frac: float = 0.1,
    tie_levels: Optional[int] = None,
    metric_shape: str = "perm",
    shuffle_blocks: bool = True,
    steady_cap: Optional[int] = None,
) -> PagedState:
    """Build a synthetic state.

    seq_lens      ``seq.data.get_len()`` per sequence: includes the token sampled
                  last step whose KV is not cached yet, so an uncompressed head
                  holds ``seq_len - 1`` KVs and ``seq_pos = seq_len - 1``
                  (reference scheduler.py:256-260).
    compressed    if True emulate a "second compression" state: every head keeps
                  a random multiple of ``block_size`` survivors with arbitrary
                  (sorted) positions, plus a random number of appended decode
                  tokens.
    steady_cap    continual-compression steady state: every head holds ``steady_cap`` survivors
                  (a multiple of block_size, arbitrary sorted positions) plus ONE freshly
                  appended token, i.e. the state right before the next compression step.
    tie_levels    if set, metrics are drawn from that many distinct values (to
                  exercise the canonical tie order); default is tie-free.
    metric_shape  "perm": per-sequence random permutation cast to f32;
                  "decay": permutation re-ranked so that older positions tend to
                  carry smaller metrics (still tie-free);
                  "oldest": metric rank == position rank (evict-oldest policy, the
                  fully clustered extreme; tie-free).
    """
    rng = np.random.default_rng(seed)
    L, H, bs, B = num_layers, num_kv_heads, block_size, len(seq_lens)
    seq_lens = np.asarray(seq_lens, dtype=np.int64)
    seq_pos = (seq_lens - 1).astype(np.int32)
    if np.isscalar(protected):
        protected = [int(protected)] * B
    protected = [int(p) for p in protected]

    ctx = np.zeros((L, B, H), dtype=np.int32)
    for b in range(B):
        full = int(seq_lens[b]) - 1
        if steady_cap is not None:
            assert steady_cap % bs == 0
            ctx[:, b, :] = min(steady_cap + 1, full)
        elif not compressed:
            ctx[:, b, :] = full
        else:
            # survivors: a multiple of bs in [0, full], then 0..bs-1 appended tokens
            max_blocks = max(full // bs, 0)
            kept_blocks = rng.integers(0, max_blocks + 1, size=(L, H))
            appended = rng.integers(0, bs, size=(L, H))
            ctx[:, b, :] = np.minimum(kept_blocks * bs + appended, full)
    nblk = (ctx + bs - 1) // bs                      # [L,B,H]
    total_blocks = int(nblk.sum())
    NB = total_blocks + int(np.ceil(total_blocks * spare_block_frac)) + 1
    M = max(int(nblk.max()), 1)

    phys = rng.permutation(NB)[:total_blocks] if shuffle_blocks else np.arange(total_blocks)
    phys = phys.astype(np.int32)

    seq_by = np.full(NB, -1, dtype=np.int32)
    lay_by = np.zeros(NB, dtype=np.int32)
    head_by = np.zeros(NB, dtype=np.int32)
    lbn_by = np.zeros(NB, dtype=np.int32)
    positions = np.zeros((NB, bs), dtype=np.int32)
    metrics = np.zeros((NB, bs), dtype=np.float32)
    block_tables = np.zeros((L, B, H, M), dtype=np.int32)

    # allocation order mimics prefill allocation: per sequence a contiguous slice
    # of the (shuffled) free list reshaped [L,H,nblk] (block_manager.py:196-222)
    cursor = 0
    ar = np.arange(bs, dtype=np.int32)
    for b in range(B):
        seq_slots = []
        for l in range(L):
            for h in range(H):
                n = int(nblk[l, b, h])
                blocks = phys[cursor:cursor + n]
                cursor += n
                block_tables[l, b, h, :n] = blocks
                seq_by[blocks] = b
                lay_by[blocks] = l
                head_by[blocks] = h
                lbn_by[blocks] = np.arange(n, dtype=np.int32)
                c = int(ctx[l, b, h])
                lam = (np.arange(n, dtype=np.int32)[:, None] * bs + ar[None, :])  # [n,bs]
                if not compressed and steady_cap is None:
                    pos = lam.copy()                 # positions == logical index
                else:
                    # survivors: sorted distinct positions below seq_pos; the empty
                    # tail continues from seq_pos upward (block.py:536-569)
                    hi = max(int(seq_pos[b]), 1)
                    if c > 0:
                        live = np.sort(rng.choice(hi, size=min(c, hi), replace=False))
                        if live.size < c:
                            live = np.concatenate([live, np.full(c - live.size, hi - 1)])
                    else:
                        live = np.zeros(0, dtype=np.int64)
                    tail = int(seq_pos[b]) + np.arange(n * bs - c)
                    pos = np.concatenate([live, tail]).astype(np.int32).reshape(n, bs)
                positions[blocks] = pos
                seq_slots.append((blocks, n))
        # tie-free metrics per sequence: a permutation of 0..N_seq-1 as f32
        n_seq = sum(n for _, n in seq_slots) * bs
        assert n_seq <= 2 ** 24, "float32 permutation is exact only up to 2^24 slots/seq"
        if tie_levels is None:
            vals = rng.permutation(n_seq).astype(np.float32)
            if metric_shape == "oldest":
                allpos = np.concatenate([positions[blk].reshape(-1) for blk, _ in seq_slots])
                order = np.argsort(allpos, kind="stable")
                ranked = np.empty(n_seq, dtype=np.float32)
                ranked[order] = np.arange(n_seq, dtype=np.float32)
                vals = ranked
            if metric_shape == "decay":
                # re-rank: smaller metrics go (noisily) to older positions
                allpos = np.concatenate([positions[blk].reshape(-1) for blk, _ in seq_slots])
                noise = rng.normal(0.0, 0.25 * max(int(seq_lens[b]), 1), size=n_seq)
                order = np.argsort(allpos + noise, kind="stable")
                ranked = np.empty(n_seq, dtype=np.float32)
                ranked[order] = np.arange(n_seq, dtype=np.float32)
                vals = ranked
        else:
            vals = rng.integers(0, tie_levels, size=n_seq).astype(np.float32)
        o = 0
        for blocks, n in seq_slots:
            metrics[blocks] = vals[o:o + n * bs].reshape(n, bs)
            o += n * bs
    assert cursor == total_blocks

    hang = hanging_tokens(ctx.transpose(1, 0, 2), bs)
    offs = kv_offsets(ctx, bs)
    return PagedState(
        block_size=bs, num_layers=L, num_kv_heads=H, num_seqs=B, num_blocks=NB,
        metrics=metrics, token_positions=positions, seq_index_by_block=seq_by,
        layer_index_by_block=lay_by, head_index_by_block=head_by,
        logical_block_num_by_block=lbn_by, context_lens=ctx, block_tables=block_tables,
        hanging_token_count=hang, evicted_kv_offsets=offs,
        seq_indices=list(range(B)), seq_positions=seq_pos, protected=protected,
    )


# --------------------------------------------------------------------------------------
# host policy: how many blocks to free per sequence (reference scheduler.py:100-181)
# --------------------------------------------------------------------------------------
def evict_block_count(
    *,
    context_lens_lh: np.ndarray,        # [L,H] context lens of ONE sequence
    seq_len: int,
    block_size: int,
    protected_window_size: int,
    max_cache_tokens: int = -1,
    target_compression_rate: float = 1.0,
    even_layer_evict: bool = False,
) -> int:
    """Blocks to free for one sequence; mirrors ``_schedule_seq_evictions``."""
    import math
    bs = block_size
    L, H = context_lens_lh.shape
    TH = L * H
    if max_cache_tokens > 0:
        max_cache_tokens = (max_cache_tokens + bs - 1) // bs * bs
    if target_compression_rate < 1.0 and max_cache_tokens > 0:
        raise RuntimeError("both compression_rate and max_cache_tokens "
                           "specified during compression")
    seq_blocks = int(((context_lens_lh.astype(np.int64) + bs - 1) // bs).sum())
    seq_kvs = int(context_lens_lh.astype(np.int64).sum())
    if max_cache_tokens >= 0:
        max_cache_blocks = (max_cache_tokens * TH + bs - 1) // bs
        n = max(0, seq_blocks - max_cache_blocks)
    else:
        protected_tokens = (protected_window_size + bs - 1) // bs * bs
        compressible = seq_len - protected_tokens
        if compressible <= 0:
            return 0
        target_kv = math.ceil(compressible * TH * target_compression_rate) + protected_tokens * TH
        evict_kv = max(0, seq_kvs - target_kv)
        n = (evict_kv + bs - 1) // bs
    if even_layer_evict:
        n = n // L * L
    limit = max(seq_blocks - (protected_window_size + bs - 1) // bs * TH, 0)
    assert n <= limit, (n, limit)
    return n


def random_kv_bytes(num_blocks: int, block_bytes: int, seed: int) -> np.ndarray:
    """Random bytes for a unified cache ``[2, NB, block_bytes]`` (bitwise parity)."""
    rng = np.random.default_rng(seed + 7919)
    return rng.integers(0, 256, size=(2, num_blocks, block_bytes), dtype=np.uint8)


def make_caches_u16(seed: int, num_blocks: int, head_size: int, block_size: int):
    """Random 2-byte-element K/V caches in the KVCAttention layout
    (reference vllm/attention/ops/paged_attn.py:272-284): K ``[NB, hd/x, bs, x]``
    with ``x = 16 // 2``, V ``[NB, hd, bs]``.  Returned as int16 bit patterns."""
    x = 8
    rng = np.random.default_rng(seed)
    k = rng.integers(0, 2 ** 16, size=(num_blocks, head_size // x, block_size, x),
                     dtype=np.uint16).view(np.int16)
    v = rng.integers(0, 2 ** 16, size=(num_blocks, head_size, block_size),
                     dtype=np.uint16).view(np.int16)
    return k, v
