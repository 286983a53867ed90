"""CPU oracle for the KV-Compress eviction/compaction hot path.

TEST INFRASTRUCTURE ONLY.  This module is the *checker*: a NumPy restatement of
the reference algorithm, function by function, each citing the reference
file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product
(``vllm_kvcompress_amd``) never does and has no CPU fallback.

Parity pinning: every function here is checked in ``tests/test_oracle_golden.py``
against golden vectors under ``tests/golden/`` that were produced by importing
the reference's own Python (``vllm/_custom_ops.py`` ``ref_*`` twins and
``vllm/kvcompress/metrics.py::CompressionMetrics.schedule_evictions``) in the
build container with ``oracle/gen_golden.py`` (committed).  The native kernels
(`count_block_evictions_kernel`, `single_tier_schedule_cache_moves_kernel`,
`execute_cache_moves_kernel`) are CUDA and cannot be built here (no nvcc; a
hipify would be a port, which this repo must not contain), so for those the
anchor is the reference's Python twins, which the reference's own test asserts
equal to the kernels (tests/kernels/test_kvcompress_eviction.py:900-901,967).

Tie order.  The reference sorts float metrics with ``torch.sort`` (unstable);
its result on tied metrics is implementation defined.  The oracle (and the HIP
path) define the canonical order as *stable by masked flat index* ``f``
(SURVEY.md section 8(a) step 3), chunk thresholds tie-break by
(head, chunk).  On tie-free inputs this is irrelevant and the oracle is
bit-identical to the reference (all golden vectors are tie-free).
"""
from __future__ import annotations

import numpy as np

MAX_INT = 2147483000  # vllm/kvcompress/metrics.py:12

INF32 = np.float32(np.inf)


# --------------------------------------------------------------------------------------
# KVHeadBias.get_bias_for_position        vllm/kvcompress/metrics.py:54-81
# --------------------------------------------------------------------------------------
def bias_for_position(bias, position_bins, positions, layer_idx, head_idx):
    """bias [L,H,nbins] f32, position_bins [nbins] i32, positions [nb,bs] i32,
    layer_idx/head_idx [nb] -> [nb,bs] f32."""
    nbins = position_bins.shape[0]
    # number of bins with position >= bin, minus one; -1 wraps to the last bin
    # (Python negative indexing in the reference gather, metrics.py:72-77)
    cnt = (positions[..., None] >= position_bins[None, None, :]).sum(-1) - 1
    cnt = np.where(cnt < 0, cnt + nbins, cnt)
    out = bias[layer_idx[:, None], head_idx[:, None], cnt].astype(np.float32)
    out = np.where(positions < 0, np.float32(0), out)
    return out.astype(np.float32)


# --------------------------------------------------------------------------------------
# count_block_evictions_kernel            csrc/kvcompress_eviction_kernels.cu:190-221
# --------------------------------------------------------------------------------------
def count_block_evictions(evicted_block_count, evicted_logical_indices, evicted_kv_offsets,
                          hanging_token_count, block_size, null_value):
    """In place on ``evicted_block_count`` and ``evicted_logical_indices``."""
    offs = evicted_kv_offsets.reshape(-1)
    hang = hanging_token_count.reshape(-1)
    out = evicted_block_count.reshape(-1)
    total_heads = offs.shape[0]
    total_kvs = evicted_logical_indices.shape[0]
    for g in range(total_heads):
        start = int(offs[g])
        end = total_kvs if g + 1 >= total_heads else int(offs[g + 1])
        n = 0
        i = start
        while i < end:
            if evicted_logical_indices[i] != null_value:
                n += 1
            else:
                break
            i += block_size
        out[g] = n
        if n > 0:
            last_end = start + n * block_size
            evicted_logical_indices[last_end - block_size + int(hang[g]):last_end] = null_value


# --------------------------------------------------------------------------------------
# CompressionMetrics.schedule_evictions   vllm/kvcompress/metrics.py:441-847
# --------------------------------------------------------------------------------------
def schedule_evictions(
    *,
    metrics, token_positions, seq_index_by_block, layer_index_by_block,
    head_index_by_block, logical_block_num_by_block,
    block_size, num_layers, num_kv_heads,
    seq_indices, seq_positions, evicted_blocks_per_seq,
    context_lens, hanging_token_count, evicted_kv_offsets, num_protected,
    use_average=False, num_sinks=0,
    bias=None, position_bins=None, bias_weight=0.0,
    mode="reference", uniform_evict=False,
):
    """Returns (evicted_logical_indices [N] i32, evicted_kv_count [B,L,H] i32,
    evicted_block_count [B,L,H] i32).

    uniform_evict        the reference's other selection rule (metrics.py:639-666; its scheduler
                         never passes it, scheduler.py:492-501): every head of sequence i frees
                         evicted_blocks_per_seq[i] // (L*H) chunks -- its own lowest ones --
                         instead of the sequence's lowest chunks wherever they lie.  The
                         reference needs the heads of a sequence to hold equally many blocks
                         (a reshape) and asserts the freed thresholds finite.

    mode="reference"     bit-exact to the reference including its batch>1
                         inf-count quirk (metrics.py:718-721).
    mode="per_sequence"  every sequence scheduled as if it were alone (== the
                         reference called with B=1 per sequence).
    """
    bs, L, H = block_size, num_layers, num_kv_heads
    seq_indices = [int(s) for s in seq_indices]
    assert sorted(seq_indices) == seq_indices          # metrics.py:459
    B = len(seq_indices)
    seq_positions = np.asarray(seq_positions, dtype=np.int32).reshape(-1)
    num_protected = np.asarray(num_protected, dtype=np.int32).reshape(-1)
    evicted_blocks_per_seq = [int(x) for x in np.asarray(evicted_blocks_per_seq).reshape(-1)]

    # candidate blocks in ascending physical order            metrics.py:465-493
    slot_of_seq = np.full(max(seq_indices) + 1, -1, dtype=np.int64)
    for i, s in enumerate(seq_indices):
        slot_of_seq[s] = i
    sib = seq_index_by_block.astype(np.int64)
    seq_mask = (sib >= 0) & (sib <= max(seq_indices))
    seq_mask &= slot_of_seq[np.clip(sib, 0, max(seq_indices))] >= 0
    blocks = np.nonzero(seq_mask)[0]
    m = metrics[blocks].astype(np.float32).reshape(-1).copy()
    pos = token_positions[blocks].astype(np.int32)              # [nb,bs]
    lay = layer_index_by_block[blocks].astype(np.int64)
    head = head_index_by_block[blocks].astype(np.int64)
    lbn = logical_block_num_by_block[blocks].astype(np.int64)
    bslot = slot_of_seq[sib[blocks]]                             # batch slot i
    lam = (lbn[:, None] * bs + np.arange(bs)[None, :]).astype(np.int32)

    if use_average:                                              # metrics.py:495-501
        qcount = (seq_positions[bslot][:, None] - pos).astype(np.float32).reshape(-1)
        with np.errstate(divide="ignore", invalid="ignore"):
            m = (m / qcount).astype(np.float32)
    if bias is not None:                                         # metrics.py:503-506
        b = bias_for_position(bias, position_bins, pos, lay, head).reshape(-1)
        m = (m + (b * np.float32(bias_weight)).astype(np.float32)).astype(np.float32)

    slh = bslot * (L * H) + lay * H + head                       # metrics.py:520-524
    ctx_of = context_lens.transpose(1, 0, 2).reshape(-1)[slh].astype(np.int64)
    max_in_range = (seq_positions - num_protected)[bslot]        # metrics.py:513-515
    in_range = ((lbn < (ctx_of + bs - 1) // bs)[:, None]
                & (pos <= max_in_range[:, None])
                & (pos >= num_sinks))                            # metrics.py:539-543
    m[~in_range.reshape(-1)] = INF32

    # 1. order by (head, metric, flat index f)                   metrics.py:562-570
    nslots = m.shape[0]
    f = np.arange(nslots)
    slh_slot = np.repeat(slh, bs)
    order = np.lexsort((f, m, slh_slot))
    sorted_m = m[order]
    sorted_slh = slh_slot[order]

    # 2. chunk thresholds                                        metrics.py:583-596
    chunk_head = sorted_slh.reshape(-1, bs)[:, 0]
    hang_chunk = hanging_token_count.reshape(-1)[chunk_head].astype(np.int64)
    nchunks = chunk_head.shape[0]
    thr = sorted_m.reshape(-1, bs)[np.arange(nchunks), hang_chunk - 1]

    # 3. per-sequence selection                                  metrics.py:604-755
    sorted_lam = lam.reshape(-1)[order].reshape(-1, bs).copy()
    total_blocks_per_seq = ((context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)
    # seq of a chunk: reference takes it from the first slot's block (metrics.py:672-677)
    chunk_seq = bslot[order.reshape(-1, bs)[:, 0] // bs]
    corder = np.lexsort((np.arange(nchunks), thr, chunk_seq))    # (seq, thr, position)
    seq_sorted_thr = thr[corder]
    keep_mask_sorted = np.zeros(nchunks, dtype=bool)             # True = NOT evicted
    offset = 0
    if uniform_evict:                                            # metrics.py:639-666
        keep_chunk = np.zeros(nchunks, dtype=bool)               # chunks are in (head, metric) order here
        for i, k in enumerate(evicted_blocks_per_seq):
            per_head = max(k, 0) // (L * H)                      # :649
            end_offset = offset + int(total_blocks_per_seq[i])
            assert (end_offset - offset) % (L * H) == 0, "uniform_evict: heads of unequal length (the reference's reshape fails)"
            cur = keep_chunk[offset:end_offset].reshape(L * H, -1)
            cur[:, per_head:] = True                             # :656
            assert np.all(thr[offset:end_offset].reshape(L * H, -1)[:, :per_head] < INF32)   # :660
            offset = end_offset
        evicted_blocks_per_seq = []                              # (the variable-rate loop below is the else branch)
    for i, k in enumerate(evicted_blocks_per_seq):
        end_offset = offset + int(total_blocks_per_seq[i])
        un = offset + k
        if mode == "reference":
            ninf = int(np.count_nonzero(seq_sorted_thr[:un] == INF32))       # :718 (from 0)
        elif mode == "per_sequence":
            ninf = int(np.count_nonzero(seq_sorted_thr[offset:un] == INF32))
        else:
            raise ValueError(mode)
        un -= ninf
        assert un >= 0
        keep_mask_sorted[un:end_offset] = True                   # :723
        assert np.all(seq_sorted_thr[offset:un] < INF32)         # :725
        offset = end_offset
    if not uniform_evict:
        keep_chunk = np.zeros(nchunks, dtype=bool)
        keep_chunk[corder] = keep_mask_sorted                    # :755 (remap)
    sorted_lam[keep_chunk, :] = MAX_INT
    flat = sorted_lam.reshape(-1).astype(np.int32)

    # count leading evicted chunks per head, truncate hanging tail   metrics.py:773-792
    evicted_block_count = np.empty_like(evicted_kv_offsets, dtype=np.int32)
    count_block_evictions(evicted_block_count, flat, evicted_kv_offsets,
                          hanging_token_count, bs, MAX_INT)
    evicted_kv_count = np.where(
        evicted_block_count > 0,
        (evicted_block_count - 1) * bs + hanging_token_count, 0).astype(np.int32)
    assert np.all(context_lens.transpose(1, 0, 2) >= evicted_kv_count)   # :798

    # 4. per head ascending logical index                        metrics.py:822-834
    final = np.lexsort((flat, sorted_slh))
    return flat[final].astype(np.int32), evicted_kv_count, evicted_block_count


# --------------------------------------------------------------------------------------
# The reference's batch > 1 rule in two stages        vllm/kvcompress/metrics.py:671-729
#
# ``schedule_evictions(mode="reference")`` above restates the reference literally: ONE sort over the whole
# batch, then the host loop.  At 256 coupled sequences (BASELINE configs[2]) that sort is minutes of NumPy, so
# the HIP path's coupled mode had no oracle verdict at size.  The loop couples the sequences through one thing
# only -- how many chunks with an infinite threshold lie in front of position ``offset_i + k_i`` of the
# (sequence, threshold)-ordered chunk list (:718, counted from 0: the reference's bug) -- and that needs no sort:
# a head's thresholds are every bs-th order statistic of its masked metrics, so the number of FINITE ones follows
# from the number of finite masked metrics of the head.  Hence:
#   stage 1  finite_threshold_chunks():  per sequence, its chunks with a finite threshold (one counting pass);
#   stage 2  coupled_eviction_counts():  :709-729 as interval arithmetic on (nblk_i, fin_i, k_i) -> the number of
#            chunks every sequence REALLY frees, later sequences un-evicting earlier ones included;
#   stage 3  the per-sequence schedule (mode="per_sequence", a sequence alone) with those counts.
# ``schedule_evictions_two_stage`` composes them and is pinned to the literal restatement and to the
# reference-generated b2_* / b3_* fixtures in tests/test_oracle_golden.py (and against the imported reference by
# oracle/crosscheck_reference.py); bench.py's reference-mode gate and tests/test_gpu_scale.py use stages 1 + 2 on
# the whole batch and stage 3 on a sample of its sequences.
# --------------------------------------------------------------------------------------
def finite_threshold_chunks(
    *, metrics, token_positions, seq_index_by_block, layer_index_by_block, head_index_by_block,
    logical_block_num_by_block, block_size, num_layers, num_kv_heads, seq_indices, seq_positions,
    context_lens, hanging_token_count, num_protected, use_average=False, num_sinks=0, bias=None,
    position_bins=None, bias_weight=0.0, blocks_per_pass=1 << 20,
):
    """Stage 1.  Returns ``(fin [B], nblk [B])``: per sequence of the batch the number of chunks whose threshold
    (metrics.py:583-596: the ``hang_g``-th entry of every ``bs`` of a head's ascending masked metrics) is finite, and
    its number of chunks.  Head g has ``R_g`` finite masked metrics (the mask of metrics.py:539-544), so its chunks
    ``c`` with ``c * bs + hang_g <= R_g`` are the finite ones."""
    bs, L, H = block_size, num_layers, num_kv_heads
    seq_indices = [int(s) for s in seq_indices]
    B = len(seq_indices)
    seq_positions = np.asarray(seq_positions, dtype=np.int32).reshape(-1)
    num_protected = np.asarray(num_protected, dtype=np.int32).reshape(-1)
    slot_of_seq = np.full(max(seq_indices) + 1, -1, dtype=np.int64)
    slot_of_seq[seq_indices] = np.arange(B)
    sib = seq_index_by_block.astype(np.int64)
    cand = (sib >= 0) & (sib <= max(seq_indices))
    cand &= slot_of_seq[np.clip(sib, 0, max(seq_indices))] >= 0
    blocks_all = np.nonzero(cand)[0]
    ctx_blh = context_lens.transpose(1, 0, 2).reshape(-1).astype(np.int64)
    R = np.zeros(B * L * H, dtype=np.int64)
    for lo in range(0, blocks_all.shape[0], blocks_per_pass):
        blocks = blocks_all[lo:lo + blocks_per_pass]
        m = metrics[blocks].astype(np.float32)
        pos = token_positions[blocks].astype(np.int32)
        lay = layer_index_by_block[blocks].astype(np.int64)
        head = head_index_by_block[blocks].astype(np.int64)
        lbn = logical_block_num_by_block[blocks].astype(np.int64)
        bslot = slot_of_seq[sib[blocks]]
        if use_average:                                              # metrics.py:495-501
            with np.errstate(divide="ignore", invalid="ignore"):
                m = (m / (seq_positions[bslot][:, None] - pos).astype(np.float32)).astype(np.float32)
        if bias is not None:                                         # metrics.py:503-506
            b = bias_for_position(bias, position_bins, pos, lay, head)
            m = (m + (b * np.float32(bias_weight)).astype(np.float32)).astype(np.float32)
        slh = bslot * (L * H) + lay * H + head
        in_range = ((lbn < (ctx_blh[slh] + bs - 1) // bs)[:, None]
                    & (pos <= (seq_positions - num_protected)[bslot][:, None])
                    & (pos >= num_sinks))                            # metrics.py:539-543
        finite = in_range & (m < INF32)
        R += np.bincount(slh, weights=finite.sum(1), minlength=B * L * H).astype(np.int64)
    hang = hanging_token_count.reshape(-1).astype(np.int64)
    fin_head = np.where(R >= hang, (R - hang) // bs + 1, 0)
    nblk_head = (ctx_blh + bs - 1) // bs
    assert np.all(fin_head <= nblk_head)
    return fin_head.reshape(B, L * H).sum(1), nblk_head.reshape(B, L * H).sum(1)


def coupled_eviction_counts(evicted_blocks_per_seq, nblk, fin, mode="reference"):
    """Stage 2: the host loop of metrics.py:709-729 on counts.  The chunk list is ordered by (sequence, threshold),
    so inside sequence j's range ``[o_j, o_j + nblk_j)`` the ``fin_j`` finite thresholds come first.  For
    ``i = 0..B-1``: ``un_i = o_i + k_i - #inf in [0, o_i + k_i)`` (:718 -- from 0, not from ``o_i``) and everything in
    ``[un_i, o_i + nblk_i)`` is marked kept (:723).  A chunk at position x of sequence j therefore stays evicted iff
    ``x < un_i`` for every ``i >= j``.  Returns the number of chunks each sequence frees."""
    k = [int(x) for x in np.asarray(evicted_blocks_per_seq).reshape(-1)]
    nblk = np.asarray(nblk, dtype=np.int64)
    fin = np.asarray(fin, dtype=np.int64)
    B = len(k)
    o = np.concatenate([[0], np.cumsum(nblk)[:-1]]).astype(np.int64)
    inf_lo, inf_hi = o + fin, o + nblk                            # the infinite thresholds of each sequence
    un = np.empty(B, dtype=np.int64)
    for i in range(B):
        x = o[i] + k[i]
        if mode == "reference":
            ninf = int(np.clip(np.minimum(x, inf_hi) - inf_lo, 0, None).sum())
        elif mode == "per_sequence":
            ninf = int(max(0, min(x, inf_hi[i]) - inf_lo[i]))
        else:
            raise ValueError(mode)
        un[i] = x - ninf
        assert un[i] >= 0                                         # (the literal restatement's assertion)
        assert un[i] <= o[i] + fin[i], "an infinite threshold inside the evicted range (metrics.py:725)"
    suffix_min = np.minimum.accumulate(un[::-1])[::-1]
    return np.clip(suffix_min - o, 0, nblk).astype(np.int64)


def schedule_evictions_two_stage(*, seq_indices, seq_positions, evicted_blocks_per_seq, context_lens,
                                 hanging_token_count, evicted_kv_offsets, num_protected, block_size, num_layers,
                                 num_kv_heads, mode="reference", only=None, **store):
    """``schedule_evictions(mode=...)`` of a batch, computed as stage 1 + stage 2 + one per-sequence run per
    sequence.  ``only``: batch positions to run stage 3 for (default all) -- the return value then holds, per
    listed position, ``(evicted_logical_indices of the sequence's heads, evicted_kv_count [L,H],
    evicted_block_count [L,H])``, plus the effective counts of the WHOLE batch; with ``only=None`` the three
    arrays of ``schedule_evictions`` are returned."""
    bs, L, H = block_size, num_layers, num_kv_heads
    seq_indices = [int(s) for s in seq_indices]
    B = len(seq_indices)
    seq_positions = np.asarray(seq_positions, dtype=np.int32).reshape(-1)
    num_protected = np.asarray(num_protected, dtype=np.int32).reshape(-1)
    common = dict(block_size=bs, num_layers=L, num_kv_heads=H, **store)
    fin, nblk = finite_threshold_chunks(seq_indices=seq_indices, seq_positions=seq_positions, context_lens=context_lens,
                                        hanging_token_count=hanging_token_count, num_protected=num_protected, **common)
    keff = coupled_eviction_counts(evicted_blocks_per_seq, nblk, fin, mode)
    pieces = {}
    for j in (range(B) if only is None else only):
        ctx_j = np.ascontiguousarray(context_lens[:, j:j + 1, :])
        per_head = (((ctx_j.astype(np.int64) + bs - 1) // bs) * bs).transpose(1, 0, 2).reshape(-1)
        offs_j = (np.cumsum(per_head) - per_head).astype(np.int32).reshape(1, L, H)
        pieces[j] = schedule_evictions(
            seq_indices=[seq_indices[j]], seq_positions=seq_positions[j:j + 1], evicted_blocks_per_seq=[int(keff[j])],
            context_lens=ctx_j, hanging_token_count=np.ascontiguousarray(hanging_token_count[j:j + 1]),
            evicted_kv_offsets=offs_j, num_protected=num_protected[j:j + 1], mode="per_sequence", **common)
    if only is not None:
        return {j: (e, kc[0], bc[0]) for j, (e, kc, bc) in pieces.items()}, keff
    eli = np.concatenate([pieces[j][0] for j in range(B)])
    ekc = np.concatenate([pieces[j][1] for j in range(B)])
    ebc = np.concatenate([pieces[j][2] for j in range(B)])
    assert np.array_equal(np.asarray(evicted_kv_offsets).reshape(-1)[::L * H],
                          np.concatenate([[0], np.cumsum(nblk * bs)[:-1]]))
    return eli, ekc, ebc


# --------------------------------------------------------------------------------------
# single_tier_schedule_cache_moves_kernel  csrc/kvcompress_eviction_kernels.cu:223-289
# (Python twin: vllm/_custom_ops.py:1108-1154; wrapper zero-fill :1168)
# --------------------------------------------------------------------------------------
def schedule_cache_moves(out_cache_moves_idx, out_cache_moves_count, evicted_logical_indices,
                         evicted_kv_count, evicted_kv_offsets, block_tables, context_lens,
                         block_size):
    """In place. ``out_cache_moves_idx [R,2]`` is zero-filled first like the wrapper."""
    out_cache_moves_idx[...] = 0
    B, L, H = evicted_kv_count.shape
    bs = block_size
    E = evicted_logical_indices
    for b in range(B):
        for l in range(L):
            for h in range(H):
                cnt = int(evicted_kv_count[b, l, h])
                off = int(evicted_kv_offsets[b, l, h])
                ctx = int(context_lens[l, b, h])
                bt = block_tables[l, b, h]
                mc = 0
                ec = 0
                for i in range(cnt):
                    src = ctx - 1 - i
                    stop = int(E[off + cnt - 1 - ec])
                    dst = int(E[off + mc])
                    if dst >= src:
                        break
                    if src <= stop:
                        ec += 1
                        continue
                    out_cache_moves_idx[off + mc, 0] = int(bt[dst // bs]) * bs + dst % bs
                    out_cache_moves_idx[off + mc, 1] = int(bt[src // bs]) * bs + src % bs
                    mc += 1
                out_cache_moves_count[b, l, h] = mc


# --------------------------------------------------------------------------------------
# execute_cache_moves_kernel               csrc/kvcompress_eviction_kernels.cu:359-435
# (Python twin: vllm/_custom_ops.py:1182-1216)
# --------------------------------------------------------------------------------------
def execute_cache_moves(k_cache, v_cache, kv_metrics, kv_position, cache_moves_idx,
                        cache_moves_count, evicted_kv_offsets):
    """k_cache [NB, hd/x, bs, x], v_cache [NB, hd, bs] (any dtype; byte copy),
    kv_metrics [NB,bs] f32, kv_position [NB,bs] i32.  In place."""
    bs = v_cache.shape[2]
    cnt = cache_moves_count.reshape(-1)
    off = evicted_kv_offsets.reshape(-1)
    met = kv_metrics.reshape(-1)
    posv = kv_position.reshape(-1)
    for g in range(cnt.shape[0]):
        o = int(off[g])
        for j in range(int(cnt[g])):
            dst = int(cache_moves_idx[o + j, 0])
            src = int(cache_moves_idx[o + j, 1])
            db, do = divmod(dst, bs)
            sb, so = divmod(src, bs)
            met[dst] = met[src]
            posv[dst] = posv[src]
            k_cache[db, :, do, :] = k_cache[sb, :, so, :]
            v_cache[db, :, do] = v_cache[sb, :, so]


def execute_cache_moves_vectorized(k_cache, v_cache, kv_metrics, kv_position, cache_moves_idx,
                                   cache_moves_count, evicted_kv_offsets):
    """Same result as :func:`execute_cache_moves` for independent moves (no dst is
    also a src, dsts distinct) -- the only case the reference kernel defines
    (csrc/kvcompress_eviction_kernels.cu:358).  Used at sizes where the scalar loop
    is too slow."""
    bs = v_cache.shape[2]
    cnt = cache_moves_count.reshape(-1).astype(np.int64)
    off = evicted_kv_offsets.reshape(-1).astype(np.int64)
    rows = np.concatenate([np.arange(o, o + c) for o, c in zip(off, cnt)]) if cnt.sum() else \
        np.zeros(0, dtype=np.int64)
    dst = cache_moves_idx[rows, 0].astype(np.int64)
    src = cache_moves_idx[rows, 1].astype(np.int64)
    assert np.intersect1d(dst, src).size == 0 and np.unique(dst).size == dst.size
    kv_metrics.reshape(-1)[dst] = kv_metrics.reshape(-1)[src]
    kv_position.reshape(-1)[dst] = kv_position.reshape(-1)[src]
    k_cache[dst // bs, :, dst % bs, :] = k_cache[src // bs, :, src % bs, :]
    v_cache[dst // bs, :, dst % bs] = v_cache[src // bs, :, src % bs]


# --------------------------------------------------------------------------------------
# aggregation                               vllm/kvcompress/metrics.py:337-342,396-439
# --------------------------------------------------------------------------------------
def aggregate_decode(metrics, temp_metrics, use_l2=True):
    """metrics [NB,bs] f32 += sum_q temp^2 (or temp); float32, sum over the last
    (qpk) axis in index order (torch sum over a contiguous last dim of 4)."""
    t = temp_metrics.astype(np.float32)
    if use_l2:
        t = (t * t).astype(np.float32)
    acc = np.zeros(t.shape[:-1], dtype=np.float32)
    for q in range(t.shape[-1]):
        acc = (acc + t[..., q]).astype(np.float32)
    metrics += acc


def aggregate_prefill(metrics, prefill_metrics, slot_mapping, num_kv_heads):
    """prefill_metrics [T, H*qpk] f32, slot_mapping [T,H] i64 (unique slots)."""
    T = prefill_metrics.shape[0]
    v = prefill_metrics.astype(np.float32).reshape(T, num_kv_heads, -1)
    acc = np.zeros((T, num_kv_heads), dtype=np.float32)
    for q in range(v.shape[-1]):
        acc = (acc + v[..., q]).astype(np.float32)
    flat = metrics.reshape(-1)
    slots = slot_mapping.reshape(-1).astype(np.int64)
    flat[slots] = (flat[slots] + acc.reshape(-1)).astype(np.float32)


# --------------------------------------------------------------------------------------
# prefill metric epilogue   vllm/attention/backends/flash_attn.py:1147-1211
# (square -> mask -> column sum -> [avg scale] -> maxpool7), per q-block, then accumulate
# --------------------------------------------------------------------------------------
def prefill_metric_epilogue(out_kh, probs_hqk, q_offset, buffer_len, use_l2=True,
                            use_average=False, use_maxpool=True):
    """out_kh [K,Hq] f32 += epilogue(probs_hqk [Hq,qb,K] f32).  ``q_offset`` is the
    absolute position (within the sequence) of the first query row of the tile."""
    Hq, qb, K = probs_hqk.shape
    p = probs_hqk.astype(np.float32)
    if use_l2:
        p = (p * p).astype(np.float32)
    q = np.arange(qb)[:, None]
    k = np.arange(K)[None, :]
    mask = (k - q) <= (q_offset - int(buffer_len))      # tril(diagonal=q_offset-buffer_len)
    colsum = (p * mask[None].astype(np.float32)).sum(axis=1, dtype=np.float32)   # [Hq,K]
    if use_average:
        scale = (np.arange(1, K + 1, dtype=np.float32) / np.float32(qb)).astype(np.float32)
        colsum = (colsum * scale[None]).astype(np.float32)
    if use_maxpool:
        pad = np.full((Hq, 3), -np.inf, dtype=np.float32)
        ext = np.concatenate([pad, colsum, pad], axis=1)
        win = np.stack([ext[:, i:i + K] for i in range(7)], axis=0)
        colsum = win.max(axis=0)
    out_kh += colsum.T.astype(np.float32)


# --------------------------------------------------------------------------------------
# single_tier_reshape_and_cache_kernel     csrc/kvcompress_cache_kernels.cu:27-89
# --------------------------------------------------------------------------------------
def reshape_and_cache_kvc(key, value, key_cache, value_cache, kv_metrics, slot_mapping,
                          kv_metric_head_bias):
    """key/value [T,H,hd] (same dtype as the caches: "auto" path), key_cache
    [NB,hd/x,bs,x], value_cache [NB,hd,bs], slot_mapping [T*H] i64 (<0 = skip)."""
    T, H, hd = key.shape
    bs = value_cache.shape[2]
    x = key_cache.shape[3]
    sm = slot_mapping.reshape(T, H)
    met = kv_metrics.reshape(-1)
    for t in range(T):
        for h in range(H):
            s = int(sm[t, h])
            if s < 0:
                continue
            met[s] = kv_metric_head_bias[h]
            b, o = divmod(s, bs)
            key_cache[b, :, o, :] = key[t, h].reshape(hd // x, x)
            value_cache[b, :, o] = value[t, h]


# --------------------------------------------------------------------------------------
# fp8 KV cache (reshape_and_cache with kv_cache_dtype fp8 / fp8_e5m2)
# csrc/quantization/fp8/nvidia/quant_utils.cuh:456-489:  fp8 = cvt(float(x) / scale,
# __NV_SATFINITE, E4M3 | E5M2): round to nearest even, saturate to the largest finite.
# Restated by definition: enumerate every finite code's value, pick the nearest, ties to the
# code with an even mantissa LSB.
# --------------------------------------------------------------------------------------
def _fp8_table(kind):
    mbits, bias, maxcode = (3, 7, 0x7E) if kind == "e4m3" else (2, 15, 0x7B)
    codes = np.arange(0, maxcode + 1, dtype=np.int64)
    exp = codes >> mbits
    man = codes & ((1 << mbits) - 1)
    vals = np.where(exp == 0, man * 2.0 ** (1 - bias - mbits),
                    (1.0 + man / float(1 << mbits)) * 2.0 ** (exp.astype(np.float64) - bias))
    return codes, vals


def fp8_encode_satfinite(x, kind):
    """x float32 array -> uint8 codes (OCP e4m3fn / e5m2)."""
    codes, vals = _fp8_table(kind)
    x = np.asarray(x, dtype=np.float32)
    a = np.abs(x).astype(np.float64)
    sign = (np.signbit(x)).astype(np.uint8) << 7
    a_clip = np.minimum(a, vals[-1])                       # saturate (incl. inf)
    hi = np.searchsorted(vals, a_clip, side="left")        # first value >= a
    hi = np.clip(hi, 0, len(vals) - 1)
    lo = np.clip(hi - 1, 0, len(vals) - 1)
    dlo, dhi = np.abs(a_clip - vals[lo]), np.abs(vals[hi] - a_clip)
    pick_hi = (dhi < dlo) | ((dhi == dlo) & ((codes[hi] & 1) == 0))
    code = np.where(pick_hi, codes[hi], codes[lo]).astype(np.uint8)
    code = np.where(np.isnan(x), np.uint8(0x7F), code)
    return (code | sign).astype(np.uint8)


def reshape_and_cache_kvc_fp8(key, value, key_cache, value_cache, kv_metrics, slot_mapping,
                              kv_metric_head_bias, kind, k_scale, v_scale):
    """key/value [T,H,hd] float arrays (already widened to float32 exactly), caches uint8:
    key_cache [NB, hd/16, bs, 16], value_cache [NB, hd, bs]."""
    T, H, hd = key.shape
    bs = value_cache.shape[2]
    kq = fp8_encode_satfinite((key.astype(np.float32) / np.float32(k_scale)).astype(np.float32), kind)
    vq = fp8_encode_satfinite((value.astype(np.float32) / np.float32(v_scale)).astype(np.float32), kind)
    sm = slot_mapping.reshape(T, H)
    met = kv_metrics.reshape(-1)
    for t in range(T):
        for h in range(H):
            s = int(sm[t, h])
            if s < 0:
                continue
            met[s] = kv_metric_head_bias[h]
            b, o = divmod(s, bs)
            key_cache[b, :, o, :] = kq[t, h].reshape(hd // 16, 16)
            value_cache[b, :, o] = vq[t, h]


# --------------------------------------------------------------------------------------
# block-state side of a compression step (F2)
#   free_compressed_blocks          vllm/kvcompress/block_manager.py:466-530
#   last_n_allocated_block_mask     vllm/kvcompress/block.py:367-379
#   remove_trailing_blocks          vllm/kvcompress/block.py:184-210
#   ParallelBlockAllocator.free     vllm/kvcompress/block_manager.py:112-118
#   remove_metadata                 vllm/kvcompress/metrics.py:366-370
# --------------------------------------------------------------------------------------
def free_compressed_blocks(block_tables, context_lens, seq_indices, freed_block_count_blh,
                           seq_index_by_block, block_size, free_mask=None):
    """In place on context_lens / seq_index_by_block / free_mask; returns freed blocks in the
    reference's order (boolean mask over [L,B,H,M] -> (l, b, h, j ascending))."""
    bs = block_size
    L, S, H, M = block_tables.shape
    sel = list(seq_indices)
    ctx = context_lens[:, sel, :].astype(np.int64)                      # [L,B,H]
    block_counts = (ctx + bs - 1) // bs                                 # get_block_counts
    n = freed_block_count_blh.transpose(1, 0, 2).astype(np.int64)       # stack(dim=1) -> [L,B,H]
    j = np.arange(M)[None, None, None, :]
    mask = (j < block_counts[..., None]) & (j >= (block_counts - n)[..., None])
    freed = block_tables[:, sel][mask]
    if free_mask is not None:
        free_mask[freed] = True
    rem = ctx % bs
    hang = np.where(rem == 0, bs, rem)                                  # get_hanging_token_counts
    removed = np.clip(n * bs - (bs - hang), 0, None)
    context_lens[:, sel, :] -= removed.astype(context_lens.dtype)
    seq_index_by_block[freed] = -1
    return freed.astype(np.int32)


def append_slots(block_tables, context_lens, seq_indices, last_token_position, free_mask,
                 seq_index_by_block, layer_index_by_block, head_index_by_block,
                 logical_block_num_by_block, token_positions, block_size,
                 write_token_position=False):
    """One decode step's append (token_count = 1) for the batch slots ``seq_indices``: in place on
    every argument; returns the number of newly allocated blocks.

    BlockSpaceManagerKVC._append_to_sequence_batch  vllm/kvcompress/block_manager.py:269-294
    ParallelBlockAllocator.allocate                 vllm/kvcompress/block_manager.py:103-110
    BlockStateView.get_batch_new_block_metadata     vllm/kvcompress/block.py:513-620
    CompressionMetrics.insert_metadata              vllm/kvcompress/metrics.py:344-361
    A head whose context length sits on a block boundary gets the next logical block; the new
    blocks are the lowest-numbered free ones, handed out in (layer, batch position, head) order
    (boolean mask over [L,B,H,M]); their metadata rows are written and their position row is
    ``last_token_position + arange(bs)``.  ``write_token_position`` (not in the reference, which
    relies on the row written when the block was allocated) also stores the appended token's
    position in its slot -- what vllm_kvcompress_amd/harness/engine_sim.py models."""
    bs = block_size
    L, S, H, M = block_tables.shape
    sel = [int(s) for s in seq_indices]
    ctx_old = context_lens[:, sel, :].astype(np.int64)                   # [L,B,H]
    need = ctx_old % bs == 0
    idx = np.argwhere(need)                                              # row-major (l, b, h)
    n = idx.shape[0]
    free = np.nonzero(free_mask)[0][:n]
    if free.shape[0] < n:
        raise ValueError(f"Out of memory! Requested {n} out of {int(free_mask.sum())} available blocks.")
    assert (ctx_old[need] // bs < M).all(), "block table too short for the appended token"
    ar = np.arange(bs, dtype=token_positions.dtype)
    for r, (l, b, h) in enumerate(idx):
        blk, m = int(free[r]), int(ctx_old[l, b, h] // bs)
        block_tables[l, sel[b], h, m] = blk
        seq_index_by_block[blk] = sel[b]
        layer_index_by_block[blk] = l
        head_index_by_block[blk] = h
        logical_block_num_by_block[blk] = m
        token_positions[blk] = int(last_token_position[b]) + ar
    free_mask[free] = False
    if write_token_position:
        for l in range(L):
            for b, s_ in enumerate(sel):
                for h in range(H):
                    c = int(ctx_old[l, b, h])
                    token_positions[block_tables[l, s_, h, c // bs], c % bs] = int(last_token_position[b])
    context_lens[:, sel, :] += 1
    return n


def add_sequence(block_tables, context_lens, seq_slot, seq_len, free_mask, seq_index_by_block,
                 layer_index_by_block, head_index_by_block, logical_block_num_by_block, token_positions,
                 block_size):
    """A prefill sequence's first allocation, in place on every argument; returns its prefill slot mapping
    ``[L, seq_len, H]`` int64.

    BlockSpaceManagerKVC._add_sequence                vllm/kvcompress/block_manager.py:196-222
    ParallelBlockAllocator.allocate                   vllm/kvcompress/block_manager.py:103-110
    BlockStateView.get_allocated_block_metadata       vllm/kvcompress/block.py:414-446
    CompressionMetrics.insert_metadata                vllm/kvcompress/metrics.py:344-361
    BlockStateView.get_prefill_slot_mapping           vllm/kvcompress/block.py:275-303
    ``ceil(seq_len / bs)`` blocks for every (layer, head): the lowest-numbered free blocks, reshaped
    ``[L, H, nblk]``; every head's context length becomes ``seq_len``; the blocks' metadata rows are written and
    their position rows are ``logical block * bs + arange(bs)`` (positions == logical indices: nothing was
    compressed yet); token t of (layer, head) goes to slot ``block[l, h, t // bs] * bs + t % bs``."""
    bs = block_size
    L, S, H, M = block_tables.shape
    cnt = (int(seq_len) + bs - 1) // bs
    total = L * H * cnt
    free = np.nonzero(free_mask)[0]
    if free.shape[0] < total:
        raise ValueError(f"Out of memory! Requested {total} out of {int(free.shape[0])} available blocks.")
    assert cnt <= M, "block table too short for the sequence"
    blocks = free[:total].reshape(L, H, cnt)
    free_mask[free[:total]] = False
    context_lens[:, seq_slot, :] = seq_len
    block_tables[:, seq_slot, :, :cnt] = blocks
    ar = np.arange(bs, dtype=token_positions.dtype)
    for l in range(L):
        for h in range(H):
            for j in range(cnt):
                blk = int(blocks[l, h, j])
                seq_index_by_block[blk] = seq_slot
                layer_index_by_block[blk] = l
                head_index_by_block[blk] = h
                logical_block_num_by_block[blk] = j
                token_positions[blk] = j * bs + ar
    t = np.arange(int(seq_len))
    slot_mapping = blocks.transpose(0, 2, 1)[:, t // bs, :].astype(np.int64) * bs + (t % bs)[None, :, None]
    return slot_mapping


# --------------------------------------------------------------------------------------
# F3  single-query paged attention with per-key metric output
#     csrc/attention/kvcompress_attention_kernels.cu:97-455 (main), :532-651 (v2 reduce);
#     twin: tests/kernels/test_kvcompress_attention.py:41-145
# --------------------------------------------------------------------------------------
def round_to_bf16(x):
    """float32 -> nearest-even bfloat16, returned as float32"""
    b = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((b + np.uint32(0x7FFF) + ((b >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000))
    return r.view(np.float32)


def paged_attention_decode(out, kv_metric_out, query, key_cache, value_cache, num_kv_heads, scale,
                           block_tables, context_lens, kv_position, last_position,
                           kv_metric_buffer_len, alibi_slopes=None, record_kv_metrics=True,
                           k_scale=1.0, v_scale=1.0, p_dtype=np.float16):
    """query [S,Hq,hd] (f16 bits as np.float16, or f32 holding bf16 values), key_cache
    [NB,hd/x,bs,x], value_cache [NB,hd,bs] (same element type, already dequantised for fp8),
    block_tables [S,Hkv,M] i32, context_lens [S,Hkv] i32, kv_position [NB,bs] i32,
    last_position / kv_metric_buffer_len [S] i32; out [S,Hq,hd] f32 (caller rounds to the
    output type), kv_metric_out [NB,bs,qpk] f32 written in place.

    Per (seq, query head): logits = scale * q.k (+ alibi_slope * (i - ctx + 1), .cu:265),
    p = exp(l - max) / (sum + 1e-6) (.cu:291-303), out = sum_i p_i v_i with p rounded to
    the value type first (.cu:332-420 `from_float(logits_vec, ...)`; twin :53
    `attn_weights.to(value.dtype)`).  p is stored at [phys_block, offset, q % qpk] only
    where kv_position <= last_position - kv_metric_buffer_len (.cu:124, :305-312); every
    other entry of kv_metric_out is left untouched (v1 semantics; the v2 reduce kernel
    copies whatever its tmp buffer holds for such slots, :563-568 - an artefact that is
    not reproduced)."""
    S, Hq, hd = query.shape
    qpk = Hq // num_kv_heads
    NB, _, bs = value_cache.shape
    kc = key_cache.astype(np.float32).transpose(0, 2, 1, 3).reshape(NB, bs, hd)    # [NB,bs,hd]
    vc = value_cache.astype(np.float32).transpose(0, 2, 1)                         # [NB,bs,hd]
    q32 = query.astype(np.float32)
    for s in range(S):
        max_pos = int(last_position[s]) - int(kv_metric_buffer_len[s])
        for h in range(num_kv_heads):
            ctx = int(context_lens[s, h])
            if ctx <= 0:
                continue
            nblk = (ctx + bs - 1) // bs
            blocks = block_tables[s, h, :nblk].astype(np.int64)
            keys = (kc[blocks].reshape(nblk * bs, hd)[:ctx] * np.float32(k_scale)).astype(np.float32)
            vals = (vc[blocks].reshape(nblk * bs, hd)[:ctx] * np.float32(v_scale)).astype(np.float32)
            phys = (blocks[:, None] * bs + np.arange(bs)[None, :]).reshape(-1)[:ctx]
            rec = kv_position.reshape(-1)[phys] <= max_pos
            for qo in range(qpk):
                qh = h * qpk + qo
                logits = (np.float32(scale) * (keys @ q32[s, qh])).astype(np.float32)
                if alibi_slopes is not None and alibi_slopes[qh] != 0:
                    logits = logits + np.float32(alibi_slopes[qh]) * (
                        np.arange(ctx, dtype=np.float32) - np.float32(ctx - 1))
                e = np.exp(logits - logits.max(), dtype=np.float32)
                p = (e * (np.float32(1.0) / (e.sum(dtype=np.float32) + np.float32(1e-6)))).astype(np.float32)
                if p_dtype == "bf16":
                    pv = round_to_bf16(p)
                elif p_dtype is not None:
                    pv = p.astype(p_dtype).astype(np.float32)
                else:
                    pv = p
                out[s, qh] = pv @ vals
                if record_kv_metrics:
                    flat = kv_metric_out.reshape(-1, qpk)
                    flat[phys[rec], qo] = p[rec]


def naive_kvc_attention(query, key, prompt_lens, scale, kv_metric_buffer_len, n_observed=32,
                        max_observed_block_size=4096, use_l2=True, use_average=False,
                        use_maxpool=True, logit_round="f16"):
    """The whole of ``_naive_kvc_attention`` + ``_naive_kvc_masked_attention``
    (vllm/attention/backends/flash_attn.py:1120-1211) in NumPy: ``query`` / ``key``
    [T, Hq, hd] as float32 arrays holding the fp16 / bf16 input values.  The reference's
    einsum returns the input type, so the logits are rounded to it (``logit_round``: "f16",
    "bf16" or None) before ``scale *`` and the fp32 softmax (:1189); masked logits get the
    type's most negative finite value added, which the softmax turns into exact zeros.
    Returns kv_metric_output [T, Hq] float32."""
    T, Hq, hd = query.shape
    out = np.zeros((T, Hq), np.float32)
    fmin = np.float32(-65504.0) if logit_round != "bf16" else np.float32(-3.3895313892515355e38)
    start = 0
    for i, plen in enumerate(int(x) for x in prompt_lens):
        end = start + plen
        first = end - min(plen, int(n_observed))
        kk = key[start:end].astype(np.float32)
        for l in range(first, end, int(max_observed_block_size)):
            qq = query[l:min(l + int(max_observed_block_size), end)].astype(np.float32)
            nq = qq.shape[0]
            q_off = l - start
            w = np.einsum("qhd,khd->hqk", qq, kk, dtype=np.float32)
            if logit_round == "f16":
                w = w.astype(np.float16).astype(np.float32)
            elif logit_round == "bf16":
                w = round_to_bf16(w)
            w = (np.float32(scale) * w).astype(np.float32)
            masked = (np.arange(plen)[None, :] - np.arange(nq)[:, None]) > q_off   # triu(q_off + 1)
            w = w + np.where(masked, fmin, np.float32(0.0))[None].astype(np.float32)
            w = w - w.max(axis=-1, keepdims=True)
            e = np.exp(w, dtype=np.float32)
            probs = (e / e.sum(axis=-1, keepdims=True, dtype=np.float32)).astype(np.float32)
            prefill_metric_epilogue(out[start:end], probs, q_off, int(kv_metric_buffer_len[i]),
                                    use_l2, use_average, use_maxpool)
        start = end
    return out
