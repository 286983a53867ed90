"""Parity at the shapes of the other BASELINE.json configs (scaled to oracle-friendly
sizes): config 3 (continual compression, batch > 1, protected_window 32, several
iterations with the block-state transitions in between), config 4 (80-layer 70B shape),
config 5 (fp8 cache, block_size 32, full-query-range prefill metrics)."""
import copy

import numpy as np
import pytest
import torch

from oracle import kvc_oracle as orc
from tests.helpers import oracle_pipeline
from vllm_kvcompress_amd import _custom_ops as ops
from vllm_kvcompress_amd.harness import device as hdev
from vllm_kvcompress_amd.harness import synth
from vllm_kvcompress_amd.harness.engine_sim import EngineSim

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _gpu_step(st, evicted, k_t, v_t, mode):
    ds = hdev.upload(st, DEV, mode=mode)
    eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, st, evicted)
    ops.execute_cache_moves(k_t, v_t, ds.cm.metrics, ds.cm.token_positions, cmi, cmc,
                            ds.evicted_kv_offsets, 1, 16)
    return dict(eli=eli.cpu().numpy(), ekc=ekc.cpu().numpy(), ebc=ebc.cpu().numpy(),
                cmi=cmi.cpu().numpy(), cmc=cmc.cpu().numpy(),
                metrics=ds.cm.metrics.cpu().numpy(), positions=ds.cm.token_positions.cpu().numpy())


@pytest.mark.parametrize("mode", ["per_sequence", "reference"])
def test_config3_continual_compression(mode):
    """4 sequences, cap of 48 tokens per head, protected_window 32, compression every step
    for 40 decode steps; GPU and oracle carry their own state and must stay identical."""
    L, H, bs, hd, cap = 2, 4, 16, 128, 48
    seq_lens = [200, 130, 77, 161]
    st_o = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=3,
                            protected=32, spare_block_frac=0.5)
    st_g = copy.deepcopy(st_o)
    sim_o, sim_g = EngineSim(st_o, seq_lens), EngineSim(st_g, seq_lens)
    k_np, v_np = synth.make_caches_u16(3, st_o.num_blocks, hd, bs)
    k_t, v_t = torch.from_numpy(k_np.copy()).to(DEV), torch.from_numpy(v_np.copy()).to(DEV)
    rng = np.random.default_rng(0)
    for it in range(40):
        evicted = [synth.evict_block_count(context_lens_lh=st_o.context_lens[:, b, :],
                                           seq_len=int(sim_o.seq_lens[b]), block_size=bs,
                                           protected_window_size=32, max_cache_tokens=cap)
                   for b in range(len(seq_lens))]
        want = oracle_pipeline(st_o, evicted, k_np, v_np, mode=mode)
        got = _gpu_step(st_g, evicted, k_t, v_t, mode)
        for key in ("eli", "ekc", "ebc", "cmi", "cmc", "metrics", "positions"):
            np.testing.assert_array_equal(got[key], want[key], err_msg=f"iter {it}: {key}")
        k_np, v_np = want["k"], want["v"]
        np.testing.assert_array_equal(k_t.cpu().numpy(), k_np, err_msg=f"iter {it}: K")
        np.testing.assert_array_equal(v_t.cpu().numpy(), v_np, err_msg=f"iter {it}: V")
        # carry the compacted metric/position stores, free blocks, append one token per head
        for st, sim, res in ((st_o, sim_o, want), (st_g, sim_g, got)):
            st.metrics, st.token_positions = res["metrics"].copy(), res["positions"].copy()
            sim.apply_compression(res["ekc"], res["ebc"])
            sim.append_token()
        # decode attention mass lands on live slots (same increments on both sides; kept
        # tie-free by adding a distinct tiny rank term)
        inc = rng.random(st_o.metrics.shape).astype(np.float32)
        st_o.metrics = (st_o.metrics + inc).astype(np.float32)
        st_g.metrics = (st_g.metrics + inc).astype(np.float32)
        new_k, new_v = synth.make_caches_u16(100 + it, st_o.num_blocks, hd, bs)
        # the appended KVs: overwrite the just-written slot of every head with fresh bytes
        for b in range(len(seq_lens)):
            for l in range(L):
                for h in range(H):
                    ctx = int(st_o.context_lens[l, b, h]) - 1
                    blk = int(st_o.block_tables[l, b, h, ctx // bs])
                    k_np[blk, :, ctx % bs, :] = new_k[blk, :, ctx % bs, :]
                    v_np[blk, :, ctx % bs] = new_v[blk, :, ctx % bs]
        k_t.copy_(torch.from_numpy(k_np))
        v_t.copy_(torch.from_numpy(v_np))
        assert np.array_equal(st_o.context_lens, st_g.context_lens)
    if mode == "per_sequence":
        # every sequence is held at its cap; in "reference" mode the batch>1 quirk lets later
        # sequences under-evict -- reproduced above bit for bit
        assert int(st_o.context_lens.max()) <= cap + bs


def test_config4_70b_shape():
    """80 layers x 8 KV heads, two sequences, per_sequence mode (the sharded multi-GPU case)"""
    st = synth.make_state(num_layers=80, num_kv_heads=8, block_size=16, seq_lens=[260, 145], seed=4,
                          protected=32)
    bs = 16
    nblk = ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)
    evicted = [int(n) // 2 for n in nblk]
    k, v = synth.make_caches_u16(4, st.num_blocks, 128, bs)
    want = oracle_pipeline(st, evicted, k, v, mode="per_sequence")
    kt, vt = torch.from_numpy(k.copy()).to(DEV), torch.from_numpy(v.copy()).to(DEV)
    got = _gpu_step(st, evicted, kt, vt, "per_sequence")
    for key in ("eli", "ekc", "ebc", "cmi", "cmc", "metrics", "positions"):
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)
    np.testing.assert_array_equal(kt.cpu().numpy(), want["k"])
    np.testing.assert_array_equal(vt.cpu().numpy(), want["v"])


def test_config5_fp8_block32():
    """1-byte cache elements (x = 16), block_size 32: the reference CUDA kernel rejects this
    shape; parity is byte-copy semantics of the Python twin."""
    L, H, bs, hd = 4, 8, 32, 128
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=[2100], seed=5,
                          protected=32)
    nblk = ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)
    evicted = [int(nblk[0]) * 3 // 4]
    rng = np.random.default_rng(5)
    k = rng.integers(0, 256, size=(st.num_blocks, hd // 16, bs, 16), dtype=np.uint8)
    v = rng.integers(0, 256, size=(st.num_blocks, hd, bs), dtype=np.uint8)
    want = oracle_pipeline(st, evicted, k, v, mode="reference")
    kt, vt = torch.from_numpy(k.copy()).to(DEV), torch.from_numpy(v.copy()).to(DEV)
    got = _gpu_step(st, evicted, kt, vt, "reference")
    for key in ("eli", "ekc", "ebc", "cmi", "cmc", "metrics", "positions"):
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)
    np.testing.assert_array_equal(kt.cpu().numpy(), want["k"])
    np.testing.assert_array_equal(vt.cpu().numpy(), want["v"])


def test_config5_full_query_range_prefill_metrics():
    """prefill_metric_collection_block_size smaller than the observed window: several query
    blocks, max-pooled per block before accumulation, then aggregate_prefill."""
    from vllm_kvcompress_amd.kvcompress.prefill import naive_kvc_attention
    torch.manual_seed(0)
    T, Hq, Hkv, hd, qblk = 300, 8, 2, 32, 64
    q = torch.randn((T, Hq, hd), device=DEV, dtype=torch.float16)
    kk = torch.randn((T, Hq, hd), device=DEV, dtype=torch.float16)
    scale = hd ** -0.5
    buf = torch.tensor([3], dtype=torch.int32)
    _, got = naive_kvc_attention(q, kk, None, [T], scale, buf, n_observed=T,
                                 max_observed_block_size=qblk, use_l2=True, use_average=False,
                                 use_maxpool=True)
    # float32 restatement of the reference loop (flash_attn.py:1122-1211) on the host
    qf, kf = q.float().cpu(), kk.float().cpu()
    want = np.zeros((T, Hq), dtype=np.float32)
    for l in range(0, T, qblk):
        qq = qf[l:l + qblk]
        w = scale * torch.einsum("qhd,khd->hqk", qq, kf)
        nq = qq.shape[0]
        mask = torch.triu(torch.ones(nq, T), diagonal=l + 1) * torch.finfo(torch.float16).min
        probs = torch.softmax(w + mask, dim=-1).numpy()
        orc.prefill_metric_epilogue(want, probs, l, 3, True, False, True)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-3, atol=1e-5)


def test_config5_aggregation_at_its_own_size():
    """BASELINE configs[4]'s aggregation half at ITS size: 65 536 keys, query blocks of
    prefill_metric_collection_block_size = 1024, 32 query heads (reference
    vllm/attention/backends/flash_attn.py:1122-1211).  (i) the A2c epilogue behind the library
    GEMM + softmax and the fused collector (F4) agree on the last two query blocks (the tolerance
    of tests/test_gpu_prefill_fused.py: the unfused path rounds its logits through fp16 einsum
    output like the reference, the fused one reproduces that rounding); (ii) full query range, L1
    metrics, no pooling, buffer 0: every query row hands out exactly 1, so a head's metrics sum to
    the number of observed queries; (iii) SURVEY Q10: pooling happens per query block BEFORE the
    accumulation -- two adjacent blocks pooled separately are not what one block of twice the
    size gives, although their unpooled sums are identical."""
    import gc
    from vllm_kvcompress_amd.kvcompress.prefill import fused_kvc_attention, naive_kvc_attention
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 48 << 30:
        pytest.skip("needs ~48 GiB of free HBM (one 1024 x 65536 x 32 probability tile is 8.6 GB)")
    torch.manual_seed(5)
    K, Hq, Hk, hd, qb = 65536, 32, 8, 128, 1024
    q = (torch.randn(K, Hq, hd, device=DEV) * 0.7).half()
    kk = (torch.randn(K, Hk, hd, device=DEV) * 0.7).half()
    k_rep = kk.repeat_interleave(Hq // Hk, dim=1)
    scale = hd ** -0.5
    buf = torch.tensor([3], dtype=torch.int32)
    # (i) two query blocks at the end of the sequence, L2 metrics, pooled
    _, fused = fused_kvc_attention(q, kk, None, [K], scale, buf, n_observed=2 * qb, max_observed_block_size=qb)
    _, naive = naive_kvc_attention(q, k_rep, None, [K], scale, buf, n_observed=2 * qb, max_observed_block_size=qb)
    np.testing.assert_allclose(fused.cpu().numpy(), naive.cpu().numpy(), rtol=1e-2, atol=1e-4)
    del naive, k_rep
    torch.cuda.empty_cache()
    # (iii) the same two blocks as ONE block of 2048 rows
    _, one = fused_kvc_attention(q, kk, None, [K], scale, buf, n_observed=2 * qb, max_observed_block_size=2 * qb)
    _, two_raw = fused_kvc_attention(q, kk, None, [K], scale, buf, n_observed=2 * qb, max_observed_block_size=qb,
                                     use_maxpool=False)
    _, one_raw = fused_kvc_attention(q, kk, None, [K], scale, buf, n_observed=2 * qb, max_observed_block_size=2 * qb,
                                     use_maxpool=False)
    np.testing.assert_allclose(two_raw.cpu().numpy(), one_raw.cpu().numpy(), rtol=2e-4, atol=1e-7)
    assert bool((fused >= one * (1 - 1e-4)).all())    # max(a) + max(b) >= max(a + b)
    assert float(((fused - one) / fused.clamp_min(1e-30)).max()) > 0.01     # and it does matter (values ~1e-6)
    # (ii) full query range (64 query blocks), L1, unpooled, buffer 0
    _, l1 = fused_kvc_attention(q, kk, None, [K], scale, torch.zeros(1, dtype=torch.int32), n_observed=K,
                                max_observed_block_size=qb, use_l2=False, use_maxpool=False)
    sums = l1.double().sum(0).cpu().numpy()
    np.testing.assert_allclose(sums, np.full(Hq, float(K)), rtol=1e-4)
    assert bool((l1 >= 0).all())


@pytest.mark.parametrize("case", [0, 1, 2])
def test_free_compressed_blocks_device(case):
    """F2 on device vs the reference-generated block-state vectors"""
    from tests.helpers import load_golden
    from vllm_kvcompress_amd.kvcompress.block_state import free_compressed_blocks
    g = load_golden(f"blockstate_{case}")
    NB = int(g["num_blocks"])
    ctx = torch.from_numpy(g["context_lens"].copy()).to(DEV)
    bt = torch.from_numpy(g["block_tables"].copy()).to(DEV)
    seq_by = torch.zeros(NB, dtype=torch.int32, device=DEV)
    free_mask = torch.zeros(NB, dtype=torch.bool, device=DEV)
    freed = free_compressed_blocks(bt, ctx, [int(s) for s in g["seq_indices"]],
                                   torch.from_numpy(g["freed_block_count"]).to(DEV), seq_by,
                                   int(g["block_size"]), free_mask)
    np.testing.assert_array_equal(freed.cpu().numpy(), g["ref_freed_blocks"])
    np.testing.assert_array_equal(ctx.cpu().numpy(), g["ref_context_lens"])
    want = np.zeros(NB, dtype=bool)
    want[g["ref_freed_blocks"]] = True
    np.testing.assert_array_equal(free_mask.cpu().numpy(), want)
    np.testing.assert_array_equal(seq_by.cpu().numpy() == -1, want)


@pytest.mark.parametrize("zero_fill", [True, False])
def test_compression_scheduler_mirror_end_to_end(zero_fill):
    """CompressionScheduler (host glue mirror) over the FULL block state with a subset of
    slots compressing, max_kv_per_compression cut-off, staleness order, persistent move
    workspace and the device block-state update -- against the oracle driven by hand.
    ``zero_fill=False`` is the opt-out of the reference's whole-workspace clear: the rows a
    consumer reads (offset_g .. offset_g + count_g) and everything downstream are unchanged."""
    from vllm_kvcompress_amd.kvcompress.scheduler import CompressionScheduler, SeqCompressionRequest
    L, H, bs, hd = 2, 4, 16, 128
    seq_lens = [200, 130, 77, 161, 90]
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=31,
                          protected=32, spare_block_frac=0.3)
    ds = hdev.upload(st, DEV, mode="per_sequence")
    k_np, v_np = synth.make_caches_u16(31, st.num_blocks, hd, bs)
    k_t, v_t = torch.from_numpy(k_np.copy()).to(DEV), torch.from_numpy(v_np.copy()).to(DEV)
    bt_full = torch.from_numpy(st.block_tables).to(DEV)
    ctx_full = torch.from_numpy(st.context_lens.copy()).to(DEV)
    free_mask = torch.from_numpy(st.seq_index_by_block < 0).to(DEV)
    total_rows = 40000
    sched = CompressionScheduler(bs, L, H, total_rows, ds.cm, device=DEV, zero_fill_moves=zero_fill)
    sched.cache_move_indices.fill_(-7)                 # stale junk from earlier iterations
    nblk = ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)
    kvs = st.context_lens.astype(np.int64).sum(0).sum(-1)
    reqs = [SeqCompressionRequest(seq_id=100 + i, slot_index=i, seq_len=seq_lens[i],
                                  block_count=int(nblk[i]), kv_count=int(kvs[i]),
                                  max_cache_tokens=48, protected_window_size=32)
            for i in (4, 0, 3, 1)]                     # slot 2 does not compress
    reqs[1].max_cache_tokens = 10 ** 6                 # slot 0: nothing to evict -> skipped
    out = sched.schedule_compression(reqs, bt_full, ctx_full, free_mask=free_mask)
    assert out is not None and out.slot_indices == [1, 3, 4]
    ops.execute_cache_moves(k_t, v_t, ds.cm.metrics, ds.cm.token_positions, out.cache_moves.index,
                            out.cache_moves.count, out.cache_moves.offsets, 1, 16)
    # ---- the same by hand on the oracle
    sel = [1, 3, 4]
    ctx = np.ascontiguousarray(st.context_lens[:, sel, :])
    sub = synth.PagedState(
        block_size=bs, num_layers=L, num_kv_heads=H, num_seqs=3, num_blocks=st.num_blocks,
        metrics=st.metrics, token_positions=st.token_positions,
        seq_index_by_block=st.seq_index_by_block, layer_index_by_block=st.layer_index_by_block,
        head_index_by_block=st.head_index_by_block,
        logical_block_num_by_block=st.logical_block_num_by_block, context_lens=ctx,
        block_tables=np.ascontiguousarray(st.block_tables[:, sel]),
        hanging_token_count=synth.hanging_tokens(ctx.transpose(1, 0, 2), bs),
        evicted_kv_offsets=synth.kv_offsets(ctx, bs), seq_indices=sel,
        seq_positions=np.ascontiguousarray(st.seq_positions[sel]), protected=[32, 32, 32])
    evicted = [synth.evict_block_count(context_lens_lh=st.context_lens[:, i, :], seq_len=seq_lens[i],
                                       block_size=bs, protected_window_size=32, max_cache_tokens=48)
               for i in sel]
    want = oracle_pipeline(sub, evicted, k_np, v_np, mode="per_sequence")
    N = sub.total_slots
    np.testing.assert_array_equal(out.cache_moves.count.cpu().numpy(), want["cmc"])
    np.testing.assert_array_equal(out.cache_moves.offsets.cpu().numpy(), sub.evicted_kv_offsets)
    got_idx = out.cache_moves.index.cpu().numpy()
    if zero_fill:
        np.testing.assert_array_equal(got_idx[:N], want["cmi"])
        assert not got_idx[N:].any()                   # rest of the workspace zero-filled
    else:
        offs, cnt = sub.evicted_kv_offsets.reshape(-1), want["cmc"].reshape(-1)
        for o, c in zip(offs, cnt):
            np.testing.assert_array_equal(got_idx[o:o + c], want["cmi"][o:o + c])
    np.testing.assert_array_equal(k_t.cpu().numpy(), want["k"])
    np.testing.assert_array_equal(v_t.cpu().numpy(), want["v"])
    np.testing.assert_array_equal(ds.cm.metrics.cpu().numpy(), want["metrics"])
    for i, sid in enumerate(out.seq_ids):
        np.testing.assert_array_equal(out.freed_block_count[sid].cpu().numpy(), want["ebc"][i])
    # block-state side
    ctx_want = st.context_lens.copy()
    seq_by = st.seq_index_by_block.copy()
    fm = st.seq_index_by_block < 0
    freed_want = orc.free_compressed_blocks(st.block_tables, ctx_want, sel, want["ebc"], seq_by, bs, fm)
    np.testing.assert_array_equal(out.freed_blocks.cpu().numpy(), freed_want)
    np.testing.assert_array_equal(ctx_full.cpu().numpy(), ctx_want)
    np.testing.assert_array_equal(ds.cm.seq_index_by_block.cpu().numpy(), seq_by)
    np.testing.assert_array_equal(free_mask.cpu().numpy(), fm)
    assert int(ctx_full[:, sel].max()) <= 48


def test_compression_step_is_graph_capturable():
    """S1 + S2 + S3 enqueue kernels only (no host sync when N is passed, no allocation inside
    the library): the whole step records into a HIP graph and a replay reproduces the eager
    result bit for bit.  (On MI355X the replay is not faster - the eager step is already
    GPU-bound - the point is that an engine that captures its iteration can include it.)"""
    B = 3
    st = synth.make_state(num_layers=4, num_kv_heads=2, block_size=16, seq_lens=[700, 333, 1200], seed=4,
                          protected=[8, 3, 20])
    evicted = [20, 9, 37]
    ds = hdev.upload(st, DEV, num_queries_per_kv=1, mode="per_sequence")
    k, v = synth.make_caches_u16(4, st.num_blocks, 128, 16)
    k_cache = torch.from_numpy(k.copy()).to(DEV).view(torch.float16)
    v_cache = torch.from_numpy(v.copy()).to(DEV).view(torch.float16)
    k0, v0 = k_cache.clone(), v_cache.clone()
    m0, p0 = ds.cm.metrics.clone(), ds.cm.token_positions.clone()
    N = st.total_slots
    wm, wp = m0.clone(), p0.clone()
    cmi = torch.zeros((N, 2), dtype=torch.int32, device=DEV)
    cmc = torch.zeros((B, 4, 2), dtype=torch.int32, device=DEV)
    ev = torch.tensor(evicted, dtype=torch.int32, device=DEV)
    seq_idx, prot = list(st.seq_indices), list(st.protected)
    out = {}

    def step():
        eli, ekc, ebc = ds.cm.schedule_evictions(seq_idx, ds.seq_positions, ev, ds.context_lens,
                                                 ds.hanging_token_count, ds.evicted_kv_offsets, prot,
                                                 total_slots=N)
        ops.schedule_cache_moves(cmi, cmc, eli, ekc, ds.evicted_kv_offsets, ds.block_tables,
                                 ds.context_lens, 16)
        ops.execute_cache_moves(k_cache, v_cache, wm, wp, cmi, cmc, ds.evicted_kv_offsets, 1, 16)
        out["eli"], out["ekc"], out["ebc"] = eli, ekc, ebc

    step()                                           # eager: warms workspaces and small caches
    torch.cuda.synchronize()
    want = [t.clone() for t in (out["eli"], out["ekc"], out["ebc"], cmi, cmc, k_cache, v_cache, wm, wp)]
    # reset the mutable state, capture, replay
    k_cache.copy_(k0); v_cache.copy_(v0); wm.copy_(m0); wp.copy_(p0); cmi.zero_(); cmc.zero_()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    k_cache.copy_(k0); v_cache.copy_(v0); wm.copy_(m0); wp.copy_(p0); cmi.zero_(); cmc.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    k_cache.copy_(k0); v_cache.copy_(v0); wm.copy_(m0); wp.copy_(p0); cmi.zero_(); cmc.zero_()
    # several replays: a schedule clears its counters first thing, and on ROCm 7.2 a hipMemsetAsync node of
    # a graph does that on the FIRST replay only -- the library fills with kernels of its own for that reason
    for rep in range(3):
        k_cache.copy_(k0); v_cache.copy_(v0); wm.copy_(m0); wp.copy_(p0); cmi.zero_(); cmc.zero_()
        graph.replay()
        torch.cuda.synchronize()
        got = [out["eli"], out["ekc"], out["ebc"], cmi, cmc, k_cache, v_cache, wm, wp]
        for a, b in zip(got, want):          # caches hold random bit patterns (NaNs): compare bits
            if a.dtype == torch.float16:
                a, b = a.view(torch.int16), b.view(torch.int16)
            assert torch.equal(a, b), rep


def test_steady_state_step_with_kept_buffers_is_graph_capturable():
    """the continual step as an engine would hold it -- the small-eviction schedule with its output
    list in the kept buffer, the move table registered (only the previous call's rows are cleared),
    execute_cache_moves on the move scheduler's plan -- recorded into a HIP graph and replayed
    several times: the dirty maps live on the device, so every replay leaves exactly what the eager
    step leaves (the whole table and the whole list compared, not just the rows a consumer reads)"""
    L, H, bs, B, cap = 2, 4, 16, 3, 512
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=[3 * cap] * B, seed=11,
                          protected=bs + 1, steady_cap=cap, spare_block_frac=0.05)
    evicted = [synth.evict_block_count(context_lens_lh=st.context_lens[:, b, :], seq_len=3 * cap, block_size=bs,
                                       protected_window_size=bs + 1, max_cache_tokens=cap) for b in range(B)]
    want = oracle_pipeline(st, evicted, mode="per_sequence")
    ds = hdev.upload(st, DEV, mode="per_sequence")
    ds.cm.schedule_path = 0
    k, v = synth.make_caches_u16(4, st.num_blocks, 128, bs)
    k_cache = torch.from_numpy(k.copy()).to(DEV).view(torch.float16)
    v_cache = torch.from_numpy(v.copy()).to(DEV).view(torch.float16)
    wm, wp = ds.cm.metrics.clone(), ds.cm.token_positions.clone()
    N = st.total_slots
    rows = N + 777
    table = ops.track_move_table(torch.empty((rows, 2), dtype=torch.int32, device=DEV))
    table.fill_(9)
    cmc = torch.zeros((B, L, H), dtype=torch.int32, device=DEV)
    seq_idx, prot = list(st.seq_indices), list(st.protected)
    out = {}

    def step():
        out.clear()
        eli, ekc, ebc = ds.cm.schedule_evictions(seq_idx, ds.seq_positions, evicted, ds.context_lens,
                                                 ds.hanging_token_count, ds.evicted_kv_offsets, prot, total_slots=N)
        ops.schedule_cache_moves(table, cmc, eli, ekc, ds.evicted_kv_offsets, ds.block_tables, ds.context_lens, bs)
        assert ops._plan_of(k_cache, table, cmc, ds.evicted_kv_offsets, B * L * H, bs) is not None
        ops.execute_cache_moves(k_cache, v_cache, wm, wp, table, cmc, ds.evicted_kv_offsets, 1, 16)
        out["eli"], out["ekc"] = eli, ekc

    def check(tag):
        torch.cuda.synchronize()
        expect = np.zeros((rows, 2), np.int32)
        expect[:N] = want["cmi"]
        got_t = table.cpu().numpy()
        bad = np.nonzero((got_t != expect).any(axis=1))[0]
        assert bad.size == 0, f"{tag}: {bad.size} rows of the move table differ, first {bad[:6]}"
        np.testing.assert_array_equal(out["eli"].cpu().numpy(), want["eli"], err_msg=f"{tag}: evicted list")
        np.testing.assert_array_equal(out["ekc"].cpu().numpy(), want["ekc"], err_msg=tag)
        np.testing.assert_array_equal(cmc.cpu().numpy(), want["cmc"], err_msg=tag)

    step(); check("eager 1")                         # sets the table and the kept buffer up (full fills)
    assert ds.cm.last_schedule_path() == "small_eviction"
    step(); check("eager 2")                         # ... and this one runs on the maps
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    check("side stream")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    captured = dict(out)                             # the tensors the captured step writes
    for rep in range(3):
        table[rep * 5 + 1] = 4                       # (a foreign write lands in rows the map does not know: put
        table[rep * 5 + 1] = 0                       #  the value back -- a replay cannot notice version counters)
        graph.replay()
        out.update(captured)
        check(f"replay {rep}")
    # the compaction moved what the oracle moves (the same moves every time: sources are never written)
    got = oracle_pipeline(st, evicted, k, v, mode="per_sequence")
    np.testing.assert_array_equal(k_cache.view(torch.int16).cpu().numpy(), got["k"].view(np.int16))
    np.testing.assert_array_equal(v_cache.view(torch.int16).cpu().numpy(), got["v"].view(np.int16))


@pytest.mark.parametrize("mode", ["per_sequence", "reference"])
def test_lean_outputs_leave_everything_downstream_unchanged(mode):
    """CompressionMetrics.lean_outputs (extension): no MAX_INT padding behind a head's evicted
    indices and no defensive clear of the key scratch.  Counts, the evicted indices a consumer
    reads, the move schedule and the compacted caches are identical to the default."""
    st = synth.make_state(num_layers=3, num_kv_heads=2, block_size=16, seq_lens=[900, 333, 1500], seed=9,
                          protected=[5, 40, 12], compressed=True)
    evicted = [25, 7, 60]
    k, v = synth.make_caches_u16(9, st.num_blocks, 128, 16)
    res = {}
    for lean in (False, True):
        ds = hdev.upload(st, DEV, num_queries_per_kv=1, mode=mode)
        ds.cm.lean_outputs = lean
        ev = torch.tensor(evicted, dtype=torch.int32, device=DEV)
        # poison the scratch so that a missing clear would show
        from vllm_kvcompress_amd import _custom_ops as _ops
        for buf in _ops._WORKSPACES.values():
            buf.fill_(0x5A)
        eli, ekc, ebc = ds.cm.schedule_evictions(list(st.seq_indices), ds.seq_positions, ev, ds.context_lens,
                                                 ds.hanging_token_count, ds.evicted_kv_offsets,
                                                 list(st.protected), total_slots=st.total_slots)
        cmi = torch.zeros((st.total_slots, 2), dtype=torch.int32, device=DEV)
        cmc = torch.zeros((3, 3, 2), dtype=torch.int32, device=DEV)
        ops.schedule_cache_moves(cmi, cmc, eli, ekc, ds.evicted_kv_offsets, ds.block_tables,
                                 ds.context_lens, 16)
        kc = torch.from_numpy(k.copy()).to(DEV)
        vc = torch.from_numpy(v.copy()).to(DEV)
        ops.execute_cache_moves(kc.view(torch.float16), vc.view(torch.float16), ds.cm.metrics,
                                ds.cm.token_positions, cmi, cmc, ds.evicted_kv_offsets, 1, 16)
        torch.cuda.synchronize()
        res[lean] = dict(eli=eli.cpu().numpy(), ekc=ekc.cpu().numpy(), ebc=ebc.cpu().numpy(),
                         cmi=cmi.cpu().numpy(), cmc=cmc.cpu().numpy(), k=kc.cpu().numpy(),
                         v=vc.cpu().numpy(), m=ds.cm.metrics.cpu().numpy(),
                         p=ds.cm.token_positions.cpu().numpy())
    a, b = res[False], res[True]
    for key in ("ekc", "ebc", "cmi", "cmc", "k", "v", "m", "p"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    offs, cnt = st.evicted_kv_offsets.reshape(-1), a["ekc"].reshape(-1)
    for o, c in zip(offs, cnt):
        np.testing.assert_array_equal(a["eli"][o:o + c], b["eli"][o:o + c])
    assert (a["eli"] == 2147483000).any()              # the default pads with MAX_INT


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_append_slots_device(case):
    """the append side of F2 on device vs vectors produced by the reference's own
    _append_to_sequence_batch / ParallelBlockAllocator / BlockState / insert_metadata"""
    from types import SimpleNamespace
    from tests.helpers import load_golden
    from vllm_kvcompress_amd.kvcompress.block_state import append_slots
    g = load_golden(f"append_{case}")
    t = lambda k: torch.from_numpy(g[k].copy()).to(DEV)
    bt, ctx, fm = t("block_tables"), t("context_lens"), t("free_mask")
    cm = SimpleNamespace(seq_index_by_block=t("seq_index_by_block"), layer_index_by_block=t("layer_index_by_block"),
                         head_index_by_block=t("head_index_by_block"),
                         logical_block_num_by_block=t("logical_block_num_by_block"),
                         token_positions=t("token_positions"))
    n = append_slots(bt, ctx, [int(s) for s in g["seq_indices"]], [int(p) for p in g["last_token_position"]],
                     fm, cm, int(g["block_size"]))
    assert n == int(g["free_mask"].sum()) - int(g["ref_free_count"])
    for got, ref in ((bt, "ref_block_tables"), (ctx, "ref_context_lens"), (fm, "ref_free_mask"),
                     (cm.seq_index_by_block, "ref_seq_index_by_block"),
                     (cm.layer_index_by_block, "ref_layer_index_by_block"),
                     (cm.head_index_by_block, "ref_head_index_by_block"),
                     (cm.logical_block_num_by_block, "ref_logical_block_num_by_block"),
                     (cm.token_positions, "ref_token_positions")):
        np.testing.assert_array_equal(got.cpu().numpy(), g[ref], err_msg=ref)
    # out of blocks: ValueError like ParallelBlockAllocator.allocate, and nothing is modified
    bt2, ctx2 = t("block_tables"), t("context_lens")
    fm2 = torch.zeros_like(fm)
    fm2[:max(n - 1, 0)] = True
    before = (bt2.clone(), ctx2.clone(), fm2.clone(), cm.token_positions.clone())
    if n > 0:
        with pytest.raises(ValueError, match="Out of memory"):
            append_slots(bt2, ctx2, [int(s) for s in g["seq_indices"]], [int(p) for p in g["last_token_position"]],
                         fm2, cm, int(g["block_size"]))
        for a, b in zip((bt2, ctx2, fm2, cm.token_positions), before):
            assert torch.equal(a, b)


@pytest.mark.parametrize("mode", ["per_sequence", "reference"])
def test_config3_continual_compression_all_state_on_device(mode):
    """The continual loop with NO host state on the device side: metric store, block tables,
    context lengths, free list and K/V stay in HBM for 30 decode steps; every transition is a
    device op (schedule -> moves -> compaction -> free_compressed_blocks -> append_slots ->
    reshape_and_cache of the new token).  The oracle side runs the reference-pinned NumPy
    restatements of the same transitions; all state is compared every step."""
    from vllm_kvcompress_amd.kvcompress.block_state import append_slots, free_compressed_blocks
    L, H, bs, hd, cap = 2, 4, 16, 128, 48
    seq_lens = [200, 130, 77, 161]
    B = len(seq_lens)
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=5,
                          protected=32, spare_block_frac=0.6)
    NB = st.num_blocks
    M = st.block_tables.shape[3] + 2
    bt_np = np.zeros((L, B, H, M), np.int32)
    bt_np[..., :st.block_tables.shape[3]] = st.block_tables
    ctx_np = st.context_lens.copy()
    free_np = st.seq_index_by_block < 0
    k_np, v_np = synth.make_caches_u16(5, NB, hd, bs)
    o = dict(metrics=st.metrics.copy(), pos=st.token_positions.copy(), seq=st.seq_index_by_block.copy(),
             lay=st.layer_index_by_block.copy(), head=st.head_index_by_block.copy(),
             lbn=st.logical_block_num_by_block.copy())
    # ---- device side
    ds = hdev.upload(st, DEV, mode=mode)
    cm = ds.cm
    bt = torch.from_numpy(bt_np.copy()).to(DEV)
    ctx = torch.from_numpy(ctx_np.copy()).to(DEV)
    fm = torch.from_numpy(free_np.copy()).to(DEV)
    k_t, v_t = torch.from_numpy(k_np.copy()).to(DEV), torch.from_numpy(v_np.copy()).to(DEV)
    slots = list(range(B))
    lens = np.asarray(seq_lens, np.int64).copy()
    rng = np.random.default_rng(1)
    bias = torch.zeros(H, device=DEV)
    for it in range(30):
        seq_pos = (lens - 1).astype(np.int32)
        # host policy on the (small) context-length table, like the reference's scheduler
        ctx_h = ctx.cpu().numpy()
        assert np.array_equal(ctx_h, ctx_np), f"iter {it}: context_lens"
        evicted = [synth.evict_block_count(context_lens_lh=ctx_h[:, b, :], seq_len=int(lens[b]), block_size=bs,
                                           protected_window_size=32, max_cache_tokens=cap) for b in range(B)]
        hang_np = synth.hanging_tokens(ctx_np.transpose(1, 0, 2), bs)
        offs_np = synth.kv_offsets(ctx_np, bs)
        N = int(((ctx_np.astype(np.int64) + bs - 1) // bs).sum()) * bs
        # ---- oracle
        eli, ekc, ebc = orc.schedule_evictions(
            metrics=o["metrics"], token_positions=o["pos"], seq_index_by_block=o["seq"],
            layer_index_by_block=o["lay"], head_index_by_block=o["head"], logical_block_num_by_block=o["lbn"],
            block_size=bs, num_layers=L, num_kv_heads=H, seq_indices=slots, seq_positions=seq_pos,
            evicted_blocks_per_seq=evicted, context_lens=ctx_np, hanging_token_count=hang_np,
            evicted_kv_offsets=offs_np, num_protected=[32] * B, mode=mode)
        cmi_o = np.zeros((N, 2), np.int32)
        cmc_o = np.zeros(ekc.shape, np.int32)
        orc.schedule_cache_moves(cmi_o, cmc_o, eli, ekc, offs_np, bt_np, ctx_np, bs)
        orc.execute_cache_moves(k_np, v_np, o["metrics"], o["pos"], cmi_o, cmc_o, offs_np)
        orc.free_compressed_blocks(bt_np, ctx_np, slots, ebc, o["seq"], bs, free_np)
        orc.append_slots(bt_np, ctx_np, slots, seq_pos, free_np, o["seq"], o["lay"], o["head"], o["lbn"],
                         o["pos"], bs, write_token_position=True)
        # ---- device (the derived per-step tensors are torch ops on device, as in the reference)
        hang = torch.from_numpy(hang_np).to(DEV)
        offs = torch.from_numpy(offs_np).to(DEV)
        g_eli, g_ekc, g_ebc = cm.schedule_evictions(slots, torch.from_numpy(seq_pos).to(DEV), evicted, ctx, hang,
                                                    offs, [32] * B, total_slots=N)
        cmi = torch.full((N, 2), 77, dtype=torch.int32, device=DEV)
        cmc = torch.empty_like(g_ekc)
        ops.schedule_cache_moves(cmi, cmc, g_eli, g_ekc, offs, bt, ctx, bs)
        ops.execute_cache_moves(k_t, v_t, cm.metrics, cm.token_positions, cmi, cmc, offs, 1, 16)
        free_compressed_blocks(bt, ctx, slots, g_ebc, cm.seq_index_by_block, bs, fm)
        append_slots(bt, ctx, slots, [int(p) for p in seq_pos], fm, cm, bs, write_token_position=True)
        for name, got, want in (("eli", g_eli, eli), ("ekc", g_ekc, ekc), ("ebc", g_ebc, ebc), ("cmi", cmi, cmi_o),
                                ("cmc", cmc, cmc_o), ("context_lens", ctx, ctx_np), ("free_mask", fm, free_np),
                                ("positions", cm.token_positions, o["pos"]), ("seq_index", cm.seq_index_by_block, o["seq"]),
                                ("lbn", cm.logical_block_num_by_block, o["lbn"])):
            np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f"iter {it}: {name}")
        # allocated part of the block tables (entries past a head's blocks are don't-care)
        nblk = (ctx_np + bs - 1) // bs
        live = np.arange(M)[None, None, None, :] < nblk[..., None]
        assert np.array_equal(bt.cpu().numpy()[live], bt_np[live]), f"iter {it}: block_tables"
        # ---- the new token's KV: reshape_and_cache per layer on device, NumPy writes on the oracle side
        key = rng.standard_normal((L, B, H, hd)).astype(np.float16)
        val = rng.standard_normal((L, B, H, hd)).astype(np.float16)
        c1 = ctx_np - 1
        slot_map = (np.take_along_axis(bt_np, (c1 // bs)[..., None], axis=3)[..., 0].astype(np.int64) * bs + c1 % bs)
        for l in range(L):
            ops.reshape_and_cache_kvc(torch.from_numpy(key[l]).to(DEV), torch.from_numpy(val[l]).to(DEV),
                                      k_t.view(torch.float16), v_t.view(torch.float16), cm.metrics,
                                      torch.from_numpy(slot_map[l].reshape(-1)).to(DEV), bias, "auto", 1.0, 1.0)
            orc.reshape_and_cache_kvc(key[l], val[l], k_np.view(np.float16), v_np.view(np.float16), o["metrics"],
                                      slot_map[l].reshape(-1), np.zeros(H, np.float32))
        inc = rng.random(o["metrics"].shape).astype(np.float32)
        o["metrics"] = (o["metrics"] + inc).astype(np.float32)
        cm.metrics.add_(torch.from_numpy(inc).to(DEV))
        np.testing.assert_array_equal(cm.metrics.cpu().numpy(), o["metrics"], err_msg=f"iter {it}: metrics")
        np.testing.assert_array_equal(k_t.cpu().numpy(), k_np, err_msg=f"iter {it}: K")
        np.testing.assert_array_equal(v_t.cpu().numpy(), v_np, err_msg=f"iter {it}: V")
        lens += 1
    if mode == "per_sequence":
        assert int(ctx_np.max()) <= cap + bs
