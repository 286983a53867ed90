"""Full-size checks on the GPU.

* C1 (Llama-3-8B shape, 4k cache, 1 M slots): complete oracle comparison (NumPy schedule +
  C move/compaction restatement), bit-exact.
* C2 (fp16, bs16, 32k cache, 8.4 M slots, 4 GiB of K/V) and C5 (fp8, bs32, 64k cache,
  16.8 M slots, 4 GiB): complete oracle comparison at their own size as well (a few seconds of
  oracle each), and size-independent properties, after the
  reference's own test (tests/kernels/test_kvcompress_eviction.py:906-924, 1107-1218,
  1224-1226): freed blocks == requested; per-head evicted indices ascending + padded; no
  evicted KV is a move source; EVERY surviving KV is bit-equal at its final slot (K/V are
  filled with a hash of the KV's identity, so the expectation is recomputed from the final
  position table); final positions are exactly the survivors; and the compaction took the
  block path for every destination block.
"""
import numpy as np
import pytest
import torch

from oracle import kvc_oracle as orc
from oracle import kvc_oracle_c as orc_c
from vllm_kvcompress_amd import _custom_ops as ops
from vllm_kvcompress_amd.harness import device as hdev
from vllm_kvcompress_amd.harness import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
L, H, BS, HD = 32, 8, 16, 128
MAX_INT = 2147483000


def _evict(st, keep, T):
    return [synth.evict_block_count(context_lens_lh=st.context_lens[:, b, :], seq_len=T + 1,
                                    block_size=BS, protected_window_size=st.protected[b],
                                    max_cache_tokens=int(T * keep)) for b in range(st.num_seqs)]


def test_c1_full_oracle_parity():
    T = 4096
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=BS, seq_lens=[T + 1], seed=0,
                          protected=32, spare_block_frac=0.02)
    evicted = _evict(st, 0.5, T)
    k_np, v_np = synth.make_caches_u16(0, st.num_blocks, HD, BS)
    eli, ekc, ebc = orc.schedule_evictions(
        metrics=st.metrics, token_positions=st.token_positions,
        seq_index_by_block=st.seq_index_by_block, layer_index_by_block=st.layer_index_by_block,
        head_index_by_block=st.head_index_by_block,
        logical_block_num_by_block=st.logical_block_num_by_block, block_size=BS, num_layers=L,
        num_kv_heads=H, seq_indices=st.seq_indices, seq_positions=st.seq_positions,
        evicted_blocks_per_seq=evicted, context_lens=st.context_lens,
        hanging_token_count=st.hanging_token_count, evicted_kv_offsets=st.evicted_kv_offsets,
        num_protected=st.protected, mode="reference")
    cmi = np.zeros((st.total_slots, 2), np.int32)
    cmc = np.zeros(ekc.shape, np.int32)
    orc_c.schedule_cache_moves(cmi, cmc, eli, ekc, st.evicted_kv_offsets,
                               np.ascontiguousarray(st.block_tables),
                               np.ascontiguousarray(st.context_lens), BS)
    wk, wv = np.ascontiguousarray(k_np.copy()), np.ascontiguousarray(v_np.copy())
    wm, wp = st.metrics.copy(), st.token_positions.copy()
    orc_c.execute_cache_moves(wk, wv, wm, wp, cmi, cmc, st.evicted_kv_offsets)

    ds = hdev.upload(st, DEV)
    g_eli, g_ekc, g_ebc, g_cmi, g_cmc = hdev.schedule(ds, st, evicted)
    k, v = torch.from_numpy(k_np.copy()).to(DEV), torch.from_numpy(v_np.copy()).to(DEV)
    ops.execute_cache_moves(k, v, ds.cm.metrics, ds.cm.token_positions, g_cmi, g_cmc,
                            ds.evicted_kv_offsets, 1, 16)
    for name, got, want in (("eli", g_eli, eli), ("ekc", g_ekc, ekc), ("ebc", g_ebc, ebc),
                            ("cmi", g_cmi, cmi), ("cmc", g_cmc, cmc), ("k", k, wk), ("v", v, wv),
                            ("metrics", ds.cm.metrics, wm), ("positions", ds.cm.token_positions, wp)):
        assert np.array_equal(got.cpu().numpy(), want), name


FULL_ORACLE_CASES = [
    # BASELINE configs[1] (the headline: fp16, bs 16, 32k, 8.4 M slots) and configs[4] (fp8, bs 32, 64k,
    # 16.8 M slots) at their own size; the reference's 1 / 8 keep ratio of the headline as well
    ("c2_keep_half", 32768, 16, np.uint16, 0.5, "reference"),
    ("c2_keep_eighth", 32768, 16, np.uint16, 0.125, "per_sequence"),
    ("c5_fp8_bs32", 65536, 32, np.uint8, 0.5, "reference"),
]


@pytest.mark.parametrize("name,T,bs,cdtype,keep,mode", FULL_ORACLE_CASES, ids=[c[0] for c in FULL_ORACLE_CASES])
def test_full_size_oracle_parity(name, T, bs, cdtype, keep, mode):
    """the headline configuration and the fp8 one, compared with the oracle AT THEIR OWN SIZE, every
    output bit for bit: evicted indices, counts, the move list, 4 GiB of compacted K / V, metrics
    and positions (the reference's twin-equality pattern, tests/kernels/
    test_kvcompress_eviction.py:900-901, 967, 1007-1008; the oracle needs ~4 s for the schedule and
    ~1 s for the moves and the compaction on the host's cores)"""
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=[T + 1], seed=2,
                          protected=32, spare_block_frac=0.02)
    evicted = [synth.evict_block_count(context_lens_lh=st.context_lens[:, 0, :], seq_len=T + 1, block_size=bs,
                                       protected_window_size=32, max_cache_tokens=int(T * keep))]
    e = np.dtype(cdtype).itemsize
    x = 16 // e
    rng = np.random.default_rng(7)
    k_np = rng.integers(0, 1 << (8 * e), size=(st.num_blocks, HD // x, bs, x), dtype=cdtype)
    v_np = rng.integers(0, 1 << (8 * e), size=(st.num_blocks, HD, bs), dtype=cdtype)
    tdt = torch.uint8 if e == 1 else torch.int16
    k = torch.from_numpy(k_np.view(np.uint8 if e == 1 else np.int16)).to(DEV)
    v = torch.from_numpy(v_np.view(np.uint8 if e == 1 else np.int16)).to(DEV)
    if e == 2:
        k, v = k.view(torch.float16), v.view(torch.float16)
    ds = hdev.upload(st, DEV, mode=mode)
    ds.cm.schedule_path = 0            # (the automatic choice is what the bench line runs on: KVC_SCHEDULE_PATH must not decide)
    g_eli, g_ekc, g_ebc, g_cmi, g_cmc = hdev.schedule(ds, st, evicted)
    how = ds.cm.last_schedule_path()
    ops.execute_cache_moves(k, v, ds.cm.metrics, ds.cm.token_positions, g_cmi, g_cmc, ds.evicted_kv_offsets, 1, 16)
    torch.cuda.synchronize()
    assert how.startswith("bracket"), how   # the schedule the bench line is measured on

    eli, ekc, ebc = orc.schedule_evictions(
        metrics=st.metrics, token_positions=st.token_positions,
        seq_index_by_block=st.seq_index_by_block, layer_index_by_block=st.layer_index_by_block,
        head_index_by_block=st.head_index_by_block,
        logical_block_num_by_block=st.logical_block_num_by_block, block_size=bs, num_layers=L,
        num_kv_heads=H, seq_indices=st.seq_indices, seq_positions=st.seq_positions,
        evicted_blocks_per_seq=evicted, context_lens=st.context_lens,
        hanging_token_count=st.hanging_token_count, evicted_kv_offsets=st.evicted_kv_offsets,
        num_protected=st.protected, mode=mode)
    cmi = np.zeros((st.total_slots, 2), np.int32)
    cmc = np.zeros(ekc.shape, np.int32)
    orc_c.set_threads(orc_c.max_threads())
    orc_c.schedule_cache_moves(cmi, cmc, eli, ekc, st.evicted_kv_offsets, np.ascontiguousarray(st.block_tables),
                               np.ascontiguousarray(st.context_lens), bs)
    wm, wp = st.metrics.copy(), st.token_positions.copy()
    orc_c.execute_cache_moves(k_np, v_np, wm, wp, cmi, cmc, st.evicted_kv_offsets)      # (in place: k_np / v_np are the expectation now)
    orc_c.set_threads(1)
    assert int(ebc.sum()) == evicted[0] and int(cmc.sum()) > 100000
    for nm, got, want in (("eli", g_eli, eli), ("ekc", g_ekc, ekc), ("ebc", g_ebc, ebc), ("cmi", g_cmi, cmi),
                          ("cmc", g_cmc, cmc), ("metrics", ds.cm.metrics, wm), ("positions", ds.cm.token_positions, wp)):
        assert np.array_equal(got.cpu().numpy(), want), nm
    assert np.array_equal(k.view(tdt).cpu().numpy().view(cdtype), k_np), "k_cache"
    assert np.array_equal(v.view(tdt).cpu().numpy().view(cdtype), v_np), "v_cache"


def _hash16(ids, salt):
    """int64 ids -> int16 pattern, different for every (id, salt)"""
    x = (ids * 0x9E3779B97F4A7C1 + salt * 0x85EBCA6B) & 0x7FFFFFFFFFFFFFFF
    x = x ^ (x >> 29)
    return ((x >> 7) & 0xFFFF).to(torch.int16)


FULL_SIZE_CASES = [
    # (T, block_size, cache dtype, keep)   config 2 (fp16, bs16, 32k) and config 5 (fp8, bs32, 64k)
    (32768, 16, torch.int16, 0.5),
    (32768, 16, torch.int16, 0.125),
    (65536, 32, torch.uint8, 0.5),
]


@pytest.mark.parametrize("T,BS,cdtype,keep", FULL_SIZE_CASES)
def test_full_size_properties(T, BS, cdtype, keep):
    X = 16 // torch.empty((), dtype=cdtype).element_size()      # elements per 16 B K vector
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=BS, seq_lens=[T + 1], seed=1,
                          protected=32, spare_block_frac=0.02)
    evicted = [synth.evict_block_count(context_lens_lh=st.context_lens[:, 0, :], seq_len=T + 1,
                                       block_size=BS, protected_window_size=32,
                                       max_cache_tokens=int(T * keep))]
    NB, N = st.num_blocks, st.total_slots
    ds = hdev.upload(st, DEV)
    dev = torch.device(DEV)
    # identity of the KV in (blk, off): id = off_g + position (position == logical index here)
    seq_b = ds.cm.seq_index_by_block.long()
    g_of_blk = (seq_b * L + ds.cm.layer_index_by_block.long()) * H + ds.cm.head_index_by_block.long()
    offs_flat = ds.evicted_kv_offsets.reshape(-1).long()
    alloc = seq_b >= 0
    off_of_blk = torch.where(alloc, offs_flat[g_of_blk.clamp(min=0)], torch.zeros_like(g_of_blk))
    ids0 = off_of_blk[:, None] + ds.cm.token_positions.long()                      # [NB,bs]
    k = torch.empty((NB, HD // X, BS, X), dtype=cdtype, device=dev)
    v = torch.empty((NB, HD, BS), dtype=cdtype, device=dev)
    for r in range(HD // X):
        for e in range(X):
            k[:, r, :, e] = _hash16(ids0, 1000 + r * X + e).to(cdtype)
    for d in range(HD):
        v[:, d, :] = _hash16(ids0, 5000 + d).to(cdtype)
    pos0 = ds.cm.token_positions.clone()

    eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, st, evicted)
    ops.execute_cache_moves(k, v, ds.cm.metrics, ds.cm.token_positions, cmi, cmc,
                            ds.evicted_kv_offsets, 1, 16)
    torch.cuda.synchronize()

    # freed blocks == requested
    assert int(ebc.sum()) == sum(evicted)
    G = L * H
    nblk = (ds.context_lens.transpose(0, 1).reshape(-1).long() + BS - 1) // BS
    seg = eli.view(G, -1)                                                           # equal heads
    cnt = ekc.reshape(-1).long()
    ar = torch.arange(seg.shape[1], device=dev)[None, :]
    live = ar < cnt[:, None]
    assert bool((seg[~live] == MAX_INT).all())
    assert bool(((seg[:, 1:] > seg[:, :-1]) | ~live[:, 1:]).all())                  # ascending
    assert bool(((cnt - ds.hanging_token_count.reshape(-1)) % BS == 0).all())       # test :759

    # moves: no evicted slot is a source, destinations are evicted slots
    rows = torch.arange(N, device=dev)
    j = rows - offs_flat.repeat_interleave(nblk * BS)
    gid = torch.arange(G, device=dev).repeat_interleave(nblk * BS)
    is_move = j < cmc.reshape(-1).long()[gid]
    evicted_mask = torch.zeros(NB * BS, dtype=torch.bool, device=dev)
    bt = ds.block_tables.permute(1, 0, 2, 3).reshape(G, -1).long()                 # [G,M] (B=1)
    lam = seg[live].long()
    g_live = torch.arange(G, device=dev)[:, None].expand_as(seg)[live]
    phys_evicted = bt[g_live, lam // BS] * BS + lam % BS
    evicted_mask[phys_evicted] = True
    dst, src = cmi[is_move, 0].long(), cmi[is_move, 1].long()
    assert not bool(evicted_mask[src].any())
    assert bool(evicted_mask[dst].all())
    assert dst.unique().numel() == dst.numel() and src.unique().numel() == src.numel()

    # every surviving KV bit-equal at its final slot
    new_len = ds.context_lens.transpose(0, 1).reshape(-1).long() - cnt              # [G]
    lbn = ds.cm.logical_block_num_by_block.long()
    lam_blk = lbn[:, None] * BS + torch.arange(BS, device=dev)[None, :]
    live_slot = alloc[:, None] & (lam_blk < new_len[g_of_blk.clamp(min=0)][:, None])
    ids1 = off_of_blk[:, None] + ds.cm.token_positions.long()
    for r in range(HD // X):
        for e in range(X):
            assert bool((k[:, r, :, e] == _hash16(ids1, 1000 + r * X + e).to(cdtype))[live_slot].all()), (r, e)
    for d in range(HD):
        assert bool((v[:, d, :] == _hash16(ids1, 5000 + d).to(cdtype))[live_slot].all()), d
    # final live positions are exactly the survivors: distinct per head, none evicted
    was_evicted_pos = torch.zeros(N, dtype=torch.bool, device=dev)
    was_evicted_pos[offs_flat[g_live] + lam] = True                                # position == lambda
    final_ids = ids1[live_slot]
    assert final_ids.unique().numel() == final_ids.numel()
    assert not bool(was_evicted_pos[final_ids].any())
    assert final_ids.numel() == int((ds.context_lens.long().sum() - cnt.sum()))
    # untouched slots keep their position
    moved_dst = torch.zeros(NB * BS, dtype=torch.bool, device=dev)
    moved_dst[dst] = True
    assert bool((ds.cm.token_positions.reshape(-1)[~moved_dst] == pos0.reshape(-1)[~moved_dst]).all())

    # The op ran on the plan schedule_cache_moves left behind: one launch, every run taken as the only
    # writer of its destination block.  That this holds for the list is checked with the planning
    # pass a list of unknown origin gets (a copy of the same list): it counts the runs per block.
    from vllm_kvcompress_amd import _lib
    assert ops._plan_of(k, cmi, cmc, ds.evicted_kv_offsets, G, BS) is not None
    ops._execute_cache_moves(k, v, ds.cm.metrics, ds.cm.token_positions, cmi[:], cmc[:],
                             ds.evicted_kv_offsets[:], "plan")
    ws = ops.workspace(torch.device(dev), 0, "execute_cache_moves")   # the buffer the planning pass just filled
    coff = int(_lib.load().kvc_cache_moves_plan_bytes())              # [plan][claim bytes]
    claims = ws[coff:coff + NB]
    assert int(claims.max()) == 1
    assert int((claims == 1).sum()) == (dst // BS).unique().numel()


def test_config4_full_size_properties():
    """BASELINE configs[3] as one GPU of the 8 sees it: Llama-3-70B shape (80 layers, 8 KV heads,
    hd 128), 16 384-token cache, 32 sequences (336 M candidate slots, 172 GB of K/V), compress_once
    to half the cache, per_sequence scheduling."""
    _multi_sequence_full_size(80, 8, 16384, 32, 0.5)


def test_config3_initial_compression_full_size_properties():
    """BASELINE configs[2], phase i: a wave of 16 sequences with 32k-token
    histories compressed to the 4k-token cap in one call (134 M candidate slots, 69 GB of K/V,
    7/8 of every head evicted), per_sequence scheduling."""
    _multi_sequence_full_size(32, 8, 32768, 16, 0.125)


def _multi_sequence_full_size(Lc, Hc, T, Bc, keep):
    """Size-independent properties: every sequence frees
    exactly what was asked; per head the evicted indices are ascending, distinct and padded; no
    evicted slot is a move source and every destination is an evicted slot; EVERY surviving KV
    sits bit-equal at its final slot (K/V rows carry a hash of the KV's identity); the block path
    was taken for every destination block."""
    need = 2 * Lc * Hc * Bc * (T // BS + 2) * BS * HD * 2 + (24 << 30)
    import gc
    gc.collect()
    torch.cuda.empty_cache()            # blocks cached by earlier tests do not count as used
    free, _ = torch.cuda.mem_get_info()
    if free < need:
        pytest.skip(f"needs {need >> 30} GiB of free HBM")
    st = synth.make_state(num_layers=Lc, num_kv_heads=Hc, block_size=BS, seq_lens=[T + 1] * Bc, seed=4,
                          protected=32, spare_block_frac=0.01)
    evicted = _evict(st, keep, T)
    NB, N = st.num_blocks, st.total_slots
    ds = hdev.upload(st, DEV, mode="per_sequence")
    dev = torch.device(DEV)
    seq_b = ds.cm.seq_index_by_block.long()
    g_of_blk = (seq_b * Lc + ds.cm.layer_index_by_block.long()) * Hc + ds.cm.head_index_by_block.long()
    offs_flat = ds.evicted_kv_offsets.reshape(-1).long()
    alloc = seq_b >= 0
    off_of_blk = torch.where(alloc, offs_flat[g_of_blk.clamp(min=0)], torch.zeros_like(g_of_blk))
    ids0 = off_of_blk[:, None] + ds.cm.token_positions.long()                      # [NB,bs]
    # K[blk, r, s, e] = h(id, r); V[blk, d, s] = h(id, d / 8): one broadcast copy per cache
    k = torch.empty((NB, HD // 8, BS, 8), dtype=torch.int16, device=dev)
    v = torch.empty((NB, HD, BS), dtype=torch.int16, device=dev)
    for r in range(HD // 8):
        hr = _hash16(ids0, 1000 + r)
        k[:, r] = hr[:, :, None]
        v[:, r * 8:(r + 1) * 8] = hr[:, None, :]
    del hr
    eli, ekc, ebc = ds.cm.schedule_evictions(list(st.seq_indices), ds.seq_positions, evicted, ds.context_lens,
                                             ds.hanging_token_count, ds.evicted_kv_offsets, list(st.protected),
                                             total_slots=N)
    cmi = torch.empty((N, 2), dtype=torch.int32, device=dev)
    cmc = torch.empty_like(ekc)
    ops.schedule_cache_moves(cmi, cmc, eli, ekc, ds.evicted_kv_offsets, ds.block_tables, ds.context_lens, BS)
    ops.execute_cache_moves(k, v, ds.cm.metrics, ds.cm.token_positions, cmi, cmc, ds.evicted_kv_offsets, 1, 16)
    torch.cuda.synchronize()
    assert ebc.sum(dim=(1, 2)).cpu().tolist() == evicted
    G = Bc * Lc * Hc
    n = N // G
    seg = eli.view(G, n)
    cnt = ekc.reshape(-1).long()
    ar = torch.arange(n, device=dev)[None, :]
    live = ar < cnt[:, None]
    assert bool((seg[~live] == MAX_INT).all())
    assert bool(((seg[:, 1:] > seg[:, :-1]) | ~live[:, 1:]).all())
    assert bool(((cnt - ds.hanging_token_count.reshape(-1)) % BS == 0).all())
    # moves
    nmov = cmc.reshape(-1).long()
    rows = offs_flat.repeat_interleave(nmov) + (torch.arange(int(nmov.sum()), device=dev)
                                                - (torch.cumsum(nmov, 0) - nmov).repeat_interleave(nmov))
    dst, src = cmi[rows, 0].long(), cmi[rows, 1].long()
    evicted_mask = torch.zeros(NB * BS, dtype=torch.bool, device=dev)
    bt = ds.block_tables.permute(1, 0, 2, 3).reshape(G, -1).long()                 # [G,M] in (b,l,h) order
    lam = seg.clamp(max=n - 1).long()
    phys = bt.gather(1, lam // BS) * BS + lam % BS
    evicted_mask[phys[live]] = True
    assert not bool(evicted_mask[src].any()) and bool(evicted_mask[dst].all())
    del phys, lam
    # every surviving KV bit-equal at its final slot
    new_len = ds.context_lens.permute(1, 0, 2).reshape(-1).long() - cnt            # [G] in (b,l,h) order
    lam_blk = ds.cm.logical_block_num_by_block.long()[:, None] * BS + torch.arange(BS, device=dev)[None, :]
    live_slot = alloc[:, None] & (lam_blk < new_len[g_of_blk.clamp(min=0)][:, None])
    ids1 = off_of_blk[:, None] + ds.cm.token_positions.long()
    for r in range(HD // 8):
        hr = _hash16(ids1, 1000 + r)
        assert bool(((k[:, r] == hr[:, :, None]).all(dim=2))[live_slot].all()), ("K", r)
        assert bool(((v[:, r * 8:(r + 1) * 8] == hr[:, None, :]).all(dim=1))[live_slot].all()), ("V", r)
    final_ids = ids1[live_slot]
    assert final_ids.numel() == int(ds.context_lens.long().sum() - cnt.sum())
    assert final_ids.unique().numel() == final_ids.numel()
    # one run per destination block: counted by the planning pass a list of unknown origin gets (views of
    # the same list: other tensor objects, so the plan schedule_cache_moves left behind does not vouch for them)
    from vllm_kvcompress_amd import _lib
    assert ops._plan_of(k, cmi, cmc, ds.evicted_kv_offsets, G, BS) is not None
    ops._execute_cache_moves(k, v, ds.cm.metrics, ds.cm.token_positions, cmi[:], cmc[:], ds.evicted_kv_offsets[:], "plan")
    ws = ops.workspace(dev, 0, "execute_cache_moves")
    coff = int(_lib.load().kvc_cache_moves_plan_bytes())              # [plan][claim bytes]
    claims = ws[coff:coff + NB]
    assert int(claims.max()) == 1 and int((claims == 1).sum()) == (dst // BS).unique().numel()


def test_one_process_drives_every_visible_device():
    """Per-device state (the > 64 KiB dynamic-LDS opt-ins of select_emit / seq_select_topk, the
    occupancy-sized grid of the compaction, the side stream of the small-eviction schedule, the
    scratch caches) must be per device: ONE process runs the long-head pipeline (heads of 16 k
    slots: 80 KiB of staged keys) and a steady-state step on cuda:0, cuda:1, ... in turn and
    again in reverse order, each against the oracle."""
    from tests.helpers import oracle_pipeline
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one visible device (the 8-GPU node runs this)")
    long_st = synth.make_state(num_layers=2, num_kv_heads=2, block_size=16, seq_lens=[16385], seed=3, protected=32)
    long_ev = _evict(long_st, 0.5, 16384)
    steady = synth.make_state(num_layers=4, num_kv_heads=4, block_size=16, seq_lens=[3000] * 2, seed=4,
                              protected=17, steady_cap=1024, spare_block_frac=0.05)
    steady_ev = [synth.evict_block_count(context_lens_lh=steady.context_lens[:, b, :], seq_len=3000, block_size=16,
                                         protected_window_size=17, max_cache_tokens=1024) for b in range(2)]
    k, v = synth.make_caches_u16(3, long_st.num_blocks, HD, BS)
    want_long = oracle_pipeline(long_st, long_ev, k, v)
    want_steady = oracle_pipeline(steady, steady_ev, mode="per_sequence")
    for d in list(range(n)) + list(range(n - 1, -1, -1)):
        dev = f"cuda:{d}"
        with torch.cuda.device(dev):
            ds = hdev.upload(long_st, dev)
            eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, long_st, long_ev)
            kd, vd = torch.from_numpy(k.copy()).to(dev), torch.from_numpy(v.copy()).to(dev)
            ops.execute_cache_moves(kd, vd, ds.cm.metrics, ds.cm.token_positions, cmi, cmc, ds.evicted_kv_offsets, 1, 16)
            for name, got in (("eli", eli), ("ekc", ekc), ("cmi", cmi), ("cmc", cmc), ("k", kd), ("v", vd)):
                np.testing.assert_array_equal(got.cpu().numpy(), want_long[name], err_msg=f"{name} on {dev}")
            ds2 = hdev.upload(steady, dev, mode="per_sequence")
            out = hdev.schedule(ds2, steady, steady_ev)
            assert ds2.cm.last_schedule_path() == "small_eviction"
            for name, got in zip(("eli", "ekc", "ebc", "cmi", "cmc"), out):
                np.testing.assert_array_equal(got.cpu().numpy(), want_steady[name], err_msg=f"{name} on {dev}")


@pytest.mark.parametrize("kind", ["steady_state", "bulk"])
def test_coupled_reference_mode_at_64_sequences_of_4k_tokens(kind):
    """The fork's default mode -- the reference's batch > 1 rule (metrics.py:709-729: the inf count runs from
    position 0, so a sequence's eviction depends on every sequence in front of it) -- over 64 coupled sequences
    of 4k tokens, against the oracle in BOTH its forms: the literal restatement (one sort over the 8.4 M slots
    of the batch) and the two-stage form bench.py's reference-mode gate uses at configs[2] size (finite-threshold
    counts -> the rule as arithmetic -> per-sequence runs).  "steady_state": every head one token over its cap,
    a block per head asked for -- the rule lets only the first sequence free anything; "bulk": 4k -> 2k tokens
    per sequence -- every sequence frees what it asked for less the infinite thresholds in front of it."""
    Lc, Hc, B, T = 4, 8, 64, 4096
    if kind == "steady_state":
        st = synth.make_state(num_layers=Lc, num_kv_heads=Hc, block_size=BS, seq_lens=[3 * T] * B, seed=5,
                              protected=32, steady_cap=T, spare_block_frac=0.05)
        evicted = [synth.evict_block_count(context_lens_lh=st.context_lens[:, b, :], seq_len=3 * T, block_size=BS,
                                           protected_window_size=32, max_cache_tokens=T) for b in range(B)]
    else:
        st = synth.make_state(num_layers=Lc, num_kv_heads=Hc, block_size=BS, seq_lens=[T + 1] * B, seed=6,
                              protected=32, spare_block_frac=0.05)
        evicted = _evict(st, 0.5, T)
    kw = dict(metrics=st.metrics, token_positions=st.token_positions, seq_index_by_block=st.seq_index_by_block,
              layer_index_by_block=st.layer_index_by_block, head_index_by_block=st.head_index_by_block,
              logical_block_num_by_block=st.logical_block_num_by_block, block_size=BS, num_layers=Lc, num_kv_heads=Hc,
              seq_indices=st.seq_indices, seq_positions=st.seq_positions, evicted_blocks_per_seq=evicted,
              context_lens=st.context_lens, hanging_token_count=st.hanging_token_count,
              evicted_kv_offsets=st.evicted_kv_offsets, num_protected=st.protected)
    want = orc.schedule_evictions(**kw, mode="reference")
    two = orc.schedule_evictions_two_stage(**kw, mode="reference")
    for a, b, name in zip(two, want, ("eli", "ekc", "ebc")):
        np.testing.assert_array_equal(a, b, err_msg=f"oracle, two-stage form vs literal restatement: {name}")
    freed = want[2].reshape(B, -1).sum(1)
    asked = np.asarray(evicted)
    if kind == "steady_state":
        assert freed[0] == asked[0] and not freed[1:].any()          # the quirk: only the first sequence frees anything
    else:
        assert freed[0] == asked[0] and (freed[1:] < asked[1:]).all() and (freed > 0).sum() > 40
    ds = hdev.upload(st, DEV, mode="reference")
    ds.cm.schedule_path = 0
    for form in ("list", "fork"):
        if form == "list":
            got = ds.cm.schedule_evictions(list(st.seq_indices), ds.seq_positions, evicted, ds.context_lens,
                                           ds.hanging_token_count, ds.evicted_kv_offsets, list(st.protected),
                                           total_slots=st.total_slots)
        else:       # the fork's call: counts as a device tensor, no N
            got = ds.cm.schedule_evictions(list(st.seq_indices), ds.seq_positions.clone(),
                                           torch.tensor(evicted, dtype=torch.int, device=DEV), ds.context_lens,
                                           ds.hanging_token_count, ds.evicted_kv_offsets, tuple(st.protected))
        path = ds.cm.last_schedule_path()
        assert path.startswith("small_eviction" if kind == "steady_state" else "bracket"), (form, ds.cm.last_schedule_reason)
        for a, b, name in zip(got, want, ("eli", "ekc", "ebc")):
            np.testing.assert_array_equal(a.cpu().numpy(), b, err_msg=f"{kind}, {form} form ({path}): {name}")
        del got
