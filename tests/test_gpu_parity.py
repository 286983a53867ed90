"""GPU parity tests: the HIP path (called through the reference-shaped Python surface and
the C ABI underneath) against the oracle and the committed golden vectors.

Bars: bit-exact for every index / count / move table; bitwise equal K/V, metrics and
positions after compaction (pure copies); float32 aggregation bit-equal to the oracle's
sequential float32 restatement (tolerance vs torch reference: 1e-6 relative).
"""
import os

import numpy as np
import pytest
import torch

from oracle import kvc_oracle as orc
from tests.conftest import golden_cases
from tests.helpers import golden_caches, load_golden, oracle_pipeline, sha
from vllm_kvcompress_amd import _custom_ops as ops
from vllm_kvcompress_amd.harness import device as hdev
from vllm_kvcompress_amd.harness import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _state_from_golden(g):
    return synth.PagedState(
        block_size=int(g["block_size"]), num_layers=int(g["num_layers"]),
        num_kv_heads=int(g["num_kv_heads"]), num_seqs=len(g["seq_indices"]),
        num_blocks=int(g["num_blocks"]), metrics=g["metrics"],
        token_positions=g["token_positions"], seq_index_by_block=g["seq_index_by_block"],
        layer_index_by_block=g["layer_index_by_block"],
        head_index_by_block=g["head_index_by_block"],
        logical_block_num_by_block=g["logical_block_num_by_block"],
        context_lens=g["context_lens"], block_tables=g["block_tables"],
        hanging_token_count=g["hanging_token_count"],
        evicted_kv_offsets=g["evicted_kv_offsets"],
        seq_indices=[int(s) for s in g["seq_indices"]], seq_positions=g["seq_positions"],
        protected=[int(p) for p in g["protected"]])


def _gpu_pipeline(st, evicted, k_np=None, v_np=None, mode="reference", uniform_evict=False, **kw):
    ds = hdev.upload(st, DEV, mode=mode, **kw)
    eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, st, evicted, uniform_evict=uniform_evict)
    out = dict(eli=eli.cpu().numpy(), ekc=ekc.cpu().numpy(), ebc=ebc.cpu().numpy(),
               cmi=cmi.cpu().numpy(), cmc=cmc.cpu().numpy())
    if k_np is not None:
        k = torch.from_numpy(k_np.copy()).to(DEV)
        v = torch.from_numpy(v_np.copy()).to(DEV)
        ops.execute_cache_moves(k, v, ds.cm.metrics, ds.cm.token_positions, cmi, cmc,
                                ds.evicted_kv_offsets, 1, 16)
        out.update(k=k.cpu().numpy(), v=v.cpu().numpy(), metrics=ds.cm.metrics.cpu().numpy(),
                   positions=ds.cm.token_positions.cpu().numpy())
    return out


# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_cases())
def test_golden_end_to_end(name):
    """A3 -> A5 -> A6 on the reference-generated fixtures (bit-exact)."""
    g = load_golden(name)
    st = _state_from_golden(g)
    kw = dict(use_average=bool(int(g["use_average"])), num_sinks=int(g["num_sinks"]))
    if "bias" in g:
        kw.update(bias=g["bias"], position_bins=g["position_bins"],
                  bias_weight=float(g["bias_weight"]))
    if "uniform_evict" in g and int(g["uniform_evict"]):
        kw.update(uniform_evict=True)          # the reference's other selection rule (metrics.py:639-666)
    k, v = golden_caches(g)
    out = _gpu_pipeline(st, g["evicted_blocks_per_seq"], k, v, **kw)
    np.testing.assert_array_equal(out["eli"], g["ref_evicted_logical_indices"])
    np.testing.assert_array_equal(out["ekc"], g["ref_evicted_kv_count"])
    np.testing.assert_array_equal(out["ebc"], g["ref_evicted_block_count"])
    np.testing.assert_array_equal(out["cmi"], g["ref_cache_moves_idx"])
    np.testing.assert_array_equal(out["cmc"], g["ref_cache_moves_count"])
    np.testing.assert_array_equal(sha(out["k"]), g["ref_k_sha256"])
    np.testing.assert_array_equal(sha(out["v"]), g["ref_v_sha256"])
    np.testing.assert_array_equal(out["metrics"], g["ref_metrics"])
    np.testing.assert_array_equal(out["positions"], g["ref_positions"])


def _limit(st, frac):
    bs = st.block_size
    nblk = ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)
    TH = st.num_layers * st.num_kv_heads
    return [int(max(int(nblk[b]) - (st.protected[b] + bs - 1) // bs * TH, 0) * frac)
            for b in range(st.num_seqs)]


RANDOM_CASES = [
    # (L, H, bs, seq_lens, protected, compressed, hd, frac, tie_levels)
    (2, 2, 4, [37], 1, False, 8, 0.5, None),
    (3, 2, 4, [37, 12, 55], [1, 3, 8], False, 8, 0.6, None),
    (2, 4, 16, [300, 171], 32, False, 128, 0.5, None),
    (2, 4, 16, [300, 171, 90], [32, 5, 17], True, 128, 0.7, None),
    (4, 8, 16, [700], 32, False, 128, 0.875, None),
    (2, 2, 32, [260, 100], 33, False, 128, 0.5, None),
    (2, 2, 1, [40, 9], 2, False, 8, 0.5, None),
    (2, 3, 2, [41, 23], 3, True, 8, 0.4, None),
    (2, 2, 16, [2100], 16, False, 64, 0.5, None),       # heads longer than one histogram tile
    (1, 1, 16, [5000], 1, False, 128, 0.9, None),
    (1, 1, 16, [5000, 100, 100, 100], 3, False, 128, 0.7, None),   # one head larger than the LDS staging buffer
    (1, 2, 16, [6000, 64, 64], 2, False, 128, 0.5, 4),             # same, with metric ties
    # ties: canonical order (metric, physical block, offset) / (threshold, head, chunk)
    (2, 2, 4, [37, 50], 2, False, 8, 0.5, 3),
    (2, 4, 16, [300, 171], 20, True, 128, 0.6, 5),
    (2, 2, 16, [400], 7, False, 128, 0.5, 1),
]


@pytest.mark.parametrize("mode", ["reference", "per_sequence"])
@pytest.mark.parametrize("case", range(len(RANDOM_CASES)))
def test_random_states_vs_oracle(case, mode):
    L, H, bs, seq_lens, prot, compressed, hd, frac, ties = RANDOM_CASES[case]
    for seed in (0, 1):
        st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens,
                              seed=seed, protected=prot, compressed=compressed, tie_levels=ties)
        evicted = _limit(st, frac)
        k, v = synth.make_caches_u16(seed, st.num_blocks, hd, bs)
        want = oracle_pipeline(st, evicted, k, v, mode=mode)
        got = _gpu_pipeline(st, evicted, k, v, mode=mode)
        for key in ("eli", "ekc", "ebc", "cmi", "cmc", "k", "v", "metrics", "positions"):
            np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} seed={seed}")


@pytest.mark.parametrize("L,H,bs,seq_lens,prot", [(2, 2, 4, [37], 2), (3, 2, 16, [300, 171, 90], [32, 5, 17]),
                                                  (1, 1, 16, [9000], 3), (2, 4, 32, [700, 40], 33), (2, 2, 1, [40, 9], 2)])
def test_uniform_evict_vs_oracle(L, H, bs, seq_lens, prot):
    """uniform_evict (metrics.py:639-666): the same number of chunks from every head, random prefill
    states against the oracle, from nothing to every finite-threshold chunk; heads of unequal length
    (which the reference's reshape cannot take) against the per-head definition"""
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=len(seq_lens) + bs,
                          protected=prot)
    ds = hdev.upload(st, DEV)
    TH = L * H
    for frac in (0.0, 0.3, 0.6, 0.9):
        evicted = [int(x * frac) // TH * TH + (TH - 1 if frac else 0) for x in _limit(st, 1.0)]   # (k need not divide by L*H)
        want = oracle_pipeline(st, evicted, uniform_evict=True)
        eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, st, evicted, uniform_evict=True)
        assert ds.cm.last_schedule_path() == "general" and "uniform_evict" in ds.cm.last_schedule_reason
        for key, got in (("eli", eli), ("ekc", ekc), ("ebc", ebc), ("cmi", cmi), ("cmc", cmc)):
            np.testing.assert_array_equal(got.cpu().numpy(), want[key], err_msg=f"{key} frac={frac}")
        assert int(ebc.sum()) == sum(e // TH * TH for e in evicted)
    # a compressed state: heads of different lengths (the reference's reshape cannot take them) -- every
    # head frees min(k / (L*H), its finite-threshold chunks)
    st2 = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=7, protected=prot,
                           compressed=True)
    ds2 = hdev.upload(st2, DEV)
    B = len(seq_lens)
    finite = hdev.schedule(ds2, st2, [10 ** 6] * B, uniform_evict=True)[2]
    two = hdev.schedule(ds2, st2, [TH * 2 + TH - 1] * B, uniform_evict=True)[2]
    assert torch.equal(two, torch.clamp(finite, max=2))
    nblk = (ds2.context_lens + bs - 1) // bs
    assert bool((finite <= nblk.permute(1, 0, 2)).all()) and int(finite.sum()) > 0


def test_overask_and_zero_evictions():
    st = synth.make_state(num_layers=2, num_kv_heads=2, block_size=4, seq_lens=[33, 21, 40],
                          seed=3, protected=[5, 2, 7])
    nblk = ((st.context_lens.astype(np.int64) + 3) // 4).sum(0).sum(-1)
    for evicted in ([int(n) for n in nblk], [0, 0, 0], [int(nblk[0]), 0, 3]):
        for mode in ("reference", "per_sequence"):
            want = oracle_pipeline(st, evicted, mode=mode)
            got = _gpu_pipeline(st, evicted, mode=mode)
            for key in ("eli", "ekc", "ebc", "cmi", "cmc"):
                np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} {evicted} {mode}")


def test_options_average_sinks_bias():
    st = synth.make_state(num_layers=2, num_kv_heads=3, block_size=4, seq_lens=[45, 30], seed=5,
                          protected=3)
    evicted = _limit(st, 0.5)
    bins = np.array([0, 6, 19], dtype=np.int32)
    bias = (np.random.default_rng(4).normal(size=(2, 3, 3)) * 30).astype(np.float32)
    for kw in (dict(use_average=True), dict(num_sinks=4),
               dict(bias=bias, position_bins=bins, bias_weight=0.25),
               dict(use_average=True, num_sinks=2, bias=bias, position_bins=bins, bias_weight=1.5)):
        want = oracle_pipeline(st, evicted, **kw)
        got = _gpu_pipeline(st, evicted, **kw)
        for key in ("eli", "ekc", "ebc", "cmi", "cmc"):
            np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} {list(kw)}")


def test_protected_zero_quirk_moves():
    """with protected=0 the first empty tail slot is evictable and the move walk
    then drops a real KV.  The move schedule must still equal the serial kernel's."""
    st = synth.make_state(num_layers=2, num_kv_heads=2, block_size=4, seq_lens=[31], seed=8,
                          protected=0)
    evicted = [6]
    want = oracle_pipeline(st, evicted)
    got = _gpu_pipeline(st, evicted)
    for key in ("eli", "ekc", "ebc", "cmi", "cmc"):
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)


# ---------------------------------------------------------------------------------------
def test_count_block_evictions_op():
    rng = np.random.default_rng(0)
    for bs in (1, 2, 4, 16, 32):
        G = 37
        nchunks = rng.integers(0, 200, size=G)
        nchunks[3] = 0
        offs = np.concatenate([[0], np.cumsum(nchunks * bs)[:-1]]).astype(np.int32).reshape(1, 1, G)
        total = int((nchunks * bs).sum())
        idx = rng.integers(0, 50, size=total).astype(np.int32)
        # leading runs of random length, then nulls sprinkled
        for g in range(G):
            s = int(offs.reshape(-1)[g])
            run = rng.integers(0, nchunks[g] + 1)
            idx[s + run * bs: s + nchunks[g] * bs][rng.random((nchunks[g] - run) * bs) < 0.5] = 99
            if run < nchunks[g]:
                idx[s + run * bs] = 99
        hang = rng.integers(1, bs + 1, size=(1, 1, G)).astype(np.int32)
        want_idx, want = idx.copy(), np.zeros((1, 1, G), np.int32)
        orc.count_block_evictions(want, want_idx, offs, hang, bs, 99)
        d_idx = torch.from_numpy(idx).to(DEV)
        d_out = torch.zeros((1, 1, G), dtype=torch.int32, device=DEV)
        ops.count_block_evictions(d_out, d_idx, torch.from_numpy(offs).to(DEV),
                                  torch.from_numpy(hang).to(DEV), bs, 99)
        np.testing.assert_array_equal(d_out.cpu().numpy(), want)
        np.testing.assert_array_equal(d_idx.cpu().numpy(), want_idx)


def _random_moves(rng, nb, bs, G, max_moves, sort_dst):
    """independent moves: disjoint dst/src slot sets; per head a random count"""
    slots = rng.permutation(nb * bs)
    counts = rng.integers(0, max_moves + 1, size=G)
    seg = rng.integers(0, 5, size=G) + counts          # segment >= count
    offs = np.concatenate([[0], np.cumsum(seg)[:-1]]).astype(np.int32)
    rows = int(seg.sum()) + 3
    moves = np.zeros((rows, 2), dtype=np.int32)
    cur = 0
    for g in range(G):
        c = int(counts[g])
        dst = slots[cur:cur + c]
        src = slots[cur + c:cur + 2 * c]
        cur += 2 * c
        if sort_dst:
            dst = np.sort(dst)
        moves[offs[g]:offs[g] + c, 0] = dst
        moves[offs[g]:offs[g] + c, 1] = src
    return moves, counts.astype(np.int32).reshape(1, 1, G), offs.reshape(1, 1, G)


@pytest.mark.parametrize("dtype,hd,bs", [
    (torch.float16, 128, 16), (torch.bfloat16, 128, 16), (torch.uint8, 128, 32),
    (torch.uint8, 128, 16), (torch.float16, 128, 32), (torch.float32, 128, 16),
    (torch.float16, 64, 16), (torch.float16, 256, 16),
    (torch.float16, 8, 4), (torch.float16, 16, 2), (torch.float32, 4, 1), (torch.uint8, 16, 4),
    (torch.float16, 80, 16), (torch.float16, 128, 8),
])
@pytest.mark.parametrize("sort_dst", [True, False])
def test_execute_cache_moves_general(dtype, hd, bs, sort_dst):
    """arbitrary independent move lists, all dtypes/shapes: fast row path (sorted dst),
    element-wise fallback (unsorted dst) and the generic kernel (odd shapes)."""
    rng = np.random.default_rng(hd * 1000 + bs)
    nb, G = 96, 9
    e = torch.empty((), dtype=dtype).element_size()
    x = 16 // e
    npdt = {1: np.uint8, 2: np.uint16, 4: np.uint32}[e]
    k_np = rng.integers(0, 256, size=(nb, hd // x, bs, x * e), dtype=np.uint8).view(npdt)
    v_np = rng.integers(0, 256, size=(nb, hd, bs * e), dtype=np.uint8).view(npdt)
    m_np = rng.random((nb, bs)).astype(np.float32)
    p_np = rng.integers(0, 10 ** 6, size=(nb, bs)).astype(np.int32)
    moves, counts, offs = _random_moves(rng, nb, bs, G, min(40, nb * bs // (2 * G)), sort_dst)
    wk, wv, wm, wp = k_np.copy(), v_np.copy(), m_np.copy(), p_np.copy()
    orc.execute_cache_moves(wk, wv, wm, wp, moves, counts, offs)
    k = torch.from_numpy(k_np.view(np.uint8)).to(DEV).view(dtype).view(nb, hd // x, bs, x)
    v = torch.from_numpy(v_np.view(np.uint8)).to(DEV).view(dtype).view(nb, hd, bs)
    m = torch.from_numpy(m_np).to(DEV)
    p = torch.from_numpy(p_np).to(DEV)
    ops.execute_cache_moves(k, v, m, p, torch.from_numpy(moves).to(DEV),
                            torch.from_numpy(counts).to(DEV), torch.from_numpy(offs).to(DEV), 1, 16)
    np.testing.assert_array_equal(k.view(torch.uint8).cpu().numpy().reshape(-1), wk.view(np.uint8).reshape(-1))
    np.testing.assert_array_equal(v.view(torch.uint8).cpu().numpy().reshape(-1), wv.view(np.uint8).reshape(-1))
    np.testing.assert_array_equal(m.cpu().numpy(), wm)
    np.testing.assert_array_equal(p.cpu().numpy(), wp)


def test_unsupported_shapes_raise():
    k = torch.zeros((4, 2, 4, 8), dtype=torch.float64, device=DEV)
    v = torch.zeros((4, 16, 4), dtype=torch.float64, device=DEV)
    m = torch.zeros((4, 4), dtype=torch.float32, device=DEV)
    p = torch.zeros((4, 4), dtype=torch.int32, device=DEV)
    z = torch.zeros((1, 1, 1), dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="Unsupported"):
        ops.execute_cache_moves(k, v, m, p, torch.zeros((4, 2), dtype=torch.int32, device=DEV), z, z, 1, 1)
    with pytest.raises(RuntimeError, match="Unsupported block size"):
        ops.count_block_evictions(z, z.view(-1), z, z, 0, 99)


def test_dispatcher_path():
    """the fork's own call form: torch.ops._C_kvc_ops.* (vllm/_custom_ops.py:1074,1169,1247)"""
    from vllm_kvcompress_amd import torch_ops
    torch_ops.register()
    st = synth.make_state(num_layers=2, num_kv_heads=2, block_size=16, seq_lens=[200, 90], seed=2,
                          protected=16)
    evicted = _limit(st, 0.5)
    k, v = synth.make_caches_u16(2, st.num_blocks, 128, 16)
    want = oracle_pipeline(st, evicted, k, v)
    ds = hdev.upload(st, DEV)
    eli, ekc, ebc = ds.cm.schedule_evictions(list(st.seq_indices), ds.seq_positions, evicted,
                                             ds.context_lens, ds.hanging_token_count,
                                             ds.evicted_kv_offsets, list(st.protected))
    cmi = torch.zeros((st.total_slots, 2), dtype=torch.int32, device=DEV)
    cmc = torch.empty_like(ekc)
    torch.ops._C_kvc_ops.schedule_t1_cache_moves(cmi, cmc, eli, ekc, ds.evicted_kv_offsets,
                                                 ds.block_tables, ds.context_lens, 16)
    kd, vd = torch.from_numpy(k.copy()).to(DEV), torch.from_numpy(v.copy()).to(DEV)
    torch.ops._C_kvc_ops.execute_cache_moves(kd, vd, ds.cm.metrics, ds.cm.token_positions, cmi, cmc,
                                             ds.evicted_kv_offsets, 1, 16)
    np.testing.assert_array_equal(cmi.cpu().numpy(), want["cmi"])
    np.testing.assert_array_equal(kd.cpu().numpy(), want["k"])
    np.testing.assert_array_equal(vd.cpu().numpy(), want["v"])
    if torch_ops._REGISTERED == "compiled":
        # the compiled kernels: the list schedule_t1_cache_moves just made runs on its plan (one launch);
        # once anything touched it through torch, or for a copy of it, the op plans for itself
        n0 = torch.ops._kvc_mi355x.planned_compactions()
        torch.ops._C_kvc_ops.schedule_t1_cache_moves(cmi, cmc, eli, ekc, ds.evicted_kv_offsets,
                                                     ds.block_tables, ds.context_lens, 16)
        for args, planned in (((cmi, cmc, ds.evicted_kv_offsets), 1), ((cmi.clone(), cmc, ds.evicted_kv_offsets), 0)):
            kd2, vd2 = torch.from_numpy(k.copy()).to(DEV), torch.from_numpy(v.copy()).to(DEV)
            m2, p2 = torch.from_numpy(st.metrics.copy()).to(DEV), torch.from_numpy(st.token_positions.copy()).to(DEV)
            torch.ops._C_kvc_ops.execute_cache_moves(kd2, vd2, m2, p2, *args, 1, 16)
            assert torch.ops._kvc_mi355x.planned_compactions() - n0 == planned
            n0 += planned
            np.testing.assert_array_equal(kd2.cpu().numpy(), want["k"])
            np.testing.assert_array_equal(vd2.cpu().numpy(), want["v"])
            np.testing.assert_array_equal(m2.cpu().numpy(), want["metrics"])
        cmc.add_(0)
        kd2, vd2 = torch.from_numpy(k.copy()).to(DEV), torch.from_numpy(v.copy()).to(DEV)
        m2, p2 = torch.from_numpy(st.metrics.copy()).to(DEV), torch.from_numpy(st.token_positions.copy()).to(DEV)
        torch.ops._C_kvc_ops.execute_cache_moves(kd2, vd2, m2, p2, cmi, cmc, ds.evicted_kv_offsets, 1, 16)
        assert torch.ops._kvc_mi355x.planned_compactions() == n0
        np.testing.assert_array_equal(kd2.cpu().numpy(), want["k"])


# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("slots,clear,l2", [((1 << 21) + 16 * 37, False, True), ((1 << 21) + 16 * 37, True, True),
                                            ((1 << 20) + 2048 * 3 + 5, False, False),
                                            ((1 << 28) + 16 * 5, False, True), ((1 << 28) + 16 * 5, True, True)])
def test_aggregate_decode_on_large_stores(slots, clear, l2):
    """the forms kvc_aggregate_decode takes for qpk 4 on large stores (csrc/kvc_aggregate.hip: tiles of eight rows
    from 1 M slots on, a non-temporal stream from 1 GiB of metrics on, each with a tail): the same sums in the same
    order -- the oracle where it takes seconds, the same float32 additions spelled out in torch (separate
    elementwise kernels: no contraction) at config 3's size"""
    from vllm_kvcompress_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=DEV)
    g.manual_seed(slots % 1000)
    m = torch.rand((slots,), device=DEV, generator=g) * 100.0
    t = torch.rand((slots, 4), device=DEV, generator=g)
    v = t * t if l2 else t
    want = m + ((((0.0 + v[:, 0]) + v[:, 1]) + v[:, 2]) + v[:, 3])
    if slots < 1 << 24:
        m_np = m.cpu().numpy().copy()
        orc.aggregate_decode(m_np, t.cpu().numpy(), use_l2=l2)
        np.testing.assert_array_equal(want.cpu().numpy(), m_np)      # (the torch spelling is the oracle's)
    del v
    _lib.check(lib.kvc_aggregate_decode(m.data_ptr(), t.data_ptr(), slots, 4, 1 if l2 else 0, 1 if clear else 0,
                                        torch.cuda.current_stream().cuda_stream))
    assert torch.equal(m.view(torch.int32), want.view(torch.int32))
    assert bool(t.any()) != clear


def test_aggregate_decode_and_clear():
    from vllm_kvcompress_amd.kvcompress.metrics import CompressionMetrics
    rng = np.random.default_rng(1)
    for qpk, l2 in ((4, True), (4, False), (1, True), (8, True), (3, False)):
        cm = CompressionMetrics(16, 2, 2, qpk, 1000, None, 0.0, device=DEV, use_l2=l2)
        cm.init_kv_metadata(50)
        m0 = rng.random((50, 16)).astype(np.float32)
        t0 = rng.random((50, 16, qpk)).astype(np.float32)
        cm.metrics.copy_(torch.from_numpy(m0))
        cm.temp_metrics.copy_(torch.from_numpy(t0))
        want = m0.copy()
        orc.aggregate_decode(want, t0, use_l2=l2)
        cm.aggregate_decode()
        np.testing.assert_array_equal(cm.metrics.cpu().numpy(), want)
        # float32 tolerance against the torch formulation of the reference (metrics.py:436-439)
        tt = torch.from_numpy(t0)
        ref = torch.from_numpy(m0) + ((tt ** 2) if l2 else tt).sum(dim=-1)
        np.testing.assert_allclose(cm.metrics.cpu().numpy(), ref.numpy(), rtol=1e-6, atol=0)
        assert float(cm._temp_metrics.abs().max()) == 0.0      # fused clear
        cm.clear_temp_metrics()                                # free; contract: zeros
        assert float(cm.temp_metrics.abs().max()) == 0.0
        cm.temp_metrics.fill_(1.0)                             # attention wrote into it
        cm.clear_temp_metrics()
        assert float(cm._temp_metrics.abs().max()) == 0.0


def test_aggregate_prefill():
    from vllm_kvcompress_amd.kvcompress.metrics import CompressionMetrics
    rng = np.random.default_rng(2)
    H, qpk, T, NB, bs = 4, 4, 70, 40, 16
    cm = CompressionMetrics(bs, 2, H, qpk, 1000, None, 0.0, device=DEV)
    cm.init_kv_metadata(NB)
    m0 = rng.random((NB, bs)).astype(np.float32)
    pm = rng.random((T, H * qpk)).astype(np.float32)
    slots = rng.permutation(NB * bs)[:T * H].astype(np.int64).reshape(T, H)
    cm.metrics.copy_(torch.from_numpy(m0))
    want = m0.copy()
    orc.aggregate_prefill(want, pm, slots, H)
    cm.aggregate_prefill(torch.from_numpy(pm).to(DEV), torch.from_numpy(slots).to(DEV))
    np.testing.assert_array_equal(cm.metrics.cpu().numpy(), want)


@pytest.mark.parametrize("l2", [True, False])
@pytest.mark.parametrize("avg", [True, False])
@pytest.mark.parametrize("pool", [True, False])
def test_prefill_metric_epilogue(l2, avg, pool):
    from vllm_kvcompress_amd.kvcompress.prefill import accumulate_prefill_tile
    rng = np.random.default_rng(3)
    Hq, qb, K = 4, 24, 96
    for q_offset, buf in ((72, 0), (72, 3), (10, 3), (0, 0)):
        probs = rng.random((Hq, qb, K)).astype(np.float32)
        out0 = rng.random((K, Hq)).astype(np.float32)
        want = out0.copy()
        orc.prefill_metric_epilogue(want, probs, q_offset, buf, l2, avg, pool)
        out = torch.from_numpy(out0).to(DEV)
        accumulate_prefill_tile(out, torch.from_numpy(probs).to(DEV), q_offset, buf, l2, avg, pool)
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("Hq,qb,K", [(4, 24, 96), (3, 70, 100), (2, 17, 97), (2, 50, 40), (1, 5, 8), (2, 33, 1024),
                                     # workgroups of 256, 512 and 1024 threads (host heuristic)
                                     (64, 7, 4096), (64, 9, 8192), (128, 5, 8192)])
def test_prefill_metric_epilogue_shapes_and_both_column_kernels(Hq, qb, K):
    """the four-columns-per-thread kernel (K % 4 == 0, 16 B aligned tile) and the scalar one sum
    every column in the same order: bit-identical results, and both within float rounding of the
    oracle; diagonals inside, left and right of the tile"""
    from vllm_kvcompress_amd.kvcompress.prefill import accumulate_prefill_tile
    rng = np.random.default_rng(Hq * 1000 + K)
    for q_offset, buf in ((max(K - qb, 0), 0), (max(K - qb, 0), 5), (K // 2, 1), (3, 7), (0, 0), (K + 9, 2)):
        for l2, avg, pool in ((True, False, True), (False, True, False)):
            probs = rng.random((Hq, qb, K)).astype(np.float32)
            out0 = rng.random((K, Hq)).astype(np.float32)
            want = out0.copy()
            orc.prefill_metric_epilogue(want, probs, q_offset, buf, l2, avg, pool)
            out = torch.from_numpy(out0).to(DEV)
            accumulate_prefill_tile(out, torch.from_numpy(probs).to(DEV), q_offset, buf, l2, avg, pool)
            np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-6, atol=1e-7)
            # the same tile at an address that is not 16 B aligned takes the scalar kernel
            flat = torch.empty(probs.size + 1, dtype=torch.float32, device=DEV)
            shifted = flat[1:].view(Hq, qb, K)
            shifted.copy_(torch.from_numpy(probs))
            assert shifted.data_ptr() % 16 != 0 and shifted.is_contiguous()
            out2 = torch.from_numpy(out0).to(DEV)
            accumulate_prefill_tile(out2, shifted, q_offset, buf, l2, avg, pool)
            assert torch.equal(out, out2)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_reshape_and_cache_kvc(dtype):
    rng = np.random.default_rng(4)
    T, H, hd, bs, NB = 37, 4, 128, 16, 30
    e = torch.empty((), dtype=dtype).element_size()
    x = 16 // e
    npdt = {2: np.uint16, 4: np.uint32}[e]
    key = rng.integers(0, 2 ** 16, size=(T, H, hd)).astype(npdt)
    val = rng.integers(0, 2 ** 16, size=(T, H, hd)).astype(npdt)
    kc = rng.integers(0, 2 ** 16, size=(NB, hd // x, bs, x)).astype(npdt)
    vc = rng.integers(0, 2 ** 16, size=(NB, hd, bs)).astype(npdt)
    met = rng.random((NB, bs)).astype(np.float32)
    slots = rng.permutation(NB * bs)[:T * H].astype(np.int64)
    slots[5] = -1
    bias = rng.random(H).astype(np.float32)
    wk, wv, wm = kc.copy(), vc.copy(), met.copy()
    orc.reshape_and_cache_kvc(key, val, wk, wv, wm, slots, bias)
    to = lambda a: torch.from_numpy(a.view(np.uint8)).to(DEV).view(dtype).view(a.shape)
    dk, dv = to(kc), to(vc)
    dm = torch.from_numpy(met).to(DEV)
    ops.reshape_and_cache_kvc(to(key), to(val), dk, dv, dm, torch.from_numpy(slots).to(DEV),
                              torch.from_numpy(bias).to(DEV), "auto", 1.0, 1.0)
    np.testing.assert_array_equal(dk.view(torch.uint8).cpu().numpy().reshape(-1), wk.view(np.uint8).reshape(-1))
    np.testing.assert_array_equal(dv.view(torch.uint8).cpu().numpy().reshape(-1), wv.view(np.uint8).reshape(-1))
    np.testing.assert_array_equal(dm.cpu().numpy(), wm)


def test_reshape_and_cache_kvc_block_path_and_ragged():
    """prefill-shaped slot mapping (aligned whole blocks -> block path) mixed with ragged
    tails, padding tokens and a decode-style scattered tail"""
    rng = np.random.default_rng(7)
    T, H, hd, bs = 75, 8, 128, 16
    NB = H * 6 + 10
    key = rng.integers(0, 2 ** 16, size=(T, H, hd)).astype(np.uint16)
    val = rng.integers(0, 2 ** 16, size=(T, H, hd)).astype(np.uint16)
    kc = rng.integers(0, 2 ** 16, size=(NB, hd // 8, bs, 8)).astype(np.uint16)
    vc = rng.integers(0, 2 ** 16, size=(NB, hd, bs)).astype(np.uint16)
    met = rng.random((NB, bs)).astype(np.float32)
    blocks = rng.permutation(NB)
    slots = np.full((T, H), -1, dtype=np.int64)
    for h in range(H):
        for t in range(64):                                   # 4 aligned blocks per head
            slots[t, h] = int(blocks[h * 6 + t // bs]) * bs + t % bs
        for t in range(64, 70):                               # ragged tail in a 5th block
            slots[t, h] = int(blocks[h * 6 + 4]) * bs + (t - 64)
        slots[71, h] = int(blocks[h * 6 + 5]) * bs + 3        # lone decode-style token
    slots[10, 2] = -1                                         # padding inside a block -> slot path
    bias = rng.random(H).astype(np.float32)
    wk, wv, wm = kc.copy(), vc.copy(), met.copy()
    orc.reshape_and_cache_kvc(key, val, wk, wv, wm, slots.reshape(-1), bias)
    to = lambda a: torch.from_numpy(a.view(np.uint8)).to(DEV).view(torch.float16).view(a.shape)
    dk, dv, dm = to(kc), to(vc), torch.from_numpy(met).to(DEV)
    ops.reshape_and_cache_kvc(to(key), to(val), dk, dv, dm, torch.from_numpy(slots.reshape(-1)).to(DEV),
                              torch.from_numpy(bias).to(DEV), "auto", 1.0, 1.0)
    np.testing.assert_array_equal(dk.view(torch.uint8).cpu().numpy().reshape(-1), wk.view(np.uint8).reshape(-1))
    np.testing.assert_array_equal(dv.view(torch.uint8).cpu().numpy().reshape(-1), wv.view(np.uint8).reshape(-1))
    np.testing.assert_array_equal(dm.cpu().numpy(), wm)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("kv_dtype,kind", [("fp8", "e4m3"), ("fp8_e4m3", "e4m3"), ("fp8_e5m2", "e5m2")])
def test_reshape_and_cache_kvc_fp8(dtype, kv_dtype, kind):
    rng = np.random.default_rng(8)
    T, H, hd, bs, NB = 21, 4, 128, 32, 12
    scale_k, scale_v = 0.37, 2.5
    base = torch.from_numpy(rng.normal(0, 30, size=(2, T, H, hd)).astype(np.float32))
    base[0, 0, 0, :8] = torch.tensor([0.0, -0.0, 1e9, -1e9, float("inf"), -float("inf"), 2e-4, -3e-7])
    key_t, val_t = base[0].to(dtype), base[1].to(dtype)
    kc = rng.integers(0, 256, size=(NB, hd // 16, bs, 16), dtype=np.uint8)
    vc = rng.integers(0, 256, size=(NB, hd, bs), dtype=np.uint8)
    met = rng.random((NB, bs)).astype(np.float32)
    slots = rng.permutation(NB * bs)[:T * H].astype(np.int64)
    slots[3] = -1
    bias = rng.random(H).astype(np.float32)
    wk, wv, wm = kc.copy(), vc.copy(), met.copy()
    orc.reshape_and_cache_kvc_fp8(key_t.float().numpy(), val_t.float().numpy(), wk, wv, wm, slots,
                                  bias, kind, scale_k, scale_v)
    dk, dv = torch.from_numpy(kc).to(DEV), torch.from_numpy(vc).to(DEV)
    dm = torch.from_numpy(met).to(DEV)
    ops.reshape_and_cache_kvc(key_t.to(DEV), val_t.to(DEV), dk, dv, dm, torch.from_numpy(slots).to(DEV),
                              torch.from_numpy(bias).to(DEV), kv_dtype, scale_k, scale_v)
    np.testing.assert_array_equal(dk.cpu().numpy(), wk)
    np.testing.assert_array_equal(dv.cpu().numpy(), wv)
    np.testing.assert_array_equal(dm.cpu().numpy(), wm)


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_aggregation_golden_gpu(case):
    """A2a/A2b on device vs the reference-generated vectors (1e-6 relative)"""
    from tests.helpers import load_golden as lg
    from vllm_kvcompress_amd.kvcompress.metrics import CompressionMetrics
    g = lg(f"agg_decode_prefill_{case}")
    NB, bs = g["metrics0"].shape
    H, qpk = int(g["num_kv_heads"]), int(g["qpk"])
    cm = CompressionMetrics(bs, 2, H, qpk, 1000, None, 0.0, device=DEV, use_l2=bool(int(g["use_l2"])))
    cm.init_kv_metadata(NB)
    cm.metrics.copy_(torch.from_numpy(g["metrics0"]))
    cm.temp_metrics.copy_(torch.from_numpy(g["temp"]))
    cm.aggregate_decode()
    np.testing.assert_allclose(cm.metrics.cpu().numpy(), g["ref_after_decode"], rtol=1e-6, atol=0)
    cm.metrics.copy_(torch.from_numpy(g["ref_after_decode"]))
    cm.aggregate_prefill(torch.from_numpy(g["prefill_metrics"]).to(DEV),
                         torch.from_numpy(g["slot_mapping"]).to(DEV))
    np.testing.assert_allclose(cm.metrics.cpu().numpy(), g["ref_after_prefill"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("case", range(24))
def test_prefill_metrics_golden_gpu(case):
    """A2c on device (library GEMM + softmax through torch, HIP epilogue) vs the reference's
    _naive_kvc_attention output; fp16 QK^T on the GPU rounds differently from the CPU
    reference, hence 5e-3 relative on values >= 1e-4"""
    from tests.helpers import load_golden as lg
    from vllm_kvcompress_amd.kvcompress.prefill import naive_kvc_attention
    g = lg(f"agg_prefill_attn_{case:02d}")
    q = torch.from_numpy(g["q"].view(np.float16).copy()).to(DEV)
    k = torch.from_numpy(g["k"].view(np.float16).copy()).to(DEV)
    hd = q.shape[2]
    _, got = naive_kvc_attention(q, k, None, [int(x) for x in g["prompt_lens"]], hd ** -0.5,
                                 torch.from_numpy(g["buffer_len"]), n_observed=int(g["n_observed"]),
                                 max_observed_block_size=int(g["block"]),
                                 use_l2=bool(int(g["use_l2"])), use_average=bool(int(g["use_average"])),
                                 use_maxpool=bool(int(g["use_maxpool"])))
    np.testing.assert_allclose(got.cpu().numpy(), g["ref_kv_metric_output"], rtol=5e-3, atol=1e-4)


def test_batch_is_a_subset_of_the_resident_sequences():
    """the cache holds 5 sequences, the compression batch is slots [1, 3, 4]: blocks of the
    other sequences must be ignored (seq_mask of metrics.py:465-481) and batch position !=
    slot index"""
    full = synth.make_state(num_layers=2, num_kv_heads=3, block_size=4, seq_lens=[30, 41, 17, 58, 26],
                            seed=21, protected=[2, 3, 1, 5, 2])
    sel = [1, 3, 4]
    bs = 4
    ctx = np.ascontiguousarray(full.context_lens[:, sel, :])
    sub = synth.PagedState(
        block_size=bs, num_layers=2, num_kv_heads=3, num_seqs=len(sel), num_blocks=full.num_blocks,
        metrics=full.metrics, token_positions=full.token_positions,
        seq_index_by_block=full.seq_index_by_block, layer_index_by_block=full.layer_index_by_block,
        head_index_by_block=full.head_index_by_block,
        logical_block_num_by_block=full.logical_block_num_by_block, context_lens=ctx,
        block_tables=np.ascontiguousarray(full.block_tables[:, sel]),
        hanging_token_count=synth.hanging_tokens(ctx.transpose(1, 0, 2), bs),
        evicted_kv_offsets=synth.kv_offsets(ctx, bs), seq_indices=sel,
        seq_positions=np.ascontiguousarray(full.seq_positions[sel]),
        protected=[full.protected[i] for i in sel])
    evicted = _limit(sub, 0.6)
    k, v = synth.make_caches_u16(21, full.num_blocks, 8, bs)
    for mode in ("reference", "per_sequence"):
        want = oracle_pipeline(sub, evicted, k, v, mode=mode)
        got = _gpu_pipeline(sub, evicted, k, v, mode=mode)
        for key in ("eli", "ekc", "ebc", "cmi", "cmc", "k", "v", "metrics", "positions"):
            np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} {mode}")


def test_metadata_management_and_profile():
    """insert/remove metadata (metrics.py:344-370) and profile_schedule_evictions (:277-335)"""
    from types import SimpleNamespace
    from vllm_kvcompress_amd.kvcompress.metrics import CompressionMetrics
    cm = CompressionMetrics(16, 2, 2, 1, 50_000, None, 0.0, device=DEV)
    extra = cm.profile_schedule_evictions()
    assert extra >= 0 and cm.num_blocks is None
    cm.init_kv_metadata(64)
    assert int((cm.seq_index_by_block == -1).sum()) == 64
    md = SimpleNamespace(
        physical_blocks=torch.tensor([5, 9, 40], device=DEV),
        seq_indices=torch.tensor([2, 2, 7], dtype=torch.int32, device=DEV),
        logical_blocks=torch.tensor([0, 1, 0], device=DEV),
        layer_indices=torch.tensor([1, 1, 0], dtype=torch.int32, device=DEV),
        head_indices=torch.tensor([0, 0, 1], dtype=torch.int32, device=DEV),
        token_positions=torch.arange(48, dtype=torch.int32, device=DEV).view(3, 16))
    cm.insert_metadata(md)
    assert cm.seq_index_by_block[[5, 9, 40]].tolist() == [2, 2, 7]
    assert cm.logical_block_num_by_block[9].item() == 1 and cm.layer_index_by_block[5].item() == 1
    assert cm.token_positions[40, 3].item() == 35
    cm.remove_metadata(torch.tensor([9], device=DEV))
    assert cm.seq_index_by_block[9].item() == -1 and cm.seq_index_by_block[5].item() == 2
    cm.validate_metadata()


def test_empty_inputs():
    """nothing to evict / nothing cached / no moves: every op must be a clean no-op"""
    from vllm_kvcompress_amd.kvcompress.metrics import CompressionMetrics
    # a batch whose heads are all empty (ctx = 0): N = 0
    cm = CompressionMetrics(16, 2, 2, 1, 1000, None, 0.0, device=DEV)
    cm.init_kv_metadata(8)
    ctx = torch.zeros((2, 1, 2), dtype=torch.int32, device=DEV)
    hang = torch.full((1, 2, 2), 16, dtype=torch.int32, device=DEV)
    offs = torch.zeros((1, 2, 2), dtype=torch.int32, device=DEV)
    eli, ekc, ebc = cm.schedule_evictions([0], [0], [0], ctx, hang, offs, [1])
    assert eli.numel() == 0 and int(ekc.abs().sum()) == 0 and int(ebc.abs().sum()) == 0
    # zero moves: caches untouched
    k = torch.randint(0, 100, (8, 16, 16, 8), dtype=torch.int16, device=DEV)
    v = torch.randint(0, 100, (8, 128, 16), dtype=torch.int16, device=DEV)
    k0, v0 = k.clone(), v.clone()
    cmi = torch.zeros((4, 2), dtype=torch.int32, device=DEV)
    cmc = torch.zeros((1, 2, 2), dtype=torch.int32, device=DEV)
    ops.schedule_cache_moves(cmi, cmc, torch.zeros(4, dtype=torch.int32, device=DEV), ekc, offs,
                             torch.zeros((2, 1, 2, 1), dtype=torch.int32, device=DEV), ctx, 16)
    assert int(cmc.abs().sum()) == 0 and int(cmi.abs().sum()) == 0
    ops.execute_cache_moves(k, v, cm.metrics, cm.token_positions, cmi, cmc, offs, 1, 16)
    assert torch.equal(k, k0) and torch.equal(v, v0)
    # zero tokens
    ops.reshape_and_cache_kvc(torch.zeros((0, 2, 128), dtype=torch.float16, device=DEV),
                              torch.zeros((0, 2, 128), dtype=torch.float16, device=DEV),
                              k.view(torch.float16), v.view(torch.float16), cm.metrics,
                              torch.zeros(0, dtype=torch.int64, device=DEV),
                              torch.zeros(2, device=DEV), "auto", 1.0, 1.0)
    assert torch.equal(k, k0) and torch.equal(v, v0)


# KVC_FUZZ_SEEDS=<n> widens the sweep (used for the 4000-seed soak runs noted in DESIGN.md 4)
@pytest.mark.parametrize("seed", range(int(os.environ.get("KVC_FUZZ_SEEDS", "120"))))
def test_fuzz_small_states(seed):
    """deterministic fuzz: random shapes (incl. block sizes outside every fast path), ragged
    and empty heads, random eviction requests, ties, both modes -- full pipeline vs oracle"""
    rng = np.random.default_rng(1000 + seed)
    L, H = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    bs = int(rng.choice([1, 2, 4, 8, 16, 32]))
    B = int(rng.integers(1, 5))
    seq_lens = [int(rng.integers(2, 40 * bs // 2 + 8)) for _ in range(B)]
    prot = [int(rng.integers(1, 2 * bs + 2)) for _ in range(B)]
    compressed = bool(rng.random() < 0.5)
    ties = int(rng.integers(1, 6)) if rng.random() < 0.3 else None
    hd = int(rng.choice([8, 16, 64, 128])) if bs != 32 else 128
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=seed,
                          protected=prot, compressed=compressed, tie_levels=ties)
    nblk = ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)
    # up to ALL blocks of a sequence: more than is evictable (protected window);
    # beyond the sequence's own block count the reference itself asserts (metrics.py:725)
    evicted = [int(rng.integers(0, int(n) + 1)) for n in nblk]
    k, v = synth.make_caches_u16(seed, st.num_blocks, hd, bs)
    mode = "reference" if seed % 2 == 0 else "per_sequence"
    want = oracle_pipeline(st, evicted, k, v, mode=mode)
    got = _gpu_pipeline(st, evicted, k, v, mode=mode)
    for key in ("eli", "ekc", "ebc", "cmi", "cmc", "k", "v", "metrics", "positions"):
        np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} seed={seed} {mode}")
