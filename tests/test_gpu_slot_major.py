"""KVC_LAYOUT_SLOT_MAJOR (include/kvc_mi355x.h, ABI version 7): the three ops that interpret the bytes inside a
cache block -- reshape_and_cache_kvc, paged_attention_kvc_*, execute_cache_moves -- with slot-major blocks
(K [bs][hd], V [bs][hd]) against the SAME oracle and the SAME reference-generated fixtures as the reference layout.

The oracle computes in the reference's layout (K [hd/x][bs][x], V [hd][bs], csrc/kvcompress_cache_kernels.cu:57-77);
`to_slot_major` / `to_reference` below are the test's own NumPy statement of the permutation between the two (the
product's converter, vllm_kvcompress_amd.layout.convert_block_layout, is itself checked against it).  Bars as for the
reference layout: bit-exact caches / metrics / positions after writes and compaction; the attention within the
reference test's tolerances."""
import os

import numpy as np
import pytest
import torch

from oracle import kvc_oracle as orc
from tests.attn_helpers import decode_golden, make_state, oracle_decode
from tests.conftest import golden_cases
from tests.helpers import GOLDEN_DIR, golden_caches, load_golden, oracle_pipeline, sha
from tests.test_gpu_parity import _state_from_golden
from vllm_kvcompress_amd import _custom_ops as ops
from vllm_kvcompress_amd import _lib, layout
from vllm_kvcompress_amd.harness import device as hdev
from vllm_kvcompress_amd.harness import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def to_slot_major(k, v):
    """[NB, hd/x, bs, x], [NB, hd, bs] (reference) -> the same shapes holding slot-major blocks"""
    nb, kg, bs, x = k.shape
    ks = np.ascontiguousarray(k.transpose(0, 2, 1, 3)).reshape(k.shape)          # [bs][hd/x][x] = [bs][hd]
    vs = np.ascontiguousarray(v.transpose(0, 2, 1)).reshape(v.shape)
    return ks, vs


def to_reference(k, v):
    nb, kg, bs, x = k.shape
    hd = kg * x
    kr = np.ascontiguousarray(k.reshape(nb, bs, kg, x).transpose(0, 2, 1, 3))
    vr = np.ascontiguousarray(v.reshape(nb, bs, hd).transpose(0, 2, 1))
    return kr, vr


@pytest.fixture
def slot_major():
    _lib.set_block_layout("slot_major")
    yield
    _lib.set_block_layout("reference")


def test_converter_is_the_permutation():
    rng = np.random.default_rng(0)
    k = rng.integers(-30000, 30000, size=(37, 16, 16, 8)).astype(np.int16)
    v = rng.integers(-30000, 30000, size=(37, 128, 16)).astype(np.int16)
    ks, vs = to_slot_major(k, v)
    # a slot's K / V row is one contiguous run
    assert np.array_equal(ks.reshape(37, 16, 128)[5, 3], k[5, :, 3, :].reshape(-1))
    assert np.array_equal(vs.reshape(37, 16, 128)[5, 3], v[5, :, 3])
    kr, vr = to_reference(ks, vs)
    assert np.array_equal(kr, k) and np.array_equal(vr, v)
    kd, vd = torch.from_numpy(k.copy()).to(DEV), torch.from_numpy(v.copy()).to(DEV)
    layout.convert_block_layout(kd, vd, "reference", "slot_major", blocks_per_pass=10)
    assert np.array_equal(kd.cpu().numpy(), ks) and np.array_equal(vd.cpu().numpy(), vs)
    layout.convert_block_layout(kd, vd, "slot_major", "reference", blocks_per_pass=16)
    assert np.array_equal(kd.cpu().numpy(), k) and np.array_equal(vd.cpu().numpy(), v)


# ------------------------------------------------------------------------------------- A7
@pytest.mark.parametrize("dtype,hd,bs", [("f16", 128, 16), ("bf16", 128, 32), ("f32", 64, 16), ("f16", 96, 16),
                                         ("f16", 256, 16)])
def test_reshape_and_cache_slot_major(dtype, hd, bs, slot_major):
    """the cache write, checked against the oracle's (reference layout) through the permutation -- including
    padding tokens (slot < 0), a strided key and the metric initialisation"""
    rng = np.random.default_rng(hd + bs)
    T, H, NB = 53, 4, 40
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[dtype]
    it = torch.int32 if dtype == "f32" else torch.int16
    npi = np.int32 if dtype == "f32" else np.int16
    x = 16 // np.dtype(npi).itemsize
    key = rng.integers(-2 ** 15, 2 ** 15, size=(T, H, hd)).astype(npi)
    val = rng.integers(-2 ** 15, 2 ** 15, size=(T, H, hd)).astype(npi)
    slots = rng.permutation(NB * bs)[:T * H].astype(np.int64)
    slots[rng.random(T * H) < 0.1] = -1
    bias = rng.random(H).astype(np.float32)
    k0 = rng.integers(-2 ** 15, 2 ** 15, size=(NB, hd // x, bs, x)).astype(npi)
    v0 = rng.integers(-2 ** 15, 2 ** 15, size=(NB, hd, bs)).astype(npi)
    m0 = rng.random((NB, bs)).astype(np.float32)
    wk, wv, wm = k0.copy(), v0.copy(), m0.copy()
    orc.reshape_and_cache_kvc(key, val, wk, wv, wm, slots, bias)
    ks, vs = to_slot_major(k0, v0)
    kd = torch.from_numpy(ks).to(DEV).view(tdt)
    vd = torch.from_numpy(vs).to(DEV).view(tdt)
    md = torch.from_numpy(m0.copy()).to(DEV)
    # a key that is a view with a token stride of its own (the fork's qkv split), a contiguous value
    kv = torch.from_numpy(np.stack([key, val], axis=1)).to(DEV).view(tdt)          # [T, 2, H, hd]
    ops.reshape_and_cache_kvc(kv[:, 0], torch.from_numpy(val).to(DEV).view(tdt), kd, vd, md,
                              torch.from_numpy(slots).to(DEV), torch.from_numpy(bias).to(DEV), "auto", 1.0, 1.0)
    torch.cuda.synchronize()
    gk, gv = to_reference(kd.view(it).cpu().numpy(), vd.view(it).cpu().numpy())
    assert np.array_equal(gk, wk) and np.array_equal(gv, wv) and np.array_equal(md.cpu().numpy(), wm)


@pytest.mark.parametrize("kind", ["fp8_e4m3", "fp8_e5m2"])
@pytest.mark.parametrize("src", ["f16", "bf16", "f32"])
def test_reshape_and_cache_fp8_slot_major(kind, src, slot_major):
    rng = np.random.default_rng(5)
    T, H, hd, bs, NB = 31, 3, 128, 32, 20
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[src]
    key = torch.from_numpy((rng.standard_normal((T, H, hd)) * 3).astype(np.float32)).to(tdt)
    val = torch.from_numpy((rng.standard_normal((T, H, hd)) * 300).astype(np.float32)).to(tdt)
    slots = rng.permutation(NB * bs)[:T * H].astype(np.int64)
    bias = rng.random(H).astype(np.float32)
    wk = np.zeros((NB, hd // 16, bs, 16), np.uint8)
    wv = np.zeros((NB, hd, bs), np.uint8)
    wm = np.zeros((NB, bs), np.float32)
    orc.reshape_and_cache_kvc_fp8(key.float().numpy(), val.float().numpy(), wk, wv, wm, slots, bias, kind[4:], 0.5, 2.0)
    kd = torch.zeros((NB, hd // 16, bs, 16), dtype=torch.uint8, device=DEV)
    vd = torch.zeros((NB, hd, bs), dtype=torch.uint8, device=DEV)
    md = torch.zeros((NB, bs), dtype=torch.float32, device=DEV)
    ops.reshape_and_cache_kvc(key.to(DEV), val.to(DEV), kd, vd, md, torch.from_numpy(slots).to(DEV),
                              torch.from_numpy(bias).to(DEV), kind, 0.5, 2.0)
    torch.cuda.synchronize()
    gk, gv = to_reference(kd.cpu().numpy(), vd.cpu().numpy())
    assert np.array_equal(gk, wk) and np.array_equal(gv, wv) and np.array_equal(md.cpu().numpy(), wm)


# ------------------------------------------------------------------------------------- A6
def _compact_slot_major(st, evicted, k_np, v_np, mode="reference", foreign_list=False, **kw):
    ds = hdev.upload(st, DEV, mode=mode, **kw)
    eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, st, evicted)
    ks, vs = to_slot_major(k_np, v_np)
    k, v = torch.from_numpy(ks).to(DEV), torch.from_numpy(vs).to(DEV)
    if foreign_list:             # a list of unknown origin (no plan of its own): the op plans for itself
        cmi, cmc = cmi.clone(), cmc.clone()
    ops.execute_cache_moves(k, v, ds.cm.metrics, ds.cm.token_positions, cmi, cmc, ds.evicted_kv_offsets, 1, 16)
    torch.cuda.synchronize()
    gk, gv = to_reference(k.cpu().numpy(), v.cpu().numpy())
    return dict(k=gk, v=gv, metrics=ds.cm.metrics.cpu().numpy(), positions=ds.cm.token_positions.cpu().numpy(),
                cmi=cmi.cpu().numpy(), cmc=cmc.cpu().numpy())


@pytest.mark.parametrize("foreign_list", [False, True], ids=["planned", "own_plan"])
@pytest.mark.parametrize("name", golden_cases())
def test_golden_compaction_slot_major(name, foreign_list, slot_major):
    """A6 on the reference-generated end-to-end fixtures: the compacted cache, brought back to the reference's
    layout, hashes to what the reference's own ref_execute_cache_moves produced"""
    g = load_golden(name)
    if "uniform_evict" in g and int(g["uniform_evict"]):
        pytest.skip("schedule-only fixture")
    st = _state_from_golden(g)
    kw = dict(use_average=bool(int(g["use_average"])), num_sinks=int(g["num_sinks"]))
    if "bias" in g:
        kw.update(bias=g["bias"], position_bins=g["position_bins"], bias_weight=float(g["bias_weight"]))
    k, v = golden_caches(g)
    out = _compact_slot_major(st, g["evicted_blocks_per_seq"], k, v, foreign_list=foreign_list, **kw)
    np.testing.assert_array_equal(out["cmi"], g["ref_cache_moves_idx"])
    np.testing.assert_array_equal(sha(out["k"]), g["ref_k_sha256"])
    np.testing.assert_array_equal(sha(out["v"]), g["ref_v_sha256"])
    np.testing.assert_array_equal(out["metrics"], g["ref_metrics"])
    np.testing.assert_array_equal(out["positions"], g["ref_positions"])


@pytest.mark.parametrize("case", [
    # L, H, bs, seq_lens, protected, compressed, hd, elem bytes, frac
    (2, 4, 16, [300, 171, 90], [32, 5, 17], True, 128, 2, 0.7),
    (4, 8, 16, [700], 32, False, 128, 2, 0.875),
    (2, 2, 32, [260, 100], 33, False, 128, 1, 0.5),          # fp8 bytes, bs 32 (BASELINE configs[4]'s block)
    (2, 2, 16, [2100], 16, False, 64, 2, 0.5),
    (1, 2, 16, [900, 64], 2, False, 256, 2, 0.5),
    (2, 2, 16, [400], 7, False, 128, 4, 0.5),                # fp32 cache: 512-byte slots
    (2, 2, 16, [333], 3, False, 96, 2, 0.6),                 # 192-byte slots: the piece-wise kernel
    (2, 3, 4, [41, 23], 3, True, 8, 2, 0.4),                 # 16-byte slots
])
def test_random_compaction_slot_major(case, slot_major):
    L, H, bs, seq_lens, prot, compressed, hd, e, frac = case
    for seed in (0, 1):
        st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=seed,
                              protected=prot, compressed=compressed)
        nblk = ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)
        TH = L * H
        evicted = [int(max(int(nblk[b]) - (st.protected[b] + bs - 1) // bs * TH, 0) * frac) for b in range(st.num_seqs)]
        rng = np.random.default_rng(seed)
        npi = {1: np.uint8, 2: np.int16, 4: np.int32}[e]
        x = 16 // e
        info = np.iinfo(npi)
        k = rng.integers(info.min, info.max, size=(st.num_blocks, hd // x, bs, x), dtype=npi)
        v = rng.integers(info.min, info.max, size=(st.num_blocks, hd, bs), dtype=npi)
        want = oracle_pipeline(st, evicted, k.copy(), v.copy(), mode="reference")
        out = _compact_slot_major(st, evicted, k, v)
        for name in ("k", "v", "metrics", "positions"):
            assert np.array_equal(out[name], want[name]), (case, seed, name)


def test_single_move_heads_fill_whole_batches(slot_major):
    """the continual steady state: thousands of heads with one or two moves each (the kernel's LDS queue) -- every
    moved slot equals its source, nothing else changes"""
    st = synth.make_state(num_layers=8, num_kv_heads=8, block_size=16, seq_lens=[400] * 12, seed=3, protected=17,
                          steady_cap=320)
    evicted = [8 * 8] * 12                                           # one block per head
    k, v = synth.make_caches_u16(1, st.num_blocks, 128, 16)
    want = oracle_pipeline(st, evicted, k.copy(), v.copy(), mode="per_sequence")
    out = _compact_slot_major(st, evicted, k, v, mode="per_sequence")
    assert int(out["cmc"].sum()) > 500 and int(out["cmc"].max()) <= 16
    for name in ("k", "v", "metrics", "positions"):
        assert np.array_equal(out[name], want[name]), name


def test_compaction_through_the_c_abi_pieces(slot_major):
    """kvc_execute_cache_moves_slot_major_plan + kvc_execute_cache_moves_slot_major (what bench.py calls as the op's
    two halves) == the op"""
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=16, seq_lens=[500, 260], seed=4, protected=20)
    evicted = [100, 40]
    k, v = synth.make_caches_u16(2, st.num_blocks, 128, 16)
    want = oracle_pipeline(st, evicted, k.copy(), v.copy(), mode="reference")
    ds = hdev.upload(st, DEV, mode="reference")
    eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, st, evicted)
    ks, vs = to_slot_major(k, v)
    kd, vd = torch.from_numpy(ks).to(DEV), torch.from_numpy(vs).to(DEV)
    cmi2, cmc2 = cmi.clone(), cmc.clone()
    ops._execute_cache_moves(kd, vd, ds.cm.metrics, ds.cm.token_positions, cmi2, cmc2, ds.evicted_kv_offsets, "plan")
    ops._execute_cache_moves(kd, vd, ds.cm.metrics, ds.cm.token_positions, cmi2, cmc2, ds.evicted_kv_offsets, "apply")
    torch.cuda.synchronize()
    gk, gv = to_reference(kd.cpu().numpy(), vd.cpu().numpy())
    assert np.array_equal(gk, want["k"]) and np.array_equal(gv, want["v"])
    assert np.array_equal(ds.cm.metrics.cpu().numpy(), want["metrics"])


def test_unsupported_slot_size_raises(slot_major):
    z = lambda *s: torch.zeros(s, dtype=torch.int32, device=DEV)
    k = torch.zeros((4, 1, 16, 4), dtype=torch.float16, device=DEV)          # head size 4: 8-byte slots
    v = torch.zeros((4, 4, 16), dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError, match="Unsupported head size"):
        ops.execute_cache_moves(k, v, torch.zeros((4, 16), device=DEV), z(4, 16), z(8, 2), z(1, 1, 1), z(1, 1, 1), 1, 16)


# ------------------------------------------------------------------------------------- F3
def _attn(g, c, pos, last, buf, version="v2", record=True, fill=-1.0, max_ctx=None, fp8=None):
    """like tests/test_gpu_attention.py::_run_gpu, with the caches permuted into slot-major blocks"""
    tdt = torch.float16 if c["dtype"] == "f16" else torch.bfloat16
    t = lambda bits: torch.from_numpy(np.ascontiguousarray(bits)).to(DEV).view(tdt)
    q = t(g["query_bits"])
    if fp8 is None:
        ks, vs = to_slot_major(np.ascontiguousarray(g["key_cache_bits"]), np.ascontiguousarray(g["value_cache_bits"]))
        kc, vc = t(ks), t(vs)
        kind, k_scale, v_scale = "auto", 1.0, 1.0
    else:
        kq, vq, kind, k_scale, v_scale = fp8
        ks, vs = to_slot_major(kq, vq)
        kc, vc = torch.from_numpy(ks).to(DEV), torch.from_numpy(vs).to(DEV)
    S, Hq, hd = q.shape
    Hkv = int(g["num_kv_heads"])
    NB, _, bs = vc.shape
    qpk = Hq // Hkv
    out = torch.full_like(q, 7.0)
    km = torch.full((NB, bs, qpk), fill, dtype=torch.float32, device=DEV)
    mx = int(g["context_lens"].max()) if max_ctx is None else max_ctx
    slopes = None if c["slopes"] is None else torch.from_numpy(c["slopes"]).to(DEV)
    args = (q, kc, vc, Hkv, float(g["scale"]), torch.from_numpy(g["block_tables"]).to(DEV),
            torch.from_numpy(g["context_lens"]).to(DEV), torch.from_numpy(pos).to(DEV), torch.from_numpy(last).to(DEV),
            torch.from_numpy(buf).to(DEV), bs, mx, slopes, kind, k_scale, v_scale, record)
    if version == "v1":
        ops.paged_attention_kvc_v1(out, km, *args)
    else:
        parts = (mx + 511) // 512
        es = torch.empty((S, Hq, parts), dtype=torch.float32, device=DEV)
        ops.paged_attention_kvc_v2(out, km, es, torch.empty_like(es), torch.empty((S, Hq, parts, hd), dtype=tdt, device=DEV),
                                   torch.full_like(km, 123.0), *args)
    torch.cuda.synchronize()
    return out.float().cpu().numpy(), km.cpu().numpy()


@pytest.fixture(params=[1, 2], ids=["partitioned", "single_pass"])
def attn_mode(request):
    ops.set_attention_schedule(request.param)
    yield request.param
    ops.set_attention_schedule(0)


ATTN_CASES = sorted(n for n in os.listdir(GOLDEN_DIR) if n.startswith("attn_decode_"))


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("case", [n[:-4] for n in ATTN_CASES])
def test_attention_matches_reference_twin_slot_major(case, version, attn_mode, slot_major):
    """the fixtures of the reference test's PyTorch twin (tests/kernels/test_kvcompress_attention.py), the reference's
    own tolerances (:145, :356-357)"""
    g = load_golden(case)
    c = decode_golden(g)
    NB, hd, bs = c["vc"].shape
    S, Hq = c["q"].shape[:2]
    qpk = Hq // int(g["num_kv_heads"])
    pos = np.zeros((NB, bs), np.int32)
    run = lambda: _attn(g, c, pos, np.full(S, 10, np.int32), np.zeros(S, np.int32), version)
    if hd not in (64, 128, 256) or bs not in (16, 32):
        with pytest.raises(RuntimeError, match="slot-major"):
            run()
        return
    out, km = run()
    rtol = 1e-5 if c["dtype"] == "f16" else 1e-4
    assert np.allclose(km, g["ref_probs"], rtol=rtol, atol=1e-8)
    assert np.allclose(out, g["ref_out"], atol=1e-3, rtol=1e-5)
    assert ((g["ref_probs"] == -1.0) == (km == -1.0)).all()


@pytest.mark.parametrize("shape", [
    # S, Hq, Hkv, hd, bs, ctx_lo, ctx_hi, dtype, alibi
    (3, 8, 2, 128, 16, 1, 300, "f16", False),
    (2, 32, 8, 128, 16, 400, 1500, "f16", False),       # Llama-3-8B GQA, several partitions
    (2, 8, 1, 128, 32, 100, 2100, "f16", True),          # qpk 8 (Llama-3-70B), bs 32, ALiBi
    (2, 4, 4, 64, 16, 30, 700, "bf16", False),           # MHA (qpk 1), 128-byte slots
    (2, 8, 2, 256, 16, 20, 530, "bf16", False),          # 512-byte slots
    (2, 16, 2, 256, 32, 20, 900, "f16", False),          # qpk 8 at head size 256
    (1, 16, 2, 128, 16, 3000, 4100, "f16", False),       # qpk 8 at a 4k cap: 8-wave single pass
    (1, 4, 1, 128, 16, 6000, 8300, "bf16", False),       # qpk 4 at 8k
    (2, 6, 2, 64, 32, 5, 1100, "f16", True),             # qpk 3
    (2, 24, 2, 128, 16, 40, 1300, "f16", False),         # qpk 12: all query heads of a KV head in one MFMA operand
    (1, 40, 2, 128, 32, 300, 800, "bf16", False),        # qpk 20: two query groups per KV head
])
def test_attention_matches_oracle_slot_major(shape, attn_mode, slot_major):
    S, Hq, Hkv, hd, bs, lo, hi, dt, alibi = shape
    rng = np.random.default_rng(hash(shape) % (2 ** 31))
    g, c, pos, last = make_state(rng, S, Hq, Hkv, hd, bs, lo, hi, dtype=dt, magnitude=1.0, alibi=alibi)
    buf = rng.integers(0, 40, size=S).astype(np.int32)
    ref_out, ref_km = oracle_decode(c, g, pos, last, buf)
    for version in ("v1", "v2"):
        out, km = _attn(g, c, pos, last, buf, version)
        assert ((ref_km == -1.0) == (km == -1.0)).all()
        rec = ref_km != -1.0
        assert np.allclose(km[rec], ref_km[rec], rtol=2e-4, atol=1e-9)
        tol = 2e-3 if dt == "f16" else 1.6e-2
        assert np.allclose(out, ref_out, atol=tol, rtol=tol)


def test_attention_unsupported_shapes_raise_slot_major(slot_major):
    rng = np.random.default_rng(1)
    for shape in ((2, 6, 2, 96, 32), (2, 8, 2, 128, 8)):      # head size 96, block size 8
        S, Hq, Hkv, hd, bs = shape
        g, c, pos, last = make_state(rng, S, Hq, Hkv, hd, bs, 10, 100)
        with pytest.raises(RuntimeError, match="slot-major"):
            _attn(g, c, pos, last, np.zeros(S, np.int32), "v1")


@pytest.mark.parametrize("scales", [(0.5, 2.0), (1.0, 1.0)], ids=["scaled", "unit_scale"])
@pytest.mark.parametrize("kind,bs,dt", [("fp8_e4m3", 16, "f16"), ("fp8_e5m2", 32, "f16"), ("fp8", 32, "bf16"),
                                        ("fp8_e5m2", 16, "bf16")])
def test_attention_fp8_cache_slot_major(kind, bs, dt, scales, attn_mode, slot_major):
    rng = np.random.default_rng(21)
    S, Hq, Hkv, hd, lo, hi = 2, 8, 2, 128, 40, 900
    g, c, pos, last = make_state(rng, S, Hq, Hkv, hd, bs, lo, hi, dtype=dt)
    NB = c["vc"].shape[0]
    tf8 = torch.float8_e5m2 if kind == "fp8_e5m2" else torch.float8_e4m3fn
    k_scale, v_scale = scales
    kq = torch.from_numpy(rng.standard_normal((NB, hd // 16, bs, 16)).astype(np.float32)).to(tf8)
    vq = torch.from_numpy(rng.standard_normal((NB, hd, bs)).astype(np.float32)).to(tf8)
    tdt = torch.float16 if dt == "f16" else torch.bfloat16
    kd = (kq.float() * k_scale).to(tdt).float().numpy()
    vd = (vq.float() * v_scale).to(tdt).float().numpy()
    c = dict(c, kc=kd.reshape(NB, hd // 16, bs, 16).transpose(0, 1, 3, 2).reshape(NB, hd, bs)
             .transpose(0, 2, 1).reshape(NB, bs, hd // 8, 8).transpose(0, 2, 1, 3).copy(), vc=vd)
    buf = np.zeros(S, np.int32)
    ref_out, ref_km = oracle_decode(c, g, pos, last, buf)
    out, km = _attn(g, c, pos, last, buf, "v2", fp8=(kq.view(torch.uint8).numpy(), vq.view(torch.uint8).numpy(),
                                                    kind, k_scale, v_scale))
    rec = ref_km != -1.0
    assert ((ref_km == -1.0) == (km == -1.0)).all()
    assert np.allclose(km[rec], ref_km[rec], rtol=2e-4, atol=1e-9)
    tol = 4e-3 if dt == "f16" else 3e-2
    assert np.allclose(out, ref_out, atol=tol, rtol=tol)


@pytest.mark.parametrize("shape", [(3, 8, 2, 128, 16, 1, 700, "f16"), (2, 16, 2, 128, 32, 100, 1300, "bf16"),
                                   (1, 16, 2, 128, 16, 2500, 4100, "f16")])
def test_attention_fused_metrics_slot_major(shape, attn_mode, slot_major):
    """metrics += sum_q p^2 inside the attention == kv_metric_out + the oracle's aggregate_decode, bit for bit (the
    weights do not depend on how V is laid out; the K addressing is what this checks)"""
    S, Hq, Hkv, hd, bs, lo, hi, dt = shape
    rng = np.random.default_rng(hash(shape) % (2 ** 31))
    g, c, pos, last = make_state(rng, S, Hq, Hkv, hd, bs, lo, hi, dtype=dt)
    buf = rng.integers(0, 30, size=S).astype(np.int32)
    out_ref, km = _attn(g, c, pos, last, buf, "v1", fill=0.0)
    NB, qpk = km.shape[0], Hq // Hkv
    m0 = rng.random((NB, bs)).astype(np.float32)
    want = m0.copy()
    orc.aggregate_decode(want, km, use_l2=True)
    tdt = torch.float16 if dt == "f16" else torch.bfloat16
    t = lambda bits: torch.from_numpy(np.ascontiguousarray(bits)).to(DEV).view(tdt)
    ks, vs = to_slot_major(np.ascontiguousarray(g["key_cache_bits"]), np.ascontiguousarray(g["value_cache_bits"]))
    q = t(g["query_bits"])
    out = torch.zeros_like(q)
    got = torch.from_numpy(m0.copy()).to(DEV)
    ops.paged_attention_kvc_fused_metrics(
        out, got, q, t(ks), t(vs), Hkv, float(g["scale"]), torch.from_numpy(g["block_tables"]).to(DEV),
        torch.from_numpy(g["context_lens"]).to(DEV), torch.from_numpy(pos).to(DEV), torch.from_numpy(last).to(DEV),
        torch.from_numpy(buf).to(DEV), bs, int(g["context_lens"].max()), None, "auto", 1.0, 1.0, use_l2=True,
        temp_metrics=torch.empty((NB, bs, qpk), dtype=torch.float32, device=DEV))
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(out.float().cpu().numpy(), out_ref)


def test_written_cache_is_attended_to_in_the_same_layout(slot_major):
    """writer and reader together: tokens written by reshape_and_cache_kvc are what the attention reads (the oracle
    attends to the same tokens in the reference's layout)"""
    rng = np.random.default_rng(9)
    S, Hq, Hkv, hd, bs = 2, 8, 2, 128, 16
    g, c, pos, last = make_state(rng, S, Hq, Hkv, hd, bs, 200, 420, dtype="f16")
    NB = c["vc"].shape[0]
    ctx = g["context_lens"]
    # write every cached token through the op (token-major slot mapping per sequence and head)
    kd = torch.zeros((NB, hd // 8, bs, 8), dtype=torch.float16, device=DEV)
    vd = torch.zeros((NB, hd, bs), dtype=torch.float16, device=DEV)
    md = torch.zeros((NB, bs), dtype=torch.float32, device=DEV)
    kref = np.ascontiguousarray(g["key_cache_bits"]).view(np.float16)
    vref = np.ascontiguousarray(g["value_cache_bits"]).view(np.float16)
    for s in range(S):
        T = int(ctx[s].max())
        key = np.zeros((T, Hkv, hd), np.float16)
        val = np.zeros((T, Hkv, hd), np.float16)
        slots = np.full((T, Hkv), -1, np.int64)
        for h in range(Hkv):
            for tkn in range(int(ctx[s, h])):
                blk = int(g["block_tables"][s, h, tkn // bs])
                slots[tkn, h] = blk * bs + tkn % bs
                key[tkn, h] = kref[blk, :, tkn % bs, :].reshape(-1)
                val[tkn, h] = vref[blk, :, tkn % bs]
        ops.reshape_and_cache_kvc(torch.from_numpy(key).to(DEV), torch.from_numpy(val).to(DEV), kd, vd, md,
                                  torch.from_numpy(slots.reshape(-1)).to(DEV), torch.zeros(Hkv, device=DEV), "auto", 1.0, 1.0)
    torch.cuda.synchronize()
    # the written cache, in the reference's layout, holds the state's tokens (slots past the contexts stay zero)
    gk, gv = to_reference(kd.view(torch.int16).cpu().numpy(), vd.view(torch.int16).cpu().numpy())
    g2 = dict(g, key_cache_bits=gk, value_cache_bits=gv)
    c2 = dict(c, kc=gk.view(np.float16), vc=gv.view(np.float16))
    buf = np.zeros(S, np.int32)
    ref_out, ref_km = oracle_decode(c2, g2, pos, last, buf)
    out, km = _attn(g2, c2, pos, last, buf, "v2")
    rec = ref_km != -1.0
    assert np.allclose(km[rec], ref_km[rec], rtol=2e-4, atol=1e-9) and np.allclose(out, ref_out, atol=2e-3, rtol=2e-3)
