"""F3 parity: the HIP decode attention (paged_attention_kvc_v1 / _v2) against the golden
vectors from the reference test's PyTorch twin and against the oracle on seeded states.

Floating point, so the bar is a tolerance (stated per assert), the reference's own:
  * softmax weights (the KV metric): rtol 1e-5 + atol 1e-8 on the reference-recipe inputs
    (tests/kernels/test_kvcompress_attention.py:145), 2e-4 relative on N(0,1) inputs where
    fp32 summation order matters;
  * attention output: atol 1e-3, rtol 1e-5 (:356-357) on the recipe inputs."""
import os

import numpy as np
import pytest

from tests.attn_helpers import decode_golden, make_state, oracle_decode
from tests.helpers import GOLDEN_DIR, load_golden

pytestmark = pytest.mark.gpu

ATTN_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith("attn_decode_"))


def _set_mode(mode):
    """0 automatic, 1 partitioned (+ reduce / rescale kernels), 2 single pass when it fits;
    travels per call in kvc_attention_params.schedule"""
    from vllm_kvcompress_amd import _custom_ops as ops
    ops.set_attention_schedule(mode)


@pytest.fixture(params=[1, 2], ids=["partitioned", "single_pass"])
def attn_mode(request):
    _set_mode(request.param)
    yield request.param
    _set_mode(0)


def _run_gpu(g, c, pos, last, buf, version="v2", record=True, fill=-1.0, max_ctx=None):
    import torch
    from vllm_kvcompress_amd import _custom_ops as ops
    dev = "cuda:0"
    tdt = torch.float16 if c["dtype"] == "f16" else torch.bfloat16
    t = lambda bits: torch.from_numpy(np.ascontiguousarray(bits)).to(dev).view(tdt)
    q, kc, vc = t(g["query_bits"]), t(g["key_cache_bits"]), t(g["value_cache_bits"])
    S, Hq, hd = q.shape
    Hkv = int(g["num_kv_heads"])
    NB, _, bs = vc.shape
    qpk = Hq // Hkv
    out = torch.full_like(q, 7.0)                    # every output element must be written
    km = torch.full((NB, bs, qpk), fill, dtype=torch.float32, device=dev)
    bt = torch.from_numpy(g["block_tables"]).to(dev)
    ctx = torch.from_numpy(g["context_lens"]).to(dev)
    mx = int(g["context_lens"].max()) if max_ctx is None else max_ctx
    slopes = None if c["slopes"] is None else torch.from_numpy(c["slopes"]).to(dev)
    args = (q, kc, vc, Hkv, float(g["scale"]), bt, ctx, torch.from_numpy(pos).to(dev),
            torch.from_numpy(last).to(dev), torch.from_numpy(buf).to(dev), bs, mx, slopes, "auto",
            1.0, 1.0, record)
    if version == "v1":
        ops.paged_attention_kvc_v1(out, km, *args)
    else:
        parts = (mx + 511) // 512
        exp_sums = torch.empty((S, Hq, parts), dtype=torch.float32, device=dev)
        max_logits = torch.empty_like(exp_sums)
        tmp_out = torch.empty((S, Hq, parts, hd), dtype=tdt, device=dev)
        tmp_km = torch.full_like(km, 123.0)          # stale junk must never leak out
        ops.paged_attention_kvc_v2(out, km, exp_sums, max_logits, tmp_out, tmp_km, *args)
    torch.cuda.synchronize()
    return out.float().cpu().numpy(), km.cpu().numpy()


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("case", ATTN_CASES)
def test_decode_attention_matches_reference_twin(case, version, attn_mode):
    g = load_golden(case)
    c = decode_golden(g)
    NB, _, bs = c["vc"].shape
    S = c["q"].shape[0]
    pos = np.zeros((NB, bs), np.int32)
    out, km = _run_gpu(g, c, pos, np.full(S, 10, np.int32), np.zeros(S, np.int32), version)
    rtol = 1e-5 if c["dtype"] == "f16" else 1e-4          # bf16 twin rounds its logits
    assert np.allclose(km, g["ref_probs"], rtol=rtol, atol=1e-8)
    # the reference compares the kernel's fp16/bf16 output with the twin at atol 1e-3
    assert np.allclose(out, g["ref_out"], atol=1e-3, rtol=1e-5)
    assert ((g["ref_probs"] == -1.0) == (km == -1.0)).all()     # nothing else is written


@pytest.mark.parametrize("shape", [
    # S, Hq, Hkv, hd, bs, ctx_lo, ctx_hi, dtype, alibi
    (3, 8, 2, 128, 16, 1, 300, "f16", False),
    (2, 32, 8, 128, 16, 400, 1500, "f16", False),       # Llama-3-8B GQA, several partitions
    (2, 8, 1, 128, 32, 100, 2100, "f16", True),          # qpk 8, bs 32, ALiBi
    (2, 4, 4, 64, 16, 30, 700, "bf16", False),           # MHA (qpk 1)
    (1, 40, 2, 128, 16, 50, 600, "f16", False),          # qpk 20 -> two query groups
    (2, 8, 2, 256, 16, 20, 530, "bf16", False),
    (2, 6, 2, 96, 32, 20, 530, "f16", False),
    (1, 16, 2, 128, 16, 3000, 4100, "f16", False),       # qpk 8 at a 4k cap: 8-wave single pass
    (1, 4, 1, 128, 16, 6000, 8300, "bf16", False),       # qpk 4 at 8k: 8-wave single pass
    (3, 8, 2, 128, 8, 1, 700, "f16", False),             # block size 8 (a 16-token sub-block spans two blocks)
    (2, 4, 4, 64, 8, 5, 1100, "bf16", True),
    (2, 8, 2, 128, 1, 1, 600, "f16", False),             # block size 1: every token in a block of its own
    (2, 4, 1, 128, 1, 300, 900, "bf16", True),
])
def test_decode_attention_matches_oracle(shape, attn_mode):
    S, Hq, Hkv, hd, bs, lo, hi, dt, alibi = shape
    rng = np.random.default_rng(hash(shape) % (2 ** 31))
    g, c, pos, last = make_state(rng, S, Hq, Hkv, hd, bs, lo, hi, dtype=dt, magnitude=1.0, alibi=alibi)
    buf = rng.integers(0, 40, size=S).astype(np.int32)
    ref_out, ref_km = oracle_decode(c, g, pos, last, buf)
    for version in ("v1", "v2"):
        out, km = _run_gpu(g, c, pos, last, buf, version)
        assert ((ref_km == -1.0) == (km == -1.0)).all()        # same slots recorded
        rec = ref_km != -1.0
        # fp32 softmax over up to 2100 N(0,1)-scaled logits: summation order differs
        assert np.allclose(km[rec], ref_km[rec], rtol=2e-4, atol=1e-9)
        # output is rounded to fp16 / bf16 (1 ulp = 2^-11 / 2^-8 relative)
        tol = 2e-3 if dt == "f16" else 1.6e-2
        assert np.allclose(out, ref_out, atol=tol, rtol=tol)


def test_decode_attention_without_metrics_and_empty_batch():
    import torch
    from vllm_kvcompress_amd import _custom_ops as ops
    rng = np.random.default_rng(3)
    g, c, pos, last = make_state(rng, 2, 8, 2, 128, 16, 10, 700)
    buf = np.zeros(2, np.int32)
    out, km = _run_gpu(g, c, pos, last, buf, "v2", record=False)
    ref_out, _ = oracle_decode(c, g, pos, last, buf, record=False)
    assert (km == -1.0).all()
    assert np.allclose(out, ref_out, atol=2e-3, rtol=2e-3)
    # unsupported shapes raise like the reference's TORCH_CHECKs
    q = torch.zeros((1, 4, 80), dtype=torch.float16, device="cuda:0")
    kc = torch.zeros((4, 10, 16, 8), dtype=torch.float16, device="cuda:0")
    vc = torch.zeros((4, 80, 16), dtype=torch.float16, device="cuda:0")
    z = torch.zeros((1, 1, 2), dtype=torch.int32, device="cuda:0")
    one = torch.ones((1, 1), dtype=torch.int32, device="cuda:0")
    with pytest.raises(RuntimeError, match="Unsupported head size"):
        ops.paged_attention_kvc_v1(torch.zeros_like(q), torch.zeros((4, 16, 4), device="cuda:0"), q,
                                   kc, vc, 1, 0.1, z, one, torch.zeros((4, 16), dtype=torch.int32,
                                                                       device="cuda:0"),
                                   one[0], one[0], 16, 16, None, "auto", 1.0, 1.0, True)


def test_decode_attention_full_size_properties():
    """BASELINE configs[2] shape of one decode step of one layer: 64 sequences x 8 KV heads at
    a 4k-token cap.  Size-independent properties: weights of every (seq, query head) sum to
    1 - 1e-6/(sum+1e-6); v1 == v2 bit for bit; output is a convex combination of V rows."""
    import torch
    from vllm_kvcompress_amd import _custom_ops as ops
    dev = "cuda:0"
    S, Hq, Hkv, hd, bs, ctx_len = 64, 32, 8, 128, 16, 4097
    nblk = (ctx_len + bs - 1) // bs
    NB = S * Hkv * nblk
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    kc = (torch.randn((NB, hd // 8, bs, 8), device=dev, generator=gen) * 0.5).half()
    vc = torch.rand((NB, hd, bs), device=dev, generator=gen).half()          # in [0, 1)
    q = torch.randn((S, Hq, hd), device=dev, generator=gen).half()
    bt = torch.randperm(NB, device=dev, generator=gen).to(torch.int32).view(S, Hkv, nblk)
    ctx = torch.full((S, Hkv), ctx_len, dtype=torch.int32, device=dev)
    pos = torch.zeros((NB, bs), dtype=torch.int32, device=dev)
    last = torch.full((S,), 10, dtype=torch.int32, device=dev)
    buf = torch.zeros((S,), dtype=torch.int32, device=dev)
    outs, kms = [], []
    for version in ("v1", "v2", "single_pass"):
        _set_mode(2 if version == "single_pass" else 1)
        out = torch.zeros_like(q)
        km = torch.zeros((NB, bs, Hq // Hkv), dtype=torch.float32, device=dev)
        args = (q, kc, vc, Hkv, hd ** -0.5, bt, ctx, pos, last, buf, bs, ctx_len, None, "auto", 1.0,
                1.0, True)
        if version == "v1":
            ops.paged_attention_kvc_v1(out, km, *args)
        else:
            parts = (ctx_len + 511) // 512
            es = torch.empty((S, Hq, parts), dtype=torch.float32, device=dev)
            ops.paged_attention_kvc_v2(out, km, es, torch.empty_like(es),
                                       torch.empty((S, Hq, parts, hd), dtype=q.dtype, device=dev),
                                       torch.empty_like(km), *args)
        outs.append(out)
        kms.append(km)
    torch.cuda.synchronize()
    _set_mode(0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(kms[0], kms[1])
    # the single-pass kernel (wave-local running max) agrees with the partitioned one
    assert torch.allclose(kms[2], kms[0], rtol=1e-5, atol=1e-9)
    assert torch.allclose(outs[2].float(), outs[0].float(), atol=2e-3, rtol=2e-3)
    for km in (kms[0], kms[2]):
        sums = km[bt.long().view(-1)].view(S, Hkv, nblk * bs, Hq // Hkv).sum(2)
        assert bool(((sums - 1.0).abs() < 1e-4).all()), float((sums - 1.0).abs().max())
    km = kms[0]
    # per (seq, kv head): gather its blocks, sum weights per query head
    sums = km[bt.long().view(-1)].view(S, Hkv, nblk * bs, Hq // Hkv).sum(2)
    assert bool(((sums - 1.0).abs() < 1e-4).all()), float((sums - 1.0).abs().max())
    assert bool((km >= 0).all())
    # tail slots beyond the context are never written
    tail = km[bt[:, :, -1].long().view(-1)][:, (ctx_len - 1) % bs + 1:, :]
    assert bool((tail == 0).all())
    o = outs[0].float()
    assert bool((o >= -1e-3).all()) and bool((o <= 1.0 + 1e-3).all())


def test_decode_attention_through_dispatcher():
    """torch.ops._C.kvcompress_paged_attention_v1 (csrc/torch_bindings.cpp:52-64) resolves to
    the HIP kernel after torch_ops.register()"""
    import torch
    from vllm_kvcompress_amd import torch_ops
    torch_ops.register()
    rng = np.random.default_rng(11)
    g, c, pos, last = make_state(rng, 2, 8, 2, 128, 16, 10, 90)
    buf = np.zeros(2, np.int32)
    ref_out, ref_km = oracle_decode(c, g, pos, last, buf)
    dev = "cuda:0"
    t = lambda bits: torch.from_numpy(np.ascontiguousarray(bits)).to(dev).view(torch.float16)
    q, kc, vc = t(g["query_bits"]), t(g["key_cache_bits"]), t(g["value_cache_bits"])
    out = torch.zeros_like(q)
    km = torch.full((vc.shape[0], 16, 4), -1.0, dtype=torch.float32, device=dev)
    torch.ops._C.kvcompress_paged_attention_v1(
        out, km, q, kc, vc, 2, float(g["scale"]), torch.from_numpy(g["block_tables"]).to(dev),
        torch.from_numpy(g["context_lens"]).to(dev), torch.from_numpy(pos).to(dev),
        torch.from_numpy(last).to(dev), torch.from_numpy(buf).to(dev), 16,
        int(g["context_lens"].max()), None, "auto", 1.0, 1.0, True)
    torch.cuda.synchronize()
    rec = ref_km != -1.0
    assert np.allclose(km.cpu().numpy()[rec], ref_km[rec], rtol=2e-4, atol=1e-9)
    assert np.allclose(out.float().cpu().numpy(), ref_out, atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("scales", [(0.5, 2.0), (1.0, 1.0)], ids=["scaled", "unit_scale"])
@pytest.mark.parametrize("kind,bs,dt", [("fp8_e4m3", 16, "f16"), ("fp8_e5m2", 32, "f16"),
                                        ("fp8", 32, "bf16"), ("fp8_e5m2", 16, "bf16")])
def test_decode_attention_fp8_cache(kind, bs, dt, scales, attn_mode):
    """fp8 K/V cache ([NB, hd/16, bs, 16] / [NB, hd, bs] bytes): the kernel dequantises
    T(float(fp8) * scale) on load; the oracle gets the same dequantised values."""
    import torch
    from vllm_kvcompress_amd import _custom_ops as ops
    rng = np.random.default_rng(21)
    S, Hq, Hkv, hd, lo, hi = 2, 8, 2, 128, 40, 900
    g, c, pos, last = make_state(rng, S, Hq, Hkv, hd, bs, lo, hi, dtype=dt)
    NB = c["vc"].shape[0]
    tf8 = torch.float8_e5m2 if kind == "fp8_e5m2" else torch.float8_e4m3fn
    k_scale, v_scale = scales
    # quantise random values (saturating cast of N(0,1)) -> bytes; x = 16 for the K layout
    kq = torch.from_numpy(rng.standard_normal((NB, hd // 16, bs, 16)).astype(np.float32)).to(tf8)
    vq = torch.from_numpy(rng.standard_normal((NB, hd, bs)).astype(np.float32)).to(tf8)
    tdt = torch.float16 if dt == "f16" else torch.bfloat16
    kd = (kq.float() * k_scale).to(tdt).float().numpy()            # what the kernel must see
    vd = (vq.float() * v_scale).to(tdt).float().numpy()
    c = dict(c, kc=kd.reshape(NB, hd // 16, bs, 16).transpose(0, 1, 3, 2).reshape(NB, hd, bs)
             .transpose(0, 2, 1).reshape(NB, bs, hd // 8, 8).transpose(0, 2, 1, 3).copy(), vc=vd)
    buf = np.zeros(S, np.int32)
    ref_out, ref_km = oracle_decode(c, g, pos, last, buf)
    dev = "cuda:0"
    q = torch.from_numpy(np.ascontiguousarray(g["query_bits"])).to(dev).view(tdt)
    out = torch.zeros_like(q)
    km = torch.full((NB, bs, Hq // Hkv), -1.0, dtype=torch.float32, device=dev)
    mx = int(g["context_lens"].max())
    parts = (mx + 511) // 512
    es = torch.empty((S, Hq, parts), dtype=torch.float32, device=dev)
    ops.paged_attention_kvc_v2(
        out, km, es, torch.empty_like(es), torch.empty((S, Hq, parts, hd), dtype=tdt, device=dev),
        torch.empty_like(km), q, kq.view(torch.uint8).to(dev), vq.view(torch.uint8).to(dev), Hkv,
        float(g["scale"]), torch.from_numpy(g["block_tables"]).to(dev),
        torch.from_numpy(g["context_lens"]).to(dev), torch.from_numpy(pos).to(dev),
        torch.from_numpy(last).to(dev), torch.from_numpy(buf).to(dev), bs, mx, None, kind, k_scale,
        v_scale, True)
    torch.cuda.synchronize()
    rec = ref_km != -1.0
    assert ((ref_km == -1.0) == (km.cpu().numpy() == -1.0)).all()
    assert np.allclose(km.cpu().numpy()[rec], ref_km[rec], rtol=2e-4, atol=1e-9)
    tol = 4e-3 if dt == "f16" else 3e-2
    assert np.allclose(out.float().cpu().numpy(), ref_out, atol=tol, rtol=tol)


@pytest.mark.parametrize("shape", [(3, 8, 2, 128, 16, 1, 700, "f16"), (2, 16, 2, 128, 32, 100, 1300, "bf16"),
                                   (2, 6, 2, 64, 16, 10, 400, "f16"), (1, 16, 2, 128, 16, 2500, 4100, "f16")])
@pytest.mark.parametrize("use_l2", [True, False])
def test_decode_attention_fused_metric_aggregation(shape, use_l2, attn_mode):
    """metrics += sum_q p^2 inside the attention == kv_metric_out + aggregate_decode, bit for
    bit (same products, same summation order)"""
    import torch
    from vllm_kvcompress_amd import _custom_ops as ops
    from vllm_kvcompress_amd import _lib
    S, Hq, Hkv, hd, bs, lo, hi, dt = shape
    rng = np.random.default_rng(hash(shape) % (2 ** 31))
    g, c, pos, last = make_state(rng, S, Hq, Hkv, hd, bs, lo, hi, dtype=dt)
    buf = rng.integers(0, 30, size=S).astype(np.int32)
    out_ref, km = _run_gpu(g, c, pos, last, buf, "v1", fill=0.0)
    dev = "cuda:0"
    NB = km.shape[0]
    qpk = Hq // Hkv
    m0 = torch.from_numpy(rng.random((NB, bs)).astype(np.float32)).to(dev)
    # unfused: aggregate the stored weights with the A2a kernel
    want = m0.clone()
    temp = torch.from_numpy(km).to(dev).contiguous()
    lib = _lib.load()
    _lib.check(lib.kvc_aggregate_decode(want.data_ptr(), temp.data_ptr(), NB * bs, qpk, int(use_l2), 0,
                                        torch.cuda.current_stream().cuda_stream))
    # fused
    tdt = torch.float16 if dt == "f16" else torch.bfloat16
    t = lambda bits: torch.from_numpy(np.ascontiguousarray(bits)).to(dev).view(tdt)
    q, kc, vc = t(g["query_bits"]), t(g["key_cache_bits"]), t(g["value_cache_bits"])
    out = torch.zeros_like(q)
    got = m0.clone()
    ops.paged_attention_kvc_fused_metrics(
        out, got, q, kc, vc, Hkv, float(g["scale"]), torch.from_numpy(g["block_tables"]).to(dev),
        torch.from_numpy(g["context_lens"]).to(dev), torch.from_numpy(pos).to(dev),
        torch.from_numpy(last).to(dev), torch.from_numpy(buf).to(dev), bs,
        int(g["context_lens"].max()), None, "auto", 1.0, 1.0, use_l2=use_l2,
        temp_metrics=torch.empty((NB, bs, qpk), dtype=torch.float32, device=dev))
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert np.array_equal(out.float().cpu().numpy(), out_ref)


def test_decode_attention_empty_head(attn_mode):
    """a head with nothing cached (context_len 0) gets a zero output and writes no metric;
    the other heads of the same call are unaffected"""
    rng = np.random.default_rng(8)
    g, c, pos, last = make_state(rng, 2, 8, 2, 128, 16, 30, 650)
    g["context_lens"] = g["context_lens"].copy()
    g["context_lens"][0, 1] = 0
    buf = np.zeros(2, np.int32)
    ref_out, ref_km = oracle_decode(c, g, pos, last, buf)           # the oracle skips empty heads (zeros)
    out, km = _run_gpu(g, c, pos, last, buf, "v2", max_ctx=650)
    assert ((ref_km == -1.0) == (km == -1.0)).all()
    rec = ref_km != -1.0
    assert np.allclose(km[rec], ref_km[rec], rtol=2e-4, atol=1e-9)
    assert np.allclose(out, ref_out, atol=2e-3, rtol=2e-3)
    assert (out[0, 4:8] == 0).all()
