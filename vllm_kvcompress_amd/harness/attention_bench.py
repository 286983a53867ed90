"""Synthetic decode-attention workload: one layer step of
``paged_attention_kvc_v2`` over a per-head paged cache; reports algorithmic HBM GB/s.

Algorithmic bytes per cached token and KV head: K row + V row (2 * hd * e), and with metric
output the position read (4) and the metric write (4 * qpk)."""
from __future__ import annotations


def run(S, ctx_len, Hq=32, Hkv=8, hd=128, bs=16, iters=20, record=True, dtype="f16", kv_dtype="auto",
        k_scale=1.0, fused=False):
    import torch
    from vllm_kvcompress_amd import _custom_ops as ops
    dev = "cuda:0"
    tdt = torch.float16 if dtype == "f16" else torch.bfloat16
    nblk = (ctx_len + bs - 1) // bs
    NB = S * Hkv * nblk
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    if kv_dtype == "auto":
        kv = torch.randint(-20000, 20000, (2, NB, bs * hd), dtype=torch.int16, device=dev, generator=gen)
        kc = kv[0].view(tdt).view(NB, hd // 8, bs, 8)
        vc = kv[1].view(tdt).view(NB, hd, bs)
        kc.mul_(1e-3)
    else:                                   # fp8 bytes below the NaN / inf encodings
        kv = torch.randint(0, 0x78, (2, NB, bs * hd), dtype=torch.uint8, device=dev, generator=gen)
        kc = kv[0].view(NB, hd // 16, bs, 16)
        vc = kv[1].view(NB, hd, bs)
    q = torch.randn((S, Hq, hd), device=dev, generator=gen).to(tdt)
    bt = torch.randperm(NB, device=dev, generator=gen).to(torch.int32).view(S, Hkv, nblk)
    ctx = torch.full((S, Hkv), ctx_len, dtype=torch.int32, device=dev)
    pos = torch.zeros((NB, bs), dtype=torch.int32, device=dev)
    last = torch.full((S,), 10, dtype=torch.int32, device=dev)
    buf = torch.zeros((S,), dtype=torch.int32, device=dev)
    out = torch.zeros_like(q)
    qpk = Hq // Hkv
    km = torch.zeros((NB, bs, qpk), dtype=torch.float32, device=dev)
    parts = (ctx_len + 511) // 512
    es = torch.empty((S, Hq, parts), dtype=torch.float32, device=dev)
    ml = torch.empty_like(es)
    to = torch.empty((S, Hq, parts, hd), dtype=tdt, device=dev)
    tkm = torch.empty_like(km)

    metrics = torch.zeros((NB, bs), dtype=torch.float32, device=dev)

    def step_fused():
        ops.paged_attention_kvc_fused_metrics(out, metrics, q, kc, vc, Hkv, hd ** -0.5, bt, ctx, pos, last,
                                              buf, bs, ctx_len, None, kv_dtype, k_scale, 1.0, True,
                                              temp_metrics=tkm)

    def step():
        if fused:
            return step_fused()
        ops.paged_attention_kvc_v2(out, km, es, ml, to, tkm, q, kc, vc, Hkv, hd ** -0.5, bt, ctx, pos,
                                   last, buf, bs, ctx_len, None, kv_dtype, k_scale, 1.0, record)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tokens = S * Hkv * ctx_len
    # fused: position read + metrics read-modify-write instead of the qpk-wide store
    alg = tokens * (2 * hd * kc.element_size() + ((4 + 8) if fused else (4 * record + 4 * qpk * record)))
    from vllm_kvcompress_amd import _lib
    return {"block_layout": _lib.block_layout(), "num_seqs": S, "context_len": ctx_len, "num_heads": Hq, "num_kv_heads": Hkv,
            "head_size": hd, "block_size": bs, "dtype": dtype, "kv_cache_dtype": kv_dtype, "k_scale": k_scale, "record_kv_metrics": record, "fused_metric_aggregation": fused,
            "ms_per_layer_step": ms, "algorithmic_bytes": alg, "GBps": alg / ms / 1e6,
            "frac_of_8TBps": alg / ms / 1e6 / 8000.0, "cached_tokens_per_s": tokens / ms * 1e3}
