"""Fixture loading / synthetic states for the decode-attention tests (F3)."""
from __future__ import annotations

import numpy as np

from oracle import kvc_oracle as orc


def bf16_bits_to_f32(bits):
    return (bits.astype(np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def f32_to_bf16_bits(x):
    return (orc.round_to_bf16(x).view(np.uint32) >> np.uint32(16)).astype(np.uint16).view(np.int16)


def decode_golden(g):
    """-> dict of numpy inputs (query/caches as float arrays the oracle accepts) + raw bits"""
    dt = str(g["dtype"])
    if dt == "f16":
        conv = lambda b: b.view(np.float16)
    else:
        conv = bf16_bits_to_f32
    return dict(dtype=dt, q=conv(g["query_bits"]), kc=conv(g["key_cache_bits"]),
                vc=conv(g["value_cache_bits"]), p_dtype=np.float16 if dt == "f16" else "bf16",
                slopes=g["alibi_slopes"] if g["alibi_slopes"].size else None)


def oracle_decode(c, g, kv_position, last_position, buffer_len, record=True, fill=-1.0):
    q, kc, vc = c["q"], c["kc"], c["vc"]
    S, Hq, hd = q.shape
    Hkv = int(g["num_kv_heads"])
    NB, _, bs = vc.shape
    out = np.zeros((S, Hq, hd), np.float32)
    km = np.full((NB, bs, Hq // Hkv), fill, np.float32)
    orc.paged_attention_decode(out, km, q, kc, vc, Hkv, float(g["scale"]), g["block_tables"],
                               g["context_lens"], kv_position, last_position, buffer_len,
                               alibi_slopes=c["slopes"], record_kv_metrics=record,
                               p_dtype=c["p_dtype"])
    return out, km


def make_state(rng, num_seqs, num_q_heads, num_kv_heads, hd, bs, ctx_lo, ctx_hi, dtype="f16",
               magnitude=1.0, alibi=False):
    """Random per-head paged cache in the reference test's recipe
    (tests/kernels/test_kvcompress_attention.py:214-262) with N(0, magnitude) data."""
    x = 8
    ctx = rng.integers(ctx_lo, ctx_hi + 1, size=(num_seqs, num_kv_heads)).astype(np.int32)
    ctx[-1, -1] = ctx_hi
    nblk = (ctx + bs - 1) // bs
    NB = int(nblk.sum()) + 5
    perm = rng.permutation(NB)
    M = int(nblk.max()) + 2
    bt = np.zeros((num_seqs, num_kv_heads, M), np.int32)
    cur = 0
    for s in range(num_seqs):
        for h in range(num_kv_heads):
            n = int(nblk[s, h])
            bt[s, h, :n] = perm[cur:cur + n]
            cur += n
    f = lambda shape: (rng.standard_normal(shape) * magnitude).astype(np.float32)
    q, kc, vc = f((num_seqs, num_q_heads, hd)), f((NB, hd // x, bs, x)), f((NB, hd, bs))
    if dtype == "f16":
        qb, kb, vb = (a.astype(np.float16) for a in (q, kc, vc))
        bits = lambda a: a.view(np.int16)
        vals = lambda a: a
        pd = np.float16
    else:
        qb, kb, vb = (f32_to_bf16_bits(a) for a in (q, kc, vc))
        bits = lambda a: a
        vals = bf16_bits_to_f32
        pd = "bf16"
    # positions: a sorted random subset per head, like a compressed cache
    pos = np.zeros((NB, bs), np.int32)
    last = np.zeros(num_seqs, np.int32)
    for s in range(num_seqs):
        top = int(ctx[s].max()) * 2 + 8
        last[s] = top
        for h in range(num_kv_heads):
            n = int(ctx[s, h])
            p = np.sort(rng.choice(top, size=n, replace=False)).astype(np.int32)
            blocks = bt[s, h, :int(nblk[s, h])]
            flat = pos[blocks].reshape(-1)
            flat[:n] = p
            flat[n:] = top + np.arange(flat.size - n)
            pos[blocks] = flat.reshape(-1, bs)
    # ALiBi slopes are 2^-k in (0, 1]; the bias grows with the context, and so does the fp32
    # rounding of logit + bias (1 ulp of a bias of -1000 is 6e-5 relative on the weight)
    slopes = rng.uniform(0.002, 0.5, num_q_heads).astype(np.float32) if alibi else None
    g = dict(num_kv_heads=np.int32(num_kv_heads), scale=np.float32(1.0 / hd ** 0.5),
             block_tables=bt, context_lens=ctx, query_bits=bits(qb), key_cache_bits=bits(kb),
             value_cache_bits=bits(vb))
    c = dict(dtype=dtype, q=vals(qb), kc=vals(kb), vc=vals(vb), p_dtype=pd, slopes=slopes)
    return g, c, pos, last
