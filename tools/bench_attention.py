#!/usr/bin/env python3
"""Decode attention with KV-metric output (F3): HBM GB/s on synthetic per-head paged caches.

Algorithmic bytes per cached token and KV head: K row + V row (2 * hd * e) + position (4)
+ metric write (4 * qpk).  One "layer step" = one call of paged_attention_kvc_v2 for all
sequences.  Run on the GPU box:  python tools/bench_attention.py [--json out.json]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(S, ctx_len, Hq=32, Hkv=8, hd=128, bs=16, iters=20, record=True, dtype="f16"):
    import torch
    from vllm_kvcompress_amd import _custom_ops as ops
    dev = "cuda:0"
    tdt = torch.float16 if dtype == "f16" else torch.bfloat16
    nblk = (ctx_len + bs - 1) // bs
    NB = S * Hkv * nblk
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    kv = torch.randint(-20000, 20000, (2, NB, bs * hd), dtype=torch.int16, device=dev, generator=gen)
    kc = kv[0].view(tdt).view(NB, hd // 8, bs, 8)
    vc = kv[1].view(tdt).view(NB, hd, bs)
    kc.mul_(1e-3)
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

    def step():
        ops.paged_attention_kvc_v2(out, km, es, ml, to, tkm, q, kc, vc, Hkv, hd ** -0.5, bt, ctx, pos,
                                   last, buf, bs, ctx_len, None, "auto", 1.0, 1.0, record)
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
    alg = tokens * (2 * hd * kc.element_size() + 4 * record + 4 * qpk * record)
    return {"num_seqs": S, "context_len": ctx_len, "num_heads": Hq, "num_kv_heads": Hkv,
            "head_size": hd, "block_size": bs, "dtype": dtype, "record_kv_metrics": record,
            "ms_per_layer_step": ms, "algorithmic_bytes": alg, "GBps": alg / ms / 1e6,
            "frac_of_8TBps": alg / ms / 1e6 / 8000.0, "cached_tokens_per_s": tokens / ms * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default=None, help="S,ctx: run a single configuration")
    args = ap.parse_args()
    res = []
    cfgs = [(256, 4097), (64, 4097), (16, 32769), (1, 32769), (256, 512)]
    if args.only:
        cfgs = [tuple(int(v) for v in args.only.split(","))]
    for (S, ctx) in cfgs:
        for record in (True, False):
            r = run(S, ctx, record=record)
            res.append(r)
            print(json.dumps(r))
    if args.json:
        json.dump(res, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
