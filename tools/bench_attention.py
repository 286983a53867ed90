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


from vllm_kvcompress_amd.harness.attention_bench import run  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default=None, help="S,ctx: run a single configuration")
    ap.add_argument("--block-layout", default=None, choices=["reference", "slot_major"],
                    help="in-block cache layout (default: KVC_BLOCK_LAYOUT or reference); the cache holds random bits, "
                         "so the same buffers serve either layout")
    args = ap.parse_args()
    if args.block_layout:
        import vllm_kvcompress_amd
        vllm_kvcompress_amd.set_block_layout(args.block_layout)
    res = []
    cfgs = [(256, 4097), (64, 4097), (16, 32769), (1, 32769), (256, 512)]
    if args.only:
        cfgs = [tuple(int(v) for v in args.only.split(","))]
    for (S, ctx) in cfgs:
        for record in (True, False):
            r = run(S, ctx, record=record)
            res.append(r)
            print(json.dumps(r))
    if not args.only:
        for S, ctx in ((256, 4097), (16, 32769)):
            r = run(S, ctx, fused=True)
            res.append(r)
            print(json.dumps(r))
        # Llama-3-70B heads (64 query / 8 KV, qpk 8) at a 4k cap, and qpk 4 at an 8k cap: the
        # 8-wave single-pass schedule
        for (S, ctx, hq) in ((128, 4097, 64), (128, 8193, 32)):
            for rec in (True, False):
                r = run(S, ctx, Hq=hq, record=rec)
                res.append(r)
                print(json.dumps(r))
        for kvd, bs, ks in (("fp8_e4m3", 16, 1.0), ("fp8_e5m2", 32, 1.0), ("fp8_e4m3", 16, 0.01)):
            r = run(256, 4097, bs=bs, kv_dtype=kvd, k_scale=ks)
            res.append(r)
            print(json.dumps(r))
    if args.json:
        json.dump(res, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
