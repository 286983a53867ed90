#!/usr/bin/env python3
"""Fused prefill metric collector (F4) at BASELINE configs[4] scale: full-query-range metric
collection (every prompt query observed, prefill_metric_collection_block_size = 1024) for one
sequence, Llama-3-8B heads.  Reports time and matrix-core TFLOP/s (2 passes x causal QK^T)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(K, Hq=32, Hk=8, hd=128, blk=1024, iters=3, unfused=False):
    import torch
    from vllm_kvcompress_amd.kvcompress.prefill import fused_kvc_attention, naive_kvc_attention
    dev = "cuda:0"
    torch.manual_seed(0)
    q = (torch.randn(K, Hq, hd, device=dev) * 0.7).half()
    k = (torch.randn(K, Hk, hd, device=dev) * 0.7).half()
    buf = torch.zeros(1, dtype=torch.int32)
    if unfused:
        k = k.repeat_interleave(Hq // Hk, dim=1)
    fn = naive_kvc_attention if unfused else fused_kvc_attention
    fn(q, k, None, [K], hd ** -0.5, buf, n_observed=K, max_observed_block_size=blk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn(q, k, None, [K], hd ** -0.5, buf, n_observed=K, max_observed_block_size=blk)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    pairs = K * (K + 1) / 2                      # causal (query, key) pairs
    flops = 2 * 2.0 * pairs * hd * Hq            # two passes of QK^T
    return {"keys": K, "num_q_heads": Hq, "num_k_heads": Hk, "head_size": hd, "q_block": blk,
            "path": "unfused (library GEMM + softmax + HIP epilogue)" if unfused else "fused",
            "ms": ms, "mfma_TFLOPs": flops / ms / 1e9, "frac_of_2.5PF": flops / ms / 1e9 / 2500.0,
            "probability_bytes_avoided": 4.0 * Hq * blk * K}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    res = []
    for K, unf in [(16384, False), (16384, True), (65536, False)]:
        r = run(K, unfused=unf, iters=2)
        res.append(r)
        print(json.dumps(r))
    if args.json:
        json.dump(res, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
