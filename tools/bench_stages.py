#!/usr/bin/env python3
"""Stage S0 (metric aggregation) and A7 rates on MI355X at the BASELINE shapes
(SURVEY.md section 8(d)): aggregate_decode (+fused clear), aggregate_prefill, the prefill
metric epilogue, reshape_and_cache.  Prints one JSON object."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from vllm_kvcompress_amd import _custom_ops as ops
from vllm_kvcompress_amd.kvcompress.metrics import CompressionMetrics
from vllm_kvcompress_amd.kvcompress.prefill import accumulate_prefill_tile

DEV = "cuda:0"


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    res = {}
    L, H, qpk, bs, hd, T = 32, 8, 4, 16, 128, 32768
    NB = L * H * T // bs
    cm = CompressionMetrics(bs, L, H, qpk, 10 ** 9, None, 0.0, device=DEV)
    cm.init_kv_metadata(NB)
    cm.metrics.zero_()
    cm._temp_metrics.uniform_()
    slots = NB * bs
    ms = timeit(lambda: cm.aggregate_decode(fuse_clear=True))
    res["aggregate_decode_fused_clear"] = {"slots": slots, "ms": ms, "slots_per_s": slots / ms * 1e3,
                                           "GBps": slots * (4 * qpk * 2 + 8) / ms / 1e6,
                                           "bytes_per_slot": 4 * qpk * 2 + 8}
    ms = timeit(lambda: cm.aggregate_decode(fuse_clear=False))
    res["aggregate_decode"] = {"slots": slots, "ms": ms, "slots_per_s": slots / ms * 1e3,
                               "GBps": slots * (4 * qpk + 8) / ms / 1e6, "bytes_per_slot": 4 * qpk + 8}
    def ref_two_pass():
        cm.metrics.add_((cm._temp_metrics ** 2).sum(dim=-1))
        cm._temp_metrics.zero_()
    ms = timeit(ref_two_pass, iters=5)
    res["torch_reference_formulation_decode_plus_clear"] = {"ms": ms}
    # aggregate_prefill: T tokens of one layer
    pm = torch.rand((T, H * qpk), device=DEV)
    sm = torch.randperm(NB * bs, device=DEV)[:T * H].reshape(T, H)
    ms = timeit(lambda: cm.aggregate_prefill(pm, sm))
    res["aggregate_prefill"] = {"tokens": T, "ms": ms, "GBps": T * H * (4 * qpk + 16) / ms / 1e6}
    del cm, pm, sm
    torch.cuda.empty_cache()
    # epilogue: C5 tile  Hq=32, qb=1024, K=65536  (8 GiB of probabilities)
    for (Hq, qb, K) in ((32, 1024, 65536), (32, 1024, 32768)):
        probs = torch.rand((Hq, qb, K), device=DEV)
        out = torch.zeros((K, Hq), device=DEV)
        ms = timeit(lambda: accumulate_prefill_tile(out, probs, K - qb, 0, True, False, True), iters=5)
        # causal tile at the end of the sequence: all keys visible to (almost) all rows
        res[f"prefill_epilogue_Hq{Hq}_qb{qb}_K{K}"] = {
            "ms": ms, "GBps": (Hq * qb * K * 4 + Hq * K * 12) / ms / 1e6, "tile_bytes": Hq * qb * K * 4}
        del probs, out
        torch.cuda.empty_cache()
    # reshape_and_cache: one prefill of T tokens into one layer's heads
    key = torch.randn((T, H, hd), device=DEV, dtype=torch.float16)
    val = torch.randn((T, H, hd), device=DEV, dtype=torch.float16)
    nb = H * T // bs
    kv = torch.zeros((2, nb, bs * hd), device=DEV, dtype=torch.float16)
    kc = kv[0].view(nb, hd // 8, bs, 8)
    vc = kv[1].view(nb, hd, bs)
    met = torch.zeros((nb, bs), device=DEV)
    # head h owns blocks [h*T/bs, (h+1)*T/bs): token t -> slot (h*T + t)
    slot_map = (torch.arange(H, device=DEV)[None, :] * T + torch.arange(T, device=DEV)[:, None]).reshape(-1)
    bias = torch.zeros(H, device=DEV)
    ms = timeit(lambda: ops.reshape_and_cache_kvc(key, val, kc, vc, met, slot_map, bias, "auto", 1.0, 1.0))
    res["reshape_and_cache_kvc"] = {"tokens": T, "ms": ms, "GBps": T * H * hd * 2 * 2 * 2 / ms / 1e6}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
