#!/usr/bin/env python3
"""How fast can MI355X gather/scatter randomly placed chunks of a given size?  (sets the
practical ceiling for block-granular compaction; profiling aid)"""
import json, sys, torch

def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

total = 4 << 30
res = {}
for chunk in (16, 64, 256, 1024, 4096, 65536):
    n = total // chunk
    w = chunk // 4
    src = torch.empty((n, w), dtype=torch.int32, device="cuda")
    src.random_()
    dst = torch.empty_like(src)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    perm = torch.randperm(n, device="cuda", generator=g)
    half = n // 2
    idx = perm[:half]
    sidx = torch.sort(idx).values
    out = dst[:half]
    ms_g = timeit(lambda: torch.index_select(src, 0, idx, out=out))
    ms_gs = timeit(lambda: torch.index_select(src, 0, sidx, out=out))
    ms_s = timeit(lambda: dst.index_copy_(0, idx, src[:half]))
    bytes_rw = 2 * half * chunk
    res[chunk] = {"gather_random_GBps": bytes_rw / ms_g / 1e6, "gather_sorted_GBps": bytes_rw / ms_gs / 1e6,
                  "scatter_random_GBps": bytes_rw / ms_s / 1e6}
    del src, dst
print(json.dumps(res, indent=1))
