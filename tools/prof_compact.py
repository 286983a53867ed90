#!/usr/bin/env python3
"""Phase breakdown of execute_cache_moves on the bench workload (profiling aid)."""
import argparse, ctypes, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from vllm_kvcompress_amd import _custom_ops as ops, _lib


def timeit(fn, iters=10):
    for _ in range(2):
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
    sys.argv = [sys.argv[0]] + sys.argv[1:]
    args = bench.parse_args()
    lib = _lib.load()
    lib.kvc_debug_set_compact_phases.argtypes = [ctypes.c_int]
    st, ds, evicted, k, v = bench.build_workload(args, 0, "cuda:0")
    N = st.total_slots
    eli, ekc, ebc = ds.cm.schedule_evictions(list(st.seq_indices), ds.seq_positions, evicted,
                                             ds.context_lens, ds.hanging_token_count,
                                             ds.evicted_kv_offsets, list(st.protected), total_slots=N)
    cmi = torch.empty((N, 2), dtype=torch.int32, device="cuda:0")
    cmc = torch.empty_like(ekc)
    ops.schedule_cache_moves(cmi, cmc, eli, ekc, ds.evicted_kv_offsets, ds.block_tables,
                             ds.context_lens, st.block_size)
    moves = int(cmc.sum())
    wm, wp = ds.cm.metrics.clone(), ds.cm.token_positions.clone()
    res = {"moves": moves}
    for name, mask in (("all", 7), ("metrics", 1), ("K", 2), ("V", 4), ("K+V", 6), ("none", 0)):
        lib.kvc_debug_set_compact_phases(mask)
        ms = timeit(lambda: ops.execute_cache_moves(k, v, wm, wp, cmi, cmc, ds.evicted_kv_offsets, 1, 16))
        res[name] = {"ms": ms, "alg_GBps": moves * 1048 / ms / 1e6}
    lib.kvc_debug_set_compact_phases(7)
    # copy bandwidth reference: 2 GiB device->device
    src = torch.empty(2 << 30, dtype=torch.uint8, device="cuda:0")
    dst = torch.empty_like(src)
    ms = timeit(lambda: dst.copy_(src))
    res["copy_2GiB"] = {"ms": ms, "GBps_rw": 2 * (2 << 30) / ms / 1e6}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
