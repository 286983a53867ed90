#!/usr/bin/env python3
"""Where the HOST spends its time inside schedule_evictions (the fork's call form, configs[1] shape): cProfile over many
calls with the device idle at entry -- what an engine that has just read back its sampled tokens pays before the first
launch.  Run on the GPU box:  python tools/profile_host_path.py [calls]"""
import cProfile
import os
import pstats
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from vllm_kvcompress_amd.harness import device as hdev, synth    # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    L, H, T, bs = 32, 8, 32768, 16
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=[T + 1], seed=1, protected=32)
    evicted = [synth.evict_block_count(context_lens_lh=st.context_lens[:, 0, :], seq_len=T + 1, block_size=bs,
                                       protected_window_size=32, max_cache_tokens=T // 2)]
    ds = hdev.upload(st, "cuda:0", mode="per_sequence")
    cm = ds.cm
    seq_idx, prot = list(st.seq_indices), tuple(st.protected)

    def call():
        k_t = torch.tensor(evicted, dtype=torch.int, device="cuda:0")
        pos_t = ds.seq_positions.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = cm.schedule_evictions(seq_idx, pos_t, k_t, ds.context_lens, ds.hanging_token_count, ds.evicted_kv_offsets, prot)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        return t1 - t0, time.perf_counter() - t0, out

    for _ in range(5):
        call()
    host, wall = [], []
    for _ in range(n):
        h, w, _ = call()
        host.append(h); wall.append(w)
    host.sort(); wall.sort()
    print(f"without the profiler: host time inside the call median {host[n // 2] * 1e6:.1f} us (min {host[0] * 1e6:.1f}), "
          f"wall to results median {wall[n // 2] * 1e6:.1f} us")
    # ... the same through the list form with total_slots= (never waits)
    hl = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cm.schedule_evictions(seq_idx, ds.seq_positions, evicted, ds.context_lens, ds.hanging_token_count,
                              ds.evicted_kv_offsets, list(prot), total_slots=st.total_slots)
        hl.append(time.perf_counter() - t0)
    hl.sort()
    print(f"list form with total_slots=: host time inside the call median {hl[n // 2] * 1e6:.1f} us (min {hl[0] * 1e6:.1f})")
    host, wall = [], []
    pr = cProfile.Profile()
    for _ in range(n):
        pr.enable()
        h, w, _ = call()
        pr.disable()
        host.append(h); wall.append(w)
    host.sort(); wall.sort()
    print(f"{cm.last_schedule_reason}: host time inside the call median {host[n // 2] * 1e6:.1f} us, "
          f"wall to results median {wall[n // 2] * 1e6:.1f} us (deferred calls {cm.deferred_calls})")
    st_ = pstats.Stats(pr).sort_stats("tottime")
    st_.print_stats(22)
    st_.print_callers("get")
    st_.print_callers("_get_device_index")


if __name__ == "__main__":
    main()
