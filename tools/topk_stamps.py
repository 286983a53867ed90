#!/usr/bin/env python3
"""where topk_fused_kernel spends its time (experiment build with -DKVC_TOPK_STAMPS): wall_clock64 stamps of one wave of
workgroup 1 at the phase boundaries, read back from the schedule's workspace.
    KVC_OUT=/tmp/libkvc_stamps.so KVC_EXTRA_FLAGS=-DKVC_TOPK_STAMPS bash vllm_kvcompress_amd/csrc/build.sh
    KVC_MI355X_LIB=/tmp/libkvc_stamps.so python tools/topk_stamps.py [B]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

from vllm_kvcompress_amd.harness import device as hdev, synth


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    L, H, bs, cap = 32, 8, 16, 4096
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=[3 * cap] * B, seed=1, protected=32,
                          steady_cap=cap, spare_block_frac=0.02)
    evicted = [synth.evict_block_count(context_lens_lh=st.context_lens[:, b, :], seq_len=3 * cap, block_size=bs,
                                       protected_window_size=32, max_cache_tokens=cap) for b in range(B)]
    ds = hdev.upload(st, "cuda:0", num_queries_per_kv=4, mode="per_sequence")
    cm = ds.cm
    args = (list(st.seq_indices), ds.seq_positions, evicted, ds.context_lens, ds.hanging_token_count, ds.evicted_kv_offsets,
            list(st.protected))
    for it in range(4):
        out = cm.schedule_evictions(*args, total_slots=st.total_slots)
        torch.cuda.synchronize()
        ws, off, plan = cm.last_schedule
        stamps = ws[off + 128:off + 128 + 96].view(torch.int64).cpu().numpy()
        d = np.diff(stamps[:8]) / 100.0          # 100 MHz -> us
        print(cm.last_schedule_reason, "us per phase (loads | ranks | barrier | select | counts | emit | next pivot):", np.round(d, 2), "total", round(float((stamps[7] - stamps[0]) / 100.0), 2))
        del out


if __name__ == "__main__":
    main()
