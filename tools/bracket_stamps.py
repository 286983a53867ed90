#!/usr/bin/env python3
"""Where the per-sequence kernels of the bracket schedule spend their time: an experiment build
(-DKVC_BR_STAMPS, loaded through KVC_MI355X_LIB) leaves the 100 MHz wall clock of workgroup 0's
phases in the workspace.  Config 2 shape.
  KVC_OUT=/tmp/libkvc_stamps.so KVC_EXTRA_FLAGS=-DKVC_BR_STAMPS bash vllm_kvcompress_amd/csrc/build.sh
  KVC_MI355X_LIB=/tmp/libkvc_stamps.so python tools/bracket_stamps.py"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from vllm_kvcompress_amd.harness import device as hdev, synth    # noqa: E402


def main():
    L, H, T, bs = 32, 8, 32768, 16
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=[T + 1], seed=1, protected=32)
    evicted = [synth.evict_block_count(context_lens_lh=st.context_lens[:, 0, :], seq_len=T + 1, block_size=bs,
                                       protected_window_size=32, max_cache_tokens=T // 2)]
    ds = hdev.upload(st, "cuda:0", mode="per_sequence")
    ds.cm.schedule_path = 4
    for _ in range(5):
        ds.cm.schedule_evictions(list(st.seq_indices), ds.seq_positions, evicted, ds.context_lens,
                                 ds.hanging_token_count, ds.evicted_kv_offsets, list(st.protected),
                                 total_slots=st.total_slots)
    torch.cuda.synchronize()
    print(ds.cm.last_schedule_path())
    ws, off, _ = ds.cm.last_schedule
    # head_fc sits behind st_seqrec: find it through the layout the library reports
    import ctypes
    lib = ctypes.CDLL(os.environ["KVC_MI355X_LIB"])
    lib.kvc_br_stamps_offset.restype = ctypes.c_size_t
    lib.kvc_br_stamps_offset.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    fc = int(lib.kvc_br_stamps_offset(st.total_slots, L * H, 1, bs))
    w = ws[fc:fc + 4 * 32].view(torch.int32).cpu().numpy().astype("int64") & 0xFFFFFFFF
    names = {0: "bracket: start", 1: "heads, counters", 2: "sample loaded", 3: "order statistics",
             8: "select: start", 9: "head geometry, prefix", 10: "thresholds in LDS", 11: "T*", 12: "head counts",
             13: "written", 16: "records: start", 17: "list in LDS", 18: "sorted", 19: "written"}
    names.update({4: "select + emit: start", 5: "keys staged in LDS", 6: "M known", 7: "emitted", 20: "padded"})
    names.update({21: "count + collect: start", 22: "first head found", 23: "tiles done", 14: "last batch reserved", 15: "stored"})
    for grp in ((0, 1, 2, 3), (21, 22, 23, 14, 15), (16, 17, 18, 19), (8, 9, 10, 11, 12, 13), (4, 5, 6, 7, 20)):
        for a, b in zip(grp, grp[1:]):
            print(f"  {names[b]:28s} {(int(w[b]) - int(w[a])) * 0.01:7.2f} us")
        print()
    prev = int(w[2])
    for k in range(24, 32):
        if w[k] and int(w[k]) >= int(w[2]):
            print(f"  order statistics, stamp {k}: +{(int(w[k]) - prev) * 0.01:6.2f} us")
            prev = int(w[k])


if __name__ == "__main__":
    main()
