"""Child process of tests/test_gpu_schedule_paths.py::test_next_pivots_are_the_same_wherever_they_are_made: a few
decode steps of an evolving store (aggregate -> schedule on remembered pivots / harvested lists), printing the pivots
every schedule call left in the harvest buffer and a digest of its outputs.  The parent runs it with the pivots made by
topk_fused_kernel's last phase (default), by harvest_pivot_kernel behind it (KVC_TOPK_PIVOT_LAUNCH=1) and on the launch
chain (KVC_TOPK_CHAIN=1): the three must agree word for word."""
import hashlib
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from vllm_kvcompress_amd.harness import device as hdev, synth         # noqa: E402

DEV = "cuda:0"


def main():
    bs, cap = 16, 320
    res = {}
    for name, (L, H, lens) in {"lh8": (2, 4, [cap + 200, cap + 90, cap + 150]), "lh128": (8, 16, [cap + 60, cap + 33])}.items():
        st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=lens, seed=21, protected=bs + 1, steady_cap=cap)
        ds = hdev.upload(st, DEV, num_queries_per_kv=4, mode="per_sequence")
        cm = ds.cm
        cm.strict_fallback = True
        rng = np.random.default_rng(3)
        B = len(lens)
        evicted = [max(1, (L * H) // 4)] * B
        seqs, prot = list(st.seq_indices), list(st.protected)
        pivots, digests, how = [], [], []
        for step in range(5):
            temp = rng.random((st.num_blocks, bs, 4)).astype(np.float32)
            cm.temp_metrics.copy_(torch.from_numpy(temp))
            cm.aggregate_decode()
            eli, ekc, ebc = cm.schedule_evictions(seqs, ds.seq_positions, evicted, ds.context_lens, ds.hanging_token_count,
                                                  ds.evicted_kv_offsets, prot, total_slots=st.total_slots)
            torch.cuda.synchronize()
            how.append(cm.last_schedule_reason)
            h = hashlib.sha256()
            for t in (eli, ekc, ebc):
                h.update(t.cpu().numpy().tobytes())
            digests.append(h.hexdigest())
            pivots.append(cm._hv_buf[:4 * B].view(torch.int32).cpu().numpy().astype(np.int64).tolist()
                          if cm._hv_buf is not None else None)
        res[name] = {"pivots": pivots, "digests": digests, "how": how}
    print("PIVOT_PHASE " + json.dumps(res))


if __name__ == "__main__":
    main()
