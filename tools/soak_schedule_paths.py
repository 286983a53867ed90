#!/usr/bin/env python3
"""Soak of schedule_evictions' schedules: random continual-compression-like states (bs 8 / 16 /
32, caps 128 .. 8192, 1 .. 6 sequences, ragged survivor counts, ties, skewed heads, both modes,
optional caller block tables) through the small-eviction schedule (forced) and the general one;
both must equal the oracle; the small-eviction schedule with its positions looked up lazily (path 2,
where the mode allows) and streamed (path 3), and with a random sample stride.  Prints how often
the small-eviction schedule finished on its own.
Run on the GPU box:  python tools/soak_schedule_paths.py [nseeds]"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import kvc_oracle as orc                 # noqa: E402
from vllm_kvcompress_amd.harness import device as hdev, synth    # noqa: E402

DEV = "cuda:0"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    t0 = time.time()
    how = {}
    for seed in range(n):
        rng = np.random.default_rng(90000 + seed)
        L, H = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        bs = int(rng.choice([8, 16, 32]))
        B = int(rng.integers(1, 7))
        cap = int(rng.choice([128, 512, 1024, 2048, 4096, 8192])) // bs * bs
        ties = int(rng.integers(1, 40)) if rng.random() < 0.25 else None
        compressed = bool(rng.random() < 0.4)
        if compressed:       # ragged survivor counts per head
            st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=[3 * cap] * B, seed=seed,
                                  protected=[int(rng.integers(1, 2 * bs)) for _ in range(B)], compressed=True,
                                  tie_levels=ties)
        else:
            st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=[3 * cap] * B, seed=seed,
                                  protected=bs + 1, steady_cap=cap, tie_levels=ties)
        if rng.random() < 0.15:                      # a head with far lower metrics absorbs the eviction
            blk = np.nonzero((st.layer_index_by_block == 0) & (st.head_index_by_block == 0) & (st.seq_index_by_block == 0))[0]
            st.metrics[blk] -= np.float32(1e6)
        nblk = ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)
        evicted = [int(min(nb, int(rng.integers(0, 3 * L * H + 1)))) for nb in nblk]
        mode = "reference" if seed % 2 == 0 else "per_sequence"
        eli, ekc, ebc = orc.schedule_evictions(
            metrics=st.metrics, token_positions=st.token_positions, seq_index_by_block=st.seq_index_by_block,
            layer_index_by_block=st.layer_index_by_block, head_index_by_block=st.head_index_by_block,
            logical_block_num_by_block=st.logical_block_num_by_block, block_size=bs, num_layers=L, num_kv_heads=H,
            seq_indices=st.seq_indices, seq_positions=st.seq_positions, evicted_blocks_per_seq=evicted,
            context_lens=st.context_lens, hanging_token_count=st.hanging_token_count,
            evicted_kv_offsets=st.evicted_kv_offsets, num_protected=st.protected, mode=mode)
        ds = hdev.upload(st, DEV, mode=mode)
        args = (list(st.seq_indices), ds.seq_positions, evicted, ds.context_lens, ds.hanging_token_count,
                ds.evicted_kv_offsets, list(st.protected))
        for path in (2, 3, 1, 4):
            ds.cm.schedule_path = path
            ds.cm.sample_stride = int(rng.choice([0, 0, 1, 2, 4, 8, 16, 32, 64]))
            bt = ds.block_tables if (path == 2 and seed % 3 == 0) else None
            got = ds.cm.schedule_evictions(*args, total_slots=st.total_slots, block_tables=bt)
            if path == 4:
                key = f"path4: {ds.cm.last_schedule_path()}"
                how[key] = how.get(key, 0) + 1
            elif path != 1:
                key = f"path{path} stride={'auto' if ds.cm.sample_stride == 0 else 'forced'}: {ds.cm.last_schedule_path()}"
                how[key] = how.get(key, 0) + 1
            for name, g, w in zip(("eli", "ekc", "ebc"), got, (eli, ekc, ebc)):
                if not np.array_equal(g.cpu().numpy(), w):
                    print(f"MISMATCH seed={seed} path={path} key={name} mode={mode} L={L} H={H} bs={bs} B={B} cap={cap} "
                          f"ties={ties} compressed={compressed} evicted={evicted} how={ds.cm.last_schedule_path()}")
                    sys.exit(1)
    print(f"soak ok: {n} states x 4 schedule paths identical to the oracle in {time.time() - t0:.1f} s; forced small-eviction: {how}")


if __name__ == "__main__":
    main()
