#!/usr/bin/env python3
"""Soak and timing of schedule_evictions' bracket schedule (schedule_path 4): random bulk-eviction
states (1 .. 4 sequences in both modes, bs 8 / 16 / 32, heads of 0.5 k .. 16 k slots, first and
second compressions, ties, a skewed head, eviction fractions 2 .. 98 % and over-asks) through the
bracket schedule and the digit rounds (schedule_path 1, itself pinned to the oracle by the suite);
the small states also against the oracle.  Prints how often the bracket finished on its own, then
S1 at config 2 / config 5 size under both.
Run on the GPU box:  python tools/soak_bracket.py [nseeds] [--time]"""
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


def run(ds, st, evicted, path):
    ds.cm.schedule_path = path
    return ds.cm.schedule_evictions(list(st.seq_indices), ds.seq_positions, evicted, ds.context_lens,
                                    ds.hanging_token_count, ds.evicted_kv_offsets, list(st.protected),
                                    total_slots=st.total_slots)


def soak(n):
    t0 = time.time()
    how = {}
    for seed in range(n):
        rng = np.random.default_rng(70000 + seed)
        L, H = int(rng.integers(1, 5)), int(rng.integers(1, 9))
        bs = int(rng.choice([8, 16, 32]))
        B = int(rng.integers(1, 5))
        T = int(rng.choice([512, 1024, 4096, 8192, 16384]))
        ties = int(rng.integers(1, 40)) if rng.random() < 0.25 else None
        compressed = bool(rng.random() < 0.4)
        mode = "reference" if seed % 2 == 0 else "per_sequence"      # (B > 1 in reference mode: the batch > 1 rule)
        st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs,
                              seq_lens=[T + int(rng.integers(0, 3 * bs)) for _ in range(B)], seed=seed,
                              protected=[int(rng.integers(1, 3 * bs)) for _ in range(B)], compressed=compressed,
                              tie_levels=ties, metric_shape=str(rng.choice(["perm", "decay", "oldest"])) if ties is None else "perm")
        if rng.random() < 0.15:                      # a head with far lower metrics absorbs the eviction
            blk = np.nonzero((st.layer_index_by_block == 0) & (st.head_index_by_block == 0) & (st.seq_index_by_block == 0))[0]
            st.metrics[blk] -= np.float32(1e6)
        nblk = ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)
        frac = float(rng.choice([0.02, 0.1, 0.3, 0.5, 0.7, 0.9, 0.98, 1.2]))
        evicted = [int(nb * frac * rng.uniform(0.8, 1.0)) if rng.random() > 0.1 else 0 for nb in nblk]
        ds = hdev.upload(st, DEV, mode=mode)
        want = [t.clone() for t in run(ds, st, evicted, 1)]
        got = run(ds, st, evicted, 4)
        key = ds.cm.last_schedule_path()
        how[key] = how.get(key, 0) + 1
        ref = None
        if st.total_slots <= 300000 and frac <= 1.0:
            ref = orc.schedule_evictions(
                metrics=st.metrics, token_positions=st.token_positions, seq_index_by_block=st.seq_index_by_block,
                layer_index_by_block=st.layer_index_by_block, head_index_by_block=st.head_index_by_block,
                logical_block_num_by_block=st.logical_block_num_by_block, block_size=bs, num_layers=L, num_kv_heads=H,
                seq_indices=st.seq_indices, seq_positions=st.seq_positions, evicted_blocks_per_seq=evicted,
                context_lens=st.context_lens, hanging_token_count=st.hanging_token_count,
                evicted_kv_offsets=st.evicted_kv_offsets, num_protected=st.protected, mode=mode)
        for i, name in enumerate(("eli", "ekc", "ebc")):
            g = got[i].cpu().numpy()
            bad = not np.array_equal(g, want[i].cpu().numpy()) or (ref is not None and not np.array_equal(g, ref[i]))
            if bad:
                print(f"MISMATCH seed={seed} key={name} mode={mode} L={L} H={H} bs={bs} B={B} T={T} ties={ties} "
                      f"compressed={compressed} frac={frac} evicted={evicted} how={key}")
                sys.exit(1)
        # the fork's call form with N on the device (ABI version 8; sequences that do not couple): the same answer whether
        # the bound holds (the call before learnt it), is far too generous, or does not hold (voided on the device, repeated)
        if mode == "per_sequence" or B == 1:
            cm = ds.cm
            cm.schedule_path = int(rng.choice([0, 1, 4]))
            cm._dn_plan[B] = 2
            cm._dn_bound[B] = int(rng.choice([st.total_slots, st.total_slots + bs * int(rng.integers(1, 5000)),
                                              4 * st.total_slots, max(bs, (st.total_slots // 2) // bs * bs)]))
            voided0 = cm.deferred_voided
            got2 = cm.schedule_evictions(list(st.seq_indices), ds.seq_positions.clone(),
                                         torch.tensor(evicted, dtype=torch.int, device=DEV), ds.context_lens,
                                         ds.hanging_token_count, ds.evicted_kv_offsets, tuple(st.protected))
            key2 = "N on the device: " + ("voided, repeated" if cm.deferred_voided > voided0 else cm.last_schedule_path())
            how[key2] = how.get(key2, 0) + 1
            for i, name in enumerate(("eli", "ekc", "ebc")):
                if not np.array_equal(got2[i].cpu().numpy(), want[i].cpu().numpy()):
                    print(f"MISMATCH (N on the device) seed={seed} key={name} mode={mode} L={L} H={H} bs={bs} B={B} T={T} "
                          f"bound={cm._dn_bound[B]} N={st.total_slots} evicted={evicted} how={key2}")
                    sys.exit(1)
    print(f"soak ok: {n} states, bracket schedule identical to the digit rounds (and the oracle where small) "
          f"in {time.time() - t0:.1f} s; {how}")


def timing():
    for name, L, H, T, bs, B, keep in (("c2", 32, 8, 32768, 16, 1, 0.5), ("c5", 32, 8, 65536, 32, 1, 0.5),
                                      ("c2 keep 0.9", 32, 8, 32768, 16, 1, 0.9), ("8 x 32k", 32, 8, 32768, 16, 8, 0.5),
                                      ("1 x 1k", 32, 8, 1024, 16, 1, 0.5), ("1 x 2k", 32, 8, 2048, 16, 1, 0.5),
                                      ("1 x 4k", 32, 8, 4096, 16, 1, 0.5), ("1 x 8k", 32, 8, 8192, 16, 1, 0.5),
                                      ("16 x 2k", 32, 8, 2048, 16, 16, 0.5), ("16 x 4k", 32, 8, 4096, 16, 16, 0.5),
                                      ("64 x 1k", 32, 8, 1024, 16, 64, 0.5),
                                      ("c4 shape, 4 x 16k", 80, 8, 16384, 16, 4, 0.5),
                                      ("16 x 4k, the reference's batch > 1 rule", 32, 8, 4096, 16, 16, 0.5),
                                      ("c4 shape, 4 x 16k, batch > 1 rule", 80, 8, 16384, 16, 4, 0.5)):
        mode = "reference" if "rule" in name else "per_sequence"
        st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=[T + 1] * B, seed=1, protected=32)
        evicted = [synth.evict_block_count(context_lens_lh=st.context_lens[:, b, :], seq_len=T + 1, block_size=bs,
                                           protected_window_size=32, max_cache_tokens=int(T * keep)) for b in range(B)]
        ds = hdev.upload(st, DEV, mode=mode)
        res = {}
        for path in (1, 4):
            outs = run(ds, st, evicted, path)
            how = ds.cm.last_schedule_path()
            for _ in range(5):
                run(ds, st, evicted, path)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run(ds, st, evicted, path)
            e1.record()
            torch.cuda.synchronize()
            res[path] = (e0.elapsed_time(e1) / 20, how, [t.clone() for t in outs])
        same = all(torch.equal(a, b) for a, b in zip(res[1][2], res[4][2]))
        print(f"{name}: digit rounds {res[1][0]*1e3:.1f} us, bracket {res[4][0]*1e3:.1f} us ({res[4][1]}), identical={same}")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    soak(int(args[0]) if args else 200)
    if "--time" in sys.argv:
        timing()
