#!/usr/bin/env python3
"""Soak: medium-sized random states (heads of 1k..20k slots: both select workgroup sizes, LDS
staged and unstaged keys, tiles with and without head boundaries, ties, both schedule modes)
through the HIP pipeline and the oracle; every output must be identical.
Run on the GPU box:  python tools/soak_medium.py [nseeds]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from tests.helpers import oracle_pipeline          # noqa: E402
from tests.test_gpu_parity import _gpu_pipeline     # noqa: E402
from vllm_kvcompress_amd.harness import synth       # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    t0 = time.time()
    for seed in range(n):
        rng = np.random.default_rng(50000 + seed)
        L, H = int(rng.integers(1, 3)), int(rng.integers(1, 4))
        bs = int(rng.choice([16, 32]))
        B = int(rng.integers(1, 4))
        hi = int(rng.choice([1500, 6000, 20000]))
        seq_lens = [int(rng.integers(hi // 3, hi)) for _ in range(B)]
        prot = [int(rng.integers(1, 64)) for _ in range(B)]
        compressed = bool(rng.random() < 0.5)
        ties = int(rng.integers(2, 50)) if rng.random() < 0.3 else None
        shape = str(rng.choice(["perm", "decay", "oldest"]))
        st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=seed,
                              protected=prot, compressed=compressed, tie_levels=ties, metric_shape=shape)
        nblk = ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)
        kind = rng.random()
        if kind < 0.3:        # steady-state like: a block or two per head
            evicted = [int(min(nb, L * H * int(rng.integers(1, 3)))) for nb in nblk]
        else:
            evicted = [int(rng.integers(0, int(nb) + 1)) for nb in nblk]
        k, v = synth.make_caches_u16(seed, st.num_blocks, 128, bs)
        mode = "reference" if seed % 2 == 0 else "per_sequence"
        want = oracle_pipeline(st, evicted, k, v, mode=mode)
        got = _gpu_pipeline(st, evicted, k, v, mode=mode)
        for key in ("eli", "ekc", "ebc", "cmi", "cmc", "k", "v", "metrics", "positions"):
            if not np.array_equal(got[key], want[key]):
                print(f"MISMATCH seed={seed} key={key} mode={mode} L={L} H={H} bs={bs} lens={seq_lens} "
                      f"ties={ties} shape={shape} evicted={evicted}")
                sys.exit(1)
    print(f"soak ok: {n} medium states identical to the oracle in {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main()
