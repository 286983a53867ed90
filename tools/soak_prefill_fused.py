#!/usr/bin/env python3
"""Soak: random shapes of the fused prefill metric collector against the oracle's NumPy
restatement of _naive_kvc_attention.  Integer-valued q / k make every logit exactly
representable, so every implementation sees identical logits and the fp32 pipeline has to
agree to 2e-5.  Run on the GPU box:  python tools/soak_prefill_fused.py [ncases]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch                                                          # noqa: E402
from oracle import kvc_oracle as orc                                  # noqa: E402
from vllm_kvcompress_amd.kvcompress.prefill import fused_kvc_attention  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    t0 = time.time()
    worst = 0.0
    for seed in range(n):
        rng = np.random.default_rng(9000 + seed)
        nseq = int(rng.integers(1, 4))
        lens = [int(rng.integers(1, int(rng.choice([40, 300, 1200])))) for _ in range(nseq)]
        n_obs = int(rng.choice([1, 7, 64, 500, 5000]))
        blk = int(rng.choice([1, 32, 100, 256, 4096]))
        if n_obs // blk > 64:
            blk = 64
        Hk = int(rng.integers(1, 3))
        qpk = int(rng.choice([1, 2, 4]))
        hd = int(rng.choice([64, 128]))
        buf = [int(rng.integers(0, 12)) for _ in range(nseq)]
        l2, avg, pool = (bool(rng.integers(0, 2)) for _ in range(3))
        T = sum(lens)
        q = rng.integers(-2, 3, size=(T, Hk * qpk, hd)).astype(np.float16)
        k = rng.integers(-2, 3, size=(T, Hk, hd)).astype(np.float16)
        want = orc.naive_kvc_attention(q.astype(np.float32), np.repeat(k, qpk, axis=1).astype(np.float32), lens,
                                       hd ** -0.5, buf, n_observed=n_obs, max_observed_block_size=blk,
                                       use_l2=l2, use_average=avg, use_maxpool=pool)
        _, got = fused_kvc_attention(torch.from_numpy(q).cuda(), torch.from_numpy(k).cuda(), None, lens,
                                     hd ** -0.5, torch.tensor(buf, dtype=torch.int32), n_observed=n_obs,
                                     max_observed_block_size=blk, use_l2=l2, use_average=avg, use_maxpool=pool)
        got = got.cpu().numpy()
        err = float(np.max(np.abs(got - want) / (1e-7 + 2e-5 * np.abs(want))))
        worst = max(worst, err)
        if err > 1.0:
            print(f"MISMATCH seed={seed} lens={lens} n_obs={n_obs} blk={blk} Hk={Hk} qpk={qpk} hd={hd} buf={buf} "
                  f"l2={l2} avg={avg} pool={pool}: {err:.3g} x tolerance")
            sys.exit(1)
    print(f"soak ok: {n} fused-collector cases within 2e-5 of the oracle in {time.time() - t0:.1f} s "
          f"(worst {worst:.2g} of tolerance)")


if __name__ == "__main__":
    main()
