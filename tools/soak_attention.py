#!/usr/bin/env python3
"""Soak: random decode-attention shapes (GQA group sizes 1..16, head sizes 64/128, block sizes
16/32, contexts from 1 to 9000 tokens, fp16/bf16, metric windows, ALiBi, both kernel
schedules incl. the 8-wave single-pass variant) against the oracle.
Run on the GPU box:  python tools/soak_attention.py [ncases] [--layout slot_major]
(--layout slot_major: the caches permuted into slot-major blocks and the package switched to that layout)"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from tests.attn_helpers import make_state, oracle_decode            # noqa: E402
from tests.test_gpu_attention import _run_gpu, _set_mode             # noqa: E402


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    layout = "slot_major" if "slot_major" in sys.argv[1:] else "reference"
    n = int(argv[0]) if argv and argv[0].isdigit() else 100
    run = _run_gpu
    if layout == "slot_major":
        from vllm_kvcompress_amd import _lib
        from tests.test_gpu_slot_major import _attn
        _lib.set_block_layout("slot_major")
        run = _attn
    t0 = time.time()
    worst_w, worst_o = 0.0, 0.0
    for seed in range(n):
        rng = np.random.default_rng(7000 + seed)
        S = int(rng.integers(1, 4))
        Hkv = int(rng.integers(1, 4))
        qpk = int(rng.choice([1, 2, 4, 8, 16]))
        hd = int(rng.choice([64, 128]))
        bs = int(rng.choice([16, 32]))
        hi = int(rng.choice([40, 600, 2500, 4300, 9000]))
        lo = int(rng.integers(1, max(2, hi // 2)))
        dt = str(rng.choice(["f16", "bf16"]))
        alibi = bool(rng.random() < 0.2)
        g, c, pos, last = make_state(rng, S, Hkv * qpk, Hkv, hd, bs, lo, hi, dtype=dt, alibi=alibi)
        buf = rng.integers(0, 50, size=S).astype(np.int32)
        ref_out, ref_km = oracle_decode(c, g, pos, last, buf)
        for mode in (1, 2):
            _set_mode(mode)
            out, km = run(g, c, pos, last, buf, "v1" if seed % 2 else "v2")
            same = ((ref_km == -1.0) == (km == -1.0)).all()
            rec = ref_km != -1.0
            ew = float(np.max(np.abs(km[rec] - ref_km[rec]) / (np.abs(ref_km[rec]) + 1e-9))) if rec.any() else 0.0
            tol = 2e-3 if dt == "f16" else 1.6e-2
            eo = float(np.max(np.abs(out - ref_out) / (tol + tol * np.abs(ref_out))))
            worst_w, worst_o = max(worst_w, ew), max(worst_o, eo)
            # weights: fp32 softmax, summation order + ALiBi rounding (see tests); output: 1 ulp of T
            if not same or ew > (2e-3 if alibi else 3e-4) or eo > 1.0:
                print(f"MISMATCH seed={seed} mode={mode} S={S} Hkv={Hkv} qpk={qpk} hd={hd} bs={bs} "
                      f"ctx<={hi} {dt} alibi={alibi}: slots_same={same} weights_rel={ew:.3g} out={eo:.3g}")
                _set_mode(0)
                sys.exit(1)
    _set_mode(0)
    print(f"soak ok ({layout} blocks): {n} attention cases x 2 schedules within tolerance in {time.time() - t0:.1f} s "
          f"(worst weight rel err {worst_w:.2g}, worst output err {worst_o:.2g} of tolerance)")


if __name__ == "__main__":
    main()
