#!/usr/bin/env python3
"""Golden vectors for the host policy (A8) from the REFERENCE's
``CompressionScheduler._schedule_seq_evictions`` (vllm/kvcompress/scheduler.py:100-181).

The module cannot be imported (it pulls vllm.sequence -> msgspec), so the method definition
is located with ``ast`` in the reference file and executed from there with stand-in
``self`` / ``seq`` objects that expose exactly the attributes it reads.  Nothing is copied into
the repository.  Build-container only.  Writes tests/golden/policy_cases.npz."""
import ast
import math
import os
import sys
from types import SimpleNamespace

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def main():
    path = os.path.join(REF, "vllm", "kvcompress", "scheduler.py")
    tree = ast.parse(open(path).read())
    fn = None
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "_schedule_seq_evictions":
            fn = node
    assert fn is not None
    fn.decorator_list = []                      # drop @BENCHMARKER.wrap()
    ns = {"math": math, "Tuple": tuple, "Sequence": object}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    ref = ns["_schedule_seq_evictions"]

    rng = np.random.default_rng(5)
    rows = []
    for _ in range(400):
        L, H = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        bs = int(rng.choice([1, 2, 4, 16, 32]))
        seq_len = int(rng.integers(2, 400))
        # per-head context lens: uncompressed (seq_len - 1) or arbitrary smaller
        if rng.random() < 0.5:
            ctx = np.full((L, H), seq_len - 1, dtype=np.int64)
        else:
            ctx = rng.integers(0, seq_len, size=(L, H)).astype(np.int64)
        prot = int(rng.integers(0, 80))
        use_rate = rng.random() < 0.5
        rate = float(rng.choice([0.1, 0.25, 0.5, 0.9, 1.0])) if use_rate else 1.0
        mct = -1 if use_rate else int(rng.integers(0, 300))
        even = bool(rng.random() < 0.2)
        kv_count = int(ctx.sum())
        blk_count = int(((ctx + bs - 1) // bs).sum())
        self_ = SimpleNamespace(
            block_size=bs,
            config=SimpleNamespace(num_layers=L, num_kv_heads=H, even_layer_evict=even),
            block_manager=SimpleNamespace(get_sequence_kv_count=lambda s, kv=kv_count: kv,
                                          get_sequence_block_count=lambda s, b=blk_count: b))
        seq = SimpleNamespace(compressed=False, data=SimpleNamespace(get_len=lambda sl=seq_len: sl))
        try:
            _, blocks = ref(self_, seq, rate, mct, prot, False)
            ok = 1
        except AssertionError:
            blocks, ok = -1, 0                  # the reference's own sanity assert fired
        rows.append((L, H, bs, seq_len, prot, rate, mct, int(even), blocks, ok, ctx))
    np.savez_compressed(
        os.path.join(REPO, "tests", "golden", "policy_cases.npz"),
        scalars=np.array([r[:10] for r in rows], dtype=np.float64),
        ctx_flat=np.concatenate([r[10].reshape(-1) for r in rows]).astype(np.int64))
    print(f"wrote {len(rows)} policy cases ({sum(int(r[9]) for r in rows)} without assertion)")


if __name__ == "__main__":
    main()
