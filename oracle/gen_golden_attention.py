#!/usr/bin/env python3
"""Golden vectors for the decode paged attention with metric output.

The native op is CUDA only; the reference pins it with a PyTorch twin in its own test
(``tests/kernels/test_kvcompress_attention.py``: ``ref_masked_attention`` :41-53 and the
gather loop of ``ref_single_query_cached_kv_attention`` :56-145).  ``ref_masked_attention`` is
located with ``ast`` in the reference file and executed from there, in memory (nothing is
copied into this repository); the gather loop is driven exactly as the twin drives it (keys /
values of one (seq, kv head) in logical order, one query head at a time, ALiBi bias
``slope * (i - ctx + 1)`` as :103-108).  Inputs follow the reference test's recipe
(``uniform(-scale, scale)`` queries and caches, random per-head context lengths, shuffled
physical blocks).  Expected: attention output per (seq, query head) and the softmax weights
per physical slot.  Build-container only (needs /root/reference).
Writes tests/golden/attn_decode_*.npz."""
import ast
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"


def load_twin():
    import torch
    from typing import Optional
    path = os.path.join(REF, "tests", "kernels", "test_kvcompress_attention.py")
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "ref_masked_attention"]
    assert len(body) == 1
    ns = {"torch": torch, "Optional": Optional}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns["ref_masked_attention"]


def make_case(rng, num_seqs, num_q_heads, num_kv_heads, hd, bs, max_ctx, dtype, use_alibi, part_len=None):
    import torch
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    scale = float(1.0 / (hd ** 0.5))
    x = 8
    ctx = rng.integers(1, max_ctx + 1, size=(num_seqs, num_kv_heads)).astype(np.int32)
    ctx[-1, -1] = max_ctx
    if num_seqs > 1:
        ctx[0, 0] = 1
    nblk = (ctx + bs - 1) // bs
    NB = int(nblk.sum()) + 3
    perm = rng.permutation(NB)
    M = int(nblk.max()) + 1
    bt = np.zeros((num_seqs, num_kv_heads, M), np.int32)
    cur = 0
    for s in range(num_seqs):
        for h in range(num_kv_heads):
            n = int(nblk[s, h])
            bt[s, h, :n] = perm[cur:cur + n]
            cur += n
    q = torch.from_numpy(rng.uniform(-scale, scale, (num_seqs, num_q_heads, hd)).astype(np.float32)).to(tdt)
    kc = torch.from_numpy(rng.uniform(-scale, scale, (NB, hd // x, bs, x)).astype(np.float32)).to(tdt)
    vc = torch.from_numpy(rng.uniform(-scale, scale, (NB, hd, bs)).astype(np.float32)).to(tdt)
    slopes = rng.standard_normal(num_q_heads).astype(np.float32) if use_alibi else None
    return dict(scale=scale, ctx=ctx, bt=bt, NB=NB, q=q, kc=kc, vc=vc, slopes=slopes)


def run_twin(twin, c, num_kv_heads, bs):
    import torch
    q, kc, vc = c["q"], c["kc"], c["vc"]
    S, Hq, hd = q.shape
    qpk = Hq // num_kv_heads
    NB = c["NB"]
    out = torch.zeros(S, Hq, hd, dtype=torch.float32)
    probs = np.full((NB, bs, qpk), -1.0, np.float32)          # -1 = slot not part of any head
    for s in range(S):
        for h in range(num_kv_heads):
            n = int(c["ctx"][s, h])
            blocks = torch.from_numpy(c["bt"][s, h, :(n + bs - 1) // bs].astype(np.int64))
            keys = kc[blocks].permute(0, 2, 1, 3).reshape(-1, hd)[:n].unsqueeze(1)     # [n,1,hd]
            vals = vc[blocks].permute(0, 2, 1).reshape(-1, hd)[:n].unsqueeze(1)        # [n,1,hd]
            phys = (blocks[:, None] * bs + torch.arange(bs)[None, :]).reshape(-1)[:n].numpy()
            for qo in range(qpk):
                qh = h * qpk + qo
                bias = None
                if c["slopes"] is not None:
                    pos = (torch.arange(n).int() - n + 1).float()
                    bias = torch.tensor(c["slopes"][qh]).view(1, 1, 1) * pos.view(1, 1, -1)
                o, w = twin(q[s, qh].view(1, 1, hd), keys, vals, c["scale"], bias)
                out[s, qh] = o.view(hd).float()
                probs.reshape(-1, qpk)[phys, qo] = w[0, 0].float().numpy()
    return out.numpy(), probs


def main():
    import torch
    torch.manual_seed(0)
    twin = load_twin()
    rng = np.random.default_rng(77)
    outdir = os.path.join(REPO, "tests", "golden")
    cases = [
        # S, Hq, Hkv, hd, bs, max_ctx, dtype, alibi
        (3, 8, 2, 128, 16, 70, "f16", False),
        (2, 4, 4, 64, 16, 33, "f16", True),
        (2, 16, 2, 128, 32, 200, "f16", False),
        (2, 8, 2, 128, 16, 1100, "f16", False),      # crosses the 512-token partition twice
        (2, 8, 2, 128, 16, 90, "bf16", False),
        (1, 4, 1, 256, 16, 50, "f16", True),
    ]
    for i, (S, Hq, Hkv, hd, bs, mx, dt, alibi) in enumerate(cases):
        c = make_case(rng, S, Hq, Hkv, hd, bs, mx, dt, alibi)
        out, probs = run_twin(twin, c, Hkv, bs)
        as_np = (lambda t: t.view(torch.int16).numpy())       # raw 16-bit patterns
        np.savez_compressed(
            os.path.join(outdir, f"attn_decode_{i}.npz"), dtype=dt, num_kv_heads=np.int32(Hkv),
            block_size=np.int32(bs), scale=np.float32(c["scale"]), query_bits=as_np(c["q"]),
            key_cache_bits=as_np(c["kc"]), value_cache_bits=as_np(c["vc"]), block_tables=c["bt"],
            context_lens=c["ctx"], alibi_slopes=(c["slopes"] if c["slopes"] is not None else np.zeros(0, np.float32)),
            ref_out=out, ref_probs=probs)
        print(i, S, Hq, Hkv, hd, bs, mx, dt, alibi, "NB", c["NB"])


if __name__ == "__main__":
    main()
