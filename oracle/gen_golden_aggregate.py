#!/usr/bin/env python3
"""Golden vectors for the metric-aggregation rows (A2a/A2b/A2c) from the REFERENCE.

* A2a/A2b: ``CompressionMetrics.aggregate_decode`` / ``aggregate_prefill`` of the imported
  ``vllm/kvcompress/metrics.py`` on CPU tensors.
* A2c: ``_naive_kvc_attention`` + ``_naive_kvc_masked_attention`` of
  ``vllm/attention/backends/flash_attn.py`` (:1122-1211).  That module cannot be imported here
  (it pulls the whole engine and the un-vendored ``vllm_flash_attn``), so the two function
  definitions are located with ``ast`` in the reference file and executed from there, in
  memory, with torch -- nothing is copied into this repository.
Build-container only (needs /root/reference).  Writes tests/golden/agg_*.npz."""
import ast
import contextlib
import importlib
import io
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"


def load_reference():
    import torch
    import torch.nn.functional as F
    sys.dont_write_bytecode = True
    pkg = types.ModuleType("vllm")
    pkg.__path__ = [os.path.join(REF, "vllm")]
    sys.modules["vllm"] = pkg
    torch.cuda.memory_allocated = lambda *a, **k: 0
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        met = importlib.import_module("vllm.kvcompress.metrics")
        bench = importlib.import_module("vllm.benchmark")
    path = os.path.join(REF, "vllm", "attention", "backends", "flash_attn.py")
    tree = ast.parse(open(path).read())
    wanted = {"_naive_kvc_attention", "_naive_kvc_masked_attention"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert {n.name for n in body} == wanted
    ns = {"torch": torch, "F": F, "List": list, "BENCHMARKER": bench.BENCHMARKER}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return met, ns["_naive_kvc_attention"]


def main():
    import torch
    met, naive = load_reference()
    out = os.path.join(REPO, "tests", "golden")
    rng = np.random.default_rng(2024)
    # ---- A2a / A2b
    for case, (qpk, l2) in enumerate([(4, True), (4, False), (1, True), (3, True)]):
        NB, bs, H, T = 12, 4, 3, 9
        with contextlib.redirect_stdout(io.StringIO()):
            cm = met.CompressionMetrics(bs, 2, H, qpk, 1000, None, 0.0, device="cpu", use_l2=l2)
            cm.init_kv_metadata(NB)
        m0 = rng.random((NB, bs)).astype(np.float32)
        t0 = rng.random((NB, bs, qpk)).astype(np.float32)
        cm.metrics[:] = torch.from_numpy(m0)
        cm.temp_metrics[:] = torch.from_numpy(t0)
        cm.aggregate_decode()
        after_decode = cm.metrics.numpy().copy()
        pm = rng.random((T, H * qpk)).astype(np.float32)
        slots = rng.permutation(NB * bs)[:T * H].astype(np.int64).reshape(T, H)
        cm.aggregate_prefill(torch.from_numpy(pm), torch.from_numpy(slots))
        np.savez_compressed(os.path.join(out, f"agg_decode_prefill_{case}.npz"), qpk=np.int32(qpk),
                            use_l2=np.int32(l2), num_kv_heads=np.int32(H), metrics0=m0, temp=t0,
                            ref_after_decode=after_decode, prefill_metrics=pm, slot_mapping=slots,
                            ref_after_prefill=cm.metrics.numpy().copy())
    # ---- A2c
    torch.manual_seed(0)
    case = 0
    for (l2, avg, pool) in [(a, b, c) for a in (True, False) for b in (True, False) for c in (True, False)]:
        for (lens, n_obs, blk, buf) in [([37], 16, 8, [0]), ([20, 45], 64, 16, [3, 0]), ([96], 32, 32, [3])]:
            T, Hq, hd = sum(lens), 4, 8
            q = (torch.randn(T, Hq, hd) * 1.5).half()
            k = (torch.randn(T, Hq, hd) * 1.5).half()
            with contextlib.redirect_stdout(io.StringIO()):
                _, res = naive(q, k, k, lens, hd ** -0.5, torch.tensor(buf, dtype=torch.int32),
                               n_observed=n_obs, max_observed_block_size=blk, use_l2=l2,
                               use_average=avg, use_maxpool=pool)
            np.savez_compressed(
                os.path.join(out, f"agg_prefill_attn_{case:02d}.npz"), q=q.numpy().view(np.uint16),
                k=k.numpy().view(np.uint16), prompt_lens=np.asarray(lens, np.int32),
                buffer_len=np.asarray(buf, np.int32), n_observed=np.int32(n_obs),
                block=np.int32(blk), use_l2=np.int32(l2), use_average=np.int32(avg),
                use_maxpool=np.int32(pool), ref_kv_metric_output=res.numpy().astype(np.float32))
            case += 1
    # ---- F4: the same reference function on MFMA-sized shapes (hd 64 / 128) for the fused
    # collector; keys are given per query head, as the engine passes them (flash_attn.py:988-991)
    torch.manual_seed(1)
    fcase = 0
    for (l2, avg, pool, lens, n_obs, blk, buf, Hq, hd, dt) in [
            (True, False, True, [150], 100, 64, [0], 4, 64, torch.float16),
            (True, True, True, [70, 200], 1000, 64, [3, 0], 4, 64, torch.float16),
            (False, False, False, [333], 333, 128, [5], 2, 128, torch.float16),
            (True, False, True, [260], 260, 100, [0], 4, 128, torch.bfloat16),
            (False, True, True, [97, 40], 50, 50, [0, 7], 8, 64, torch.float16)]:
        T = sum(lens)
        q = (torch.randn(T, Hq, hd) * 1.2).to(dt)
        k = (torch.randn(T, Hq, hd) * 1.2).to(dt)
        with contextlib.redirect_stdout(io.StringIO()):
            _, res = naive(q, k, k, lens, hd ** -0.5, torch.tensor(buf, dtype=torch.int32),
                           n_observed=n_obs, max_observed_block_size=blk, use_l2=l2,
                           use_average=avg, use_maxpool=pool)
        np.savez_compressed(
            os.path.join(out, f"agg_prefill_fused_{fcase}.npz"), q=q.view(torch.int16).numpy(),
            k=k.view(torch.int16).numpy(), dtype=("f16" if dt == torch.float16 else "bf16"),
            prompt_lens=np.asarray(lens, np.int32), buffer_len=np.asarray(buf, np.int32),
            n_observed=np.int32(n_obs), block=np.int32(blk), use_l2=np.int32(l2),
            use_average=np.int32(avg), use_maxpool=np.int32(pool),
            ref_kv_metric_output=res.numpy().astype(np.float32))
        fcase += 1
    print(f"wrote 4 decode/prefill cases, {case} prefill-attention cases and {fcase} fused-collector cases")


if __name__ == "__main__":
    main()
