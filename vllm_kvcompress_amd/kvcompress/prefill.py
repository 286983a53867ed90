"""Prefill metric collection: the aggregation epilogue of the reference's
``_naive_kvc_attention`` / ``_naive_kvc_masked_attention``
(``vllm/attention/backends/flash_attn.py:1122-1211``) on MI355X.

The dense part (QK^T + softmax, a contraction) stays with the GEMM library through torch
and is not in this repo's scope; what is implemented in HIP is
everything after the probabilities exist: square -> causal/buffer mask -> column sum over
the query block -> optional position rescale -> max_pool1d(7) per query block ->
accumulate into ``kv_metric_output [K, Hq]`` (note: pooled per q-block *before*
accumulation).
"""
from __future__ import annotations

from typing import List

import torch

from .. import _lib
from .._custom_ops import _stream, workspace


def accumulate_prefill_tile(out_kh: torch.Tensor, probs_hqk: torch.Tensor, q_offset: int,
                            buffer_len: int, use_l2: bool = True, use_average: bool = False,
                            use_maxpool: bool = True) -> None:
    """out_kh [K,Hq] f32 += epilogue(probs_hqk [Hq,qb,K] f32); ``q_offset`` = position of
    the tile's first query inside the sequence."""
    lib = _lib.load()
    if not (out_kh.is_cuda and probs_hqk.is_cuda):
        raise RuntimeError("accumulate_prefill_tile: tensors must be on a HIP device")
    if out_kh.dtype != torch.float32 or probs_hqk.dtype != torch.float32:
        raise RuntimeError("accumulate_prefill_tile: float32 tensors required")
    if not out_kh.is_contiguous():
        raise RuntimeError("accumulate_prefill_tile: out_kh must be contiguous (in place)")
    probs_hqk = probs_hqk.contiguous()
    Hq, qb, K = probs_hqk.shape
    assert tuple(out_kh.shape) == (K, Hq)
    ws_bytes = lib.kvc_prefill_metric_epilogue_workspace_bytes(Hq, K)
    ws = workspace(out_kh.device, ws_bytes, "prefill_epilogue")
    with torch.cuda.device(out_kh.device):
        _lib.check(lib.kvc_prefill_metric_epilogue(
            out_kh.data_ptr(), probs_hqk.data_ptr(), Hq, qb, K, int(q_offset), int(buffer_len),
            int(bool(use_l2)), int(bool(use_average)), int(bool(use_maxpool)), ws.data_ptr(),
            ws.numel(), _stream(out_kh)))


def naive_kvc_attention(query: torch.Tensor, key: torch.Tensor, value, prompt_lens: List[int],
                        scale: float, kv_metric_buffer_len: torch.Tensor, n_observed: int = 32,
                        max_observed_block_size: int = 4096, use_l2: bool = True,
                        use_average: bool = False, use_maxpool: bool = True):
    """Same contract as the reference ``_naive_kvc_attention`` (flash_attn.py:1122-1164):
    returns ``(None, kv_metric_output [T, Hq] f32)``."""
    seq_len, num_heads, _ = key.shape
    out = torch.zeros((seq_len, num_heads), dtype=torch.float32, device=key.device)
    buf = kv_metric_buffer_len.tolist()
    start = 0
    for i, prompt_len in enumerate(prompt_lens):
        end = start + prompt_len
        start_trunc = end - min(prompt_len, n_observed)
        k = key[start:end]
        for l in range(start_trunc, end, max_observed_block_size):
            q = query[l:min(l + max_observed_block_size, end)]
            q_offset = l - start
            nq = q.shape[0]
            # dense contraction + softmax (library GEMM through torch; out of scope here)
            w = scale * torch.einsum("qhd,khd->hqk", q, k).float()
            cols = torch.arange(prompt_len, device=key.device)[None, :]
            rows = torch.arange(nq, device=key.device)[:, None]
            w = w.masked_fill((cols - rows > q_offset)[None], torch.finfo(q.dtype).min)
            probs = torch.softmax(w, dim=-1)
            accumulate_prefill_tile(out[start:end], probs, q_offset, int(buf[i]), use_l2,
                                    use_average, use_maxpool)
        start += prompt_len
    return None, out


def fused_kvc_attention(query: torch.Tensor, key: torch.Tensor, value, prompt_lens: List[int],
                        scale: float, kv_metric_buffer_len: torch.Tensor, n_observed: int = 32,
                        max_observed_block_size: int = 4096, use_l2: bool = True,
                        use_average: bool = False, use_maxpool: bool = True):
    """Same contract and loop structure as the reference ``_naive_kvc_attention``
    (flash_attn.py:1120-1164) -- returns ``(None, kv_metric_output [T, Hq] f32)`` -- but the
    probabilities are never materialised: per (sequence, query block) two matrix-core passes
    (row log-sum-exp, then masked column sums of P or P^2 recomputed from it) and the pool +
    accumulate step run inside ``kvc_prefill_metric_fused``.
    ``key`` may carry one head per query head (what the engine passes after repeating the KV
    heads, flash_attn.py:988-991) or the un-repeated KV heads."""
    lib = _lib.load()
    if not (query.is_cuda and key.is_cuda):
        raise RuntimeError("fused_kvc_attention: tensors must be on a HIP device")
    dtypes = {torch.float16: 0, torch.bfloat16: 1}
    if query.dtype not in dtypes or key.dtype != query.dtype:
        raise RuntimeError(f"Unsupported data type: {query.dtype}")
    T, Hq, hd = query.shape
    Hk = key.shape[1]
    if query.stride(2) != 1 or query.stride(1) != hd or key.stride(2) != 1 or key.stride(1) != hd:
        raise RuntimeError("fused_kvc_attention: query / key must be contiguous in (head, dim)")
    out = torch.zeros((key.shape[0], Hq), dtype=torch.float32, device=key.device)
    buf = kv_metric_buffer_len.tolist()
    esz = query.element_size()
    start = 0
    with torch.cuda.device(query.device):
        for i, prompt_len in enumerate(prompt_lens):
            end = start + prompt_len
            n_obs = min(prompt_len, n_observed)
            first = end - n_obs
            ws_bytes = lib.kvc_prefill_metric_fused_workspace_bytes(Hq, n_obs, prompt_len)
            ws = workspace(query.device, ws_bytes, "prefill_fused")
            _lib.check(lib.kvc_prefill_metric_fused(
                out[start:end].data_ptr(), query.data_ptr() + first * query.stride(0) * esz,
                key.data_ptr() + start * key.stride(0) * esz, Hq, Hk, hd, n_obs,
                int(max_observed_block_size), prompt_len, first - start, int(buf[i]),
                query.stride(0), key.stride(0), float(scale), dtypes[query.dtype],
                int(bool(use_l2)), int(bool(use_average)), int(bool(use_maxpool)), ws.data_ptr(),
                ws.numel(), _stream(query)))
            start += prompt_len
    return None, out
