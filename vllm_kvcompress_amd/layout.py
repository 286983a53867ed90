"""Conversion of a KV cache between the two in-block layouts (``KVC_LAYOUT_*``, include/kvc_mi355x.h).

An engine never needs this -- a cache starts empty and every write, read and move of it goes through this package's
ops in ONE layout.  It exists for tools that hold a cache in one layout and want the other: the tests and bench.py
(the oracle computes in the reference's layout, reference ``csrc/kvcompress_cache_kernels.cu:57-77``) and anybody
migrating a saved cache.
"""
from __future__ import annotations

import torch

from . import _lib


def convert_block_layout(k_cache: torch.Tensor, v_cache: torch.Tensor, src: str, dst: str,
                         blocks_per_pass: int = 1 << 16) -> None:
    """In place: every block of ``k_cache`` ``[NB, hd/x, bs, x]`` and ``v_cache`` ``[NB, hd, bs]`` (the fork's views,
    reference vllm/attention/ops/paged_attn.py:272-284; shapes stay as they are) is re-laid-out from ``src`` to ``dst``
    (``"reference"``: K ``[hd/x][bs][x]``, V ``[hd][bs]``; ``"slot_major"``: K ``[bs][hd]``, V ``[bs][hd]``).  Pure
    byte permutation inside each block; ``blocks_per_pass`` bounds the temporary."""
    for name in (src, dst):
        if name not in _lib.LAYOUTS:
            raise ValueError(f"block layout {name!r}: expected one of {sorted(_lib.LAYOUTS)}")
    if src == dst:
        return
    if k_cache.dim() != 4 or v_cache.dim() != 3 or not k_cache.is_contiguous() or not v_cache.is_contiguous():
        raise RuntimeError("convert_block_layout: k_cache must be a contiguous [NB, hd/x, bs, x], v_cache [NB, hd, bs]")
    nb, kg, bs, x = k_cache.shape
    hd = kg * x
    to_slots = dst == "slot_major"
    for lo in range(0, nb, blocks_per_pass):
        hi = min(nb, lo + blocks_per_pass)
        k, v = k_cache[lo:hi], v_cache[lo:hi]
        if to_slots:
            k.view(hi - lo, bs, kg, x).copy_(k.permute(0, 2, 1, 3).contiguous())        # [bs][hd/x][x] = [bs][hd]
            v.view(hi - lo, bs, hd).copy_(v.permute(0, 2, 1).contiguous())
        else:
            k.copy_(k.view(hi - lo, bs, kg, x).permute(0, 2, 1, 3).contiguous())
            v.copy_(v.view(hi - lo, bs, hd).permute(0, 2, 1).contiguous())
    torch.autograd.graph.increment_version(k_cache)
    torch.autograd.graph.increment_version(v_cache)
