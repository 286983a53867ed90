"""Device-side block-state update after a compression step.

``free_compressed_blocks`` does, in two small kernels and without boolean-mask gathers,
what ``BlockSpaceManagerKVC.free_compressed_blocks`` does through ``BlockStateView``,
``ParallelBlockAllocator.free``, ``BlockState.remove_trailing_blocks`` and
``CompressionMetrics.remove_metadata`` (vllm/kvcompress/block_manager.py:466-530,
block.py:184-210, 367-379, metrics.py:366-370).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from .. import _lib
from .._custom_ops import _stream, workspace


def free_compressed_blocks(
    block_tables: torch.Tensor,          # [L, max_num_seqs, H, M] i32   (BlockState.block_tables)
    context_lens: torch.Tensor,          # [L, max_num_seqs, H]    i32   (updated in place)
    seq_indices: Sequence[int],          # batch slots of the compressed sequences (batch order)
    freed_block_count: torch.Tensor,     # [B, L, H] i32  (evicted_block_count of schedule_evictions)
    seq_index_by_block: torch.Tensor,    # [NB] i32  (CompressionMetrics; -1 written for freed blocks)
    block_size: int,
    free_mask: Optional[torch.Tensor] = None,   # [NB] bool (ParallelBlockAllocator.free_mask)
    max_freed: Optional[int] = None,            # host-known upper bound: sum(evicted_blocks_per_seq)
) -> torch.Tensor:
    """Returns the freed physical blocks (int32, the reference's order).  ``max_freed`` sizes the
    output list (default: every block of the batch, B*L*H*M entries); a bound below what was
    really freed raises instead of returning a truncated list."""
    lib = _lib.load()
    for n, t in (("block_tables", block_tables), ("context_lens", context_lens),
                 ("freed_block_count", freed_block_count), ("seq_index_by_block", seq_index_by_block)):
        if not t.is_cuda or t.dtype != torch.int32 or not t.is_contiguous():
            raise RuntimeError(f"free_compressed_blocks: {n} must be a contiguous int32 HIP tensor")
    L, S, H, M = block_tables.shape
    B = len(seq_indices)
    assert tuple(freed_block_count.shape) == (B, L, H)
    dev = block_tables.device
    slots = torch.tensor(list(seq_indices), dtype=torch.int32).to(dev, non_blocking=True)
    cap = int(B * L * H * M) if max_freed is None else max(int(max_freed), 1)
    freed = torch.empty((cap,), dtype=torch.int32, device=dev)
    total = torch.zeros((1,), dtype=torch.int32, device=dev)
    fm_ptr = None
    if free_mask is not None:
        if free_mask.dtype not in (torch.bool, torch.uint8) or not free_mask.is_cuda:
            raise RuntimeError("free_compressed_blocks: free_mask must be a bool/uint8 HIP tensor")
        fm_ptr = free_mask.data_ptr()
    ws_bytes = lib.kvc_free_compressed_blocks_workspace_bytes(L, B, H)
    ws = workspace(dev, ws_bytes, "free_compressed_blocks")
    for t in (context_lens, seq_index_by_block):       # (written through raw pointers: tell the version counters)
        torch.autograd.graph.increment_version(t)
    with torch.cuda.device(dev):
        _lib.check(lib.kvc_free_compressed_blocks(
            context_lens.data_ptr(), seq_index_by_block.data_ptr(), fm_ptr, freed.data_ptr(), cap,
            total.data_ptr(), block_tables.data_ptr(), freed_block_count.data_ptr(), slots.data_ptr(),
            L, B, S, H, M, int(block_size), ws.data_ptr(), ws.numel(), _stream(block_tables)))
    n = int(total.item())                  # exact size like the reference (one host sync)
    if n > cap:
        # the kernel freed n blocks (free_mask, metadata and context_lens say so) but could list only
        # `cap` of them: an allocator fed from the list would leak the rest
        raise RuntimeError(f"free_compressed_blocks: {n} blocks were freed but max_freed={cap} bounds the "
                           "returned list; pass the true upper bound (sum of evicted_blocks_per_seq) or omit it")
    return freed[:n]


def append_slots(
    block_tables: torch.Tensor,          # [L, max_num_seqs, H, M] i32   (BlockState.block_tables, updated)
    context_lens: torch.Tensor,          # [L, max_num_seqs, H]    i32   (updated)
    seq_indices: Sequence[int],          # batch slots of the decoding sequences (batch order)
    last_token_position: Sequence[int],  # per sequence: seq.data.get_len() - 1
    free_mask: torch.Tensor,             # [NB] bool (ParallelBlockAllocator.free_mask, updated)
    kv_metrics,                          # CompressionMetrics: metadata rows + token_positions (updated)
    block_size: int,
    write_token_position: bool = False,
) -> int:
    """``BlockSpaceManagerKVC._append_to_sequence_batch`` (vllm/kvcompress/block_manager.py:269-294)
    with ``token_count = 1`` on device: five small launches instead of two boolean masks over
    [L,B,H,M], a masked gather / scatter and five indexed stores.  Returns the number of newly
    allocated blocks (one host sync, which the reference has too: ``new_mask.sum()``); raises
    ``ValueError("Out of memory! ...")`` like ``ParallelBlockAllocator.allocate`` -- in that case
    nothing has been modified."""
    lib = _lib.load()
    for n, t in (("block_tables", block_tables), ("context_lens", context_lens),
                 ("seq_index_by_block", kv_metrics.seq_index_by_block),
                 ("token_positions", kv_metrics.token_positions)):
        if not t.is_cuda or t.dtype != torch.int32 or not t.is_contiguous():
            raise RuntimeError(f"append_slots: {n} must be a contiguous int32 HIP tensor")
    if free_mask.dtype not in (torch.bool, torch.uint8) or not free_mask.is_cuda or not free_mask.is_contiguous():
        raise RuntimeError("append_slots: free_mask must be a contiguous bool/uint8 HIP tensor")
    L, S, H, M = block_tables.shape
    B = len(seq_indices)
    assert len(last_token_position) == B
    dev = block_tables.device
    host = torch.tensor(list(seq_indices) + list(last_token_position), dtype=torch.int32).to(dev, non_blocking=True)
    status = torch.zeros((2,), dtype=torch.int32, device=dev)
    NB = int(free_mask.numel())
    ws_bytes = lib.kvc_append_slots_workspace_bytes(L, B, H, NB)
    ws = workspace(dev, ws_bytes, "append_slots")
    cm = kv_metrics
    # (written through raw pointers: say so to whoever asks the tensors' version counters)
    for t in (cm.seq_index_by_block, cm.layer_index_by_block, cm.head_index_by_block, cm.logical_block_num_by_block,
              cm.token_positions, context_lens, block_tables):
        torch.autograd.graph.increment_version(t)
    with torch.cuda.device(dev):
        _lib.check(lib.kvc_append_slots(
            context_lens.data_ptr(), block_tables.data_ptr(), free_mask.data_ptr(),
            cm.seq_index_by_block.data_ptr(), cm.layer_index_by_block.data_ptr(),
            cm.head_index_by_block.data_ptr(), cm.logical_block_num_by_block.data_ptr(),
            cm.token_positions.data_ptr(), host.data_ptr(), host[B:].data_ptr(), status.data_ptr(),
            L, B, S, H, M, NB, int(block_size), 1 if write_token_position else 0, ws.data_ptr(),
            ws.numel(), _stream(block_tables)))
    need, free = (int(x) for x in status.tolist())
    if free < 0:
        raise RuntimeError("append_slots: a head's block table is full (max_num_blocks_per_head reached)")
    if need > free:
        raise ValueError(f"Out of memory! Requested {need} out of {free} available blocks.")
    return need


def add_sequence(
    block_tables: torch.Tensor,          # [L, max_num_seqs, H, M] i32   (BlockState.block_tables, updated)
    context_lens: torch.Tensor,          # [L, max_num_seqs, H]    i32   (updated)
    seq_slot: int,                       # the batch slot the sequence takes
    seq_len: int,                        # its prompt length in tokens
    free_mask: torch.Tensor,             # [NB] bool (ParallelBlockAllocator.free_mask, updated)
    kv_metrics,                          # CompressionMetrics: metadata rows + token_positions (updated)
    block_size: int,
    slot_mapping: bool = True,
):
    """``BlockSpaceManagerKVC._add_sequence`` (vllm/kvcompress/block_manager.py:196-222: allocate ``ceil(T / bs)``
    blocks per (layer, head) from the low end of the free list, wire them into the tables, set the context lengths,
    ``get_allocated_block_metadata`` + ``CompressionMetrics.insert_metadata``) and, in the same pass,
    ``BlockStateView.get_prefill_slot_mapping`` (block.py:275-303) on device.  Returns ``(blocks allocated,
    slot_mapping [L, T, H] int64 or None)`` -- ``slot_mapping[l]`` is what ``reshape_and_cache_kvc`` takes for layer
    ``l``.  One host sync (the reference's allocator keeps its free count on the host); raises
    ``ValueError("Out of memory! ...")`` like ``ParallelBlockAllocator.allocate`` -- nothing has been modified then."""
    lib = _lib.load()
    cm = kv_metrics
    for n, t in (("block_tables", block_tables), ("context_lens", context_lens),
                 ("seq_index_by_block", cm.seq_index_by_block), ("token_positions", cm.token_positions)):
        if not t.is_cuda or t.dtype != torch.int32 or not t.is_contiguous():
            raise RuntimeError(f"add_sequence: {n} must be a contiguous int32 HIP tensor")
    if free_mask.dtype not in (torch.bool, torch.uint8) or not free_mask.is_cuda or not free_mask.is_contiguous():
        raise RuntimeError("add_sequence: free_mask must be a contiguous bool/uint8 HIP tensor")
    L, S, H, M = block_tables.shape
    dev = block_tables.device
    NB, T, bs = int(free_mask.numel()), int(seq_len), int(block_size)
    status = torch.zeros((2,), dtype=torch.int32, device=dev)
    sm = torch.empty((L, T, H), dtype=torch.int64, device=dev) if slot_mapping else None
    ws = workspace(dev, int(lib.kvc_add_sequence_workspace_bytes(L, H, T, bs, NB)), "add_sequence")
    for t in (cm.seq_index_by_block, cm.layer_index_by_block, cm.head_index_by_block, cm.logical_block_num_by_block,
              cm.token_positions, context_lens, block_tables):
        torch.autograd.graph.increment_version(t)
    cm._hv_lists = None                   # (insert_metadata: lists made before it do not know the new blocks)
    with torch.cuda.device(dev):
        _lib.check(lib.kvc_add_sequence(
            context_lens.data_ptr(), block_tables.data_ptr(), free_mask.data_ptr(),
            cm.seq_index_by_block.data_ptr(), cm.layer_index_by_block.data_ptr(), cm.head_index_by_block.data_ptr(),
            cm.logical_block_num_by_block.data_ptr(), cm.token_positions.data_ptr(),
            None if sm is None else sm.data_ptr(), status.data_ptr(), L, S, H, M, NB, bs, int(seq_slot), T,
            ws.data_ptr(), ws.numel(), _stream(block_tables)))
    need, free = (int(x) for x in status.tolist())
    if free < 0:
        raise RuntimeError("add_sequence: the sequence needs more blocks per head than the block table holds "
                           "(max_num_blocks_per_head)")
    if need > free:
        raise ValueError(f"Out of memory! Requested {need} out of {free} available blocks.")
    return need, sm
