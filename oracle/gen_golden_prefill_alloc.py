#!/usr/bin/env python3
"""Golden vectors for a prefill sequence's FIRST allocation (F2, the prefill side) from the REFERENCE's own code:
``BlockSpaceManagerKVC._add_sequence`` (vllm/kvcompress/block_manager.py:196-222) run against the reference's
``BlockState`` / ``BlockStateView`` (vllm/kvcompress/block.py: ``get_block_state_seq_view``,
``get_allocated_block_metadata`` :414-446, ``get_prefill_slot_mapping`` :275-303), ``ParallelBlockAllocator``
(block_manager.py:76-118) and ``CompressionMetrics.insert_metadata`` (vllm/kvcompress/metrics.py:344-361) on CPU
tensors.

block.py and metrics.py import under the stub parent package; block_manager.py does not
(vllm.config -> msgspec), so the allocator class and the method are located with ``ast`` in the reference file and
executed from there with a stand-in ``self``.  Nothing is copied into the repository.  Build-container only.
Writes tests/golden/prefill_alloc_*.npz."""
import argparse
import ast
import contextlib
import importlib
import io
import os
import sys
import types
from types import SimpleNamespace

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    out_dir = ap.parse_args().out
    import torch
    sys.dont_write_bytecode = True
    pkg = types.ModuleType("vllm")
    pkg.__path__ = [os.path.join(REF, "vllm")]
    sys.modules["vllm"] = pkg
    torch.cuda.memory_allocated = lambda *a, **k: 0
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        blk = importlib.import_module("vllm.kvcompress.block")
        met = importlib.import_module("vllm.kvcompress.metrics")
    path = os.path.join(REF, "vllm", "kvcompress", "block_manager.py")
    tree = ast.parse(open(path).read())
    alloc_cls = add_fn = None
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == "ParallelBlockAllocator":
            alloc_cls = node
        if isinstance(node, ast.FunctionDef) and node.name == "_add_sequence":
            add_fn = node
    assert alloc_cls is not None and add_fn is not None
    alloc_cls.bases = []                                    # drop the abstract base
    for n in ast.walk(alloc_cls):
        if isinstance(n, ast.FunctionDef):
            n.decorator_list = []                           # drop @BENCHMARKER.wrap()
    add_fn.decorator_list = []
    ns = {"torch": torch, "List": list, "Sequence": object, "BlockTableView": lambda t: t,
          "PhysicalTokenBlock": object, "Optional": object}
    exec(compile(ast.Module(body=[alloc_cls, add_fn], type_ignores=[]), path, "exec"), ns)
    Allocator, add_sequence = ns["ParallelBlockAllocator"], ns["_add_sequence"]

    rng = np.random.default_rng(2024)
    for case, (L, S, H, M, bs, slot, seq_len, resident) in enumerate([
        # layers, max_num_seqs, heads, max blocks per head, block size, the new sequence's batch slot, its length,
        # other resident sequences (slot -> tokens per head, compressed states: ragged)
        (2, 4, 2, 9, 4, 1, 13, {0: 9, 3: 30}),
        (3, 3, 4, 12, 16, 0, 160, {}),                      # a multiple of the block size, an empty cache
        (2, 5, 3, 40, 2, 4, 1, {0: 7, 1: 3, 2: 12}),        # one token
        (2, 3, 2, 6, 16, 2, 81, {0: 50, 1: 17}),
        (1, 2, 8, 70, 16, 1, 1025, {0: 300}),
    ]):
        ctx = np.zeros((L, S, H), np.int32)
        for s, n in resident.items():
            ctx[:, s, :] = rng.integers(max(1, n // 2), n + 1, size=(L, H))
        nblk = (ctx + bs - 1) // bs
        need = L * H * ((seq_len + bs - 1) // bs)
        NB = int(nblk.sum()) + need + 11
        perm = rng.permutation(NB)
        bt = rng.integers(0, NB, size=(L, S, H, M)).astype(np.int32)   # garbage beyond nblk
        free_mask = np.ones(NB, dtype=bool)
        seq_by = np.full(NB, -1, np.int32); lay_by = np.zeros(NB, np.int32)
        head_by = np.zeros(NB, np.int32); lbn_by = np.zeros(NB, np.int32)
        pos = rng.integers(0, 1000, size=(NB, bs)).astype(np.int32)
        cur = 0
        for l in range(L):
            for s in range(S):
                for h in range(H):
                    n = int(nblk[l, s, h])
                    blocks = perm[cur:cur + n]
                    bt[l, s, h, :n] = blocks
                    cur += n
                    free_mask[blocks] = False
                    seq_by[blocks], lay_by[blocks], head_by[blocks] = s, l, h
                    lbn_by[blocks] = np.arange(n)
        state = blk.BlockState.__new__(blk.BlockState)
        state.block_size = bs
        state.num_layers, state.num_kv_heads, state.max_num_seqs = L, H, S
        state.max_num_blocks_per_head = M
        state.use_tiered_block_tables = False
        state.block_tables = torch.from_numpy(bt.copy())
        state.t2_block_tables = None
        state.context_lens = torch.from_numpy(ctx.copy())
        state.block_table_indices = torch.arange(M)[None, None, None]
        alloc = Allocator.__new__(Allocator)
        alloc.num_blocks, alloc.device = NB, "cpu"
        alloc.block_numbers = torch.arange(NB)
        alloc.free_mask = torch.from_numpy(free_mask.copy())
        alloc.free_count = int(free_mask.sum())
        cm = met.CompressionMetrics.__new__(met.CompressionMetrics)
        cm.seq_index_by_block = torch.from_numpy(seq_by.copy())
        cm.layer_index_by_block = torch.from_numpy(lay_by.copy())
        cm.head_index_by_block = torch.from_numpy(head_by.copy())
        cm.logical_block_num_by_block = torch.from_numpy(lbn_by.copy())
        cm.token_positions = torch.from_numpy(pos.copy())
        self_ = SimpleNamespace(free_batch_slots={slot}, batch_slot_mapping={}, device="cpu", block_size=bs,
                                num_layers=L, num_kv_heads=H, block_state=state, gpu_allocator=alloc, kv_metrics=cm)
        with contextlib.redirect_stdout(io.StringIO()):
            add_sequence(self_, 777, seq_len)
            assert self_.batch_slot_mapping == {777: slot}
            slot_mapping = state.get_block_state_seq_view(slot).get_prefill_slot_mapping()
        np.savez_compressed(
            os.path.join(out_dir, f"prefill_alloc_{case}.npz"), block_size=np.int32(bs), block_tables=bt,
            context_lens=ctx, seq_slot=np.int32(slot), seq_len=np.int32(seq_len),
            free_mask=free_mask, seq_index_by_block=seq_by, layer_index_by_block=lay_by,
            head_index_by_block=head_by, logical_block_num_by_block=lbn_by, token_positions=pos,
            ref_block_tables=state.block_tables.numpy().astype(np.int32),
            ref_context_lens=state.context_lens.numpy().astype(np.int32),
            ref_free_mask=alloc.free_mask.numpy(), ref_free_count=np.int64(alloc.free_count),
            ref_seq_index_by_block=cm.seq_index_by_block.numpy().astype(np.int32),
            ref_layer_index_by_block=cm.layer_index_by_block.numpy().astype(np.int32),
            ref_head_index_by_block=cm.head_index_by_block.numpy().astype(np.int32),
            ref_logical_block_num_by_block=cm.logical_block_num_by_block.numpy().astype(np.int32),
            ref_token_positions=cm.token_positions.numpy().astype(np.int32),
            ref_slot_mapping=slot_mapping.numpy().astype(np.int64))
        print(f"prefill_alloc_{case}: {need} blocks, slot mapping {tuple(slot_mapping.shape)}")


if __name__ == "__main__":
    main()
