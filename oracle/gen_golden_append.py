#!/usr/bin/env python3
"""Golden vectors for the APPEND side of the block state (F2) from the REFERENCE's own code:
``BlockSpaceManagerKVC._append_to_sequence_batch`` (vllm/kvcompress/block_manager.py:269-294) run
against the reference's ``BlockState`` / ``BlockStateView`` (vllm/kvcompress/block.py),
``ParallelBlockAllocator`` (block_manager.py:76-118) and ``CompressionMetrics.insert_metadata``
(vllm/kvcompress/metrics.py:344-361) on CPU tensors.

block.py and metrics.py import under the stub parent package;
block_manager.py does not (vllm.config -> msgspec), so the allocator class and the method are
located with ``ast`` in the reference file and executed from there with a stand-in ``self``.
Nothing is copied into the repository.  Build-container only.  Writes tests/golden/append_*.npz."""
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
    alloc_cls = append_fn = None
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == "ParallelBlockAllocator":
            alloc_cls = node
        if isinstance(node, ast.FunctionDef) and node.name == "_append_to_sequence_batch":
            append_fn = node
    assert alloc_cls is not None and append_fn is not None
    alloc_cls.bases = []                                    # drop the abstract base
    for n in ast.walk(alloc_cls):
        if isinstance(n, ast.FunctionDef):
            n.decorator_list = []                           # drop @BENCHMARKER.wrap()
    append_fn.decorator_list = []

    class BlockTableView(torch.Tensor):                     # block_manager.py:67-73 stand-in (a view type)
        @staticmethod
        def __new__(cls, data):
            return data
    ns = {"torch": torch, "List": list, "Sequence": object, "BlockTableView": lambda t: t,
          "PhysicalTokenBlock": object, "Optional": object}
    exec(compile(ast.Module(body=[alloc_cls, append_fn], type_ignores=[]), path, "exec"), ns)
    Allocator, append = ns["ParallelBlockAllocator"], ns["_append_to_sequence_batch"]

    out_dir = os.path.join(REPO, "tests", "golden")
    rng = np.random.default_rng(99)
    for case, (L, S, H, M, bs, sel) in enumerate([
        (2, 4, 2, 9, 4, [0, 2, 3]),
        (3, 3, 4, 12, 16, [1]),
        (2, 5, 3, 7, 2, [0, 1, 2, 3, 4]),
        (2, 3, 2, 6, 16, [0, 2]),
    ]):
        ctx = rng.integers(0, (M - 1) * bs + 1, size=(L, S, H)).astype(np.int32)
        # plenty of heads exactly at a block boundary (they need a new block), one empty head
        bound = rng.random(ctx.shape) < 0.5
        ctx[bound] = ctx[bound] // bs * bs
        ctx[0, sel[0], 0] = 0
        nblk = (ctx + bs - 1) // bs
        NB = int(nblk.sum()) + L * len(sel) * H + 7
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
        last_pos = [int(rng.integers(50, 900)) for _ in sel]
        state = blk.BlockState.__new__(blk.BlockState)
        state.block_size = bs
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
        seqs = [SimpleNamespace(seq_id=100 + i, data=SimpleNamespace(get_len=lambda lp=lp: lp + 1))
                for i, lp in enumerate(last_pos)]
        self_ = SimpleNamespace(batch_slot_mapping={100 + i: s for i, s in enumerate(sel)}, device="cpu",
                                block_state=state, gpu_allocator=alloc, kv_metrics=cm)
        with contextlib.redirect_stdout(io.StringIO()):
            append(self_, seqs, 1)
        np.savez_compressed(
            os.path.join(out_dir, f"append_{case}.npz"), block_size=np.int32(bs), block_tables=bt,
            context_lens=ctx, seq_indices=np.asarray(sel, np.int32), last_token_position=np.asarray(last_pos, np.int32),
            free_mask=free_mask, seq_index_by_block=seq_by, layer_index_by_block=lay_by,
            head_index_by_block=head_by, logical_block_num_by_block=lbn_by, token_positions=pos,
            ref_block_tables=state.block_tables.numpy().astype(np.int32),
            ref_context_lens=state.context_lens.numpy().astype(np.int32),
            ref_free_mask=alloc.free_mask.numpy(), ref_free_count=np.int64(alloc.free_count),
            ref_seq_index_by_block=cm.seq_index_by_block.numpy().astype(np.int32),
            ref_layer_index_by_block=cm.layer_index_by_block.numpy().astype(np.int32),
            ref_head_index_by_block=cm.head_index_by_block.numpy().astype(np.int32),
            ref_logical_block_num_by_block=cm.logical_block_num_by_block.numpy().astype(np.int32),
            ref_token_positions=cm.token_positions.numpy().astype(np.int32))
        print(f"append_{case}: {int(free_mask.sum()) - alloc.free_count} new blocks")


if __name__ == "__main__":
    main()
