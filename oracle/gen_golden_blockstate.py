#!/usr/bin/env python3
"""Golden vectors for the block-state side (F2) from the REFERENCE's own block.py.

Imports ``vllm/kvcompress/block.py`` under the stub parent package,
builds a ``BlockState`` without running its CUDA-only ``__init__`` (its tensors are plain
CPU tensors here) and drives ``BlockStateView.get_last_n_allocated_blocks`` +
``BlockState.remove_trailing_blocks`` exactly as ``BlockSpaceManagerKVC.free_compressed_blocks``
does (vllm/kvcompress/block_manager.py:466-530).  Writes tests/golden/blockstate_*.npz.
Build-container only (needs /root/reference)."""
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


def main():
    import torch
    sys.dont_write_bytecode = True
    pkg = types.ModuleType("vllm")
    pkg.__path__ = [os.path.join(REF, "vllm")]
    sys.modules["vllm"] = pkg
    torch.cuda.memory_allocated = lambda *a, **k: 0
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        blk = importlib.import_module("vllm.kvcompress.block")
    out_dir = os.path.join(REPO, "tests", "golden")
    rng = np.random.default_rng(77)
    for case, (L, S, H, M, bs, sel) in enumerate([
        (2, 4, 2, 9, 4, [0, 2, 3]),
        (3, 3, 4, 12, 16, [1]),
        (2, 5, 3, 7, 2, [0, 1, 2, 3, 4]),
    ]):
        ctx = rng.integers(0, M * bs + 1, size=(L, S, H)).astype(np.int32)
        ctx[0, sel[0], 0] = 0                                   # an empty head
        ctx[-1, sel[-1], -1] = M * bs                            # a full head
        nblk = (ctx + bs - 1) // bs
        NB = int(nblk.sum()) + 5
        perm = rng.permutation(NB)
        bt = rng.integers(0, NB, size=(L, S, H, M)).astype(np.int32)   # garbage beyond nblk
        cur = 0
        for l in range(L):
            for s in range(S):
                for h in range(H):
                    n = int(nblk[l, s, h])
                    bt[l, s, h, :n] = perm[cur:cur + n]
                    cur += n
        B = len(sel)
        freed_cnt = np.zeros((B, L, H), dtype=np.int32)
        for b, s in enumerate(sel):
            for l in range(L):
                for h in range(H):
                    freed_cnt[b, l, h] = rng.integers(0, int(nblk[l, s, h]) + 1)
        state = blk.BlockState.__new__(blk.BlockState)
        state.block_size = bs
        state.use_tiered_block_tables = False
        state.block_tables = torch.from_numpy(bt.copy())
        state.t2_block_tables = None
        state.context_lens = torch.from_numpy(ctx.copy())
        state.block_table_indices = torch.arange(M)[None, None, None]
        view = state.get_block_state_batch_view(sel)
        removed = [torch.from_numpy(freed_cnt[b].copy()) for b in range(B)]       # per seq [L,H]
        freed_blocks, _ = view.get_last_n_allocated_blocks(torch.stack(removed, dim=1))
        state.remove_trailing_blocks(seq_indices=torch.tensor(sel, dtype=torch.long),
                                     removed_block_count=removed)
        np.savez_compressed(
            os.path.join(out_dir, f"blockstate_{case}.npz"), block_size=np.int32(bs),
            block_tables=bt, context_lens=ctx, seq_indices=np.asarray(sel, np.int32),
            freed_block_count=freed_cnt, num_blocks=np.int32(NB),
            ref_freed_blocks=freed_blocks.numpy().astype(np.int32),
            ref_context_lens=state.context_lens.numpy().astype(np.int32))
        print(f"blockstate_{case}: freed {freed_blocks.numel()} blocks")


if __name__ == "__main__":
    main()
