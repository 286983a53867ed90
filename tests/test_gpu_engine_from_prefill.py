"""The all-device engine loop from PREFILL on (vllm_kvcompress_amd/harness/engine_device.py: add_sequence ->
reshape_and_cache -> aggregate_prefill, then per iteration schedule_evictions -> schedule_cache_moves ->
execute_cache_moves -> free_compressed_blocks -> append_slots -> reshape_and_cache -> aggregate_decode, no NumPy
state on the device side) next to the oracle's restatements of the same transitions (each pinned to fixtures of the
reference's own code: tests/test_oracle_golden.py).  ALL state is compared after every transition, bit for bit.

Also: the device-side prefill allocation (kvc_add_sequence) against the fixtures produced by the reference's own
_add_sequence / ParallelBlockAllocator / get_allocated_block_metadata / insert_metadata / get_prefill_slot_mapping
(oracle/gen_golden_prefill_alloc.py)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import kvc_oracle as orc
from oracle.engine_oracle import OracleEngine
from tests.helpers import GOLDEN_DIR, load_golden
from vllm_kvcompress_amd import _lib
from vllm_kvcompress_amd.harness import synth
from vllm_kvcompress_amd.harness.engine_device import DeviceEngine
from vllm_kvcompress_amd.kvcompress.block_state import add_sequence

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith("prefill_alloc_"))
KEYS = (("block_tables", "ref_block_tables"), ("context_lens", "ref_context_lens"), ("free_mask", "ref_free_mask"),
        ("seq_index_by_block", "ref_seq_index_by_block"), ("layer_index_by_block", "ref_layer_index_by_block"),
        ("head_index_by_block", "ref_head_index_by_block"), ("logical_block_num_by_block", "ref_logical_block_num_by_block"),
        ("token_positions", "ref_token_positions"))


@pytest.mark.parametrize("name", CASES)
def test_add_sequence_device(name):
    g = load_golden(name)
    t = {k: torch.from_numpy(g[k].copy()).to(DEV) for k, _ in KEYS}
    cm = SimpleNamespace(**{k: t[k] for k in ("seq_index_by_block", "layer_index_by_block", "head_index_by_block",
                                              "logical_block_num_by_block", "token_positions")}, _hv_lists=1)
    n, sm = add_sequence(t["block_tables"], t["context_lens"], int(g["seq_slot"]), int(g["seq_len"]), t["free_mask"],
                         cm, int(g["block_size"]))
    assert n == int(g["free_mask"].sum()) - int(g["ref_free_count"]) and cm._hv_lists is None
    for k, r in KEYS:
        np.testing.assert_array_equal(t[k].cpu().numpy(), g[r], err_msg=k)
    np.testing.assert_array_equal(sm.cpu().numpy(), g["ref_slot_mapping"])
    assert sm.dtype == torch.int64
    # out of blocks: ValueError like ParallelBlockAllocator.allocate (block_manager.py:104-106), nothing modified
    t2 = {k: torch.from_numpy(g[k].copy()).to(DEV) for k, _ in KEYS}
    fm = t2["free_mask"]
    fm[torch.nonzero(fm).view(-1)[max(n - 1, 0):]] = False
    before = {k: v.clone() for k, v in t2.items()}
    cm2 = SimpleNamespace(**{k: t2[k] for k in ("seq_index_by_block", "layer_index_by_block", "head_index_by_block",
                                               "logical_block_num_by_block", "token_positions")}, _hv_lists=None)
    with pytest.raises(ValueError, match="Out of memory"):
        add_sequence(t2["block_tables"], t2["context_lens"], int(g["seq_slot"]), int(g["seq_len"]), fm, cm2,
                     int(g["block_size"]))
    for k in before:
        assert torch.equal(t2[k], before[k]), k
    # a sequence longer than the block table: refused, nothing modified
    M = t2["block_tables"].shape[3]
    with pytest.raises(RuntimeError, match="block table"):
        add_sequence(t2["block_tables"], t2["context_lens"], int(g["seq_slot"]), (M + 1) * int(g["block_size"]),
                     torch.ones(10 ** 6, dtype=torch.bool, device=DEV), cm2, int(g["block_size"]), slot_mapping=False)
    for k in before:
        if k != "free_mask":
            assert torch.equal(t2[k], before[k]), k


def _same_state(dev: DeviceEngine, o: OracleEngine, where):
    M = o.bt.shape[3]
    np.testing.assert_array_equal(dev.context_lens.cpu().numpy(), o.ctx, err_msg=f"{where}: context_lens")
    live = np.arange(M)[None, None, None, :] < ((o.ctx + o.bs - 1) // o.bs)[..., None]
    assert np.array_equal(dev.block_tables.cpu().numpy()[live], o.bt[live]), f"{where}: block_tables"
    cm = dev.cm
    for name, got, want in (("free_mask", dev.free_mask, o.free), ("metrics", cm.metrics, o.metrics),
                            ("seq_index", cm.seq_index_by_block, o.seq), ("K", dev.k_cache.view(torch.int16), o.k.view(np.int16)),
                            ("V", dev.v_cache.view(torch.int16), o.v.view(np.int16))):
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f"{where}: {name}")
    alloc = o.seq >= 0                       # rows of blocks that belong to somebody (freed blocks keep stale rows)
    for name, got, want in (("positions", cm.token_positions, o.pos), ("layer", cm.layer_index_by_block, o.lay),
                            ("head", cm.head_index_by_block, o.head), ("lbn", cm.logical_block_num_by_block, o.lbn)):
        assert np.array_equal(got.cpu().numpy()[alloc], want[alloc]), f"{where}: {name}"


@pytest.mark.parametrize("layout", ["reference", "slot_major"])
@pytest.mark.parametrize("mode", ["per_sequence", "reference"])
def test_engine_from_prefill_all_state_on_device(mode, layout):
    """three sequences arrive (one of them later), decode and are compressed every iteration back to a cap; one leaves
    and its slot is reused.  The slot-major run permutes the device cache back before comparing K / V."""
    L, H, hd, bs, qpk, cap, prot = 2, 4, 128, 16, 4, 64, 20
    NB, S, M = 420, 4, 24
    rng = np.random.default_rng(11)
    _lib.set_block_layout(layout)
    try:
        dev = DeviceEngine(num_layers=L, num_kv_heads=H, head_size=hd, block_size=bs, num_blocks=NB, max_num_seqs=S,
                           max_blocks_per_head=M, num_queries_per_kv=qpk, mode=mode, protected_window=prot,
                           max_cache_tokens=cap)
        o = OracleEngine(L, H, hd, bs, NB, S, M, qpk, prot, cap, mode)

        def same(where):
            if layout == "slot_major":      # the oracle computes in the reference's layout
                from vllm_kvcompress_amd.layout import convert_block_layout
                convert_block_layout(dev.k_cache, dev.v_cache, "slot_major", "reference")
                _same_state(dev, o, where)
                convert_block_layout(dev.k_cache, dev.v_cache, "reference", "slot_major")
            else:
                _same_state(dev, o, where)

        def arrive(slot, T):
            key = rng.standard_normal((L, T, H, hd)).astype(np.float16)
            val = rng.standard_normal((L, T, H, hd)).astype(np.float16)
            pm = rng.random((L, T, H * qpk)).astype(np.float32)
            sm_o = o.add_sequence(slot, key, val, pm)
            dev.add_sequence(slot, torch.from_numpy(key).to(DEV), torch.from_numpy(val).to(DEV), torch.from_numpy(pm).to(DEV))
            np.testing.assert_array_equal(dev.last["slot_mapping"].cpu().numpy(), sm_o)
            same(f"prefill of slot {slot}")

        arrive(0, 150)
        arrive(2, 97)
        compressions = 0
        for it in range(36):
            if it == 7:
                arrive(1, 64)
            if it == 20:                     # a sequence finishes; its slot is taken by a new one two iterations on
                dev.remove_sequence(2)
                o.remove_sequence(2)
                same(f"iteration {it}: slot 2 left")
            if it == 22:
                arrive(2, 33)
            r_o = o.compress()
            r_d = dev.compress()
            assert (r_o is None) == (r_d is None), f"iteration {it}"
            if r_o is not None:
                compressions += 1
                for k in ("eli", "ekc", "ebc", "cmc", "freed"):
                    np.testing.assert_array_equal(r_d[k].cpu().numpy(), r_o[k], err_msg=f"iteration {it}: {k}")
                np.testing.assert_array_equal(r_d["cmi"].cpu().numpy(), r_o["cmi"], err_msg=f"iteration {it}: cmi")
                same(f"iteration {it}: compression")
            B = len(o.slots)
            key = rng.standard_normal((L, B, H, hd)).astype(np.float16)
            val = rng.standard_normal((L, B, H, hd)).astype(np.float16)
            temp = rng.random((NB, bs, qpk)).astype(np.float32)
            n_o = o.decode(key, val, temp)
            n_d = dev.decode(torch.from_numpy(key).to(DEV), torch.from_numpy(val).to(DEV), torch.from_numpy(temp).to(DEV))
            assert n_o == n_d, f"iteration {it}"
            same(f"iteration {it}: decode")
        assert compressions >= 25, compressions
        # (the reference's batch > 1 rule hands a sequence fewer evictions than it asked for when another one's
        # heads hold few evictable keys -- metrics.py:718's count -- so only the per-sequence mode holds the cap)
        if mode == "per_sequence":
            assert int(o.ctx.max()) <= cap + bs
    finally:
        _lib.set_block_layout("reference")
