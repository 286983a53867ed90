"""Child process of tests/test_gpu_dispatch_bindings.py::test_writes_of_the_compiled_binding_are_seen:
under ``register("compiled")`` the dispatcher ops are C++ kernels that write through raw pointers.  Everything
the package keeps between calls -- harvested candidate lists, the tracked move table's dirty map, the plan a
move list brings along -- trusts tensor version counters, so a write through ``torch.ops._C_cache_ops.*`` /
``torch.ops._C_kvc_ops.*`` must be as visible as a write through torch.  Each scenario makes kept state, writes
through the compiled op in a way that kept state does not cover, and checks the next call against the oracle."""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import kvc_oracle as orc                                   # noqa: E402
from tests.helpers import oracle_pipeline                              # noqa: E402
from vllm_kvcompress_amd import _custom_ops as ops                     # noqa: E402
from vllm_kvcompress_amd import torch_ops                              # noqa: E402
from vllm_kvcompress_amd.harness import device as hdev, synth         # noqa: E402

DEV = "cuda:0"
BS, HD = 16, 128


def _steady(seed):
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=BS, seq_lens=[700, 420, 555], seed=seed,
                          protected=BS + 1, steady_cap=320, spare_block_frac=0.3)
    return st


def _harvested(st, ds, rng, evicted):
    """two decode steps; the second one's aggregation pass makes the lists.  Returns the arguments of the schedule call"""
    cm = ds.cm
    seqs, prot = list(st.seq_indices), list(st.protected)
    args = (seqs, ds.seq_positions, evicted, ds.context_lens, ds.hanging_token_count, ds.evicted_kv_offsets, prot)
    for step in range(2):
        temp = rng.random((st.num_blocks, BS, 4)).astype(np.float32)
        cm.temp_metrics.copy_(torch.from_numpy(temp))
        orc.aggregate_decode(st.metrics, temp, use_l2=True)
        made = cm.aggregate_decode_and_harvest(seqs, ds.seq_positions, prot, ds.context_lens, total_slots=st.total_slots)
        assert made == (step == 1)
        if step == 0:
            cm.schedule_evictions(*args, total_slots=st.total_slots)      # leaves the pivots
            assert cm.last_schedule_path() == "small_eviction"
    assert np.array_equal(cm.metrics.cpu().numpy(), st.metrics)
    return args


def _high_evictable_slot(st, b=0, l=1, h=2):
    """an evictable slot of head (b, l, h) whose metric is the head's LARGEST: far above any pivot, so lists made
    before a write that lowers it cannot contain it"""
    blocks = np.nonzero((st.seq_index_by_block == st.seq_indices[b]) & (st.layer_index_by_block == l)
                        & (st.head_index_by_block == h))[0]
    ok = st.token_positions[blocks] <= int(st.seq_positions[b]) - int(st.protected[b])
    ok &= st.logical_block_num_by_block[blocks][:, None] * BS + np.arange(BS)[None, :] < st.context_lens[l, b, h]
    m = np.where(ok, st.metrics[blocks], -np.inf)
    i = int(np.argmax(m))
    return int(blocks[i // BS]) * BS + i % BS


def harvest_then_reshape_and_cache(res):
    """kv_metrics[slot] = bias through torch.ops._C_cache_ops.kvcompress_reshape_and_cache (the fork's schema does not
    even mark kv_metrics mutable, csrc/torch_bindings.cpp:353-360) between the harvest and its schedule call"""
    st = _steady(3)
    ds = hdev.upload(st, DEV, num_queries_per_kv=4, mode="per_sequence")
    cm = ds.cm
    cm.strict_fallback = True
    rng = np.random.default_rng(0)
    evicted = [8, 8, 8]
    args = _harvested(st, ds, rng, evicted)
    slot = _high_evictable_slot(st)
    key = torch.zeros((1, 4, HD), dtype=torch.float16, device=DEV)
    kc = torch.zeros((st.num_blocks, HD // 8, BS, 8), dtype=torch.float16, device=DEV)
    vc = torch.zeros((st.num_blocks, HD, BS), dtype=torch.float16, device=DEV)
    slots = torch.full((4,), -1, dtype=torch.int64, device=DEV)
    slots[2] = slot                                             # (token 0, head 2)
    bias = torch.full((4,), -1.0e9, device=DEV)                 # the smallest metric of its head from now on
    v0 = cm.metrics._version
    torch.ops._C_cache_ops.kvcompress_reshape_and_cache(key, key, kc, vc, cm.metrics, slots, bias, "auto", 1.0, 1.0)
    assert cm.metrics._version > v0, "the compiled op wrote kv_metrics without bumping its version counter"
    st.metrics.reshape(-1)[slot] = np.float32(-1.0e9)
    want = oracle_pipeline(st, evicted, mode="per_sequence")
    eli, ekc, ebc = cm.schedule_evictions(*args, total_slots=st.total_slots)
    assert not cm.last_harvest_used, "lists made before the write were trusted"
    for got, k in ((eli, "eli"), (ekc, "ekc"), (ebc, "ebc")):
        assert np.array_equal(got.cpu().numpy(), want[k]), f"reshape_and_cache scenario: {k}"
    res["harvest_then_reshape_and_cache"] = "lists dropped, oracle's schedule"


def harvest_then_execute_cache_moves(res):
    """one move through torch.ops._C_kvc_ops.execute_cache_moves between the harvest and its schedule call: it copies a
    tiny metric (and its position) over the head's largest"""
    st = _steady(5)
    ds = hdev.upload(st, DEV, num_queries_per_kv=4, mode="per_sequence")
    cm = ds.cm
    cm.strict_fallback = True
    rng = np.random.default_rng(1)
    evicted = [8, 8, 8]
    args = _harvested(st, ds, rng, evicted)
    dst = _high_evictable_slot(st)
    # source: an unallocated block's slot carrying a tiny metric and an old position (evictable where it lands)
    free_blk = int(np.nonzero(st.seq_index_by_block < 0)[0][0])
    src = free_blk * BS + 3
    st.metrics.reshape(-1)[src] = np.float32(-5.0e8)
    st.token_positions.reshape(-1)[src] = st.token_positions.reshape(-1)[dst]
    cm.metrics.view(-1)[src] = -5.0e8                            # (a torch write in front of the harvest would be
    cm.token_positions.view(-1)[src] = int(st.token_positions.reshape(-1)[dst])   # seen anyway: redo the harvest)
    temp = np.zeros((st.num_blocks, BS, 4), np.float32)
    cm.temp_metrics.copy_(torch.from_numpy(temp))
    assert cm.aggregate_decode_and_harvest(args[0], args[1], args[6], args[3], total_slots=st.total_slots)
    G = 3 * 2 * 4
    cmi = torch.zeros((st.total_slots, 2), dtype=torch.int32, device=DEV)
    cmc = torch.zeros((3, 2, 4), dtype=torch.int32, device=DEV)
    g = (0 * 2 + 1) * 4 + 2
    off = int(st.evicted_kv_offsets.reshape(-1)[g])
    cmi[off, 0], cmi[off, 1] = dst, src
    cmc.view(-1)[g] = 1
    k = torch.zeros((st.num_blocks, HD // 8, BS, 8), dtype=torch.float16, device=DEV)
    v = torch.zeros((st.num_blocks, HD, BS), dtype=torch.float16, device=DEV)
    vm, vp = cm.metrics._version, cm.token_positions._version
    torch.ops._C_kvc_ops.execute_cache_moves(k, v, cm.metrics, cm.token_positions, cmi, cmc, ds.evicted_kv_offsets, 1, 16)
    assert cm.metrics._version > vm and cm.token_positions._version > vp, "execute_cache_moves did not bump the counters"
    st.metrics.reshape(-1)[dst] = st.metrics.reshape(-1)[src]
    want = oracle_pipeline(st, evicted, mode="per_sequence")
    eli, ekc, ebc = cm.schedule_evictions(*args, total_slots=st.total_slots)
    assert not cm.last_harvest_used, "lists made before the move were trusted"
    for got, kk in ((eli, "eli"), (ekc, "ekc"), (ebc, "ebc")):
        assert np.array_equal(got.cpu().numpy(), want[kk]), f"execute_cache_moves scenario: {kk}"
    res["harvest_then_execute_cache_moves"] = "lists dropped, oracle's schedule"


def tracked_table_and_plan(res):
    """the bare compiled schedule_t1_cache_moves writes rows into a tracked table (rows its dirty map does not know) and
    over a move list whose plan the Python wrapper remembers: the next wrapper call clears the whole table, and
    execute_cache_moves plans for itself"""
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=BS, seq_lens=[900, 350, 1300], seed=12,
                          protected=BS + 1, spare_block_frac=0.2)
    ds = hdev.upload(st, DEV, mode="per_sequence")
    nblk = ((st.context_lens.astype(np.int64) + BS - 1) // BS).sum(0).sum(-1)
    small, big = [1, 0, 2], [int(n * 0.5) for n in nblk]
    rows = st.total_slots + 100
    table = ops.track_move_table(torch.full((rows, 2), 5, dtype=torch.int32, device=DEV))
    sched = lambda ev: ds.cm.schedule_evictions(list(st.seq_indices), ds.seq_positions, ev, ds.context_lens,
                                                ds.hanging_token_count, ds.evicted_kv_offsets, list(st.protected),
                                                total_slots=st.total_slots)
    tail = (ds.evicted_kv_offsets, ds.block_tables, ds.context_lens, BS)
    eli_s, ekc_s, _ = sched(small)
    eli_s, ekc_s = eli_s.clone(), ekc_s.clone()
    eli_b, ekc_b, _ = sched(big)
    cmc = torch.zeros_like(ekc_s)
    ops.schedule_cache_moves(table, cmc, eli_s, ekc_s, *tail)                 # sets the map up (full fill)
    ops.schedule_cache_moves(table, cmc, eli_s, ekc_s, *tail)                 # runs on the map
    assert ops._tracked(table).version == table._version
    v0 = table._version
    torch.ops._C_kvc_ops.schedule_t1_cache_moves(table, cmc, eli_b, ekc_b, *tail)     # many rows the map does not know
    assert table._version > v0, "the compiled op wrote the move table without bumping its version counter"
    ops.schedule_cache_moves(table, cmc, eli_s, ekc_s, *tail)
    want = oracle_pipeline(st, small, mode="per_sequence")
    expect = np.zeros((rows, 2), np.int32)
    expect[:st.total_slots] = want["cmi"]
    bad = np.nonzero((table.cpu().numpy() != expect).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} rows of the tracked table are stale after the compiled op wrote to it"
    # ---- the plan: the wrapper's plan belongs to (table, cmc, offsets) as the wrapper left them
    k_np, v_np = synth.make_caches_u16(2, st.num_blocks, HD, BS)
    assert ops._plan_of(torch.empty(1, device=DEV), table, cmc, ds.evicted_kv_offsets, cmc.numel(), BS) is not None
    torch.ops._C_kvc_ops.schedule_t1_cache_moves(table, cmc, eli_b, ekc_b, *tail)     # another list in the same tensors
    assert ops._plan_of(torch.empty(1, device=DEV), table, cmc, ds.evicted_kv_offsets, cmc.numel(), BS) is None, \
        "a plan made for another move list still vouches for the tensors the compiled op rewrote"
    want = oracle_pipeline(st, big, k_np, v_np, mode="per_sequence")
    k, v = torch.from_numpy(k_np.copy()).to(DEV), torch.from_numpy(v_np.copy()).to(DEV)
    m, p = ds.cm.metrics.clone(), ds.cm.token_positions.clone()
    ops.execute_cache_moves(k, v, m, p, table, cmc, ds.evicted_kv_offsets, 1, 16)
    assert np.array_equal(k.cpu().numpy(), want["k"]) and np.array_equal(v.cpu().numpy(), want["v"])
    assert np.array_equal(m.cpu().numpy(), want["metrics"]) and np.array_equal(p.cpu().numpy(), want["positions"])
    # ... and the compiled pair keeps its own plan: schedule_t1_cache_moves -> execute_cache_moves is one launch
    before = int(torch.ops._kvc_mi355x.planned_compactions())
    k, v = torch.from_numpy(k_np.copy()).to(DEV), torch.from_numpy(v_np.copy()).to(DEV)
    m, p = ds.cm.metrics.clone(), ds.cm.token_positions.clone()
    torch.ops._C_kvc_ops.execute_cache_moves(k, v, m, p, table, cmc, ds.evicted_kv_offsets, 1, 16)
    assert int(torch.ops._kvc_mi355x.planned_compactions()) == before + 1
    assert np.array_equal(k.cpu().numpy(), want["k"]) and np.array_equal(m.cpu().numpy(), want["metrics"])
    res["tracked_table_and_plan"] = "full fill after the compiled write; stale plan refused; compiled pair planned"


def outputs_are_counted(res):
    st = synth.make_state(num_layers=2, num_kv_heads=2, block_size=BS, seq_lens=[200, 90], seed=2, protected=16)
    ds = hdev.upload(st, DEV)
    eli, ekc, ebc = ds.cm.schedule_evictions(list(st.seq_indices), ds.seq_positions, [3, 1], ds.context_lens,
                                             ds.hanging_token_count, ds.evicted_kv_offsets, list(st.protected))
    cnt = torch.empty_like(ebc)
    flat = eli.clone()
    v = (cnt._version, flat._version)
    torch.ops._C_kvc_ops.count_block_evictions(cnt, flat, ds.evicted_kv_offsets, ds.hanging_token_count, BS, 2147483000)
    assert cnt._version > v[0] and flat._version > v[1]
    res["outputs_are_counted"] = True


def main():
    assert torch_ops.register("compiled") == "compiled"
    res = {}
    harvest_then_reshape_and_cache(res)
    harvest_then_execute_cache_moves(res)
    tracked_table_and_plan(res)
    outputs_are_counted(res)
    print("COMPILED_WRITES " + json.dumps(res))


if __name__ == "__main__":
    main()
