"""schedule_cache_moves on a move table the package tracks, and execute_cache_moves on the plan it
leaves behind.

The reference clears the whole ``[max_kv_per_compression, 2]`` table in front of every
``schedule_t1_cache_moves`` (vllm/_custom_ops.py:1158-1179): afterwards every row outside
``[off_g, off_g + count_g)`` is zero.  A tracked table (``ops.track_move_table`` -- what
``CompressionScheduler.cache_move_indices`` is, reference scheduler.py:74-86) must hold EXACTLY that
after every call although only the rows the previous call wrote are cleared; compared here row for
row with the oracle over many calls with changing batch composition, eviction sizes, foreign writes
in between and tables that are larger / smaller than the batch.

``execute_cache_moves`` of a list that ``schedule_cache_moves`` just made runs as one launch on the
plan that call left behind; the same list through the self-planning op (any other provenance) must
give the same bytes."""
import numpy as np
import pytest
import torch

from tests.helpers import oracle_pipeline
from vllm_kvcompress_amd import _custom_ops as ops
from vllm_kvcompress_amd.harness import device as hdev
from vllm_kvcompress_amd.harness import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sub_state(st, sel):
    bs = st.block_size
    ctx = np.ascontiguousarray(st.context_lens[:, sel, :])
    return synth.PagedState(
        block_size=bs, num_layers=st.num_layers, num_kv_heads=st.num_kv_heads, num_seqs=len(sel),
        num_blocks=st.num_blocks, metrics=st.metrics, token_positions=st.token_positions,
        seq_index_by_block=st.seq_index_by_block, layer_index_by_block=st.layer_index_by_block,
        head_index_by_block=st.head_index_by_block, logical_block_num_by_block=st.logical_block_num_by_block,
        context_lens=ctx, block_tables=np.ascontiguousarray(st.block_tables[:, sel]),
        hanging_token_count=synth.hanging_tokens(ctx.transpose(1, 0, 2), bs),
        evicted_kv_offsets=synth.kv_offsets(ctx, bs), seq_indices=list(sel),
        seq_positions=np.ascontiguousarray(st.seq_positions[sel]), protected=[st.protected[i] for i in sel])


@pytest.mark.parametrize("bs,rows_mode", [(16, "large"), (32, "large"), (16, "odd"), (8, "tight"), (16, "short")])
def test_tracked_table_holds_what_the_reference_table_holds(bs, rows_mode):
    L, H = 3, 4
    seq_lens = [700, 260, 1500, 90, 410, 1100]
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=9, protected=bs + 1,
                          spare_block_frac=0.1)
    ds = hdev.upload(st, DEV, mode="per_sequence")
    ctx_full, bt_full = ds.context_lens, ds.block_tables
    n_all = st.total_slots
    # ("short": a table smaller than the largest batch -- moves beyond its end are dropped, as in the reference's kernel)
    rows = {"large": n_all + 5000, "odd": n_all + 13, "tight": n_all, "short": n_all - 9001}[rows_mode]
    table = ops.track_move_table(torch.empty((rows, 2), dtype=torch.int32, device=DEV))
    table.fill_(-3)                                      # junk: the first call must clear all of it
    rng = np.random.default_rng(bs)
    B = len(seq_lens)
    modes_seen = set()
    for step in range(64):
        k = int(rng.integers(1, B + 1))
        sel = sorted(rng.choice(B, size=k, replace=False).tolist())
        if step == 20:
            sel = list(range(B))                          # everything, bulk: many moves per head
        sub = _sub_state(st, sel)
        nblk = ((sub.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)
        kind = step % 4
        if step == 20:
            evicted = [int(n * 0.6) for n in nblk]
        elif kind == 0:
            evicted = [int(rng.integers(0, 3)) for _ in sel]                 # the steady state: a block or two
        elif kind == 1:
            evicted = [int(n * rng.uniform(0.05, 0.5)) for n in nblk]        # bulk
        elif kind == 2:
            evicted = [0 for _ in sel]                                       # nothing at all
        else:
            evicted = [int(rng.integers(0, max(2, n // 8))) for n in nblk]
        want = oracle_pipeline(sub, evicted, mode="per_sequence")
        if step in (11, 37):                              # somebody else writes to the table through torch
            table[int(rng.integers(0, rows))] = 123
        if step in (25, 26):                              # ... or the bare op does (no fill: rows the map does not know)
            junk_cmc = torch.empty_like(ekc_prev)
            ops._schedule_t1_cache_moves(table, junk_cmc, *prev_args, zero_fill=False)
        sub_ds = hdev.DeviceState(cm=ds.cm, context_lens=ctx_full[:, sel].contiguous(),
                                  block_tables=bt_full[:, sel].contiguous(),
                                  hanging_token_count=torch.from_numpy(sub.hanging_token_count).to(DEV),
                                  evicted_kv_offsets=torch.from_numpy(sub.evicted_kv_offsets).to(DEV),
                                  seq_positions=torch.from_numpy(sub.seq_positions).to(DEV), total_slots=sub.total_slots)
        eli, ekc, ebc = ds.cm.schedule_evictions(sel, sub_ds.seq_positions, evicted, sub_ds.context_lens,
                                                 sub_ds.hanging_token_count, sub_ds.evicted_kv_offsets, sub.protected,
                                                 total_slots=sub.total_slots)
        cmc = torch.empty_like(ekc)
        ekc_prev, prev_args = ekc, (eli, ekc, sub_ds.evicted_kv_offsets, sub_ds.block_tables, sub_ds.context_lens, bs)
        rec = ops._tracked(table)
        modes_seen.add(2 if (rec.dirty_map is not None and rec.version == table._version and rec.block_size == bs) else 1)
        ops.schedule_cache_moves(table, cmc, eli, ekc, sub_ds.evicted_kv_offsets, sub_ds.block_tables,
                                 sub_ds.context_lens, bs)
        np.testing.assert_array_equal(cmc.cpu().numpy(), want["cmc"], err_msg=f"step {step}: counts")
        expect = np.zeros((rows, 2), np.int32)
        n = min(rows, sub.total_slots)
        expect[:n] = want["cmi"][:n]
        got = table.cpu().numpy()
        bad = np.nonzero((got != expect).any(axis=1))[0]
        assert bad.size == 0, f"step {step} (sel {sel}, evicted {evicted}): {bad.size} rows differ, first {bad[:5]}"
    assert modes_seen == {1, 2}


def test_an_untracked_table_gets_the_full_fill_every_time():
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=16, seq_lens=[600, 300], seed=3, protected=32)
    ds = hdev.upload(st, DEV, mode="per_sequence")
    evicted = [9, 4]
    want = oracle_pipeline(st, evicted, mode="per_sequence")
    for junk in (5, -1, 77):
        eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, st, evicted, move_rows=st.total_slots + 1000)
        cmi2 = torch.full_like(cmi, junk)
        ops.schedule_cache_moves(cmi2, cmc, eli, ekc, ds.evicted_kv_offsets, ds.block_tables, ds.context_lens, 16)
        got = cmi2.cpu().numpy()
        np.testing.assert_array_equal(got[:st.total_slots], want["cmi"])
        assert not got[st.total_slots:].any()


@pytest.mark.parametrize("dtype,bs,hd", [(torch.float16, 16, 128), (torch.uint8, 32, 128), (torch.float16, 16, 80)])
def test_planned_and_self_planned_compaction_agree(dtype, bs, hd):
    """(hd 80: the byte-wise generic kernel on the plan's 32-move tiles)"""
    L, H = 3, 4
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=[900, 350, 40, 1300], seed=12,
                          protected=bs + 1, spare_block_frac=0.2)
    ds = hdev.upload(st, DEV, mode="per_sequence")
    nblk = ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)
    e = torch.empty((), dtype=dtype).element_size()
    x = 16 // e
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    kv = torch.randint(0, 256, (2, st.num_blocks, bs * hd * e), dtype=torch.uint8, device=DEV, generator=g)
    for evicted in ([int(n * 0.5) for n in nblk], [1, 0, 1, 2], [0, 0, 0, 0], [int(nblk[0] * 0.9), 0, 0, 3]):
        eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, st, evicted)
        outs = []
        for how in ("planned", "copied list", "version bump"):
            k = kv[0].clone().view(dtype).view(st.num_blocks, hd // x, bs, x)
            v = kv[1].clone().view(dtype).view(st.num_blocks, hd, bs)
            m, p = ds.cm.metrics.clone(), ds.cm.token_positions.clone()
            a_cmi, a_cmc, a_offs = cmi, cmc, ds.evicted_kv_offsets
            if how == "copied list":
                a_cmi, a_cmc, a_offs = cmi.clone(), cmc.clone(), ds.evicted_kv_offsets.clone()
            hit = ops._plan_of(k, a_cmi, a_cmc, a_offs, cmc.numel(), bs) is not None
            assert hit == (how == "planned"), how
            ops.execute_cache_moves(k, v, m, p, a_cmi, a_cmc, a_offs, 1, 16)
            outs.append((k.view(torch.uint8).clone(), v.view(torch.uint8).clone(), m, p))
            if how == "copied list":
                cmc.add_(0)                              # touched through torch: the plan no longer vouches for it
        for o in outs[1:]:
            for a, b in zip(outs[0], o):
                assert torch.equal(a, b)
        # ... and both are what the oracle's serial kernel makes of the list
        want_cmc = oracle_pipeline(st, evicted, mode="per_sequence")["cmc"]
        np.testing.assert_array_equal(cmc.cpu().numpy(), want_cmc)


def test_compression_scheduler_workspace_is_tracked_and_exact_over_steps():
    """CompressionScheduler's persistent workspace through schedule_compression, 50 calls with a
    changing set of compressing sequences: the whole workspace equals zeros + the oracle's moves"""
    from vllm_kvcompress_amd.kvcompress.scheduler import CompressionScheduler, SeqCompressionRequest
    L, H, bs = 2, 4, 16
    seq_lens = [500, 230, 177, 361, 90, 640]
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=21, protected=32,
                          spare_block_frac=0.3)
    nblk = ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)
    kvs = st.context_lens.astype(np.int64).sum(0).sum(-1)
    rows = 60000
    rng = np.random.default_rng(2)
    ds = hdev.upload(st, DEV, mode="per_sequence")
    sched = CompressionScheduler(bs, L, H, rows, ds.cm, device=DEV)
    assert ops._tracked(sched.cache_move_indices) is not None
    for step in range(50):
        # fresh copies of the block state every call: this test is about the move workspace
        bt = torch.from_numpy(st.block_tables).to(DEV)
        ctx = torch.from_numpy(st.context_lens.copy()).to(DEV)
        ds.cm.seq_index_by_block.copy_(torch.from_numpy(st.seq_index_by_block))
        k = int(rng.integers(1, len(seq_lens) + 1))
        sel = sorted(rng.choice(len(seq_lens), size=k, replace=False).tolist())
        caps = {i: int(rng.choice([48, 96, 160, 10 ** 6])) for i in sel}
        reqs = [SeqCompressionRequest(seq_id=100 + i, slot_index=i, seq_len=seq_lens[i], block_count=int(nblk[i]),
                                      kv_count=int(kvs[i]), max_cache_tokens=caps[i], protected_window_size=32)
                for i in sel]
        out = sched.schedule_compression(reqs, bt, ctx, force=True)
        got = sched.cache_move_indices.cpu().numpy()
        if out is None:
            continue
        sub = _sub_state(st, out.slot_indices)
        evicted = [synth.evict_block_count(context_lens_lh=st.context_lens[:, i, :], seq_len=seq_lens[i], block_size=bs,
                                           protected_window_size=32, max_cache_tokens=caps[i]) for i in out.slot_indices]
        want = oracle_pipeline(sub, evicted, mode="per_sequence")
        expect = np.zeros((rows, 2), np.int32)
        expect[:sub.total_slots] = want["cmi"]
        np.testing.assert_array_equal(out.cache_moves.count.cpu().numpy(), want["cmc"], err_msg=f"step {step}")
        bad = np.nonzero((got != expect).any(axis=1))[0]
        assert bad.size == 0, f"step {step}: {bad.size} rows differ, first {bad[:5]}"


@pytest.mark.parametrize("protected", [33, 0])
def test_move_scheduler_by_eviction_count(protected):
    """the three ways a head is walked (a lane up to 32 evictions, a wave up to 2048, the workgroup
    beyond) at their borders, regular heads and the protected = 0 quirk (an empty tail slot
    evictable), one head and several per workgroup, both instantiations"""
    bs = 16
    for L, H, seq_len in ((1, 1, 5001), (1, 1, 5008), (1, 1, 5016), (2, 3, 5001), (1, 2, 40001)):
        st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=[seq_len], seed=seq_len + L,
                              protected=protected, spare_block_frac=0.05)
        ds = hdev.upload(st, DEV, mode="per_sequence")
        nblk = int(((st.context_lens.astype(np.int64) + bs - 1) // bs).sum())
        for k in (1, 2, 3, 4, 5, 6, 127, 128, 129, 130, 131, 200, 257, nblk // 2, nblk - 3 * L * H):
            k = int(k) * (L * H if k < 300 else 1)
            if k <= 0 or k > nblk:
                continue
            want = oracle_pipeline(st, [k], mode="per_sequence")
            eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, st, [k])
            np.testing.assert_array_equal(ekc.cpu().numpy(), want["ekc"], err_msg=f"{seq_len} {k}")
            np.testing.assert_array_equal(cmc.cpu().numpy(), want["cmc"], err_msg=f"{seq_len} {k} counts")
            np.testing.assert_array_equal(cmi.cpu().numpy(), want["cmi"], err_msg=f"{seq_len} {k} moves")
        counts = sorted(set(int(c) for c in want["ekc"].reshape(-1)))
        assert counts, counts


def test_under_inference_mode_nothing_is_assumed_about_tensors_without_a_version_counter():
    """vLLM's workers run under torch.inference_mode(): tensors made there keep no version counter, so neither the
    tracked table's map nor the plan can be trusted for them -- the ops take the full fill / plan for themselves,
    and the results are the oracle's"""
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=16, seq_lens=[600, 300], seed=3, protected=32)
    evicted = [9, 4]
    want = oracle_pipeline(st, evicted, mode="per_sequence")
    with torch.inference_mode():
        ds = hdev.upload(st, DEV, mode="per_sequence")
        table = ops.track_move_table(torch.empty((st.total_slots + 77, 2), dtype=torch.int32, device=DEV))
        assert table.is_inference()
        k_np, v_np = synth.make_caches_u16(3, st.num_blocks, 32, 16)
        for rep in range(3):
            table.fill_(-5)                                   # (nobody can tell: the next call must clear all of it)
            eli, ekc, ebc = ds.cm.schedule_evictions(list(st.seq_indices), ds.seq_positions, evicted, ds.context_lens,
                                                     ds.hanging_token_count, ds.evicted_kv_offsets, list(st.protected),
                                                     total_slots=st.total_slots)
            cmc = torch.empty_like(ekc)
            ops.schedule_cache_moves(table, cmc, eli, ekc, ds.evicted_kv_offsets, ds.block_tables, ds.context_lens, 16)
            got = table.cpu().numpy()
            np.testing.assert_array_equal(got[:st.total_slots], want["cmi"])
            assert not got[st.total_slots:].any()
            np.testing.assert_array_equal(eli.cpu().numpy(), want["eli"])
            k, v = torch.from_numpy(k_np.copy()).to(DEV), torch.from_numpy(v_np.copy()).to(DEV)
            m, p = ds.cm.metrics.clone(), ds.cm.token_positions.clone()
            ops.execute_cache_moves(k, v, m, p, table, cmc, ds.evicted_kv_offsets, 1, 16)
            w = oracle_pipeline(st, evicted, k_np, v_np, mode="per_sequence")
            np.testing.assert_array_equal(k.cpu().numpy(), w["k"])
            np.testing.assert_array_equal(m.cpu().numpy(), w["metrics"])
