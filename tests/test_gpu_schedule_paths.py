"""schedule_evictions has two schedules behind one entry point (kvc_schedule_params
.schedule_path / .max_evicted_blocks_hint): the general radix-select pipeline and the
small-eviction schedule of the continual-compression steady state (the metric store streamed once in
physical order, per-head records of the keys below a sampled pivot), which raises a device flag and lets the general pipeline redo the work when it cannot
finish exactly.  Both must give the oracle's result bit for bit; these tests also pin WHICH one
produced it."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import oracle_pipeline
from vllm_kvcompress_amd.harness import device as hdev
from vllm_kvcompress_amd.harness import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KEYS = ("eli", "ekc", "ebc", "cmi", "cmc")


def _run(st, evicted, path, mode="reference", lean=False, **kw):
    ds = hdev.upload(st, DEV, mode=mode, **kw)
    ds.cm.schedule_path = path
    ds.cm.lean_outputs = lean
    eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, st, evicted)
    out = dict(eli=eli.cpu().numpy(), ekc=ekc.cpu().numpy(), ebc=ebc.cpu().numpy(),
               cmi=cmi.cpu().numpy(), cmc=cmc.cpu().numpy())
    return out, ds.cm.last_schedule_path()


def _steady(L, H, bs, B, cap, seed, spare_block_frac=0.05, **kw):
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=[3 * cap] * B, seed=seed,
                          protected=bs + 1, steady_cap=cap, spare_block_frac=spare_block_frac, **kw)
    evicted = [synth.evict_block_count(context_lens_lh=st.context_lens[:, b, :], seq_len=3 * cap,
                                       block_size=bs, protected_window_size=bs + 1, max_cache_tokens=cap)
               for b in range(B)]
    return st, evicted


@pytest.mark.parametrize("name", ["b2_bs16_steady256", "b3_bs32_steady512", "b1_bs8_steady128"])
def test_reference_generated_steady_states(name):
    """steady-state fixtures produced by the reference's own CompressionMetrics.schedule_evictions
    and move twins (oracle/gen_golden.py): the small-eviction schedule is the one that runs, and
    both schedules reproduce the reference bit for bit (reference mode, batch>1 quirk included)"""
    from tests.helpers import load_golden
    from tests.test_gpu_parity import _state_from_golden
    g = load_golden(name)
    st = _state_from_golden(g)
    evicted = [int(x) for x in g["evicted_blocks_per_seq"]]
    for path, how_want in ((0, "small_eviction"), (1, "general")):
        out, how = _run(st, evicted, path, "reference")
        assert how == how_want
        np.testing.assert_array_equal(out["eli"], g["ref_evicted_logical_indices"])
        np.testing.assert_array_equal(out["ekc"], g["ref_evicted_kv_count"])
        np.testing.assert_array_equal(out["ebc"], g["ref_evicted_block_count"])
        np.testing.assert_array_equal(out["cmi"], g["ref_cache_moves_idx"])
        np.testing.assert_array_equal(out["cmc"], g["ref_cache_moves_count"])


@pytest.mark.parametrize("mode", ["reference", "per_sequence"])
@pytest.mark.parametrize("L,H,bs,B,cap", [(4, 4, 16, 3, 512), (2, 8, 32, 2, 1024), (3, 2, 8, 2, 256),
                                          (2, 2, 16, 1, 4096)])
def test_steady_state_takes_the_small_eviction_schedule(L, H, bs, B, cap, mode):
    for seed in (0, 1):
        st, evicted = _steady(L, H, bs, B, cap, seed)
        want = oracle_pipeline(st, evicted, mode=mode)
        auto, how_auto = _run(st, evicted, 0, mode)
        gen, how_gen = _run(st, evicted, 1, mode)
        assert how_auto == "small_eviction" and how_gen == "general"
        for key in KEYS:
            np.testing.assert_array_equal(auto[key], want[key], err_msg=f"{key} small-eviction seed={seed}")
            np.testing.assert_array_equal(gen[key], want[key], err_msg=f"{key} general seed={seed}")


def _fork_call_arguments(st, evicted):
    """the arguments of schedule_evictions built the way the fork's scheduler builds them
    (reference vllm/kvcompress/scheduler.py:234-280, 491-499): slot indices as a list, last token positions and the
    eviction counts as fresh device int tensors, context_lens / hanging counts / offsets derived on the device,
    protected windows as a tuple -- no N, no host copy of the counts, no block tables"""
    bs = st.block_size
    slot_indices = [int(s) for s in st.seq_indices]
    evicted_blocks_per_seq = torch.tensor([int(e) for e in evicted], dtype=torch.int, device=DEV)
    last_token_positions = torch.tensor([int(x) + 1 for x in st.seq_positions], dtype=torch.int, device=DEV) - 1
    context_lens = torch.from_numpy(st.context_lens).to(DEV).contiguous()                        # [L, B, H]
    hanging = torch.from_numpy(st.hanging_token_count).to(DEV).contiguous()                      # [B, L, H]
    offs = (((context_lens.transpose(0, 1) + bs - 1) // bs) * bs).flatten().cumsum(dim=0)
    offs = torch.cat([torch.zeros_like(offs[:1]), offs[:-1]]).reshape(*context_lens.transpose(0, 1).shape).type(torch.int)
    return (slot_indices, last_token_positions, evicted_blocks_per_seq, context_lens, hanging, offs, tuple(st.protected))


def test_the_forks_call_form_reaches_every_schedule():
    """The fork passes evicted_blocks_per_seq as a DEVICE int tensor and neither N nor its maximum
    (vllm/kvcompress/scheduler.py:245-247, 491-499).  That call must take the schedules the list form takes
    (one wait brings N and the counts to the host, CompressionMetrics._batch_summary) and give the oracle's result."""
    # continual steady state: small-eviction schedule, then -- same batch again -- on the pivots of the call before
    st, evicted = _steady(2, 4, 16, 2, 512, 3)
    want = oracle_pipeline(st, evicted, mode="reference")
    ds = hdev.upload(st, DEV)
    ds.cm.schedule_path = 0                 # (this test is about the automatic choice: KVC_SCHEDULE_PATH must not decide)
    for call in range(2):
        eli, ekc, ebc = ds.cm.schedule_evictions(*_fork_call_arguments(st, evicted))
        assert ds.cm.last_schedule_reason.startswith("small_eviction"), ds.cm.last_schedule_reason
        assert ds.cm.last_schedule_path() == "small_eviction"
        assert ds.cm.last_pivot_memory_used == (call == 1)
        assert tuple(eli.shape) == (st.total_slots,)
        for got, key in ((eli, "eli"), (ekc, "ekc"), (ebc, "ebc")):
            np.testing.assert_array_equal(got.cpu().numpy(), want[key], err_msg=f"{key} call {call}")
    # the list form (the reference's test harnesses) gives the same tensors
    args = _fork_call_arguments(st, evicted)
    b = ds.cm.schedule_evictions(args[0], args[1], evicted, *args[3:], total_slots=st.total_slots)
    for x, key in zip(b, ("eli", "ekc", "ebc")):
        np.testing.assert_array_equal(x.cpu().numpy(), want[key])
    # bulk eviction (compress_once): the counts send it to the other schedules, tensor form or list form
    big = [e * 12 for e in evicted]
    want = oracle_pipeline(st, big, mode="reference")
    for form in ("tensor", "list"):
        args = _fork_call_arguments(st, big)
        if form == "list":
            args = args[:2] + (big,) + args[3:]
        out = ds.cm.schedule_evictions(*args)
        assert ds.cm.last_schedule_reason.startswith("general (small_eviction: bulk_eviction"), ds.cm.last_schedule_reason
        for x, key in zip(out, ("eli", "ekc", "ebc")):
            np.testing.assert_array_equal(x.cpu().numpy(), want[key], err_msg=f"{key} bulk {form}")


def test_the_forks_call_form_takes_the_bracket_schedule_for_bulk_evictions():
    """compress_once of one long sequence (BASELINE configs[1] in small): 64 Ki slots per sequence and 64 blocks per
    head -- the bracket schedule, from a device tensor of counts"""
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=16, seq_lens=[8193], seed=4, protected=32)
    nblk = int(((st.context_lens.astype(np.int64) + 15) // 16).sum())
    evicted = [nblk // 2]
    want = oracle_pipeline(st, evicted, mode="reference")
    ds = hdev.upload(st, DEV)
    ds.cm.schedule_path = 0
    out = ds.cm.schedule_evictions(*_fork_call_arguments(st, evicted))
    assert ds.cm.last_schedule_reason.startswith("bracket"), ds.cm.last_schedule_reason
    assert ds.cm.last_schedule_path() == "bracket"
    for x, key in zip(out, ("eli", "ekc", "ebc")):
        np.testing.assert_array_equal(x.cpu().numpy(), want[key], err_msg=key)


def test_a_device_tensor_of_counts_under_stream_capture_is_the_only_unknown_hint():
    """nothing can be read back while a stream is being captured: a captured call with a device tensor of counts takes
    the digit rounds (hint unknown) and needs total_slots=; its replay gives the oracle's result"""
    st, evicted = _steady(2, 4, 16, 2, 512, 3)
    want = oracle_pipeline(st, evicted, mode="reference")
    ds = hdev.upload(st, DEV)
    ds.cm.schedule_path = 0
    args = _fork_call_arguments(st, evicted)
    ds.cm.schedule_evictions(*args)                       # (warm: workspaces exist before the capture)
    ds.cm.schedule_path = 1
    ds.cm.schedule_evictions(*args, total_slots=st.total_slots)
    ds.cm.schedule_path = 0
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ds.cm.schedule_path = 1
        ds.cm.schedule_evictions(*args, total_slots=st.total_slots)      # (this stream's workspace)
        ds.cm.schedule_path = 0
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        out = ds.cm.schedule_evictions(*args, total_slots=st.total_slots)
        reason = ds.cm.last_schedule_reason
    assert "hint_unknown" in reason, reason
    g.replay()
    torch.cuda.synchronize()
    for x, key in zip(out, ("eli", "ekc", "ebc")):
        np.testing.assert_array_equal(x.cpu().numpy(), want[key], err_msg=key)


@pytest.mark.parametrize("bs,evicted", [(16, [6, 9]), (16, [3, 12]), (32, [4, 5]), (8, [12, 20])])
def test_lists_longer_than_a_wave(bs, evicted):
    """few, long heads that free several blocks each: their candidate lists hold 65 ... 256 entries -- the path of the
    one-launch records / selection / emission kernel that sorts a list through LDS instead of ranking it in registers
    (and, with KVC_TOPK_CHAIN=1, stream_records_kernel's); one short-listed sequence next to them"""
    for seed in range(3):
        st = synth.make_state(num_layers=1, num_kv_heads=2, block_size=bs, seq_lens=[6000, 9000, 700], seed=seed, protected=bs + 1,
                              steady_cap=2048 // bs * bs)
        ev = evicted + [1]
        want = oracle_pipeline(st, ev, mode="per_sequence")
        got, how = _run(st, ev, 2, "per_sequence")
        assert how.startswith("small_eviction"), how
        for key in KEYS:
            np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} bs={bs} evicted={ev} seed={seed}")


@pytest.mark.parametrize("ties", [1, 3, 6])
def test_small_eviction_schedule_with_metric_ties(ties):
    """canonical tie order: slots by (metric, physical block, offset), thresholds by (threshold,
    head, chunk) -- at the record cut, inside the records and at the sequence cut"""
    for seed in range(3):
        st, evicted = _steady(2, 4, 16, 2, 512, seed, tie_levels=ties)
        want = oracle_pipeline(st, evicted, mode="per_sequence")
        got, how = _run(st, evicted, 0, "per_sequence")
        assert how.startswith("small_eviction")
        for key in KEYS:
            np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} ties={ties} seed={seed}")


def test_skewed_head_raises_the_fallback_and_the_result_stays_exact():
    """one head whose metrics are all lowest absorbs the whole eviction: more chunks than a record
    holds -> flag -> the gated general pipeline recomputes"""
    st, evicted = _steady(2, 4, 16, 1, 1024, 5)
    blocks = np.nonzero((st.layer_index_by_block == 1) & (st.head_index_by_block == 2)
                        & (st.seq_index_by_block == 0))[0]
    st.metrics[blocks] -= np.float32(1e7)            # still tie-free within the head
    evicted = [24]                                   # > 256 / 16 chunks, all from that head
    want = oracle_pipeline(st, evicted, mode="per_sequence")
    got, how = _run(st, evicted, 2, "per_sequence")
    assert how == "small_eviction+fallback"
    for key in KEYS:
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)
    assert int(want["ebc"].reshape(-1)[1 * 4 + 2]) == 24


def test_ragged_heads_sampled_pivot_and_candidate_overflow():
    """ragged batches (long and tiny sequences side by side, a sequence that frees nothing); a long
    head whose metrics are all tied overflows its record -> flag -> general pipeline"""
    for seed in range(4):
        st = synth.make_state(num_layers=1, num_kv_heads=2, block_size=16, seq_lens=[9000, 200, 3000, 200],
                              seed=seed, protected=3)
        evicted = [2, 1, 3, 0]
        want = oracle_pipeline(st, evicted, mode="per_sequence")
        got, how = _run(st, evicted, 2, "per_sequence")
        assert how == "small_eviction"
        for key in KEYS:
            np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} seed={seed}")
    st = synth.make_state(num_layers=1, num_kv_heads=2, block_size=16, seq_lens=[6000, 64], seed=1,
                          protected=2, tie_levels=1)
    evicted = [3, 1]
    want = oracle_pipeline(st, evicted, mode="per_sequence")
    got, how = _run(st, evicted, 2, "per_sequence")
    assert how == "small_eviction+fallback"
    for key in KEYS:
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)


@pytest.mark.parametrize("path", [0, 1, 2, 3, 4])
def test_forced_paths_on_mixed_batches(path):
    """bulk and tiny evictions, compressed states, B > 1 quirk: whatever the path, the oracle's result"""
    cases = [
        dict(L=2, H=4, bs=16, seq_lens=[300, 171, 90], prot=[32, 5, 17], compressed=True, frac=0.7),
        dict(L=4, H=8, bs=16, seq_lens=[700], prot=32, compressed=False, frac=0.05),
        dict(L=2, H=2, bs=32, seq_lens=[260, 100], prot=33, compressed=False, frac=0.5),
        dict(L=2, H=2, bs=8, seq_lens=[400, 90], prot=9, compressed=True, frac=0.1),
    ]
    for c in cases:
        for mode in ("reference", "per_sequence"):
            st = synth.make_state(num_layers=c["L"], num_kv_heads=c["H"], block_size=c["bs"],
                                  seq_lens=c["seq_lens"], seed=7, protected=c["prot"], compressed=c["compressed"])
            bs = st.block_size
            nblk = ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)
            evicted = [int(n * c["frac"]) for n in nblk]
            want = oracle_pipeline(st, evicted, mode=mode)
            got, _ = _run(st, evicted, path, mode)
            for key in KEYS:
                np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} {c} {mode}")


def test_lean_outputs_on_the_small_eviction_schedule():
    st, evicted = _steady(2, 4, 16, 2, 512, 4)
    full, how = _run(st, evicted, 0, "per_sequence")
    lean, how2 = _run(st, evicted, 0, "per_sequence", lean=True)
    assert how == how2 == "small_eviction"
    for key in ("ekc", "ebc", "cmc"):
        np.testing.assert_array_equal(lean[key], full[key])
    offs = st.evicted_kv_offsets.reshape(-1)
    for g, c in enumerate(full["ekc"].reshape(-1)):
        np.testing.assert_array_equal(lean["eli"][offs[g]:offs[g] + c], full["eli"][offs[g]:offs[g] + c])
        np.testing.assert_array_equal(lean["cmi"][offs[g]:offs[g] + full["cmc"].reshape(-1)[g]],
                                      full["cmi"][offs[g]:offs[g] + full["cmc"].reshape(-1)[g]])


def test_config3_full_size_steady_state_properties():
    """BASELINE configs[2] at its real scale: 256 resident sequences x 256 heads (65 536 heads,
    270 M candidate slots), one compression step.  Size-independent properties: every sequence
    frees exactly what was asked, per head the evicted indices are ascending and distinct and are
    exactly the head's cnt lowest-metric evictable slots, the two schedules agree bit for bit."""
    L, H, bs, B, cap = 32, 8, 16, 256, 4096
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 40 << 30:
        pytest.skip("needs ~40 GB of free HBM")
    st, evicted = _steady(L, H, bs, B, cap, 11)
    ds = hdev.upload(st, DEV, mode="per_sequence")
    ds.cm.schedule_path = 0                 # (this test is about the automatic choice: KVC_SCHEDULE_PATH must not decide)
    args = (list(st.seq_indices), ds.seq_positions, evicted, ds.context_lens, ds.hanging_token_count,
            ds.evicted_kv_offsets, list(st.protected))
    eli, ekc, ebc = ds.cm.schedule_evictions(*args, total_slots=st.total_slots)
    assert ds.cm.last_schedule_path() == "small_eviction"
    ds.cm.schedule_path = 1
    eli1, ekc1, ebc1 = ds.cm.schedule_evictions(*args, total_slots=st.total_slots)
    assert ds.cm.last_schedule_path() == "general"
    assert torch.equal(eli, eli1) and torch.equal(ekc, ekc1) and torch.equal(ebc, ebc1)
    assert ebc.sum(dim=(1, 2)).cpu().tolist() == evicted
    G = B * L * H
    n = st.total_slots // G
    seg = eli.view(G, n).long()
    cnt = ekc.reshape(-1).long()
    j = torch.arange(n, device=DEV)[None, :]
    live = j < cnt[:, None]
    assert bool((seg[~live] == 2147483000).all())
    assert bool(((seg[:, 1:] > seg[:, :-1]) | ~live[:, 1:]).all())
    # evicted slots = the cnt smallest evictable metrics of the head (tie-free data)
    bt = ds.block_tables.permute(1, 0, 2, 3).reshape(G, -1).long()          # [G, M] in (b,l,h) order
    lam = seg.clamp(max=n - 1)
    slot = bt.gather(1, lam // bs) * bs + lam % bs
    m_ev = ds.cm.metrics.reshape(-1)[slot].masked_fill(~live, float("-inf"))
    worst_evicted = m_ev.max(dim=1).values
    allslots = (bt[:, :n // bs, None] * bs + torch.arange(bs, device=DEV)[None, None, :]).reshape(G, n)
    m_all = ds.cm.metrics.reshape(-1)[allslots]
    pos = ds.cm.token_positions.reshape(-1)[allslots]
    seqpos = ds.seq_positions.long().repeat_interleave(L * H)[:, None]
    prot = torch.tensor(st.protected, device=DEV).long().repeat_interleave(L * H)[:, None]
    ctx = ds.context_lens.permute(1, 0, 2).reshape(G, 1).long()
    evictable = (pos <= seqpos - prot) & (j < ctx)
    below = (evictable & (m_all <= worst_evicted[:, None])).sum(dim=1)
    assert torch.equal(below, cnt)


@pytest.mark.parametrize("mode", ["reference", "per_sequence"])
def test_block_tables_argument_is_accepted_and_changes_nothing(mode):
    """``block_tables=`` fed round 2's gathering schedule; the streaming one needs no logical ->
    physical map, and a batch that fills its cache builds its keys from the per-block metadata.
    The argument is not dereferenced there -- a table naming blocks the cache does not have
    included -- and the result is the oracle's, also for a batch that is a subset of the resident
    sequences"""
    st, evicted = _steady(3, 4, 16, 4, 512, 9)
    want = oracle_pipeline(st, evicted, mode=mode)
    ds = hdev.upload(st, DEV, mode=mode)
    ds.cm.schedule_path = 0                 # (this test is about the automatic choice: KVC_SCHEDULE_PATH must not decide)
    args = (list(st.seq_indices), ds.seq_positions, evicted, ds.context_lens, ds.hanging_token_count,
            ds.evicted_kv_offsets, list(st.protected))
    bad = ds.block_tables.clone()
    bad[1, 2, 3, 5] = st.num_blocks + 7
    bad[0, 0, 0, 0] = 2 ** 31 - 1
    for bt in (ds.block_tables, bad):
        a = ds.cm.schedule_evictions(*args, total_slots=st.total_slots, block_tables=bt)
        assert ds.cm.last_schedule_path() == "small_eviction"
        for got, key in zip(a, ("eli", "ekc", "ebc")):
            np.testing.assert_array_equal(got.cpu().numpy(), want[key], err_msg=key)
    # two of the four sequences
    sub = [1, 3]
    ctx = ds.context_lens[:, sub].contiguous()
    hang = ds.hanging_token_count[sub].contiguous()
    offs_np = synth.kv_offsets(st.context_lens[:, sub], st.block_size)
    offs = torch.from_numpy(offs_np).to(DEV)
    n_sub = int(((st.context_lens[:, sub].astype(np.int64) + 15) // 16).sum()) * 16
    ds.cm.schedule_path = 2
    b = ds.cm.schedule_evictions(sub, ds.seq_positions[sub].contiguous(), [evicted[i] for i in sub], ctx, hang, offs,
                                 [st.protected[i] for i in sub], total_slots=n_sub)
    assert ds.cm.last_schedule_path() == "small_eviction"
    ds.cm.schedule_path = 1
    c = ds.cm.schedule_evictions(sub, ds.seq_positions[sub].contiguous(), [evicted[i] for i in sub], ctx, hang, offs,
                                 [st.protected[i] for i in sub], total_slots=n_sub)
    assert ds.cm.last_schedule_path() == "general"
    for x, y in zip(b, c):
        assert torch.equal(x, y)


@pytest.mark.parametrize("path", [2, 3])
@pytest.mark.parametrize("stride", [1, 2, 4, 8, 16, 64])
@pytest.mark.parametrize("shape", ["perm", "decay", "oldest"])
def test_sample_stride_never_changes_the_result(stride, shape, path):
    """the pivots of the streaming schedule come from a sample of one physical block in `stride`;
    whatever the sample says the records hold every key below the pivot, so the result is exact or
    the flag is raised (and the general pipeline's result is exact).  Clustered metrics
    (oldest-first: a head's lowest keys all sit in its first block) are the sample's worst case.
    path 2: positions looked up for the candidates only (per_sequence mode); path 3: streamed."""
    flagged = 0
    for seed in range(3):
        st, evicted = _steady(4, 4, 16, 3, 1024, 30 + seed, metric_shape=shape)
        want = oracle_pipeline(st, evicted, mode="per_sequence")
        ds = hdev.upload(st, DEV, mode="per_sequence")
        ds.cm.schedule_path = path
        ds.cm.sample_stride = stride
        eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, st, evicted)
        how = ds.cm.last_schedule_path()
        assert how.startswith("small_eviction")
        flagged += how.endswith("+fallback")
        for key, got in zip(KEYS, (eli, ekc, ebc, cmi, cmc)):
            np.testing.assert_array_equal(got.cpu().numpy(), want[key], err_msg=f"{key} stride={stride} {shape} seed={seed}")
    if shape == "perm" or stride == 1:
        assert flagged == 0                       # unclustered metrics / a full sample never miss


@pytest.mark.parametrize("path", [2, 3])
@pytest.mark.parametrize("mode", ["per_sequence"])
def test_overask_on_the_small_eviction_schedule(path, mode):
    """asking for more chunks than a sequence has evictable ones (the reference silently
    evicts fewer): with the positions looked up lazily nobody counted the evictable keys, the
    records cannot list k thresholds and the general pipeline takes over; tiny heads whose every
    key fits a record finish on their own when the keys were counted"""
    st = synth.make_state(num_layers=2, num_kv_heads=2, block_size=16, seq_lens=[100, 60], seed=21, protected=20)
    nblk = ((st.context_lens.astype(np.int64) + 15) // 16).sum(0).sum(-1)
    for evicted in ([int(n) for n in nblk], [int(nblk[0]), 1], [0, int(nblk[1])], [int(nblk[0]) - 3, 2]):
        want = oracle_pipeline(st, evicted, mode=mode)
        got, how = _run(st, evicted, path, mode)
        assert how.startswith("small_eviction")
        for key in KEYS:
            np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} {evicted} {how}")


def test_missing_block_metadata_raises_the_fallback():
    """a logical block of the batch that no physical block claims (its metadata row was detached;
    outside what the reference defines): the streaming schedule counts the claims and hands over
    to the general pipeline, which treats the chunk as not evictable -- one behaviour, whatever
    the path"""
    st, evicted = _steady(2, 4, 16, 2, 512, 13)
    victim = int(st.block_tables[1, 1, 2, 3])
    st.seq_index_by_block[victim] = -1
    gen, how_gen = _run(st, evicted, 1, "per_sequence")
    got, how = _run(st, evicted, 2, "per_sequence")
    assert how_gen == "general" and how == "small_eviction+fallback"
    for key in KEYS:
        np.testing.assert_array_equal(got[key], gen[key], err_msg=key)


@pytest.mark.parametrize("bs,hd", [(4, 8), (8, 64), (16, 128), (32, 128)])
def test_sparse_batch_in_a_large_cache(bs, hd):
    """an engine sizes its cache to HBM: most blocks do not belong to the batch.  The key pass then
    takes its compacting form (coalesced sweeps of the sequence indices, keys only for the
    batch's blocks); bulk and small evictions, both modes, against the oracle, end to end"""
    from tests.helpers import oracle_pipeline as pipe
    from vllm_kvcompress_amd import _custom_ops as ops
    for seed, (compressed, frac) in enumerate([(False, 0.5), (True, 0.6), (False, 0.03)]):
        st = synth.make_state(num_layers=2, num_kv_heads=3, block_size=bs, seq_lens=[40 * bs, 13 * bs + 5, 70 * bs],
                              seed=20 + seed, protected=[bs + 1, 3, 2 * bs], compressed=compressed,
                              spare_block_frac=4.0)
        assert st.total_slots < st.num_blocks * bs // 2
        nblk = ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)
        evicted = [int(n * frac) for n in nblk]
        k, v = synth.make_caches_u16(seed, st.num_blocks, hd, bs)
        for mode in ("reference", "per_sequence"):
            want = pipe(st, evicted, k, v, mode=mode)
            ds = hdev.upload(st, DEV, mode=mode)
            ds.cm.schedule_path = 1
            eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, st, evicted)
            kd, vd = torch.from_numpy(k.copy()).to(DEV), torch.from_numpy(v.copy()).to(DEV)
            ops.execute_cache_moves(kd, vd, ds.cm.metrics, ds.cm.token_positions, cmi, cmc, ds.evicted_kv_offsets, 1, 16)
            for name, got, w in (("eli", eli, want["eli"]), ("ekc", ekc, want["ekc"]), ("ebc", ebc, want["ebc"]),
                                 ("cmi", cmi, want["cmi"]), ("cmc", cmc, want["cmc"]), ("k", kd, want["k"]),
                                 ("v", vd, want["v"]), ("metrics", ds.cm.metrics, want["metrics"])):
                np.testing.assert_array_equal(got.cpu().numpy(), w, err_msg=f"{name} bs={bs} seed={seed} {mode}")


def test_a_raised_flag_sends_the_next_calls_to_the_general_schedule():
    """host policy (CompressionMetrics): a small-eviction call that had to fall back costs the
    streaming pass and the single-launch general pipeline; its flag reaches the host asynchronously
    and the next automatic calls take the general schedule -- 1, then 2, 4 ... calls -- before the
    small-eviction one is tried again.  Results never change."""
    st, evicted = _steady(4, 4, 16, 1, 1024, 5)
    blocks = np.nonzero((st.layer_index_by_block == 1) & (st.head_index_by_block == 2) & (st.seq_index_by_block == 0))[0]
    st.metrics[blocks] -= np.float32(1e7)
    evicted = [30]                  # <= 2 blocks per head (the hint admits it), all from one head: more than a record holds
    want = oracle_pipeline(st, evicted, mode="per_sequence")
    ds = hdev.upload(st, DEV, mode="per_sequence")
    ds.cm.schedule_path = 0
    args = (list(st.seq_indices), ds.seq_positions, evicted, ds.context_lens, ds.hanging_token_count,
            ds.evicted_kv_offsets, list(st.protected))
    hows = []
    for _ in range(8):
        out = ds.cm.schedule_evictions(*args, total_slots=st.total_slots)
        hows.append(ds.cm.last_schedule_path())      # (synchronises: the flag copy has landed by the next call)
        for got, key in zip(out, ("eli", "ekc", "ebc")):
            np.testing.assert_array_equal(got.cpu().numpy(), want[key], err_msg=key)
    assert hows == ["small_eviction+fallback", "general", "small_eviction+fallback", "general", "general",
                    "small_eviction+fallback", "general", "general"], hows


@pytest.mark.parametrize("path", [2, 3])
@pytest.mark.parametrize("bs", [8, 16, 32])
def test_small_eviction_schedule_in_a_sparse_cache(path, bs):
    """an engine-sized cache: most blocks belong to other sequences or to nobody.  The collecting
    pass then sweeps the sequence indices and works the batch's blocks off a compacted list, the
    sampling drain sits behind a membership filter, and the single-launch fallback takes its
    compacting key pass -- steady states (both modes), a batch that is a subset of the resident
    sequences, and a skewed head that forces the fallback, against the oracle"""
    for seed, mode in ((0, "per_sequence"), (1, "reference")):
        st, evicted = _steady(3, 4, bs, 3, 16 * bs * 4, 40 + seed, spare_block_frac=5.0)
        assert st.total_slots < st.num_blocks * bs // 2
        # other sequences' blocks in between: a third of the free blocks get a foreign owner
        free = np.nonzero(st.seq_index_by_block < 0)[0]
        st.seq_index_by_block[free[::3]] = 7
        want = oracle_pipeline(st, evicted, mode=mode)
        got, how = _run(st, evicted, path, mode)
        assert how == "small_eviction"
        for key in KEYS:
            np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} bs={bs} {mode}")
    # forced fallback in the sparse cache
    st, evicted = _steady(4, 4, bs, 1, 64 * bs, 50, spare_block_frac=5.0)
    blocks = np.nonzero((st.layer_index_by_block == 1) & (st.head_index_by_block == 2) & (st.seq_index_by_block == 0))[0]
    st.metrics[blocks] -= np.float32(1e7)
    evicted = [2 * (256 // bs) + 3]                 # more chunks from one head than a record holds
    want = oracle_pipeline(st, evicted, mode="per_sequence")
    got, how = _run(st, evicted, path, "per_sequence")
    assert how == "small_eviction+fallback"
    for key in KEYS:
        np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} bs={bs} fallback")


@pytest.mark.parametrize("path", [0, 1, 4])
@pytest.mark.parametrize("bs", [8, 16, 32])
def test_sparse_batch_builds_its_keys_through_the_callers_block_tables(path, bs):
    """a batch that takes less than half of its cache (an engine sizes the cache to HBM) and a
    caller that hands over BlockState.block_tables: the keys are built in logical order through the
    tables instead of by a sweep over every block's metadata.  The oracle's result, on every
    schedule that writes keys, for the whole batch and for a subset of the resident sequences; a
    block whose metadata no longer names its sequence counts as not there on both routes"""
    for seed, (compressed, frac) in enumerate([(False, 0.5), (True, 0.6), (False, 0.1)]):
        st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=bs, seq_lens=[70 * bs, 13 * bs + 5, 150 * bs],
                              seed=40 + seed, protected=[bs + 1, 3, 2 * bs], compressed=compressed, spare_block_frac=4.0)
        assert st.total_slots < st.num_blocks * bs // 2
        evicted = [int(n * frac) for n in ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)]
        for mode in ("reference", "per_sequence"):
            want = oracle_pipeline(st, evicted, mode=mode)
            ds = hdev.upload(st, DEV, mode=mode)
            ds.cm.schedule_path = path
            args = (list(st.seq_indices), ds.seq_positions, evicted, ds.context_lens, ds.hanging_token_count,
                    ds.evicted_kv_offsets, list(st.protected))
            for bt in (ds.block_tables, None):
                got = ds.cm.schedule_evictions(*args, total_slots=st.total_slots, block_tables=bt)
                assert ds.cm.last_used_block_tables == (bt is not None and not ds.cm.last_schedule_path().startswith("small"))
                for g, key in zip(got, ("eli", "ekc", "ebc")):
                    np.testing.assert_array_equal(g.cpu().numpy(), want[key], err_msg=f"{key} {mode} tables={bt is not None}")
    # two of the three sequences, and a detached block: with and without the tables
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=bs, seq_lens=[70 * bs, 40 * bs, 150 * bs], seed=44,
                          protected=bs + 1, spare_block_frac=4.0)
    st.seq_index_by_block[int(st.block_tables[1, 2, 3, 5])] = -1
    ds = hdev.upload(st, DEV, mode="per_sequence")
    ds.cm.schedule_path = path
    sub = [0, 2]
    ctx = ds.context_lens[:, sub].contiguous()
    hang = ds.hanging_token_count[sub].contiguous()
    offs = torch.from_numpy(synth.kv_offsets(st.context_lens[:, sub], bs)).to(DEV)
    n_sub = int(((st.context_lens[:, sub].astype(np.int64) + bs - 1) // bs).sum()) * bs
    ev = [int(n * 0.5) for n in ((st.context_lens[:, sub].astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)]
    res = []
    for bt in (ds.block_tables, None):
        out = ds.cm.schedule_evictions([int(st.seq_indices[i]) for i in sub], ds.seq_positions[sub].contiguous(), ev, ctx, hang,
                                       offs, [st.protected[i] for i in sub], total_slots=n_sub, block_tables=bt)
        res.append([t.clone() for t in out])
    assert ds.cm.last_used_block_tables is False
    for x, y in zip(*res):
        assert torch.equal(x, y)


def test_output_buffer_kept_between_calls_holds_the_padded_list():
    """the small-eviction schedule returns evicted_logical_indices in a buffer CompressionMetrics keeps
    (only what earlier calls left behind is padded again): over 60 calls with changing batch
    composition, eviction sizes and a call that falls back in the middle, the returned list equals
    the reference's fully padded one entry for entry; a result somebody still holds is never
    overwritten"""
    from tests.test_gpu_move_table import _sub_state
    L, H, bs = 2, 4, 16
    seq_lens = [3 * 1024] * 5
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=8, protected=bs + 1,
                          steady_cap=1024, spare_block_frac=0.05)
    ds = hdev.upload(st, DEV, mode="per_sequence")
    cm = ds.cm
    cm.schedule_path = 2
    assert cm.reuse_output_buffer
    rng = np.random.default_rng(4)
    B = len(seq_lens)
    ptrs, held = set(), []
    for step in range(60):
        k = int(rng.integers(1, B + 1))
        sel = sorted(rng.choice(B, size=k, replace=False).tolist())
        sub = _sub_state(st, sel)
        evicted = [int(rng.integers(0, 9)) for _ in sel]
        skew = step == 30
        if skew:                                       # one head absorbs the eviction: record overflow -> fallback
            blocks = np.nonzero((st.layer_index_by_block == 1) & (st.head_index_by_block == 2)
                                & (st.seq_index_by_block == sel[0]))[0]
            saved = cm.metrics[torch.from_numpy(blocks).to(DEV)].clone()
            m2 = st.metrics.copy()
            m2[blocks] -= np.float32(1e7)
            cm.metrics.copy_(torch.from_numpy(m2))
            sub.metrics = m2
            evicted[0] = 24
        want = oracle_pipeline(sub, evicted, mode="per_sequence")
        out = cm.schedule_evictions(sel, torch.from_numpy(sub.seq_positions).to(DEV), evicted,
                                    ds.context_lens[:, sel].contiguous(),
                                    torch.from_numpy(sub.hanging_token_count).to(DEV),
                                    torch.from_numpy(sub.evicted_kv_offsets).to(DEV), sub.protected,
                                    total_slots=sub.total_slots)
        how = cm.last_schedule_path()
        assert how == ("small_eviction+fallback" if skew else "small_eviction"), (step, how)
        for got, key in zip(out, ("eli", "ekc", "ebc")):
            np.testing.assert_array_equal(got.cpu().numpy(), want[key], err_msg=f"step {step}: {key}")
        ptrs.add(out[0].data_ptr())
        if step % 10 == 3:
            held.append((out[0], want["eli"].copy()))    # somebody keeps this result
        if skew:
            cm.metrics.copy_(torch.from_numpy(st.metrics))
        del out
    for t, want_eli in held:                             # ... and finds it unchanged at the end
        np.testing.assert_array_equal(t.cpu().numpy(), want_eli)
    assert len(ptrs) <= 1 + len(held) + B, "the buffer is kept between calls unless a result is still held or it must grow"
    cm.reuse_output_buffer = False
    args = ([0], torch.from_numpy(st.seq_positions[:1]).to(DEV), [2], ds.context_lens[:, [0]].contiguous(),
            ds.hanging_token_count[[0]].contiguous(), ds.evicted_kv_offsets[[0]].contiguous(), st.protected[:1])
    a = cm.schedule_evictions(*args)
    b = cm.schedule_evictions(*args)
    assert a[0].data_ptr() != b[0].data_ptr() and torch.equal(a[0], b[0])


def test_block_tables_debug_check(monkeypatch):
    """KVC_DEBUG_TABLES=1: schedule_evictions(block_tables=...) verifies the tables against all four
    per-block metadata rows of the batch (the key pass through the tables itself only checks the owning
    sequence): consistent tables pass, a table pointing at another head's block, a stale logical block
    number and an out-of-range block each raise with the offending entry named"""
    st, evicted = _steady(2, 4, 16, 3, 512, 4)
    ds = hdev.upload(st, DEV, mode="per_sequence")
    args = (list(st.seq_indices), ds.seq_positions, evicted, ds.context_lens, ds.hanging_token_count,
            ds.evicted_kv_offsets, list(st.protected))
    monkeypatch.setenv("KVC_DEBUG_TABLES", "1")
    ds.cm.schedule_evictions(*args, total_slots=st.total_slots, block_tables=ds.block_tables)       # consistent: fine
    ds.cm.check_block_tables(ds.block_tables, list(st.seq_indices), ds.context_lens)
    bad = ds.block_tables.clone()
    bad[1, 2, 3, 5], bad[1, 2, 0, 5] = ds.block_tables[1, 2, 0, 5], ds.block_tables[1, 2, 3, 5]     # two heads' blocks swapped
    with pytest.raises(RuntimeError, match=r"block_tables\[1, seq 2, [03], 5\]"):
        ds.cm.schedule_evictions(*args, total_slots=st.total_slots, block_tables=bad)
    bad = ds.block_tables.clone()
    bad[0, 1, 1, 2], bad[0, 1, 1, 3] = ds.block_tables[0, 1, 1, 3], ds.block_tables[0, 1, 1, 2]     # logical order swapped
    with pytest.raises(RuntimeError, match="not the ones the metadata was written from"):
        ds.cm.check_block_tables(bad, list(st.seq_indices), ds.context_lens)
    bad = ds.block_tables.clone()
    bad[0, 0, 0, 0] = st.num_blocks + 5
    with pytest.raises(RuntimeError, match=r"block_tables\[0, seq 0, 0, 0\]"):
        ds.cm.check_block_tables(bad, list(st.seq_indices), ds.context_lens)
    monkeypatch.delenv("KVC_DEBUG_TABLES")
    ds.cm.schedule_evictions(*args, total_slots=st.total_slots, block_tables=bad)                     # (not checked without the switch)


def test_next_pivots_are_the_same_wherever_they_are_made():
    """the pivots a call leaves for the next decode step's harvest: topk_fused_kernel's last phase (default),
    harvest_pivot_kernel launched behind it (KVC_TOPK_PIVOT_LAUNCH=1), the launch chain (KVC_TOPK_CHAIN=1) -- one
    function over the same remaining keys, so the same words, step after step (tests/pivot_phase_driver.py)"""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    for name, env in (("phase", {}), ("launch", {"KVC_TOPK_PIVOT_LAUNCH": "1"}), ("chain", {"KVC_TOPK_CHAIN": "1"})):
        e = dict(os.environ)
        e.pop("KVC_TOPK_PIVOT_LAUNCH", None)
        e.pop("KVC_TOPK_CHAIN", None)
        e.update(env)
        out = subprocess.run([sys.executable, os.path.join(repo, "tests", "pivot_phase_driver.py")], capture_output=True, text=True,
                             timeout=600, cwd=repo, env=e)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        line = [l for l in out.stdout.splitlines() if l.startswith("PIVOT_PHASE ")][-1]
        runs[name] = json.loads(line[len("PIVOT_PHASE "):])
    for case in runs["phase"]:
        a = runs["phase"][case]
        assert all(p is not None for p in a["pivots"]) and any("the call before" in h or "lists" in h for h in a["how"]), a["how"]
        for other in ("launch", "chain"):
            b = runs[other][case]
            assert a["pivots"] == b["pivots"], (case, other, a["pivots"], b["pivots"])
            assert a["digests"] == b["digests"], (case, other)
