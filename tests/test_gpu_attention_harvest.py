"""The decode step WITHOUT a sweep of the metric store: the fused-metric attention's epilogue makes the candidate
lists of the next ``schedule_evictions`` (``CompressionMetrics.begin_attention_harvest`` /
``paged_attention_kvc_fused_metrics(..., harvest=h, layer=l)`` / ``end_attention_harvest``;
kvc_attention_harvest_begin, include/kvc_mi355x.h ABI version 6; the reference author's to-do,
vllm/kvcompress/README.md:32, 49).

Two engines on the device, stepped through many iterations of continual compression with the SAME queries and
K/V: A in the reference's flow -- the attention writes its weights to temp_metrics (``paged_attention_kvc_v1``),
``aggregate_decode`` (metrics.py:429-439), ``schedule_evictions``' own pass -- and B with the fused attention +
harvest and no aggregation at all.  After every step every piece of state must be bit-equal (metric store,
positions, block metadata, context lengths, move lists, compacted K/V); B's schedule must be the ORACLE's schedule
of the store it ran on; and B must really have run on the epilogue's lists."""
import copy

import numpy as np
import pytest
import torch

from tests.helpers import oracle_pipeline
from vllm_kvcompress_amd import _custom_ops as ops
from vllm_kvcompress_amd.harness import device as hdev
from vllm_kvcompress_amd.harness import synth
from vllm_kvcompress_amd.kvcompress.block_state import append_slots
from vllm_kvcompress_amd.kvcompress.scheduler import CompressionScheduler, SeqCompressionRequest

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class AttnEngine:
    """resident sequences in the continual steady state: compress -> append the sampled token -> forward (L attention
    launches), as LLMEngine.step orders them (llm_engine.py:1556-1634).  ``fused``: the attention folds its weights
    into the store and harvests; else it writes temp_metrics and aggregate_decode runs behind the forward."""

    def __init__(self, st, seq_lens, cap, fused, qpk=4, hd=64, buffer_len=0, protected=None, seed=7, dtype=torch.float16,
                 mode="per_sequence", speculative=False):
        self.bs, self.L, self.H, self.cap, self.fused, self.qpk, self.hd = st.block_size, st.num_layers, st.num_kv_heads, cap, fused, qpk, hd
        self.mode = mode
        self.ds = hdev.upload(st, DEV, num_queries_per_kv=qpk, mode=mode)
        self.cm = self.ds.cm
        self.cm.strict_fallback = True
        # (the reference-flow engine is the baseline: its aggregate_decode() is the plain pass -- unless `speculative`: the
        # fork's flow unchanged, aggregate_decode() harvesting for the next call by itself)
        self.cm.speculative_harvest = speculative
        self.B = len(seq_lens)
        B, M = self.B, st.block_tables.shape[3] + 6
        bt = np.zeros((self.L, B, self.H, M), np.int32)
        bt[..., :st.block_tables.shape[3]] = st.block_tables
        self.bt = torch.from_numpy(bt).to(DEV)
        self.ctx = torch.from_numpy(st.context_lens.copy()).to(DEV)
        self.fm = torch.from_numpy(st.seq_index_by_block < 0).to(DEV)
        self.lens = np.asarray(seq_lens, np.int64).copy()          # seq.data.get_len(): includes the token sampled last
        self.protected = protected if protected is not None else self.bs + 1
        self.sched = CompressionScheduler(self.bs, self.L, self.H, 4 * st.total_slots, self.cm, device=DEV)
        g = torch.Generator(device=DEV)
        g.manual_seed(seed)
        NB = st.num_blocks
        self.k = (torch.randn((NB, hd // 8, self.bs, 8), device=DEV, generator=g) * 0.7).to(dtype)
        self.v = (torch.randn((NB, hd, self.bs), device=DEV, generator=g) * 0.7).to(dtype)
        self.qgen = torch.Generator(device=DEV)
        self.qgen.manual_seed(seed + 1)
        self.buffer_len = torch.full((B,), buffer_len, dtype=torch.int32, device=DEV)
        self.dtype = dtype
        self.used = self.offered = 0
        self.paths = []

    def compress(self, sel):
        cm, bs = self.cm, self.bs
        ctx_h = self.ctx.cpu().numpy().astype(np.int64)
        reqs = [SeqCompressionRequest(seq_id=100 + i, slot_index=i, seq_len=int(self.lens[i]),
                                      block_count=int(((ctx_h[:, i] + bs - 1) // bs).sum()), kv_count=int(ctx_h[:, i].sum()),
                                      max_cache_tokens=self.cap, protected_window_size=self.protected) for i in sel]
        out = self.sched.schedule_compression(reqs, self.bt, self.ctx, force=True, free_mask=self.fm)
        res = dict(used=None)
        if out is not None:
            res.update(used=bool(cm.last_harvest_used), path=cm.last_schedule_path(), reason=cm.last_schedule_reason,
                       cmc=out.cache_moves.count.clone(), cmi=out.cache_moves.index.clone(), slots=list(out.slot_indices))
            ops.execute_cache_moves(self.k, self.v, cm.metrics, cm.token_positions, out.cache_moves.index,
                                    out.cache_moves.count, out.cache_moves.offsets, 1, 16)
            self.used += res["used"]
            self.paths.append(res["path"])
        return res

    def append(self):
        """slots for the token sampled last step (block_manager.py:269-294); its metric starts at the head bias = 0
        (csrc/kvcompress_cache_kernels.cu:55-58: reshape_and_cache_kvc does that in the engine)"""
        B = self.B
        append_slots(self.bt, self.ctx, list(range(B)), [int(n) - 1 for n in self.lens], self.fm, self.cm, self.bs,
                     write_token_position=True)
        c = (self.ctx - 1).long()                                                  # [L, B, H]: the new token's logical index
        blk = torch.gather(self.bt.long(), 3, (c // self.bs).unsqueeze(-1)).squeeze(-1)
        self.cm.metrics.view(-1)[(blk * self.bs + c % self.bs).reshape(-1)] = 0.0

    def forward(self, next_batch):
        """one decode step of the model: L attention launches over all resident sequences.  ``next_batch``: the
        sequences the NEXT iteration will compress (what an engine in continual compression knows: all of them)"""
        cm, B, L, H, bs = self.cm, self.B, self.L, self.H, self.bs
        last = torch.tensor([int(n) - 1 for n in self.lens], dtype=torch.int32, device=DEV)     # the token being processed
        max_ctx = int(self.ctx.max().item())
        h = None
        if self.fused:
            # the schedule call of the next iteration: positions of the token sampled NOW, the context lengths as they are
            nb = sorted(next_batch)
            h = cm.begin_attention_harvest(nb, [int(self.lens[i]) for i in nb], [self.protected] * len(nb),
                                           self.ctx[:, nb].contiguous(), attention_seq_indices=list(range(B)))
            self.offered += h is not None
        else:
            cm.clear_temp_metrics()
        for l in range(L):
            q = (torch.randn((B, H * self.qpk, self.hd), device=DEV, generator=self.qgen) * 0.8).to(self.dtype)
            out = torch.empty_like(q)
            args = (q, self.k, self.v, H, self.hd ** -0.5, self.bt[l].contiguous(), self.ctx[l].contiguous(),
                    cm.token_positions, last, self.buffer_len, bs, max_ctx, None, "auto", 1.0, 1.0)
            if self.fused:
                ops.paged_attention_kvc_fused_metrics(out, cm.metrics, *args, use_l2=True, temp_metrics=cm.temp_metrics,
                                                      harvest=h, layer=l)
            else:
                ops.paged_attention_kvc_v1(out, cm.temp_metrics, *args, True)
        if self.fused:
            cm.end_attention_harvest(h)
        else:
            cm.aggregate_decode()
        self.lens += 1                                                             # the token sampled by this step

    def state(self):
        cm = self.cm
        return dict(metrics=cm.metrics.clone(), pos=cm.token_positions.clone(), seq=cm.seq_index_by_block.clone(),
                    lbn=cm.logical_block_num_by_block.clone(), ctx=self.ctx.clone(), k=self.k.clone(), v=self.v.clone())


def _same(a, b, what):
    for key in a:
        x, y = a[key], b[key]
        if x is None or isinstance(x, (bool, str, list)):
            continue
        if x.dtype in (torch.float32,):
            x, y = x.view(torch.int32), y.view(torch.int32)
        elif x.dtype in (torch.float16, torch.bfloat16):
            x, y = x.view(torch.int16), y.view(torch.int16)
        assert torch.equal(x, y), f"{what}: {key} differs"


def _oracle_schedule_of(engine, sel):
    """the oracle's schedule of the store engine B is about to compress (downloaded), per_sequence mode"""
    cm = engine.cm
    st = synth.PagedState.__new__(synth.PagedState)
    ctx = np.ascontiguousarray(engine.ctx.cpu().numpy()[:, sel, :])
    bs = engine.bs
    st.block_size, st.num_layers, st.num_kv_heads, st.num_seqs = bs, engine.L, engine.H, len(sel)
    st.num_blocks = int(cm.metrics.shape[0])
    st.metrics, st.token_positions = cm.metrics.cpu().numpy(), cm.token_positions.cpu().numpy()
    st.seq_index_by_block = cm.seq_index_by_block.cpu().numpy()
    st.layer_index_by_block = cm.layer_index_by_block.cpu().numpy()
    st.head_index_by_block = cm.head_index_by_block.cpu().numpy()
    st.logical_block_num_by_block = cm.logical_block_num_by_block.cpu().numpy()
    st.context_lens = ctx
    st.block_tables = np.ascontiguousarray(engine.bt.cpu().numpy()[:, sel])
    st.hanging_token_count = synth.hanging_tokens(ctx.transpose(1, 0, 2), bs)
    st.evicted_kv_offsets = synth.kv_offsets(ctx, bs)
    st.seq_indices = list(sel)
    st.seq_positions = np.asarray([int(engine.lens[i]) - 1 for i in sel], np.int32)
    st.protected = [engine.protected] * len(sel)
    evicted = [synth.evict_block_count(context_lens_lh=ctx[:, b, :], seq_len=int(engine.lens[s]), block_size=bs,
                                       protected_window_size=engine.protected, max_cache_tokens=engine.cap)
               for b, s in enumerate(sel)]
    return oracle_pipeline(st, evicted, mode=engine.mode), st


@pytest.mark.parametrize("bs,cap,qpk,hd,buffer_len,schedule", [
    (16, 320, 4, 64, 0, 0),        # contexts of one partition: the partition kernel's own epilogue
    (16, 640, 4, 128, 8, 0),       # several partitions: the rescale pass
    (16, 640, 4, 128, 0, 2),       # the single-pass kernel (forced: it is chosen by itself from 512 (sequence, head) pairs on)
    (32, 512, 8, 64, 0, 0), (8, 160, 4, 64, 3, 0)])
def test_engine_on_the_epilogues_lists_equals_the_reference_flow(bs, cap, qpk, hd, buffer_len, schedule):
    ops.set_attention_schedule(schedule)
    try:
        _run_twin_engines(bs, cap, qpk, hd, buffer_len)
    finally:
        ops.set_attention_schedule(0)


def _run_twin_engines(bs, cap, qpk, hd, buffer_len):
    L, H = 2, 4
    seq_lens = [cap + 300, cap + 41, cap + 555, cap + 123]
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=bs + qpk, protected=bs + 1,
                          spare_block_frac=0.8, steady_cap=cap)
    a = AttnEngine(copy.deepcopy(st), seq_lens, cap, fused=False, qpk=qpk, hd=hd, buffer_len=buffer_len)
    b = AttnEngine(copy.deepcopy(st), seq_lens, cap, fused=True, qpk=qpk, hd=hd, buffer_len=buffer_len)
    all_seqs = [0, 1, 2, 3]
    for it in range(26):
        sel = all_seqs if it % 9 != 7 else [1, 3]                 # (now and then only some sequences compress)
        want, ost = _oracle_schedule_of(b, sel)
        ra, rb = a.compress(sel), b.compress(sel)
        _same({k: v for k, v in ra.items() if k in ("cmc", "cmi")}, rb, f"step {it} (schedule)")
        # B's schedule is the oracle's schedule of B's store (the move list is a function of the schedule's outputs)
        rows = np.concatenate([np.arange(o, o + c) for o, c in zip(ost.evicted_kv_offsets.reshape(-1), want["cmc"].reshape(-1))]
                              + [np.zeros(0, np.int64)]).astype(np.int64)
        np.testing.assert_array_equal(rb["cmc"].cpu().numpy(), want["cmc"], err_msg=f"step {it}: move counts vs oracle")
        np.testing.assert_array_equal(rb["cmi"].cpu().numpy()[rows], want["cmi"][rows], err_msg=f"step {it}: moves vs oracle")
        a.append(); b.append()
        nxt = all_seqs if (it + 1) % 9 != 7 else [1, 3]
        a.forward(nxt); b.forward(nxt)
        _same(a.state(), b.state(), f"step {it} (after the forward)")
    assert b.offered >= 17 and b.used >= 14, (b.offered, b.used)
    assert not a.used
    # most harvested steps needed no redo (lists that fall short are redone on the device: exact either way)
    assert sum(p == "small_eviction" for p in b.paths) >= len(b.paths) - 5, b.paths


def test_a_window_that_ends_before_the_eviction_bound_is_redone_not_trusted():
    """kv_metric_buffer_len larger than the protected window: keys between the end of the metric window and the
    eviction bound are evictable but were not walked by the epilogue -- the head's count is pushed over the record
    length, the schedule call raises its flag and the device redoes the work: the oracle's schedule all the same"""
    bs, cap, L, H = 16, 320, 2, 4
    seq_lens = [cap + 300, cap + 41, cap + 90]
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=3, protected=bs + 1,
                          spare_block_frac=0.8, steady_cap=cap)
    a = AttnEngine(copy.deepcopy(st), seq_lens, cap, fused=False, buffer_len=40)
    b = AttnEngine(copy.deepcopy(st), seq_lens, cap, fused=True, buffer_len=40)
    sel = [0, 1, 2]
    used = 0
    for it in range(8):
        ra, rb = a.compress(sel), b.compress(sel)
        _same({k: v for k, v in ra.items() if k in ("cmc", "cmi")}, rb, f"step {it}")
        if rb["used"]:                       # (after a miss the host leaves predicted pivots alone for a few calls)
            used += 1
            assert rb["path"] == "small_eviction+fallback", rb
        a.append(); b.append()
        a.forward(sel); b.forward(sel)
        _same(a.state(), b.state(), f"step {it}")
    assert used >= 1


def test_lists_of_another_batch_or_a_missing_layer_are_not_used():
    bs, cap, L, H = 16, 320, 2, 4
    seq_lens = [cap + 300, cap + 41, cap + 90]
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=5, protected=bs + 1,
                          spare_block_frac=0.8, steady_cap=cap)
    a = AttnEngine(copy.deepcopy(st), seq_lens, cap, fused=False)
    b = AttnEngine(copy.deepcopy(st), seq_lens, cap, fused=True)
    sel = [0, 1, 2]
    for it in range(3):
        a.compress(sel); b.compress(sel)
        a.append(); b.append()
        a.forward(sel); b.forward(sel)
    assert b.used >= 1
    # the store written between the forward and the schedule call: the lists are dropped on the host
    b.cm.metrics[0, 0].add_(0.0)
    a.cm.metrics[0, 0].add_(0.0)
    ra, rb = a.compress(sel), b.compress(sel)
    assert not rb["used"] and rb["path"] == "small_eviction"
    _same({k: v for k, v in ra.items() if k in ("cmc", "cmi")}, rb, "after a foreign write")
    a.append(); b.append()
    # a forward that skips a layer's harvest: end_attention_harvest refuses
    cm = b.cm
    nb = sel
    h = cm.begin_attention_harvest(nb, [int(b.lens[i]) for i in nb], [b.protected] * 3, b.ctx[:, nb].contiguous(),
                                   attention_seq_indices=[0, 1, 2])
    assert h is not None
    h.layers_done.add(0)
    assert not cm.end_attention_harvest(h) and cm._hv_lists is None
    # ... and a handle that was not the last one begun
    h1 = cm.begin_attention_harvest(nb, [int(b.lens[i]) for i in nb], [b.protected] * 3, b.ctx[:, nb].contiguous())
    h2 = cm.begin_attention_harvest(nb, [int(b.lens[i]) for i in nb], [b.protected] * 3, b.ctx[:, nb].contiguous())
    assert h1 is not None and h2 is not None and not cm.end_attention_harvest(h1)


def test_not_offered_where_keys_depend_on_more_than_the_sum():
    """averaged metrics: a key is the sum divided by the key's age -> no handle, the plain fused attention"""
    bs, cap = 16, 320
    seq_lens = [cap + 300, cap + 41]
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=bs, seq_lens=seq_lens, seed=5, protected=bs + 1,
                          spare_block_frac=0.8, steady_cap=cap)
    ds = hdev.upload(st, DEV, num_queries_per_kv=4, mode="per_sequence", use_average=True)
    cm = ds.cm
    args = (list(st.seq_indices), ds.seq_positions, [8, 8], ds.context_lens, ds.hanging_token_count,
            ds.evicted_kv_offsets, list(st.protected))
    cm.harvest_ahead = True
    cm.schedule_evictions(*args, total_slots=st.total_slots)
    assert cm.last_schedule_path().startswith("small_eviction")
    assert cm.begin_attention_harvest(list(st.seq_indices), ds.seq_positions, list(st.protected), ds.context_lens) is None


@pytest.mark.parametrize("bs,cap,schedule", [(16, 320, 0), (16, 640, 0), (16, 640, 2), (32, 512, 0)])
def test_under_the_batch_rule_of_the_reference(bs, cap, schedule):
    """mode "reference" with several sequences -- the fork's default: the rule couples the sequences through every head's
    count of evictable keys (metrics.py:709-729), so the epilogue also counts the masked slots of every head (the last
    block's tail, keys inside the protected window / outside the metric window, non-finite sums), as the schedule's full
    collecting pass does.  Two engines again, the oracle's schedule (mode "reference") of the store every step."""
    ops.set_attention_schedule(schedule)
    try:
        L, H = 2, 4
        seq_lens = [cap + 300, cap + 41, cap + 555]
        st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=bs + 1, protected=bs + 1,
                              spare_block_frac=1.5, steady_cap=cap)
        a = AttnEngine(copy.deepcopy(st), seq_lens, cap, fused=False, mode="reference", hd=64)
        b = AttnEngine(copy.deepcopy(st), seq_lens, cap, fused=True, mode="reference", hd=64)
        sel = [0, 1, 2]
        for it in range(14):
            want, ost = _oracle_schedule_of(b, sel)
            ra, rb = a.compress(sel), b.compress(sel)
            _same({k: v for k, v in ra.items() if k in ("cmc", "cmi")}, rb, f"step {it} (schedule)")
            np.testing.assert_array_equal(rb["cmc"].cpu().numpy(), want["cmc"], err_msg=f"step {it}: move counts vs oracle")
            a.append(); b.append()
            a.forward(sel); b.forward(sel)
            _same(a.state(), b.state(), f"step {it} (after the forward)")
        # (under that rule later sequences free less than they were asked to, so what they are asked grows from step to
        # step: lists made for a smaller request are not used -- the call then takes its own pass)
        assert b.offered >= 8 and b.used >= 3, (b.offered, b.used, b.paths)
        assert not a.used
    finally:
        ops.set_attention_schedule(0)


@pytest.mark.parametrize("mode,cap", [("per_sequence", 320), ("per_sequence", 640), ("reference", 320)])
def test_the_unchanged_fork_flow_with_aggregate_decode_predicting_the_next_call(mode, cap):
    """The fork's flow with NOTHING changed -- attention -> temp_metrics, ``aggregate_decode()`` at the end of the
    iteration, the scheduler at the start of the next (llm_engine.py:1556-1634) -- twice: once with the plain aggregation
    pass, once with ``CompressionMetrics.speculative_harvest`` (the default): aggregate_decode() harvests for the call the
    next iteration will most likely make (the last call's batch, positions + 1).  Identical state after every step; the
    second engine's schedule calls run on lists, checked on the device against what the call really is."""
    bs, L, H = 16, 2, 4
    seq_lens = [cap + 300, cap + 41, cap + 555]
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=21, protected=bs + 1,
                          spare_block_frac=1.5, steady_cap=cap)
    a = AttnEngine(copy.deepcopy(st), seq_lens, cap, fused=False, mode=mode)
    c = AttnEngine(copy.deepcopy(st), seq_lens, cap, fused=False, mode=mode, speculative=True)
    sel = [0, 1, 2]
    for it in range(24):
        want, ost = _oracle_schedule_of(c, sel)
        ra, rc = a.compress(sel), c.compress(sel)
        _same({k: v for k, v in ra.items() if k in ("cmc", "cmi")}, rc, f"step {it} (schedule)")
        np.testing.assert_array_equal(rc["cmc"].cpu().numpy(), want["cmc"], err_msg=f"step {it}: move counts vs oracle")
        if rc["used"]:
            assert c.cm.last_harvest_kind == "aggregation pass, ahead of the call"
        a.append(); c.append()
        a.forward(sel); c.forward(sel)
        _same(a.state(), c.state(), f"step {it} (after the forward)")
    assert not a.used
    assert c.used >= (18 if mode == "per_sequence" else 3), (c.used, c.paths)
