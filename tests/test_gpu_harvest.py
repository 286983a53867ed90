"""Harvest-ahead: ``CompressionMetrics.aggregate_decode_and_harvest`` + the ``schedule_evictions`` that
follows (kvc_aggregate_decode_harvest / kvc_schedule_params.harvest, ABI version 5).

The aggregation pass of a decode step (reference metrics.py:429-439) also makes the small-eviction
schedule's candidate lists, with pivots the previous schedule call left behind.  Whatever those
pivots are worth, every step must give (a) the reference's sums, bit for bit -- the oracle's
``aggregate_decode`` -- and (b) the oracle's schedule of the aggregated store (reference
metrics.py:441-847): lists that fall short are redone on the device, lists that no longer belong
to the call (another batch, a store somebody wrote to) are not used.  Driven through many decode
steps of the continual steady state with the host-side block-state simulator."""
import copy

import numpy as np
import pytest
import torch

from oracle import kvc_oracle as orc
from tests.helpers import oracle_pipeline
from vllm_kvcompress_amd import _lib
from vllm_kvcompress_amd.harness import device as hdev
from vllm_kvcompress_amd.harness import synth
from vllm_kvcompress_amd.harness.engine_sim import EngineSim

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KEYS = ("eli", "ekc", "ebc")


class Loop:
    """one CompressionMetrics kept over the steps; the host state (oracle side) is copied into its
    tensors in front of every step"""

    def __init__(self, L, H, bs, seq_lens, cap, qpk=4, seed=3, use_l2=True, stride=0, mode="per_sequence", speculative=False):
        self.L, self.H, self.bs, self.cap, self.qpk, self.use_l2, self.mode = L, H, bs, cap, qpk, use_l2, mode
        self.seq_lens = list(seq_lens)
        self.st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=seed,
                                   protected=bs + 1, spare_block_frac=0.6, steady_cap=cap)
        self.sim = EngineSim(self.st, seq_lens)
        self.ds = hdev.upload(self.st, DEV, num_queries_per_kv=qpk, mode=mode)
        self.cm = self.ds.cm
        self.cm.use_l2 = use_l2
        assert self.cm.harvest_ahead is None     # (the first aggregate_decode_and_harvest turns it on)
        self.cm.strict_fallback = True          # (the flag word is looked at in the call itself)
        self.cm.speculative_harvest = speculative   # (plain aggregate_decode() harvesting ahead of the call: its own tests below)
        self.cm.sample_stride = stride
        self.k_np, self.v_np = synth.make_caches_u16(seed, self.st.num_blocks, 32, bs)
        self.rng = np.random.default_rng(seed)
        self.step_no = 0

    def sub_state(self, sel):
        st = self.st
        if sel == list(range(len(self.seq_lens))):
            return st
        ctx = np.ascontiguousarray(st.context_lens[:, sel, :])
        return synth.PagedState(
            block_size=st.block_size, num_layers=st.num_layers, num_kv_heads=st.num_kv_heads, num_seqs=len(sel),
            num_blocks=st.num_blocks, metrics=st.metrics, token_positions=st.token_positions,
            seq_index_by_block=st.seq_index_by_block, layer_index_by_block=st.layer_index_by_block,
            head_index_by_block=st.head_index_by_block, logical_block_num_by_block=st.logical_block_num_by_block,
            context_lens=ctx, block_tables=np.ascontiguousarray(st.block_tables[:, sel]),
            hanging_token_count=synth.hanging_tokens(ctx.transpose(1, 0, 2), st.block_size),
            evicted_kv_offsets=synth.kv_offsets(ctx, st.block_size), seq_indices=list(sel),
            seq_positions=np.ascontiguousarray(st.seq_positions[sel]), protected=[st.protected[i] for i in sel])

    def step(self, sel=None, between=None, before_harvest=None, scale=1.0, plain=False):
        """one decode step: attention mass -> aggregate (+ harvest) -> schedule; both sides compared;
        then the host state is carried on (compaction, freed blocks, one more token per head)"""
        st, cm, bs = self.st, self.cm, self.bs
        B = len(self.seq_lens)
        sel = list(range(B)) if sel is None else sel
        sub = self.sub_state(sel)
        temp = (self.rng.random((st.num_blocks, bs, self.qpk)) * scale).astype(np.float32)
        # ---- device: the state in front of the step
        cm.metrics.copy_(torch.from_numpy(st.metrics))
        cm.token_positions.copy_(torch.from_numpy(st.token_positions))
        cm.seq_index_by_block.copy_(torch.from_numpy(st.seq_index_by_block))
        cm.layer_index_by_block.copy_(torch.from_numpy(st.layer_index_by_block))
        cm.head_index_by_block.copy_(torch.from_numpy(st.head_index_by_block))
        cm.logical_block_num_by_block.copy_(torch.from_numpy(st.logical_block_num_by_block))
        cm.temp_metrics.copy_(torch.from_numpy(temp))
        ctx_t = torch.from_numpy(sub.context_lens).to(DEV)
        hang_t = torch.from_numpy(sub.hanging_token_count).to(DEV)
        offs_t = torch.from_numpy(sub.evicted_kv_offsets).to(DEV)
        pos_t = torch.from_numpy(sub.seq_positions).to(DEV)
        seqs, prot = list(sub.seq_indices), list(sub.protected)
        # ---- oracle: the sums
        orc.aggregate_decode(st.metrics, temp, use_l2=self.use_l2)
        if before_harvest is not None:
            before_harvest()
        if plain:
            cm.aggregate_decode()
            harvested = False
        else:
            harvested = cm.aggregate_decode_and_harvest(seqs, pos_t, prot, ctx_t, total_slots=sub.total_slots)
        np.testing.assert_array_equal(cm.metrics.cpu().numpy(), st.metrics, err_msg=f"step {self.step_no}: sums")
        assert not bool(cm._temp_metrics.any()), "the fused clear"
        if between is not None:
            between()
        evicted = [synth.evict_block_count(context_lens_lh=sub.context_lens[:, b, :], seq_len=int(self.sim.seq_lens[s]),
                                           block_size=bs, protected_window_size=bs + 1, max_cache_tokens=self.cap)
                   for b, s in enumerate(sel)]
        want = oracle_pipeline(sub, evicted, self.k_np, self.v_np, mode=self.mode)
        eli, ekc, ebc = cm.schedule_evictions(seqs, pos_t, evicted, ctx_t, hang_t, offs_t, prot, total_slots=sub.total_slots)
        got = dict(eli=eli.cpu().numpy(), ekc=ekc.cpu().numpy(), ebc=ebc.cpu().numpy())
        for key in KEYS:
            np.testing.assert_array_equal(got[key], want[key], err_msg=f"step {self.step_no} (sel {sel}, k {evicted}): {key}")
        if cm.last_schedule[2] == 1:
            assert cm.last_schedule_reason.endswith(f"[lists: the {cm.last_harvest_kind}]" if cm.last_harvest_used else
                                                    "[pivots: the call before]" if cm.last_pivot_memory_used else
                                                    "[pivots: sampled]"), cm.last_schedule_reason
        info = dict(harvested=harvested, used=cm.last_harvest_used, path=cm.last_schedule_path(), evicted=evicted,
                    remembered=cm.last_pivot_memory_used)
        # ---- carry the host state on (the oracle's compaction; only the selected sequences were compressed)
        st.metrics, st.token_positions = want["metrics"].copy(), want["positions"].copy()
        self.k_np, self.v_np = want["k"], want["v"]
        full_kv = np.zeros((B, self.L, self.H), np.int32)
        full_bc = np.zeros((B, self.L, self.H), np.int32)
        full_kv[sel], full_bc[sel] = want["ekc"], want["ebc"]
        self.sim.apply_compression(full_kv, full_bc)
        self.sim.append_token()
        self.step_no += 1
        return info


@pytest.mark.parametrize("bs,qpk,use_l2,stride", [(16, 4, True, 0), (16, 4, False, 2), (32, 8, True, 0), (8, 4, True, 0), (16, 1, True, 0), (16, 7, False, 0),
                                                   (16, 8, True, 4)])
def test_continual_steps_on_harvested_lists(bs, qpk, use_l2, stride):
    lp = Loop(L=2, H=4, bs=bs, seq_lens=[40 * bs + 5, 25 * bs, 33 * bs + 9], cap=20 * bs, qpk=qpk, use_l2=use_l2,
              stride=stride, seed=bs + qpk)
    first = lp.step()
    assert not first["harvested"] and not first["used"], "nothing to harvest with before the first schedule call"
    assert first["path"].startswith("small_eviction"), first
    used = clean = 0
    for it in range(36):
        info = lp.step()
        assert info["used"] == info["harvested"] or not info["used"]
        used += info["used"]
        clean += info["used"] and info["path"] == "small_eviction"
        assert info["path"].startswith("small_eviction"), info
    # (a step is not harvested when it frees more than the pivots were made for, or right after a miss)
    assert used >= 18, f"harvested lists were used in {used} of 36 steps"
    if stride == 0:
        assert clean >= used - 4, f"{used - clean} of {used} harvested steps had to be redone on the device"


def test_lists_are_dropped_when_they_no_longer_belong_to_the_call():
    lp = Loop(L=2, H=4, bs=16, seq_lens=[700, 420, 555, 610], cap=320)
    cm = lp.cm
    lp.step()
    assert lp.step()["used"]
    # the store written through torch between the two calls
    info = lp.step(between=lambda: cm.metrics[0, 0].add_(0.0))
    assert info["harvested"] and not info["used"] and info["path"] == "small_eviction"
    assert lp.step()["used"]
    # block metadata written through the object
    info = lp.step(between=lambda: cm.remove_metadata(torch.empty((0,), dtype=torch.long, device=DEV)))
    assert info["harvested"] and not info["used"]
    assert lp.step()["used"]
    # another batch: the pivots were made for all four sequences
    info = lp.step(sel=[0, 2, 3])
    assert not info["harvested"] and not info["used"]
    info = lp.step(sel=[0, 2, 3])
    assert info["used"]
    info = lp.step()                       # (the sequence that sat out has more to free: maybe a bulk call, no pivots)
    assert not info["harvested"]
    lp.step()
    # a plain aggregate_decode in between: nothing to use, and the call leaves pivots again
    assert lp.step()["used"]
    info = lp.step(plain=True)
    assert not info["used"]
    assert lp.step()["used"]


def test_lists_that_fall_short_are_redone_on_the_device():
    lp = Loop(L=2, H=4, bs=16, seq_lens=[700, 420, 555], cap=320)
    cm = lp.cm
    lp.step()
    assert lp.step()["path"] == "small_eviction"
    lib = _lib.load()
    B, G = 3, 3 * 2 * 4
    assert cm._hv_buf.numel() >= lib.kvc_harvest_buffer_bytes(G, B)

    def no_pivots():                       # pivot 0: no key lies below it -> empty lists
        cm._hv_buf[256:256 + 4 * B] = 0

    misses, widen = cm.harvest_misses, cm.harvest_widen
    info = lp.step(before_harvest=no_pivots)
    assert info["used"] and info["path"] == "small_eviction+fallback", info
    assert cm.harvest_misses == misses + 1 and cm.harvest_widen > widen
    for _ in range(2):                     # the pivots are made anew by a usual pass; two calls without predictions
        info = lp.step()
        assert not info["harvested"] and not info["remembered"] and info["path"] == "small_eviction"
    assert lp.step()["used"]


def test_attention_mass_that_lifts_every_key_over_the_pivot():
    """increments far larger than the pivots' allowance: short lists now and then, never a wrong schedule"""
    lp = Loop(L=2, H=4, bs=16, seq_lens=[900, 640], cap=384, seed=11)
    lp.step()
    paths = [lp.step(scale=40.0 if it % 3 == 0 else 1.0)["path"] for it in range(18)]
    assert all(p.startswith("small_eviction") for p in paths)


def test_keys_that_depend_on_positions_are_harvested_with_the_position_rows():
    """use_average: a key is metric / (seq_pos - position) -- no lazy form; the aggregation pass then streams the
    position rows as well, makes every key in full and counts the masked slots per head, as the full collecting
    pass does.  A bulk eviction (not the small-eviction schedule) is not eligible: the plain sums, the usual schedule"""
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=16, seq_lens=[600, 300], seed=3, protected=17,
                          steady_cap=160)
    ds = hdev.upload(st, DEV, num_queries_per_kv=4, mode="per_sequence", use_average=True)
    cm = ds.cm
    cm.schedule_path = 0                 # (the bulk call below is about the automatic choice: KVC_SCHEDULE_PATH must not decide)
    temp = np.random.default_rng(0).random((st.num_blocks, 16, 4)).astype(np.float32)
    for it in range(4):
        cm.temp_metrics.copy_(torch.from_numpy(temp))
        orc.aggregate_decode(st.metrics, temp, use_l2=True)
        harvested = cm.aggregate_decode_and_harvest(list(st.seq_indices), ds.seq_positions, list(st.protected),
                                                    ds.context_lens, total_slots=st.total_slots)
        assert harvested == (it > 0)
        np.testing.assert_array_equal(cm.metrics.cpu().numpy(), st.metrics)
        want = oracle_pipeline(st, [8, 8], mode="per_sequence", use_average=True)
        eli, ekc, ebc = cm.schedule_evictions(list(st.seq_indices), ds.seq_positions, [8, 8], ds.context_lens,
                                              ds.hanging_token_count, ds.evicted_kv_offsets, list(st.protected),
                                              total_slots=st.total_slots)
        assert cm.last_harvest_used == harvested and cm.last_schedule_path() == "small_eviction"
        np.testing.assert_array_equal(eli.cpu().numpy(), want["eli"])
        np.testing.assert_array_equal(ekc.cpu().numpy(), want["ekc"])
    # a bulk call in between: nothing to harvest for
    nblk = ((st.context_lens.astype(np.int64) + 15) // 16).sum(0).sum(-1)
    bulk = [int(n) // 2 for n in nblk]
    eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, st, bulk)
    assert not cm.last_harvest_used and cm._hv is None
    np.testing.assert_array_equal(ekc.cpu().numpy(), oracle_pipeline(st, bulk, mode="per_sequence", use_average=True)["ekc"])
    assert not cm.aggregate_decode_and_harvest(list(st.seq_indices), ds.seq_positions, list(st.protected), ds.context_lens,
                                               total_slots=st.total_slots)


def test_switched_off_by_the_environment_variable(monkeypatch):
    monkeypatch.setenv("KVC_HARVEST_AHEAD", "0")
    monkeypatch.setenv("KVC_PIVOT_MEMORY", "0")
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=16, seq_lens=[600, 300], seed=3, protected=17, steady_cap=160)
    ds = hdev.upload(st, DEV, num_queries_per_kv=4, mode="per_sequence")
    cm = ds.cm
    assert cm.harvest_ahead is False and cm.pivot_memory is False
    temp = np.random.default_rng(0).random((st.num_blocks, 16, 4)).astype(np.float32)
    for it in range(3):
        cm.temp_metrics.copy_(torch.from_numpy(temp))
        orc.aggregate_decode(st.metrics, temp, use_l2=True)
        assert not cm.aggregate_decode_and_harvest(list(st.seq_indices), ds.seq_positions, list(st.protected),
                                                   ds.context_lens, total_slots=st.total_slots)
        np.testing.assert_array_equal(cm.metrics.cpu().numpy(), st.metrics)
        want = oracle_pipeline(st, [8, 8], mode="per_sequence")
        eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, st, [8, 8])
        assert not cm.last_harvest_used and not cm.last_pivot_memory_used and cm._hv is None and cm._hv_buf is None
        np.testing.assert_array_equal(eli.cpu().numpy(), want["eli"])


class _Engine:
    """the block state of a few resident sequences on the device, stepped with the package's own ops
    (scheduler -> compaction -> append_slots), as the fork's engine steps its own"""

    def __init__(self, st, seq_lens, cap, qpk, deferred, speculative=False):
        from vllm_kvcompress_amd.kvcompress.scheduler import CompressionScheduler
        self.bs, self.L, self.H, self.cap, self.deferred = st.block_size, st.num_layers, st.num_kv_heads, cap, deferred
        self.ds = hdev.upload(st, DEV, num_queries_per_kv=qpk, mode="per_sequence")
        self.cm = self.ds.cm
        self.cm.speculative_harvest = speculative
        B, M = len(seq_lens), st.block_tables.shape[3] + 4
        bt = np.zeros((self.L, B, self.H, M), np.int32)
        bt[..., :st.block_tables.shape[3]] = st.block_tables
        self.bt = torch.from_numpy(bt).to(DEV)
        self.ctx = torch.from_numpy(st.context_lens.copy()).to(DEV)
        self.fm = torch.from_numpy(st.seq_index_by_block < 0).to(DEV)
        self.lens = np.asarray(seq_lens, np.int64).copy()
        self.sched = CompressionScheduler(self.bs, self.L, self.H, 4 * st.total_slots, self.cm, device=DEV)
        hd = 32
        k_np, v_np = synth.make_caches_u16(1, st.num_blocks, hd, self.bs)
        self.k, self.v = torch.from_numpy(k_np).to(DEV), torch.from_numpy(v_np).to(DEV)

    def step(self, temp, sel):
        from vllm_kvcompress_amd import _custom_ops as ops
        from vllm_kvcompress_amd.kvcompress.block_state import append_slots
        from vllm_kvcompress_amd.kvcompress.scheduler import SeqCompressionRequest
        cm, bs = self.cm, self.bs
        cm.temp_metrics.copy_(temp)                      # (the decode attention's output)
        if not self.deferred:
            cm.aggregate_decode()                        # reference order: llm_engine.py:1634
        # the next iteration: slots for the sampled token first (block_manager.py:269-294) ...
        self.lens += 1
        append_slots(self.bt, self.ctx, list(range(len(self.lens))), [int(n) - 2 for n in self.lens], self.fm, cm, bs,
                     write_token_position=True)
        # ... then the compression of this iteration (scheduler.py:184-560)
        ctx_h = self.ctx.cpu().numpy().astype(np.int64)
        reqs = [SeqCompressionRequest(seq_id=100 + i, slot_index=i, seq_len=int(self.lens[i]),
                                      block_count=int(((ctx_h[:, i] + bs - 1) // bs).sum()), kv_count=int(ctx_h[:, i].sum()),
                                      max_cache_tokens=self.cap, protected_window_size=bs + 1) for i in sel]
        out = self.sched.schedule_compression(reqs, self.bt, self.ctx, force=True, free_mask=self.fm,
                                              aggregate_decode=self.deferred)
        res = dict(metrics=cm.metrics.clone(), used=cm.last_harvest_used if out is not None else None)
        if out is not None:
            res.update(cmc=out.cache_moves.count.clone(), cmi=out.cache_moves.index.clone(), freed=out.freed_blocks.clone()
                       if hasattr(out, "freed_blocks") else None, slots=list(out.slot_indices))
            ops.execute_cache_moves(self.k, self.v, cm.metrics, cm.token_positions, out.cache_moves.index,
                                    out.cache_moves.count, out.cache_moves.offsets, 1, 16)
        res.update(ctx=self.ctx.clone(), seq=cm.seq_index_by_block.clone(), after=cm.metrics.clone(),
                   pos=cm.token_positions.clone(), k=self.k.clone())
        return res


def test_scheduler_that_was_left_the_aggregate_gives_what_the_reference_order_gives():
    """``CompressionScheduler.schedule_compression(..., aggregate_decode=True)``: the engine leaves the
    decode step's aggregate_decode to the next reader of the metrics, the compression of the next
    iteration, where it runs as the harvesting pass (or as the plain one when nothing is compressed).
    Two engines on the device, one in the reference's order (llm_engine.py:1634 then scheduler.py:492),
    stepped with the same attention mass: every piece of state equal after every step."""
    L, H, bs, cap, qpk = 2, 4, 16, 320, 4
    seq_lens = [700, 420, 555, 610]
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=8, protected=bs + 1,
                          spare_block_frac=0.8, steady_cap=cap)
    a = _Engine(copy.deepcopy(st), seq_lens, cap, qpk, deferred=False)
    b = _Engine(copy.deepcopy(st), seq_lens, cap, qpk, deferred=True)
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    used = 0
    for it in range(30):
        temp = torch.rand((st.num_blocks, bs, qpk), device=DEV, generator=g)
        sel = [0, 1, 2, 3] if it % 7 != 5 else [1, 3]          # (now and then only some sequences compress)
        if it == 20:
            sel = []                                             # ... or none: the plain pass, once
        ra, rb = a.step(temp, sel), b.step(temp, sel)
        for key in ("metrics", "cmc", "cmi", "ctx", "seq", "after", "pos", "k"):
            if ra.get(key) is None:
                assert rb.get(key) is None, key
                continue
            assert torch.equal(ra[key].view(torch.int32) if ra[key].dtype == torch.float32 else ra[key],
                               rb[key].view(torch.int32) if rb[key].dtype == torch.float32 else rb[key]), f"step {it}: {key}"
        assert not ra["used"]
        used += bool(rb["used"])
    assert used >= 15, used


@pytest.mark.parametrize("bs,stride", [(16, 0), (32, 2), (8, 0)])
def test_reference_order_takes_its_pivots_from_the_call_before(bs, stride):
    """no harvest at all -- aggregate_decode at the end of a step, schedule_evictions at the start of the next,
    as the fork does: from the second call on the collecting pass takes the pivots the call before left behind
    (kvc_schedule_params.harvest bit 2) instead of sampling; the oracle's schedule every step, and a pass that
    lists too little is redone on the device"""
    lp = Loop(L=2, H=4, bs=bs, seq_lens=[40 * bs + 5, 25 * bs, 33 * bs + 9], cap=20 * bs, stride=stride, seed=bs)
    cm = lp.cm
    buf0 = cm._hv_buf                    # reserved with the store (init_kv_metadata), under harvest_buffer_max_bytes
    assert buf0 is not None and _lib.load().kvc_harvest_pivot_bytes(3) <= buf0.numel() <= cm.harvest_buffer_max_bytes
    first = lp.step(plain=True)
    assert not first["remembered"] and first["path"] == "small_eviction"
    remembered = clean = 0
    for it in range(30):
        info = lp.step(plain=True, scale=30.0 if it == 17 else 1.0)      # (once: attention that lifts keys over every pivot)
        assert not info["used"] and info["path"].startswith("small_eviction")
        remembered += info["remembered"]
        clean += info["remembered"] and info["path"] == "small_eviction"
    assert cm.harvest_ahead is None and cm._hv_buf is buf0          # nothing was allocated while "serving"
    assert remembered >= 24 and clean >= remembered - 3, (remembered, clean)
    # switched off: every call samples
    cm.pivot_memory = False
    for it in range(3):
        info = lp.step(plain=True)
        assert not info["remembered"] and info["path"] == "small_eviction"


@pytest.mark.parametrize("bs", [16, 32, 8])
def test_under_the_batch_rule_of_the_reference(bs):
    """mode "reference" with three sequences -- the fork's default: the reference's batch > 1 rule couples the
    sequences through their counts of evictable keys, so the collecting pass streams the positions (no lazy form).
    Its pivots come from the call before; and the harvesting aggregation pass streams the positions too and counts
    the masked slots per head, after which the schedule call runs on its lists: the oracle's schedule every step"""
    lp = Loop(L=2, H=4, bs=bs, seq_lens=[44 * bs, 27 * bs + 3, 35 * bs], cap=20 * bs, mode="reference", seed=bs + 1)
    lp.step(plain=True)
    remembered = used = 0
    for it in range(16):
        info = lp.step(plain=it % 4 == 3)
        # (under that rule later sequences free less than they were asked to, so what they are asked grows from step
        # to step: lists and pivots made for a smaller request are not used -- the call then samples)
        assert not info["used"] or info["harvested"], info
        remembered += info["remembered"]
        used += info["used"]
    assert used >= 3 and used + remembered >= 8, (used, remembered)


def test_pivots_that_never_suffice_stop_being_used():
    """every call's remembered pivots wiped: each predicted call lists nothing and is redone on the device (the
    oracle's schedule all the same); the host then leaves predicted pivots alone for 2, 4, 8 ... calls, so redone
    calls become rare instead of every other one"""
    lp = Loop(L=2, H=4, bs=16, seq_lens=[700, 420, 555], cap=320)
    cm = lp.cm

    def wipe():
        if cm._hv_buf is not None:
            cm._hv_buf[256:256 + 4 * 3] = 0

    lp.step(plain=True)
    predicted = []
    for it in range(40):
        info = lp.step(plain=True, between=wipe)
        assert info["path"] == ("small_eviction+fallback" if info["remembered"] else "small_eviction"), info
        predicted.append(bool(info["remembered"]))
    assert cm.harvest_misses == sum(predicted) and 3 <= sum(predicted) <= 6, predicted
    assert not any(predicted[-8:]) or sum(predicted[-20:]) <= 1, predicted


def test_under_inference_mode_no_lists_are_made():
    """tensors made under torch.inference_mode() (vLLM's workers) keep no version counter: whether somebody wrote to
    them between the harvest and the schedule call could not be known, so no lists are made -- the plain sums, and the
    call's own pass (on remembered pivots), with the oracle's result"""
    with torch.inference_mode():
        lp = Loop(L=2, H=4, bs=16, seq_lens=[700, 420, 555], cap=320)
        assert lp.cm.metrics.is_inference()
        lp.step()
        for it in range(4):
            info = lp.step()
            assert not info["harvested"] and not info["used"] and info["remembered"] and info["path"] == "small_eviction", info
    # a store made outside, per-step tensors made inside: the same
    lp = Loop(L=2, H=4, bs=16, seq_lens=[700, 420, 555], cap=320)
    lp.step()
    assert lp.step()["used"]
    with torch.inference_mode():
        info = lp.step()
        assert not info["harvested"] and not info["used"] and info["path"] == "small_eviction", info
    lp.step()
    assert lp.step()["used"]


# ---- aggregate_decode() that harvests ahead of the call by itself (CompressionMetrics.speculative_harvest) ----------------
@pytest.mark.parametrize("bs,qpk,mode", [(16, 4, "per_sequence"), (32, 8, "per_sequence"), (8, 4, "per_sequence"), (16, 4, "reference")])
def test_plain_aggregate_decode_harvests_for_the_next_call_in_the_fork_s_flow(bs, qpk, mode):
    """The fork's own flow, unchanged: ``aggregate_decode()`` at the end of an iteration, ``schedule_evictions`` at the start
    of the next -- nobody tells the aggregation what the next call will be.  It predicts the last call's batch one token
    further on and harvests for it; the lists carry the positions and windows they were made with and the schedule call
    checks them on the device.  The oracle's sums and schedule every step; the plain steps run on lists."""
    lp = Loop(L=2, H=4, bs=bs, seq_lens=[40 * bs + 5, 25 * bs, 33 * bs + 9], cap=20 * bs, qpk=qpk, seed=bs + qpk, mode=mode,
              speculative=True)
    cm = lp.cm
    first = lp.step(plain=True)
    assert not first["used"] and first["path"].startswith("small_eviction")
    used = clean = 0
    for it in range(30):
        info = lp.step(plain=True)
        used += info["used"]
        clean += info["used"] and info["path"] == "small_eviction"
        assert info["path"].startswith("small_eviction"), info
        if info["used"]:
            assert cm.last_harvest_kind == "aggregation pass, ahead of the call"
    if mode == "per_sequence":
        assert used >= 20 and clean >= used - 4, (used, clean)
    else:
        # (under the reference's batch > 1 rule later sequences free less than asked, so what they are asked grows from
        # step to step and lists made for a smaller request are not used)
        assert used >= 3, used


def test_a_prediction_that_does_not_hold_is_redone_on_the_device():
    """the next call is NOT the last one a token further on: other positions (a sequence that skipped a step), another
    protected window, another batch -- the device-side check (or the host's, for the batch) refuses the lists; the
    oracle's schedule all the same, and after a refusal on the device predictions pause"""
    lp = Loop(L=2, H=4, bs=16, seq_lens=[700, 420, 555], cap=320, speculative=True)
    cm = lp.cm
    lp.step(plain=True)
    assert lp.step(plain=True)["used"]
    # positions that moved by two: the lists were made for + 1
    lp.sim.append_token()
    lp.st.metrics[:] = lp.st.metrics                      # (state carried by the host simulator; nothing else to do)
    info = lp.step(plain=True)
    assert info["used"] and info["path"] == "small_eviction+fallback", info
    assert cm._hv_pause > 0 or cm.harvest_misses >= 1
    for _ in range(3):                                      # predictions pause, then come back
        info = lp.step(plain=True)
        assert info["path"].startswith("small_eviction")
    used_again = any(lp.step(plain=True)["used"] for _ in range(6))
    assert used_again
    # another batch: refused on the host (no lists are offered for other sequences)
    info = lp.step(plain=True, sel=[0, 2])
    assert not info["used"]


def test_speculative_harvest_is_off_when_asked():
    lp = Loop(L=2, H=4, bs=16, seq_lens=[700, 420, 555], cap=320, speculative=False)
    lp.step(plain=True)
    for _ in range(4):
        info = lp.step(plain=True)
        assert not info["used"] and info["remembered"]
    assert lp.cm._hv is not None and lp.cm._hv["full"] is False          # pivots only: nobody asks for lists



def test_prediction_with_several_decode_steps_between_two_schedule_calls():
    """compression_interval = 3: three aggregate_decode() calls (three tokens) lie between two schedule calls.  The first
    gap teaches the pattern; from then on the LAST aggregation of a gap harvests for positions + 3, and the schedule call
    runs on its lists -- the oracle's schedule of the store every time"""
    bs, cap = 16, 320
    seq_lens = [cap + 200, cap + 90]
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=bs, seq_lens=seq_lens, seed=13, protected=bs + 1,
                          steady_cap=cap)
    ds = hdev.upload(st, DEV, num_queries_per_kv=4, mode="per_sequence")
    cm = ds.cm
    cm.strict_fallback = True
    rng = np.random.default_rng(2)
    seqs, prot = list(st.seq_indices), list(st.protected)
    used = []
    for call in range(6):
        for agg in range(3):                                     # three decode steps: the sums, no change of the block state
            temp = rng.random((st.num_blocks, bs, 4)).astype(np.float32)
            cm.temp_metrics.copy_(torch.from_numpy(temp))
            orc.aggregate_decode(st.metrics, temp, use_l2=True)
            cm.aggregate_decode()
            st.seq_positions = st.seq_positions + 1
        np.testing.assert_array_equal(cm.metrics.cpu().numpy(), st.metrics)
        want = oracle_pipeline(st, [8, 8], mode="per_sequence")
        pos_t = torch.from_numpy(st.seq_positions.astype(np.int32)).to(DEV)
        eli, ekc, ebc = cm.schedule_evictions(seqs, pos_t, [8, 8], ds.context_lens, ds.hanging_token_count,
                                              ds.evicted_kv_offsets, prot, total_slots=st.total_slots)
        assert cm.last_schedule_path() == "small_eviction", (call, cm.last_schedule_path())
        np.testing.assert_array_equal(eli.cpu().numpy(), want["eli"], err_msg=f"call {call}")
        np.testing.assert_array_equal(ekc.cpu().numpy(), want["ekc"], err_msg=f"call {call}")
        used.append(bool(cm.last_harvest_used))
    # call 0: no pivots yet; call 1: the gap was not known (assumed 1: the first aggregation harvested, the next two dropped
    # its lists); from call 2 on the third aggregation harvests for positions + 3
    assert used[0] is False and all(used[2:]), used


def test_predictions_nobody_takes_are_paused():
    """an engine that writes to the store between aggregate_decode() and schedule_evictions() (here: a no-op write to
    the position table, which bumps its version counter) voids every prediction: after three in a row the predictions
    pause -- aggregate_decode() is the plain pass again -- and the schedules stay the oracle's throughout"""
    bs, cap = 16, 320
    seq_lens = [cap + 120, cap + 70]
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=bs, seq_lens=seq_lens, seed=17, protected=bs + 1,
                          steady_cap=cap)
    ds = hdev.upload(st, DEV, num_queries_per_kv=4, mode="per_sequence")
    cm = ds.cm
    cm.strict_fallback = True
    rng = np.random.default_rng(4)
    seqs, prot = list(st.seq_indices), list(st.protected)
    made = []
    for call in range(9):
        temp = rng.random((st.num_blocks, bs, 4)).astype(np.float32)
        cm.temp_metrics.copy_(torch.from_numpy(temp))
        orc.aggregate_decode(st.metrics, temp, use_l2=True)
        cm.aggregate_decode()
        made.append(cm._hv_lists is not None)
        cm.token_positions.add_(0)                                  # somebody writes to the store
        st.seq_positions = st.seq_positions + 1
        want = oracle_pipeline(st, [8, 8], mode="per_sequence")
        pos_t = torch.from_numpy(st.seq_positions.astype(np.int32)).to(DEV)
        eli, ekc, ebc = cm.schedule_evictions(seqs, pos_t, [8, 8], ds.context_lens, ds.hanging_token_count,
                                              ds.evicted_kv_offsets, prot, total_slots=st.total_slots)
        assert not cm.last_harvest_used
        np.testing.assert_array_equal(eli.cpu().numpy(), want["eli"], err_msg=f"call {call}")
        np.testing.assert_array_equal(ekc.cpu().numpy(), want["ekc"], err_msg=f"call {call}")
    np.testing.assert_array_equal(cm.metrics.cpu().numpy(), st.metrics)
    # call 0: nothing to predict from; calls 1-3: predictions made and void; then four schedule calls without
    assert made == [False, True, True, True, False, False, False, False, True], made
