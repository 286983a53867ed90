"""``CompressionMetrics`` with the reference's interface, computed by HIP kernels.

Mirror of ``vllm/kvcompress/metrics.py`` (reference lines cited per method): same
constructor, same state tensors (names, shapes, dtypes), same methods and return
values.  What differs is *how*: ``schedule_evictions`` is one sort-free device
pipeline (``csrc/kvc_schedule.hip``) instead of six ``torch.sort`` calls plus a host
loop, and the aggregation methods are single fused passes.

There is no CPU path: the state lives on a HIP device and every method that computes
calls into ``libkvc_mi355x.so``.
"""
from __future__ import annotations

import ctypes
import math
import os
import time
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .. import _lib
from .._custom_ops import _stream, on_device, workspace
from .._lib import MAX_INT, KvcScheduleParams

_BIAS_KEY = "bias"                 # reference metrics.py:13-14
_POSITION_RANGE_KEY = "pos_bins"

IntsLike = Union[Sequence[int], torch.Tensor]


@dataclass
class KVHeadBias:
    """reference metrics.py:44-81 (the lookup itself runs inside build_keys_kernel)"""
    bias: torch.Tensor             # [num_layers, num_kv_heads, num_bins] f32
    position_bins: torch.Tensor    # [num_bins] i32

    def to(self, device) -> "KVHeadBias":
        self.bias = self.bias.to(device)
        self.position_bins = self.position_bins.to(device)
        return self


def _load_kv_head_bias(path: str) -> KVHeadBias:
    """reference metrics.py:17-41"""
    ext = path.split(".")[-1]
    if ext == "safetensors":
        from safetensors import safe_open
        f = safe_open(path, framework="pt")
        return KVHeadBias(f.get_tensor(_BIAS_KEY).type(torch.float),
                          f.get_tensor(_POSITION_RANGE_KEY).type(torch.int))
    if ext in ("pt", "bin"):
        f = torch.load(path)
        return KVHeadBias(f[_BIAS_KEY].type(torch.float), f[_POSITION_RANGE_KEY].type(torch.int))
    if ext == "npz":
        import numpy as np
        f = np.load(path)
        return KVHeadBias(torch.tensor(f[_BIAS_KEY]).type(torch.float),
                          torch.tensor(f[_POSITION_RANGE_KEY]).type(torch.int))
    raise ValueError(f"Unsupported file format {ext}")


class AttentionHarvest:
    """What ``CompressionMetrics.begin_attention_harvest`` hands to the L attention launches of a decode step
    (``_custom_ops.paged_attention_kvc_fused_metrics(..., harvest=h, layer=l)``): the harvest buffer with the pivots
    of the last schedule call, the compression batch the lists are made for, and which layers have run."""

    def __init__(self, cm, buf, seq_slot, seq_positions, num_protected, num_seqs, stream, record):
        self.cm, self.buf, self.seq_slot = cm, buf, seq_slot
        self.seq_positions, self.num_protected = seq_positions, num_protected
        self.num_seqs, self.num_layers, self.num_kv_heads = num_seqs, cm.num_layers, cm.num_kv_heads
        self.block_size, self.num_sinks = cm.block_size, int(cm.num_sinks)
        self.stream, self.record = stream, record
        self.layers_done = set()

    def check_call(self, query, num_kv_heads, block_size, layer, stream) -> None:
        if (num_kv_heads != self.num_kv_heads or block_size != self.block_size or not 0 <= layer < self.num_layers
                or stream != self.stream or int(query.shape[0]) != int(self.seq_slot.numel())):
            raise RuntimeError("paged_attention_kvc_fused_metrics: this call does not belong to the harvest it was given "
                               "(KV heads / block size / layer / stream / number of sequences differ from begin_attention_harvest's)")


@dataclass
class SortedMetricOutputs:
    """reference metrics.py:83-90 (what the dead V1 front end ``sort_seq_metrics`` returned)"""
    sorted_indices: torch.Tensor
    seq_block_offsets: torch.Tensor
    layer_by_block: torch.Tensor
    head_by_block: torch.Tensor
    logical_block_num_by_block: torch.Tensor
    token_positions: torch.Tensor


class CompressionMetrics:
    """reference metrics.py:94-975"""

    def __init__(
        self,
        block_size: int,
        num_layers: int,
        num_kv_heads: int,
        num_queries_per_kv: int,
        max_kv_per_sort: int,
        kv_head_bias_file: Optional[str],
        kv_head_bias_weight: float,
        device: str = "cuda:0",
        random: bool = False,
        even_layer_evict: bool = False,
        use_l2: bool = True,
        use_average: bool = False,
        record_decoding_metrics: bool = True,
        num_attention_sinks: int = 0,
    ) -> None:
        _lib.load()                                   # fail loudly if the extension is absent
        self.block_size = block_size
        self.num_layers = num_layers
        self.num_kv_heads = num_kv_heads
        self.num_queries_per_kv = num_queries_per_kv
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("CompressionMetrics needs a HIP device (no CPU fallback exists)")
        self.num_sinks = num_attention_sinks
        self.random = random
        self.even_layer_evict = even_layer_evict
        self.use_l2 = use_l2
        self.use_average = use_average
        self.record_decoding_metrics = record_decoding_metrics
        self._has_bias = bool(kv_head_bias_file)
        if kv_head_bias_file:
            expected_shape = (num_layers, num_kv_heads)
            self.kv_metric_head_bias = _load_kv_head_bias(kv_head_bias_file).to(self.device)
            if tuple(self.kv_metric_head_bias.bias.shape[:-1]) != expected_shape:
                raise ValueError(f"expected shape {(*expected_shape, -1)} for KV head bias tensor "
                                 f"but got {self.kv_metric_head_bias.bias.shape}")
        else:
            self.kv_metric_head_bias = KVHeadBias(
                torch.zeros((num_layers, num_kv_heads, 1), dtype=torch.float, device=self.device),
                torch.zeros((1,), dtype=torch.int, device=self.device))
        self.kv_metric_bias_weight = kv_head_bias_weight
        self.max_kv_per_sort = max_kv_per_sort
        self.num_blocks = None
        self.unassigned_seq_idx = -1
        self.metrics = None
        self._temp_metrics = None
        self._temp_clean = False
        self.temp_v2_metrics = None
        self.seq_index_by_block = None
        self.layer_index_by_block = None
        self.head_index_by_block = None
        self.logical_block_num_by_block = None
        self.token_positions = None
        self.prev_seq_lens = {}
        # "reference": bit-exact to the reference including its batch>1 quirk
        # (metrics.py:718-721); "per_sequence": each sequence scheduled as if alone.
        self.schedule_mode = "reference"
        # extension (default off = reference-observable outputs): skip the MAX_INT padding of
        # evicted_logical_indices behind each head's evicted_kv_count entries and the defensive
        # clear of the key scratch (the engine keeps block metadata consistent); saves two
        # N x 4 B passes per call, which is 12 % of the schedule at 256 resident sequences
        self.lean_outputs = False
        # 0 = pick the schedule from the eviction counts and the shapes, 1 = digit rounds only, 2 / 3 =
        # small-eviction schedule, 4 = bracket schedule whenever the shapes allow
        # (kvc_schedule_params.schedule_path; tests force each)
        self.schedule_path = int(os.environ.get("KVC_SCHEDULE_PATH", "0"))
        # sample stride of the small-eviction schedule's pivots (0 = chosen from the batch size;
        # results do not depend on it, tests force every value)
        self.sample_stride = int(os.environ.get("KVC_SAMPLE_STRIDE", "0"))
        # workgroups of the single launch that redoes a call whose flag was raised (0 = what is resident
        # at once; results do not depend on it, tests launch grids that are not resident)
        self.fallback_grid = int(os.environ.get("KVC_FALLBACK_GRID", "0"))
        # KVC_STRICT_FALLBACK=1: read the flag word back after every small-eviction / bracket call
        # (one synchronisation per call) instead of one call later
        self.strict_fallback = os.environ.get("KVC_STRICT_FALLBACK", "0") not in ("", "0")
        self.last_schedule = None      # (workspace, fallback offset, plan: 0 general, 1 small-eviction, 2 bracket)
        self.last_schedule_reason = ""  # why the last call took the schedule it took (kvc_schedule_evictions_plan_reason)
        self.last_used_block_tables = False   # the last call built its keys through block_tables= (sparse batch)
        # host policy around the small-eviction schedule: a call whose flag was raised costs the
        # streaming pass AND the general pipeline in its single-launch form (3 x the general
        # pipeline at 16 sequences) -- fine as the exception, not as the rule.  The flag of every
        # small-eviction call is copied to pinned memory asynchronously and looked at by the NEXT
        # call (no synchronisation: a decode step lies in between); after a raised flag the general
        # schedule is taken for 1, 2, 4 ... up to 64 calls before the small-eviction one is tried
        # again.  Results are identical either way.
        self._fb_pin = None           # page-locked flag words, one per call in flight (FB_RING of them)
        self._fb_inflight = []        # [(word index, event or None, predicted, ticket or None)] in the order the calls were made
        self._fb_np = None
        self._fb_ticket = 0
        self._fb_penalty = 0          # general-schedule calls the last raised flag cost
        self._fb_backoff = 0          # of which still to go
        self._fb_fault = False        # a fallback launch gave up a wait (device fault): digit rounds only from now on
        # evicted_logical_indices of the small-eviction schedule: a handful of indices per head and 4 B
        # of MAX_INT padding per candidate slot (1.08 GB per decode step at 256 resident sequences).
        # The list is returned in a buffer this object keeps, with a device-side map of where earlier
        # calls left indices behind, so that a call pads only those places -- the returned tensor holds
        # exactly what the reference's holds.  The buffer is used again only when nothing else refers
        # to its storage any more (the caller dropped the previous result, as the reference's scheduler
        # does, scheduler.py:492-523); otherwise a new one is made, so a result stays valid for as long
        # as somebody holds it.  False = a fresh tensor and the full padding every call.
        self.reuse_output_buffer = os.environ.get("KVC_REUSE_OUTPUT_BUFFER", "1") not in ("", "0")
        self._eli_buf = None          # (buffer, dirty map, block size, storage use count when only we hold it, stream, version)
        self._small_cache = {}
        # The fork's call (device tensor of counts, no N) without a wait in front of the launches (ABI version 8): the
        # schedule is enqueued on an upper bound of N -- 1.25 x the largest N a batch of this size has had -- right behind
        # the summary launch that leaves the true N on the device; the host reads N while the device works and hands out
        # evicted_logical_indices[:N].  A bound that turns out too small voids the call on the device and it is repeated
        # the waiting way.  For the digit rounds and the bracket schedule (the small-eviction schedule is chosen from the
        # host-side counts); results are the same either way.  KVC_DEFERRED_N=0 / cm.deferred_n = False turns it off.
        self.deferred_n = os.environ.get("KVC_DEFERRED_N", "1") not in ("", "0")
        self._dn_bound = {}            # batch size -> bound on N (a multiple of 64 Ki slots)
        self._dn_plan = {}             # batch size -> the schedule the last call's true N and counts pick (0 / 1 / 2)
        self._dn_dev = None            # stream -> [2] int64 on the device: N, void
        self._dn_sizes = {}            # (bound, B, path) -> plan, workspace bytes, flag offset
        self.deferred_calls = 0        # calls that went this way / that were voided
        self.deferred_voided = 0
        self._summary_pin = None      # page-locked words the batch summary kernel writes (N, evicted_blocks_per_seq)
        self._summary_ticket = 0      # ... and the ticket of the last launch (the word behind the counts)
        self._summary_np = None
        # harvest-ahead (include/kvc_mi355x.h, ABI version 5): with compression every decode
        # step the metric store is swept twice per step -- by aggregate_decode and, a moment later, by the
        # small-eviction schedule's collecting pass.  ``aggregate_decode_and_harvest`` is the first sweep
        # that also makes the second one's candidate lists (with the pivots the previous
        # ``schedule_evictions`` left behind); the ``schedule_evictions`` that follows with the SAME batch
        # arguments, the store untouched in between, uses them.  Anything else -- another batch, a store
        # somebody wrote to, a larger eviction than the pivots were made for -- takes the usual pass.  Results
        # are identical either way (lists that fall short raise the flag like any short record).
        # None = from the first aggregate_decode_and_harvest on (the schedule calls of an engine that never
        # harvests do not pay for pivots nobody uses); KVC_HARVEST_AHEAD=0 / 1 = never / from the first call on
        env = os.environ.get("KVC_HARVEST_AHEAD", "")
        self.harvest_ahead = None if env == "" else env != "0"
        self.harvest_widen = float(os.environ.get("KVC_HARVEST_WIDEN", "0.25"))
        self._hv_widen0 = self.harvest_widen
        self._hv_streak = 0                # predicted calls in a row whose lists sufficed
        self._hv_pause = 0                 # calls still to go without predicted pivots (after a miss)
        self._hv_pause_len = 0
        # pivot memory (harvest bit 2; on unless KVC_PIVOT_MEMORY=0): without any harvest, a small-eviction call for
        # the batch of the call before takes the pivots that call left behind instead of sampling the store --
        # no sampling pass, no pivot kernel, half the candidates in its collecting pass.  Same results.
        self.pivot_memory = os.environ.get("KVC_PIVOT_MEMORY", "1") not in ("", "0")
        # speculative harvest (on unless KVC_SPECULATIVE_HARVEST=0 or KVC_HARVEST_AHEAD=0): the fork calls aggregate_decode()
        # at the END of an iteration (llm_engine.py:1634) without saying what the next iteration will compress.  In
        # continual compression that is the batch of the last schedule call, one token further on: the plain
        # aggregate_decode() then harvests for THAT call -- the last call's sequences and protected windows, its positions
        # + 1, block membership from the metadata alone -- and the lists carry what they were made with; the next
        # schedule_evictions for the same sequences takes them with the device-side check of harvest bit 3 (another
        # position, another window, blocks that came or went: redone on the device, and predictions pause).  One sweep of
        # the store per decode step in the UNCHANGED fork flow.  Same results either way.
        self.speculative_harvest = (os.environ.get("KVC_SPECULATIVE_HARVEST", "1") not in ("", "0")
                                    and os.environ.get("KVC_HARVEST_AHEAD", "") != "0")
        # What the lists cost in HBM: kvc_harvest_buffer_bytes(G, B) ~ 2 KiB per head of the compression batch (256 u64
        # entries + counters; 134 MB at 256 sequences x 32 layers x 8 KV heads).  The buffer is RESERVED when the store
        # is (init_kv_metadata: the engine's start-up, where its memory profiling sees it), sized for the heads the store
        # can hold and capped by this budget (KVC_HARVEST_BUFFER_MAX_MB, default 192); a batch whose lists would not fit
        # the budget keeps the pivots only -- its schedule call makes its lists in its own collecting pass, as without
        # speculative harvest.  Nothing is allocated while serving unless a batch outgrows a reservation smaller than
        # the budget (a store initialised with few blocks).
        self.harvest_buffer_max_bytes = int(float(os.environ.get("KVC_HARVEST_BUFFER_MAX_MB", "192")) * (1 << 20))
        self.last_harvest_kind = ""        # "aggregation pass" | "aggregation pass, ahead of the call" | "attention's epilogue"
        # (compression_interval > 1: several aggregate_decode() calls lie between two schedule calls.  The prediction is made
        # by the LAST of them -- the gap seen last time says which one that is -- for positions + that many tokens)
        self._aggs_since_schedule = 0
        self._last_gap = 1
        # (an engine whose order of calls makes every prediction void -- say, one that writes to the store between the
        # aggregation and the schedule call -- would pay for lists it never uses: three predictions in a row that no
        # call took pause the predictions for 4, 8 ... 256 schedule calls)
        self._spec_made = False
        self._spec_unused = 0
        self._spec_pause = 0
        self._spec_pause_len = 0
        self.last_pivot_memory_used = False
        self.last_harvest_used = False     # the last schedule_evictions ran on harvested lists
        self.harvest_misses = 0            # harvested calls whose lists fell short (flag raised, redone on device)
        self._hv_buf = None                # pivots + lists (kvc_harvest_buffer_bytes)
        self._hv = None                    # pivots in _hv_buf: the batch and eviction sizes they were made for
        self._hv_lists = None              # lists in _hv_buf: the call they were made for
        self._hv_attention = None          # the handle of begin_attention_harvest while the step's attention runs

    # temp_metrics is handed to the attention kernels, which write into it; reading the
    # attribute therefore marks it dirty so that the fused clear in aggregate_decode stays
    # invisible to callers (clear_temp_metrics keeps its contract).
    @property
    def temp_metrics(self):
        self._temp_clean = False
        return self._temp_metrics

    @temp_metrics.setter
    def temp_metrics(self, value):
        self._temp_metrics = value
        self._temp_clean = False

    # ------------------------------------------------------------------ state management
    def reinit_kv_metadata(self) -> None:
        num_blocks = self.num_blocks
        self.clear_kv_metadata()
        self.init_kv_metadata(num_blocks)

    def clear_kv_metadata(self) -> None:
        """reference metrics.py:205-214"""
        self._hv = self._hv_lists = None
        self.num_blocks = None
        self.metrics = None
        self._temp_metrics = None
        self.temp_v2_metrics = None
        self.seq_index_by_block = None
        self.layer_index_by_block = None
        self.head_index_by_block = None
        self.logical_block_num_by_block = None
        self.token_positions = None

    def init_kv_metadata(self, num_blocks: int) -> None:
        """reference metrics.py:216-275"""
        assert self.num_blocks is None, "already initialized"
        self.num_blocks = num_blocks
        dev = self.device
        self.metrics = torch.empty((num_blocks, self.block_size), dtype=torch.float32, device=dev)
        if self.random:
            self.metrics.uniform_()
        self._temp_metrics = torch.empty((num_blocks, self.block_size, self.num_queries_per_kv),
                                         dtype=torch.float32, device=dev)
        self._temp_clean = False
        self.temp_v2_metrics = torch.empty_like(self._temp_metrics)
        self.seq_index_by_block = torch.full((num_blocks,), self.unassigned_seq_idx,
                                             dtype=torch.int, device=dev)
        self.layer_index_by_block = torch.zeros((num_blocks,), dtype=torch.int, device=dev)
        self.head_index_by_block = torch.zeros((num_blocks,), dtype=torch.int, device=dev)
        self.logical_block_num_by_block = torch.zeros((num_blocks,), dtype=torch.int, device=dev)
        self.token_positions = torch.zeros((num_blocks, self.block_size), dtype=torch.int,
                                           device=dev)
        self._reserve_harvest_buffer()
        self.validate_metadata()

    def _wants_lists(self) -> bool:
        """lists are wanted once somebody harvests explicitly -- or from the start when aggregate_decode() may harvest
        ahead of the call by itself"""
        return bool(self.harvest_ahead) or (self.harvest_ahead is None and self.speculative_harvest
                                            and self.record_decoding_metrics and not self.random)

    def _reserve_harvest_buffer(self) -> None:
        """The harvest buffer (pivots + per-head candidate lists) next to the store, at start-up: sized for as many
        heads as the store can hold blocks (a head of a compression batch owns at least one), capped by
        ``harvest_buffer_max_bytes``.  See ``__init__``."""
        if self.num_blocks is None or not (self._wants_lists() or self.pivot_memory):
            return
        lib = _lib.load()
        heads_per_seq = self.num_layers * self.num_kv_heads
        B = max(1, min(self.num_blocks // heads_per_seq, 6500))
        if self._wants_lists():
            G = max(heads_per_seq, min(self.num_blocks, B * heads_per_seq))
            size = min(int(lib.kvc_harvest_buffer_bytes(G, B)), max(self.harvest_buffer_max_bytes, 0))
        else:
            size = 0
        size = max(size, int(lib.kvc_harvest_pivot_bytes(B)))
        if self._hv_buf is None or self._hv_buf.numel() < size:
            self._hv_buf = torch.zeros((size,), dtype=torch.uint8, device=self.device)
            self._hv = self._hv_lists = None

    def validate_metadata(self) -> None:
        """reference metrics.py:372-376"""
        allocated_mask = self.seq_index_by_block >= 0
        assert (self.head_index_by_block[allocated_mask] < self.num_kv_heads).all()

    def validate_metadata_even_layer_evict(self) -> None:
        """reference metrics.py:380-389 (a debug check, commented out at its only call site :372): with
        ``even_layer_evict`` every layer holds the same number of allocated blocks.  The reference samples ONE random
        layer per call; here every layer is checked (deterministic; fails whenever the reference's could)."""
        if self.even_layer_evict:
            allocated_mask = self.seq_index_by_block >= 0
            per_layer = torch.bincount(self.layer_index_by_block[allocated_mask].long(), minlength=self.num_layers)
            per_layer_count = int(allocated_mask.sum().item()) / self.num_layers
            for check_idx in range(self.num_layers):
                check_idx_count = int(per_layer[check_idx].item())
                assert check_idx_count == per_layer_count, (
                    f'{check_idx_count=}, {per_layer_count=} (are you using control_layers?)')

    def sort_seq_metrics(self, seq_indices: List[int], seq_positions: List[int], checkpoint: bool = True):
        """reference metrics.py:850-966: the global-sort front end of the V1 scheduler (``schedule_cache_evictions``),
        which is dead code in the fork (vllm/kvcompress/scheduler.py:285 ``if False:``).  The live path is
        ``schedule_evictions``; like the V1 ops this raises with a pointer to it instead of an AttributeError."""
        from .._custom_ops import V1_DEAD_MESSAGE
        raise NotImplementedError("sort_seq_metrics feeds the V1 scheduler: " + V1_DEAD_MESSAGE)

    def checkpoint(self, sink=None) -> None:
        """reference metrics.py:968-975: hands the store and its metadata to the fork's debugging CHECKPOINTER
        (vllm/debug.py, disabled unless a developer switches it on: a no-op in every run of the fork's own scripts).
        ``sink(name, tensor)`` receives the same six (name, tensor) pairs; without one this is the disabled
        checkpointer's no-op."""
        if sink is None:
            return
        for name in ("metrics", "seq_index_by_block", "layer_index_by_block", "head_index_by_block",
                     "logical_block_num_by_block", "token_positions"):
            sink("metrics__" + name, getattr(self, name))

    def clear_temp_metrics(self) -> None:
        """reference metrics.py:337-342.  A no-op when the last aggregate_decode already
        zeroed the buffer in its own pass and nobody has touched it since."""
        if self._temp_clean:
            return
        self._temp_metrics.zero_()
        self._temp_clean = True

    def insert_metadata(self, metadata) -> None:
        """reference metrics.py:344-364 (``metadata`` is a BlockMetadata-like object)"""
        pb = metadata.physical_blocks
        self._hv_lists = None
        self.seq_index_by_block[pb] = metadata.seq_indices
        self.logical_block_num_by_block[pb] = metadata.logical_blocks.type(torch.int)
        self.layer_index_by_block[pb] = metadata.layer_indices
        self.head_index_by_block[pb] = metadata.head_indices
        self.token_positions[pb] = metadata.token_positions

    def remove_metadata(self, physical_blocks: torch.Tensor) -> None:
        """reference metrics.py:366-370"""
        self._hv_lists = None
        self.seq_index_by_block[physical_blocks] = -1

    def forget_pivots(self) -> None:
        """Drop the pivots the last ``schedule_evictions`` left behind (and any lists made with them).  They are kept
        per batch slot: when a slot changes hands -- a sequence finished, another one took its index
        (CompressionScheduler.complete_seqs calls this) -- the next call samples the store instead of starting from
        a stranger's pivot (which would still give the right schedule, through a redone call)."""
        self._hv = self._hv_lists = None

    def randomize_metric_slots(self, slot_mapping: torch.Tensor) -> None:
        flat_indices = slot_mapping.flatten().type(torch.long)
        self._hv_lists = None
        self.metrics.view(-1)[flat_indices] = torch.rand(
            flat_indices.shape, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ aggregation
    def aggregate_prefill(self, prefill_metrics: torch.Tensor, slot_mapping: torch.Tensor) -> None:
        """reference metrics.py:396-427: metrics[slot[t,h]] += sum_q prefill[t, h*qpk+q]"""
        if self.random:
            return
        self._hv_lists = None
        seq_len, nq = prefill_metrics.shape
        assert nq % self.num_kv_heads == 0
        qpk = nq // self.num_kv_heads
        assert tuple(slot_mapping.shape) == (seq_len, self.num_kv_heads), \
            f"{(seq_len, self.num_kv_heads)} != {tuple(slot_mapping.shape)}"
        lib = _lib.load()
        pm = prefill_metrics.contiguous()
        sm = slot_mapping.contiguous()
        if pm.dtype != torch.float32 or sm.dtype != torch.int64 or not pm.is_cuda:
            raise RuntimeError("aggregate_prefill: need float32 metrics and int64 slots on a HIP device")
        with on_device(self.device):
            _lib.check(lib.kvc_aggregate_prefill(self.metrics.data_ptr(), pm.data_ptr(),
                                                 sm.data_ptr(), seq_len, self.num_kv_heads, qpk,
                                                 _stream(self.metrics)))

    def aggregate_decode(self, fuse_clear: bool = True, predict: bool = True) -> None:
        """reference metrics.py:429-439: metrics += sum_q temp^2 (L2) or sum_q temp (L1).
        With ``fuse_clear`` the same pass also zeroes temp_metrics, which makes the next
        ``clear_temp_metrics()`` free.  ``predict=False`` (not in the reference's signature): the
        plain pass only -- no lists for the call the next iteration will probably make (a step that is being
        aborted, CompressionScheduler.schedule_compression)."""
        if self.random or not self.record_decoding_metrics:
            return
        self._hv_lists = None
        self._aggs_since_schedule += 1
        if (predict and self._aggs_since_schedule == self._last_gap
                and self._speculative_harvest(fuse_clear, self._last_gap)):
            return
        self._plain_aggregate_decode(fuse_clear)

    def _plain_aggregate_decode(self, fuse_clear: bool) -> None:
        lib = _lib.load()
        with on_device(self.device):
            _lib.check(lib.kvc_aggregate_decode(
                self.metrics.data_ptr(), self._temp_metrics.data_ptr(), self.metrics.numel(),
                self.num_queries_per_kv, 1 if self.use_l2 else 0, 1 if fuse_clear else 0,
                _stream(self.metrics)))
        self._temp_clean = bool(fuse_clear)

    def _speculative_harvest(self, fuse_clear: bool, tokens_ahead: int = 1) -> bool:
        """``aggregate_decode()`` as the harvesting pass for the schedule call the NEXT iteration will most likely make
        (see ``speculative_harvest`` in ``__init__``).  Returns False -- nothing was launched -- when there is nothing to
        predict from; the sums are the plain pass's, bit for bit, either way."""
        hv = self._hv
        if not (self.speculative_harvest and self.harvest_ahead is not False and hv is not None and hv.get("full")
                and hv["buf"] is self._hv_buf and hv.get("seq_pos") is not None and self._hv_pause == 0 and not self._fb_fault
                and self._spec_pause == 0
                and not (self._fb_backoff > 0 and int(self.schedule_path) == 0)):
            return False
        capturing = torch.cuda.is_current_stream_capturing()
        stream = _stream(self.metrics)
        if capturing or hv["stream"] != stream or self._store_versions() is None:
            return False
        try:
            self._poll_fallback(False)
        except RuntimeError:
            self._plain_aggregate_decode(fuse_clear)     # (a fault reported by an EARLIER call: this step's sums still happen)
            raise
        if self._hv is not hv or self._hv_pause > 0 or self._fb_fault:      # (the poll may have dropped the pivots)
            return False
        lib = _lib.load()
        p = KvcScheduleParams()
        self._store_params(p, list(hv["seqs"]), hv["seq_pos"], hv["prot"], None, hv["N"])
        p.max_evicted_blocks_hint = int(hv["k"].max())
        p.schedule_path = int(self.schedule_path)
        p.harvest_buf = self._hv_buf.data_ptr()
        p.harvest_position_delta = int(tokens_ahead)      # a decode step: one more token per sequence
        if not lib.kvc_harvest_eligible(ctypes.byref(p), self.num_queries_per_kv):
            return False
        with on_device(self.device):
            _lib.check(lib.kvc_aggregate_decode_harvest(
                ctypes.byref(p), self._temp_metrics.data_ptr(), self.num_queries_per_kv,
                1 if self.use_l2 else 0, 1 if fuse_clear else 0, stream))
        self._temp_clean = bool(fuse_clear)
        torch.autograd.graph.increment_version(self.metrics)     # (written through a raw pointer)
        self._hv_lists = dict(seqs=hv["seqs"], attention=True, speculative=True, k=hv["k"], stream=stream, buf=self._hv_buf,
                              store=self._store_versions())
        self._spec_made = True
        return True

    # ------------------------------------------------------------------ harvest-ahead
    def _store_versions(self):
        """version counters of the six store tensors, or None if one of them was made under torch.inference_mode()
        (no counter: nothing can be said about who wrote to it, so lists made from it are never trusted)"""
        store = (self.metrics, self.token_positions, self.seq_index_by_block, self.layer_index_by_block,
                 self.head_index_by_block, self.logical_block_num_by_block)
        if any(t.is_inference() for t in store):
            return None
        return tuple(t._version for t in store)

    @staticmethod
    def _arg_record(x):
        """a batch argument as remembered between the harvest and its schedule call: a tensor by identity
        and version (held, so that neither can be reused), a list by value"""
        if isinstance(x, torch.Tensor):
            return (x, None if x.is_inference() else x._version)
        return (tuple(int(v) for v in x), None)

    @staticmethod
    def _arg_same(rec, x) -> bool:
        if isinstance(x, torch.Tensor):
            return rec[0] is x and rec[1] is not None and not x.is_inference() and rec[1] == x._version
        return not isinstance(rec[0], torch.Tensor) and rec[0] == tuple(int(v) for v in x)

    def _store_params(self, p, seq_indices, seq_pos, prot, context_lens, N: int) -> None:
        """the fields of kvc_schedule_params that describe the store and the batch"""
        bs, L, H, B = self.block_size, self.num_layers, self.num_kv_heads, len(seq_indices)
        slot_of_seq = self._slot_map(seq_indices)
        p.metrics = self.metrics.data_ptr()
        p.token_positions = self.token_positions.data_ptr()
        p.seq_index_by_block = self.seq_index_by_block.data_ptr()
        p.layer_index_by_block = self.layer_index_by_block.data_ptr()
        p.head_index_by_block = self.head_index_by_block.data_ptr()
        p.logical_block_num_by_block = self.logical_block_num_by_block.data_ptr()
        p.num_blocks = self.num_blocks
        p.block_size, p.num_layers, p.num_kv_heads, p.num_seqs = bs, L, H, B
        p.seq_slot_of_seq = slot_of_seq.data_ptr()
        p.seq_slot_len = slot_of_seq.numel()
        p.seq_positions = seq_pos.data_ptr()
        p.num_protected = prot.data_ptr()
        p.context_lens = None if context_lens is None else context_lens.data_ptr()
        p.total_slots = N
        p.use_average = 1 if self.use_average else 0
        p.num_sinks = int(self.num_sinks)
        if self._has_bias or float(self.kv_metric_bias_weight) != 0.0:
            hb = self.kv_metric_head_bias
            self._bias_keepalive = (hb.bias.contiguous(), hb.position_bins.contiguous())
            p.bias = self._bias_keepalive[0].data_ptr()
            p.position_bins = self._bias_keepalive[1].data_ptr()
            p.num_bins = self._bias_keepalive[1].numel()
        else:
            p.bias, p.position_bins, p.num_bins = None, None, 0
        p.bias_weight = float(self.kv_metric_bias_weight)
        p.mode = {"reference": 0, "per_sequence": 1}[self.schedule_mode]
        p.null_value = MAX_INT
        p.lean = 3 if self.lean_outputs else 0
        p.sample_stride = int(self.sample_stride)
        p.fallback_grid = int(self.fallback_grid)

    def aggregate_decode_and_harvest(self, seq_indices: List[int], seq_positions: IntsLike, num_protected: IntsLike,
                                     context_lens: torch.Tensor, total_slots: Optional[int] = None,
                                     fuse_clear: bool = True) -> bool:
        """``aggregate_decode`` (reference metrics.py:429-439; the same sums, bit for bit) for a decode step
        whose ``schedule_evictions(seq_indices, seq_positions, ..., context_lens, ..., num_protected)``
        follows: the pass that adds the step's attention to the store also lists, per head, the keys that
        fall below the pivots the previous ``schedule_evictions`` left behind, and the call that follows does
        not stream the store again (kvc_aggregate_decode_harvest, include/kvc_mi355x.h).  Not in the
        reference's surface: calling it is the opt-in (``KVC_HARVEST_AHEAD=0`` makes it a plain ``aggregate_decode``).

        The lists are used only by a ``schedule_evictions`` with these very arguments (tensors: the same
        objects, unmodified; lists: equal values), evictions at most widen / 2 larger than the ones the
        pivots were made for, and the store (metrics, positions, block metadata) not written through this object or torch in
        between -- a writer that goes around both (a custom kernel on ``metrics.data_ptr()``) must not run
        between the two calls.  Tensors made under ``torch.inference_mode()`` keep no version counter, so with a
        store or batch arguments of that kind no lists are made (allocate them outside, as an engine's start-up does).  In every other case this is ``aggregate_decode`` and the schedule call
        takes its usual pass.  Returns whether lists were made."""
        if self.random or not self.record_decoding_metrics:
            return False
        if self.harvest_ahead is None:
            self.harvest_ahead = True
        self._aggs_since_schedule += 1
        try:
            self._poll_fallback(torch.cuda.is_current_stream_capturing())
        except RuntimeError:
            self._hv_lists = None
            self._plain_aggregate_decode(fuse_clear)     # (a fault reported by an EARLIER call: this step's sums still happen)
            raise
        hv = self._hv
        self._hv_lists = None
        stream = _stream(self.metrics)
        ok = (self.harvest_ahead and self._hv_pause == 0 and hv is not None and hv["seqs"] == tuple(int(s) for s in seq_indices)
              and hv["buf"] is self._hv_buf and hv["full"] and hv["stream"] == stream and not self._fb_fault
              # (after a raised flag the next calls take the digit rounds: lists would be paid for and not used)
              and not (self._fb_backoff > 0 and int(self.schedule_path) == 0)
              and isinstance(context_lens, torch.Tensor) and context_lens.is_cuda and context_lens.dtype == torch.int32
              and context_lens.is_contiguous()
              and tuple(context_lens.shape) == (self.num_layers, len(seq_indices), self.num_kv_heads)
              and not torch.cuda.is_current_stream_capturing()
              # (tensors made under torch.inference_mode() keep no version counter: lists made from them could never
              # be trusted by the schedule call, so none are made)
              and self._store_versions() is not None
              and not any(isinstance(x, torch.Tensor) and x.is_inference() for x in (seq_positions, num_protected, context_lens)))
        p = None
        if ok:
            lib = _lib.load()
            seq_pos, prot = self._as_i32(seq_positions), self._as_i32(num_protected)
            p = KvcScheduleParams()
            self._store_params(p, seq_indices, seq_pos, prot, context_lens, int(total_slots) if total_slots else hv["N"])
            p.max_evicted_blocks_hint = int(hv["k"].max())
            p.schedule_path = int(self.schedule_path)
            p.harvest_buf = self._hv_buf.data_ptr()
            ok = bool(lib.kvc_harvest_eligible(ctypes.byref(p), self.num_queries_per_kv))
        if not ok:
            self._plain_aggregate_decode(fuse_clear)
            return False
        with on_device(self.device):
            _lib.check(lib.kvc_aggregate_decode_harvest(
                ctypes.byref(p), self._temp_metrics.data_ptr(), self.num_queries_per_kv,
                1 if self.use_l2 else 0, 1 if fuse_clear else 0, stream))
        self._temp_clean = bool(fuse_clear)
        torch.autograd.graph.increment_version(self.metrics)     # (written through a raw pointer)
        self._hv_lists = dict(seqs=hv["seqs"], seq_pos=self._arg_record(seq_positions), prot=self._arg_record(num_protected),
                              ctx=self._arg_record(context_lens), store=self._store_versions(), k=hv["k"],
                              stream=stream, buf=self._hv_buf)
        return True

    def begin_attention_harvest(self, seq_indices: List[int], seq_positions: IntsLike, num_protected: IntsLike,
                                context_lens: torch.Tensor, attention_seq_indices: Optional[List[int]] = None,
                                total_slots: Optional[int] = None) -> Optional[AttentionHarvest]:
        """The decode step WITHOUT a sweep of the metric store (the reference author's to-do,
        vllm/kvcompress/README.md:32, 49): the fused-metric attention adds a step's weights straight into ``metrics``
        (``_custom_ops.paged_attention_kvc_fused_metrics``: no temp_metrics, no aggregate_decode) and -- given the
        handle this method returns -- its epilogue also makes the candidate lists of the ``schedule_evictions(
        seq_indices, seq_positions, ..., context_lens, ..., num_protected)`` that follows, with the pivots the
        previous schedule call left behind: that call then runs records -> selection -> emission on the lists, and the
        store is not streamed at all that step.

            h = cm.begin_attention_harvest(slots, last_token_positions, protected, ctx, attention_seq_indices=...)
            for l in range(L):      # the model's forward pass
                ops.paged_attention_kvc_fused_metrics(out, cm.metrics, q[l], k_cache[l], v_cache[l], ..., harvest=h, layer=l)
            cm.end_attention_harvest(h)
            cm.schedule_evictions(slots, last_token_positions, evicted_blocks, ctx, hanging, offsets, protected)

        ``seq_positions`` / ``context_lens`` are the SCHEDULE call's (the position of the token sampled this step, the
        context lengths as they are after this step's append: what the fork's scheduler derives at the start of the
        next iteration, scheduler.py:256-280).  ``attention_seq_indices``: the sequence index (``seq_index_by_block``
        value) of every sequence of the attention batch, in its order; default: the compression batch itself.
        Returns None -- and the caller runs the attention without ``harvest=`` -- when lists cannot be made: no pivots
        yet (first step, another batch), a schedule that is not the small-eviction one, keys that are more than the
        sum (use_average, a position bias), stream capture, tensors without version counters.  The reference's
        batch > 1 rule (the fork's default mode) is served: the epilogue then also counts every head's masked slots.  Same contract as ``aggregate_decode_and_harvest``: the
        lists are used only by a schedule call with these very arguments and an untouched store in between; anything
        else, and lists that fall short, take the usual pass / are redone on the device.  Results are identical."""
        if self.random or not self.record_decoding_metrics:
            return None
        if self.harvest_ahead is None:
            self.harvest_ahead = True
        capturing = torch.cuda.is_current_stream_capturing()
        self._poll_fallback(capturing)
        hv = self._hv
        self._hv_lists = None
        stream = _stream(self.metrics)
        B = len(seq_indices)
        ok = (self.harvest_ahead and self._hv_pause == 0 and hv is not None and hv["seqs"] == tuple(int(s) for s in seq_indices)
              and hv["buf"] is self._hv_buf and hv["full"] and hv["stream"] == stream and not self._fb_fault
              and not (self._fb_backoff > 0 and int(self.schedule_path) == 0)
              and isinstance(context_lens, torch.Tensor) and context_lens.is_cuda and context_lens.dtype == torch.int32
              and context_lens.is_contiguous()
              and tuple(context_lens.shape) == (self.num_layers, B, self.num_kv_heads)
              and not capturing and self._store_versions() is not None
              and not any(isinstance(x, torch.Tensor) and x.is_inference() for x in (seq_positions, num_protected, context_lens)))
        if not ok:
            return None
        lib = _lib.load()
        seq_pos, prot = self._as_i32(seq_positions), self._as_i32(num_protected)
        p = KvcScheduleParams()
        self._store_params(p, seq_indices, seq_pos, prot, context_lens, int(total_slots) if total_slots else hv["N"])
        p.max_evicted_blocks_hint = int(hv["k"].max())
        p.schedule_path = int(self.schedule_path)
        p.harvest_buf = self._hv_buf.data_ptr()
        if not lib.kvc_attention_harvest_eligible(ctypes.byref(p)):
            return None
        att = list(seq_indices) if attention_seq_indices is None else [int(x) for x in attention_seq_indices]
        pos_of = {int(s): i for i, s in enumerate(seq_indices)}
        seq_slot = self._as_i32([pos_of.get(int(s), -1) for s in att])
        with on_device(self.device):
            _lib.check(lib.kvc_attention_harvest_begin(ctypes.byref(p), stream))
        record = dict(seqs=hv["seqs"], attention=True, k=hv["k"], stream=stream, buf=self._hv_buf)
        self._hv_attention = AttentionHarvest(self, self._hv_buf, seq_slot, seq_pos, prot, B, stream, record)
        return self._hv_attention

    def end_attention_harvest(self, harvest: Optional[AttentionHarvest]) -> bool:
        """Behind the last attention launch of the step: the lists belong to the schedule call that follows (see
        ``begin_attention_harvest``).  Returns whether they will be offered to it -- not if a layer is missing, the
        handle is not the one of the last ``begin_attention_harvest``, or the stream changed."""
        ok = (harvest is not None and harvest is getattr(self, "_hv_attention", None) and harvest.cm is self
              and harvest.layers_done == set(range(self.num_layers)) and harvest.buf is self._hv_buf
              and harvest.stream == _stream(self.metrics))
        self._hv_attention = None
        if not ok:
            self._hv_lists = None
            return False
        self._hv_lists = dict(harvest.record, store=self._store_versions())
        return True

    def _lists_usable(self, hl, seq_indices, seq_positions, num_protected, context_lens, k_list, stream) -> bool:
        # (lists made by the attention's epilogue carry the context lengths, positions and protected windows they were
        # made with, and the schedule call checks them on the device -- the fork's scheduler builds these tensors anew
        # for every call, scheduler.py:256-280; lists made by the aggregation pass are tied to the argument objects)
        return (hl is not None and hl["buf"] is self._hv_buf and hl["stream"] == stream
                and hl["seqs"] == tuple(int(s) for s in seq_indices)
                and (hl.get("attention", False)
                     or (self._arg_same(hl["seq_pos"], seq_positions) and self._arg_same(hl["prot"], num_protected)
                         and self._arg_same(hl["ctx"], context_lens)))
                and hl["store"] is not None and hl["store"] == self._store_versions()
                and self._k_within(k_list, hl["k"]))

    def _k_within(self, k_list, k_then) -> bool:
        """the pivots aim at (1 + widen) x what the step before needed: half of that allowance may go to a sequence
        that frees more blocks than it did then, the rest is for the keys the attention lifts"""
        if k_list.shape != k_then.shape:
            return False
        return bool(np.all(k_list <= np.maximum(k_then, (k_then * (1.0 + 0.5 * self.harvest_widen)).astype(np.int64))))

    FB_RING = 16

    def _poll_fallback(self, capturing: bool) -> None:
        """The flag words of earlier small-eviction / bracket calls, each in its own page-locked word -- stored there by
        the call's last launch itself (kvc_schedule_params.flag_mirror, ABI version 8: (ticket << 8) | flag) or, for the one
        chain that does not end in that launch, copied there behind the call: looked at (never waited for) by the next
        calls of this object, in the order the calls were made -- EVERY call's flag is seen (the back-off after a redone
        call and the pause / widening of predicted pivots count all of them), at the latest FB_RING calls later."""
        if capturing:
            return
        while self._fb_inflight and self._flag_arrived(self._fb_inflight[0]):
            slot, _, predicted, ticket = self._fb_inflight.pop(0)
            word = int(self._fb_np[slot])
            if ticket is not None:
                word &= 0xFF
            if word & 2:
                self._raise_fallback_fault("an earlier")
            self._note_flag(word, predicted)

    def _flag_arrived(self, entry) -> bool:
        slot, event, _, ticket = entry
        if ticket is None:
            return event.query()
        return (int(self._fb_np[slot]) >> 8) & 0xFFFFFF == ticket

    def _flag_slot(self) -> int:
        """a page-locked word nobody is waiting on"""
        if self._fb_pin is None:
            with torch.inference_mode(False):
                self._fb_pin = torch.zeros(self.FB_RING, dtype=torch.int32).pin_memory()
            self._fb_np = self._fb_pin.numpy()
        busy = {e[0] for e in self._fb_inflight}
        if len(busy) >= self.FB_RING:              # (FB_RING calls enqueued and none has run yet: wait for the oldest)
            if self._fb_inflight[0][3] is None:
                self._fb_inflight[0][1].synchronize()
            else:
                torch.cuda.synchronize(self.device)
            self._poll_fallback(False)
            busy = {e[0] for e in self._fb_inflight}
        return next(i for i in range(self.FB_RING) if i not in busy)

    def _mirror_flag(self, p) -> Tuple[int, int]:
        """the call `p` describes reports its flag word itself: slot and ticket for ``_fb_inflight`` once it is enqueued"""
        slot = self._flag_slot()
        self._fb_ticket = (self._fb_ticket % 0xFFFFFF) + 1
        p.flag_mirror = self._fb_pin.data_ptr() + 4 * slot
        p.flag_ticket = self._fb_ticket
        return slot, self._fb_ticket

    def _watch_flag(self, ws: torch.Tensor, off: int, predicted: bool) -> None:
        """copy this call's flag word to a free page-locked word behind the call (no synchronisation)"""
        slot = self._flag_slot()
        with on_device(self.device):
            self._fb_pin[slot:slot + 1].copy_(ws[off:off + 4].view(torch.int32), non_blocking=True)
            event = torch.cuda.Event()
            event.record(torch.cuda.current_stream(ws.device.index))     # (an explicit index: see _custom_ops._stream)
        self._fb_inflight.append((slot, event, predicted, None))

    def _note_flag(self, word: int, predicted: bool) -> None:
        """what a call's flag word means for the calls to come (predicted: it ran on harvested lists or on
        remembered pivots)"""
        if predicted:
            if word:
                # lists that fell short (the device redid the call, at about three times the cost of a call that
                # samples): wider pivots, made anew by a usual pass -- and predicted pivots are left alone for 2, 4,
                # ... 256 calls, so that a workload whose attention outruns every allowance costs a redone call
                # now and then, not every other call
                self.harvest_misses += 1
                self.harvest_widen = min(8.0, max(0.5, 2.0 * self.harvest_widen))
                self._hv = self._hv_lists = None
                self._hv_streak = 0
                self._hv_pause_len = min(max(2 * self._hv_pause_len, 2), 256)
                self._hv_pause = self._hv_pause_len
            else:
                self._hv_streak += 1
                if self._hv_streak >= 64:              # a long run without a miss: back towards the configured allowance
                    self._hv_streak = 0
                    self._hv_pause_len = 0
                    self.harvest_widen = max(self._hv_widen0, 0.5 * self.harvest_widen)
        elif int(self.schedule_path) == 0:
            self._fb_penalty = min(max(2 * self._fb_penalty, 1), 64) if word else 0
            self._fb_backoff = self._fb_penalty

    # ------------------------------------------------------------------ scheduling
    def _as_i32(self, x: IntsLike) -> torch.Tensor:
        """int32 device tensor; small host lists are uploaded once and cached by value (the
        scheduler passes the same protected-window / slot lists step after step)."""
        if isinstance(x, torch.Tensor):
            return x.to(device=self.device, dtype=torch.int32).contiguous()
        key = tuple(int(v) for v in x)
        hit = self._small_cache.get(key)
        if hit is None:
            if len(self._small_cache) > 256:
                self._small_cache.clear()
            hit = torch.tensor(key, dtype=torch.int32, device=self.device)
            self._small_cache[key] = hit
        return hit

    def _slot_map(self, seq_indices) -> torch.Tensor:
        key = ("slots",) + tuple(int(s) for s in seq_indices)
        hit = self._small_cache.get(key)
        if hit is None:
            if len(self._small_cache) > 256:
                self._small_cache.clear()
            m = torch.full((max(seq_indices) + 1,), -1, dtype=torch.int32)
            m[torch.tensor(list(seq_indices), dtype=torch.long)] = torch.arange(
                len(seq_indices), dtype=torch.int32)
            hit = m.to(self.device)
            self._small_cache[key] = hit
        return hit

    def _batch_summary_enqueue(self, context_lens: torch.Tensor, k_per_seq: Optional[torch.Tensor], stream: int,
                               bound: Optional[int] = None) -> int:
        """``N`` and ``evicted_blocks_per_seq`` of a batch the caller describes with device tensors -- what the fork's
        scheduler passes (reference scheduler.py:245-247, 491-499: a device int tensor of counts, no N) -- on their way
        to the host: one launch that writes into page-locked memory (kvc_schedule_batch_summary).
        ``_batch_summary_read`` waits for it; whatever the call can prepare without the numbers is done in between.
        The reference method waits for the device for the same reason, many times over (metrics.py:465-489, 709-729)."""
        lib = _lib.load()
        B = 0 if k_per_seq is None else int(k_per_seq.numel())
        if self._summary_pin is None or self._summary_pin.numel() < 2 + B:
            with torch.inference_mode(False):
                self._summary_pin = torch.zeros((max(2 + B, 320),), dtype=torch.int64).pin_memory()
            self._summary_np = self._summary_pin.numpy()
        # the kernel stores the numbers and then a ticket (system-scope release) into the page-locked buffer: the host
        # polls the ticket word instead of waiting for the stream's end-of-kernel signal (ABI version 7)
        self._summary_ticket += 1
        with on_device(self.device):
            if bound is None:
                _lib.check(lib.kvc_schedule_batch_summary_ticket(
                    context_lens.data_ptr(), int(context_lens.numel()), int(self.block_size),
                    None if k_per_seq is None else k_per_seq.data_ptr(), B, self._summary_pin.data_ptr(),
                    self._summary_ticket, stream))
            else:       # (ABI version 8: N also where the kernels behind this launch read it)
                _lib.check(lib.kvc_schedule_batch_summary_deferred(
                    context_lens.data_ptr(), int(context_lens.numel()), int(self.block_size),
                    None if k_per_seq is None else k_per_seq.data_ptr(), B, self._summary_pin.data_ptr(),
                    self._summary_ticket, self._dn_words(stream).data_ptr(), int(bound), stream))
        return B

    def _batch_summary_read(self, B: int, stream: int):
        word, ticket = self._summary_np[1 + B:2 + B], self._summary_ticket
        deadline = None
        spins = 0
        while int(word[0]) != ticket:
            spins += 1
            if spins & 0x3FFF == 0:             # (a launch that failed never writes the ticket: ask the runtime now and then)
                if deadline is None:
                    deadline = time.monotonic() + 30.0
                _lib.check(_lib.load().kvc_schedule_batch_summary_wait(stream))
                if int(word[0]) != ticket and time.monotonic() > deadline:
                    raise RuntimeError("schedule_evictions: the batch summary never arrived")
        return int(self._summary_np[0]), self._summary_np[1:1 + B].copy()

    _DN_GRAIN = 1 << 16

    def _dn_words(self, stream: int) -> torch.Tensor:
        """the two device words (N, void) of calls enqueued on `stream` (stream order keeps one call's apart from the next's)"""
        if self._dn_dev is None:
            self._dn_dev = {}
        t = self._dn_dev.get(stream)
        if t is None:
            t = self._dn_dev[stream] = torch.zeros((2,), dtype=torch.int64, device=self.device)
        return t

    def _dn_note(self, B: int, N: int, p) -> None:
        """what a call's true N and counts say for the next call of a batch of this size (deferred N, __init__)"""
        # (a quarter over N, in steps that are fine where the schedules' thresholds are: the library picks the schedule
        # from the bound, and a bracket schedule on a batch that is too small for it is redone by its fallback)
        grain = self._DN_GRAIN if N >= (1 << 20) else 1024
        grain = grain // math.gcd(grain, self.block_size) * self.block_size
        want = (N + N // 4 + grain - 1) // grain * grain
        if want > self._dn_bound.get(B, 0):
            self._dn_bound[B] = want
        if int(p.max_evicted_blocks_hint) >= 0:
            self._dn_plan[B] = int(_lib.load().kvc_schedule_evictions_plan(ctypes.byref(p)))

    def _schedule_evictions_deferred(self, seq_indices, seq_positions, k_per_seq, context_lens, hanging_token_count,
                                     evicted_kv_offsets, num_protected, stream: int):
        """``schedule_evictions`` for the fork's call without a wait in front of the launches (see ``deferred_n``).
        Returns the three outputs -- or ``(N, counts)`` when the bound did not hold and the device did nothing."""
        lib = _lib.load()
        bs, L, H, B = self.block_size, self.num_layers, self.num_kv_heads, len(seq_indices)
        dev = self.device
        bound = int(self._dn_bound[B])
        if bound >= 2147483647 - self._DN_GRAIN:
            bound = (2147483646 // bs) * bs
        pending = self._batch_summary_enqueue(context_lens, k_per_seq, stream, bound=bound)
        seq_pos = self._as_i32(seq_positions)
        prot = self._as_i32(num_protected)
        out_idx = torch.empty((bound,), dtype=torch.int32, device=dev)
        out_two = torch.empty((2, B, L, H), dtype=torch.int32, device=dev)       # (one allocation for the two count tensors)
        out_kv, out_blk = out_two[0], out_two[1]
        p = KvcScheduleParams()
        self._store_params(p, seq_indices, seq_pos, prot, context_lens, bound)
        p.evicted_blocks_per_seq = k_per_seq.data_ptr()
        p.hanging_token_count = hanging_token_count.data_ptr()
        p.evicted_kv_offsets = evicted_kv_offsets.data_ptr()
        p.schedule_path = int(self.schedule_path)
        p.uniform_evict = 0
        if p.schedule_path == 0 and self._fb_backoff > 0:
            self._fb_backoff -= 1
            p.schedule_path = 1
        p.block_tables, p.seq_index_of_slot, p.max_num_seqs, p.block_tables_width = None, None, 0, 0
        p.harvest_buf, p.harvest, p.harvest_widen, p.eli_dirty_map = None, 0, float(self.harvest_widen), None
        p.max_evicted_blocks_hint = -1
        p.total_slots_dev = self._dn_words(stream).data_ptr()
        mirror = self._mirror_flag(p)
        p.evicted_logical_indices = out_idx.data_ptr()
        p.evicted_kv_count = out_kv.data_ptr()
        p.evicted_block_count = out_blk.data_ptr()
        self._hv_lists = None          # (lists made for this call belong to the small-eviction schedule: not used here)
        self._hv = None
        self.last_harvest_used = self.last_pivot_memory_used = False
        self._spec_made = False
        key = (bound, B, int(p.schedule_path))
        known = self._dn_sizes.get(key)
        if known is None:
            known = self._dn_sizes[key] = (int(lib.kvc_schedule_evictions_plan(ctypes.byref(p))),
                                           int(lib.kvc_schedule_evictions_workspace_bytes(bound, B * L * H, B, bs)),
                                           int(lib.kvc_schedule_evictions_fallback_offset(bound, B * L * H, B, bs)))
        plan, ws_bytes, fb_off = known
        ws = workspace(dev, ws_bytes, "schedule_evictions")
        with on_device(dev):
            _lib.check(lib.kvc_schedule_evictions(ctypes.byref(p), ws.data_ptr(), ws.numel(), stream))
        # ---- the device is at work; N and the counts arrive meanwhile
        self._poll_fallback(False)
        n_read, k_read = self._batch_summary_read(pending, stream)
        self.deferred_calls += 1
        p.total_slots_dev = None
        p.total_slots, p.max_evicted_blocks_hint = int(n_read), int(k_read.max())
        self._dn_note(B, int(n_read), p)
        if n_read > bound:
            self.deferred_voided += 1
            return int(n_read), k_read
        self.last_used_block_tables = False
        self.last_schedule = (ws, fb_off, plan)
        p.total_slots, p.max_evicted_blocks_hint = bound, -1
        self.last_schedule_reason = self._describe_plan(
            plan, int(lib.kvc_schedule_evictions_plan_reason(ctypes.byref(p))),
            backoff=int(p.schedule_path) == 1 and int(self.schedule_path) != 1) + " [N on the device]"
        if plan:
            self._fb_inflight.append((mirror[0], None, False, mirror[1]))
        return out_idx[:int(n_read)], out_kv, out_blk

    def schedule_evictions(
        self,
        seq_indices: List[int],
        seq_positions: IntsLike,
        evicted_blocks_per_seq: IntsLike,
        context_lens: torch.Tensor,
        hanging_token_count: torch.Tensor,
        evicted_kv_offsets: torch.Tensor,
        num_protected: IntsLike,
        uniform_evict: bool = False,
        debug={},
        profile=False,
        total_slots: Optional[int] = None,
        block_tables: Optional[torch.Tensor] = None,
    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """reference metrics.py:441-847.  Returns ``(evicted_logical_indices [N] i32,
        evicted_kv_count [B,L,H] i32, evicted_block_count [B,L,H] i32)``.

        ``evicted_blocks_per_seq``: what the fork's scheduler passes is a device int tensor
        (scheduler.py:245-247), with no N next to it; the reference's own test harnesses pass lists.
        Both forms reach every schedule: the tensor form costs one launch and one wait that brings N
        and the counts to the host together (``_batch_summary``; the reference method waits for the
        device many times, metrics.py:465-489, 709-729), after which the call is the list form's.

        ``total_slots`` (optional, not in the reference signature): N if the caller
        already knows it.  With it AND a host list of counts the method never waits for the device;
        under stream capture that is the only form there is (a captured call with a device tensor
        of counts cannot look at them: digit rounds).

        ``block_tables`` (optional, not in the reference signature): ``BlockState.block_tables``
        ``[L, max_num_seqs, H, M]`` (rows indexed by sequence index).  The fork's scheduler has it
        next to the ``context_lens`` it already passes.  The digit-round and bracket schedules build
        their keys in logical order through it when the batch takes less than half of the cache (an
        engine-sized cache): three scattered accesses per block of the batch instead of a sweep over
        every block's metadata; the small-eviction schedule streams the store in physical order and
        ignores it (kvc_schedule_params.block_tables, ABI version 3).  Contract: the tables must be
        the ones the per-block metadata (seq / layer / head / logical block number) was written
        from -- only the owning sequence of a listed block is checked, so stale tables give
        different evictions without an error.  With consistent state the results are identical."""
        assert len(seq_indices) > 0
        assert list(sorted(seq_indices)) == list(seq_indices), (
            "schedule_evictions input not ordered by ascending index")
        lib = _lib.load()
        bs, L, H, B = self.block_size, self.num_layers, self.num_kv_heads, len(seq_indices)
        dev = self.device
        for n, t in (("context_lens", context_lens), ("hanging_token_count", hanging_token_count),
                     ("evicted_kv_offsets", evicted_kv_offsets)):
            if not t.is_cuda or t.dtype != torch.int32:
                raise RuntimeError(f"schedule_evictions: {n} must be an int32 tensor on a HIP device")
        context_lens = context_lens.contiguous()
        hanging_token_count = hanging_token_count.contiguous()
        evicted_kv_offsets = evicted_kv_offsets.contiguous()
        assert tuple(context_lens.shape) == (L, B, H)
        assert tuple(evicted_kv_offsets.shape) == (B, L, H)
        capturing = torch.cuda.is_current_stream_capturing()
        stream = _stream(self.metrics)
        self._last_gap, self._aggs_since_schedule = max(self._aggs_since_schedule, 1), 0
        k_per_seq = self._as_i32(evicted_blocks_per_seq)
        # the counts on the host (their maximum picks the schedule, include/kvc_mi355x.h) and N
        k_list, pending = None, None
        if (self.deferred_n and not capturing and total_slots is None and block_tables is None and not uniform_evict
                and isinstance(evicted_blocks_per_seq, torch.Tensor) and evicted_blocks_per_seq.is_cuda
                and (self.schedule_mode == "per_sequence" or B == 1) and self._dn_plan.get(B, 1) != 1
                and self._dn_bound.get(B, 0) > 0 and not self._fb_fault and not self.strict_fallback
                and int(self.schedule_path) in (0, 1, 4) and not self.lean_outputs):
            done = self._schedule_evictions_deferred(seq_indices, seq_positions, k_per_seq, context_lens, hanging_token_count,
                                                     evicted_kv_offsets, num_protected, stream)
            if isinstance(done, tuple) and len(done) == 3:
                return done
            total_slots, k_list = done         # (voided: N and the counts are known now -- the usual way, without a wait)
        if k_list is not None:
            pass
        elif isinstance(evicted_blocks_per_seq, torch.Tensor):
            if not evicted_blocks_per_seq.is_cuda:
                k_list = np.asarray(evicted_blocks_per_seq.tolist(), dtype=np.int64)
            elif not capturing:
                # the fork's call (scheduler.py:245-247, 491-499): a device tensor, no N -- both in one wait, further down
                pending = self._batch_summary_enqueue(context_lens, k_per_seq, stream)
        else:
            k_list = np.asarray([int(v) for v in evicted_blocks_per_seq], dtype=np.int64)
        if total_slots is None and pending is None:
            if capturing:
                raise RuntimeError("schedule_evictions: under stream capture pass total_slots= (N cannot be read "
                                   "back from context_lens while the stream is being captured)")
            pending = self._batch_summary_enqueue(context_lens, None, stream)

        # ---- everything that does not depend on N or the counts (the device is busy with the summary meanwhile)
        seq_pos = self._as_i32(seq_positions)
        prot = self._as_i32(num_protected)
        out_idx = None                 # (made below, once the schedule is known)
        out_kv = torch.empty((B, L, H), dtype=torch.int32, device=dev)
        out_blk = torch.empty((B, L, H), dtype=torch.int32, device=dev)

        p = KvcScheduleParams()
        self._store_params(p, seq_indices, seq_pos, prot, context_lens, 0)
        p.evicted_blocks_per_seq = k_per_seq.data_ptr()
        p.hanging_token_count = hanging_token_count.data_ptr()
        p.evicted_kv_offsets = evicted_kv_offsets.data_ptr()
        p.schedule_path = int(self.schedule_path)
        # the reference's other selection rule (metrics.py:639-666; its scheduler never passes it)
        p.uniform_evict = 1 if uniform_evict else 0
        self._poll_fallback(capturing)
        if self._fb_fault:
            p.schedule_path = 1        # the launch chain of the digit rounds has no device-side waits
        elif p.schedule_path == 0 and self._fb_backoff > 0 and not capturing:
            self._fb_backoff -= 1
            p.schedule_path = 1
        if block_tables is not None:
            if (not block_tables.is_cuda or block_tables.dtype != torch.int32 or block_tables.dim() != 4
                    or not block_tables.is_contiguous() or block_tables.shape[0] != L or block_tables.shape[2] != H):
                raise RuntimeError("schedule_evictions: block_tables must be a contiguous int32 HIP tensor "
                                   "[L, max_num_seqs, H, M]")
            if max(seq_indices) >= block_tables.shape[1]:
                raise RuntimeError("schedule_evictions: a sequence index lies outside block_tables")
            if os.environ.get("KVC_DEBUG_TABLES", "0") not in ("", "0"):
                self.check_block_tables(block_tables, seq_indices, context_lens)
            p.block_tables = block_tables.data_ptr()
            p.seq_index_of_slot = self._as_i32([int(s) for s in seq_indices]).data_ptr()
            p.max_num_seqs, p.block_tables_width = int(block_tables.shape[1]), int(block_tables.shape[3])
        else:
            p.block_tables, p.seq_index_of_slot = None, None
            p.max_num_seqs, p.block_tables_width = 0, 0
        # harvest-ahead: lists made by aggregate_decode_and_harvest for exactly this call / pivots for the next one
        hl, self._hv_lists = self._hv_lists, None
        p.harvest_buf, p.harvest, p.harvest_widen = None, 0, float(self.harvest_widen)
        p.eli_dirty_map = None
        seqs_key = tuple(int(x) for x in seq_indices)
        p.evicted_kv_count = out_kv.data_ptr()
        p.evicted_block_count = out_blk.data_ptr()

        # ---- N and the counts are needed from here on
        if pending is not None:
            n_read, k_read = self._batch_summary_read(pending, stream)
            if total_slots is None:
                total_slots = n_read
            if k_list is None:
                k_list = k_read
        N = int(total_slots)
        p.total_slots = N
        # the largest count picks the schedule (include/kvc_mi355x.h); unknown only for a device tensor of counts
        # under stream capture, where nothing can be read back
        p.max_evicted_blocks_hint = -1 if k_list is None else int(k_list.max())
        self._dn_note(B, N, p)
        # (lists are wanted once somebody harvests explicitly -- or from the start when aggregate_decode() may harvest
        # ahead of the call by itself)
        want_lists = self._wants_lists()
        if (not capturing and p.max_evicted_blocks_hint >= 0
                and ((want_lists and lib.kvc_harvest_eligible(ctypes.byref(p), self.num_queries_per_kv))
                     or (self.pivot_memory and lib.kvc_pivot_memory_eligible(ctypes.byref(p))))):
            # the buffer: pivots only, or pivots + lists once somebody harvests
            full = want_lists
            need = int(lib.kvc_harvest_buffer_bytes(B * L * H, B) if full else lib.kvc_harvest_pivot_bytes(B))
            have = 0 if self._hv_buf is None else self._hv_buf.numel()
            if full and need > max(have, self.harvest_buffer_max_bytes):
                # lists for this batch would not fit the reservation or the budget (__init__): pivots only, the call
                # makes its lists in its own collecting pass
                full, need = False, int(lib.kvc_harvest_pivot_bytes(B))
            if self._hv_buf is None or self._hv_buf.numel() < need:      # (kept when the batch shrinks: offsets are the call's)
                self._hv_buf = torch.zeros((need,), dtype=torch.uint8, device=dev)
                self._hv = hl = None
            p.harvest_buf = self._hv_buf.data_ptr()
            p.harvest = 2
            hv = self._hv
            if self._hv_pause > 0:
                self._hv_pause -= 1
            elif self._lists_usable(hl, seq_indices, seq_positions, num_protected, context_lens, k_list, stream):
                p.harvest |= 9 if hl.get("attention", False) else 1
                self.last_harvest_kind = ("aggregation pass, ahead of the call" if hl.get("speculative", False) else
                                          "attention's epilogue" if hl.get("attention", False) else "aggregation pass")
            elif (self.pivot_memory and hv is not None and hv["buf"] is self._hv_buf and hv["stream"] == stream
                  and hv["seqs"] == seqs_key and self._k_within(k_list, hv["k"])):
                p.harvest |= 4
            self._hv = dict(seqs=seqs_key, k=k_list, N=N, buf=self._hv_buf, stream=stream, full=full,
                            seq_pos=seq_pos, prot=prot)       # (held: what aggregate_decode() predicts the next call from)
        else:
            self._hv = None
        self.last_harvest_used = bool(p.harvest & 1)
        self.last_pivot_memory_used = bool(p.harvest & 4)
        made, self._spec_made = self._spec_made, False
        if self._spec_pause > 0:
            self._spec_pause -= 1
        if made:
            if (p.harvest & 1) and hl is not None and hl.get("speculative", False):
                self._spec_unused = self._spec_pause_len = 0
            else:
                self._spec_unused += 1
                if self._spec_unused >= 3:
                    self._spec_unused = 0
                    self._spec_pause_len = min(max(2 * self._spec_pause_len, 4), 256)
                    self._spec_pause = self._spec_pause_len
        plan = int(lib.kvc_schedule_evictions_plan(ctypes.byref(p)))
        if self.reuse_output_buffer and not self.lean_outputs and N > 0 and not capturing and plan == 1:
            out_idx = self._tracked_output(N, bs, p)
        if out_idx is None:
            out_idx = torch.empty((N,), dtype=torch.int32, device=dev)
        p.evicted_logical_indices = out_idx.data_ptr()

        ws_bytes = lib.kvc_schedule_evictions_workspace_bytes(N, B * L * H, B, bs)
        ws = workspace(dev, ws_bytes, "schedule_evictions")
        # (the call's last launch reports its flag word itself -- except the one chain that does not end in it)
        mirror = None
        if plan and not capturing and not self.strict_fallback and not (plan == 1 and int(p.mode) == 0 and B > 256):
            mirror = self._mirror_flag(p)
        with on_device(dev):
            _lib.check(lib.kvc_schedule_evictions(ctypes.byref(p), ws.data_ptr(), ws.numel(), stream))
        self.last_used_block_tables = bool(lib.kvc_schedule_evictions_uses_block_tables(ctypes.byref(p)))
        self.last_schedule = (ws, int(lib.kvc_schedule_evictions_fallback_offset(N, B * L * H, B, bs)), plan)
        self.last_schedule_reason = self._describe_plan(
            self.last_schedule[2], int(lib.kvc_schedule_evictions_plan_reason(ctypes.byref(p))),
            backoff=int(p.schedule_path) == 1 and int(self.schedule_path) != 1)
        if self.last_schedule[2] == 1:
            # ... and where the small-eviction schedule took its pivots / lists from
            self.last_schedule_reason += (f" [lists: the {self.last_harvest_kind}]" if p.harvest & 1 else
                                          " [pivots: the call before]" if p.harvest & 4 else " [pivots: sampled]")
        if self.last_schedule[2] and self.strict_fallback and not capturing:
            off = self.last_schedule[1]
            word = int(ws[off:off + 4].view(torch.int32).item())
            if word & 2:
                self._raise_fallback_fault("this")
            self._note_flag(word, bool(p.harvest & 5))
        elif mirror is not None:
            self._fb_inflight.append((mirror[0], None, bool(p.harvest & 5), mirror[1]))
        elif self.last_schedule[2] and not capturing:
            self._watch_flag(ws, self.last_schedule[1], bool(p.harvest & 5))
        return out_idx, out_kv, out_blk

    @staticmethod
    def _storage_refs(t: torch.Tensor) -> Optional[int]:
        """how many tensors share ``t``'s storage, or None when this torch cannot say (``torch._C._storage_Use_Count``
        is a private API): without it nothing proves that the previous result was dropped, and the kept buffer is
        never handed out twice"""
        fn = getattr(torch._C, "_storage_Use_Count", None)
        if fn is None:
            return None
        return int(fn(t.untyped_storage()._cdata))

    def _tracked_output(self, N: int, bs: int, p) -> Optional[torch.Tensor]:
        """evicted_logical_indices [N] as a view of the buffer this object keeps for the small-eviction
        schedule (see __init__), with its dirty map in ``p.eli_dirty_map``; None = no such buffer for this call (a
        fresh tensor and the full padding).  The buffer is handed out again only when (a) nobody else refers to its
        storage any more and (b) its version counter is where this object left it: a caller may edit the list the
        reference returns in place before dropping it (it is a fresh tensor there), which leaves entries the dirty map
        does not know about -- torch counts such writes, and a buffer that was written to is replaced.  Never under
        stream capture (the caller decides; a buffer baked into a graph would be overwritten by its replays)."""
        rec = self._eli_buf
        stream = _stream(self.metrics)
        if rec is not None:
            buf, dmap, rec_bs, refs, rec_stream, version = rec
            if (buf.numel() < N or rec_bs != bs or rec_stream != stream or refs is None
                    or self._storage_refs(buf) != refs or buf._version != version):
                rec = None              # too small, another block size / stream, a previous result still alive or edited
        if rec is None:
            cap = (N + N // 16 + 4095) // 4096 * 4096          # the batch grows and shrinks by blocks: some slack
            with torch.inference_mode(False):                  # (a tensor with a version counter, whoever calls)
                buf = torch.full((cap,), MAX_INT, dtype=torch.int32, device=self.device)
                dmap = torch.zeros(((cap // bs + 31) // 32 + 1,), dtype=torch.int32, device=self.device)
            refs = self._storage_refs(buf)
            if refs is None:
                return None
            rec = self._eli_buf = (buf, dmap, bs, refs, stream, buf._version)
        p.eli_dirty_map = rec[1].data_ptr()
        return rec[0][:N]

    def check_block_tables(self, block_tables: torch.Tensor, seq_indices, context_lens: torch.Tensor) -> None:
        """Debug aid (``KVC_DEBUG_TABLES=1`` runs it inside every ``schedule_evictions(block_tables=...)``;
        synchronises): the contract of the optional ``block_tables=`` argument is that the tables are the
        ones the per-block metadata was written from -- the key pass through the tables only checks the
        owning sequence of a listed block.  Here every listed block of the batch is checked against all
        four metadata rows (sequence, layer, head, logical block number); raises ``RuntimeError`` with the
        first offender."""
        L, H, bs = self.num_layers, self.num_kv_heads, self.block_size
        M = block_tables.shape[3]
        sel = torch.tensor([int(s) for s in seq_indices], device=self.device, dtype=torch.long)
        bt = block_tables[:, sel].long()                                        # [L, B, H, M]
        nblk = ((context_lens.long() + bs - 1) // bs)                            # [L, B, H]
        live = torch.arange(M, device=self.device)[None, None, None, :] < nblk[..., None]
        blk = bt[live]
        l_i, b_i, h_i, m_i = torch.nonzero(live, as_tuple=True)
        ok = ((blk >= 0) & (blk < self.num_blocks))
        safe = blk.clamp(0, self.num_blocks - 1)
        ok &= self.seq_index_by_block[safe].long() == sel[b_i]
        ok &= self.layer_index_by_block[safe].long() == l_i
        ok &= self.head_index_by_block[safe].long() == h_i
        ok &= self.logical_block_num_by_block[safe].long() == m_i
        if not bool(ok.all()):
            j = int(torch.nonzero(~ok)[0])
            raise RuntimeError(
                f"block_tables[{int(l_i[j])}, seq {int(sel[b_i[j]])}, {int(h_i[j])}, {int(m_i[j])}] = {int(blk[j])} does not "
                "match that block's metadata (sequence / layer / head / logical block number): the tables passed to "
                "schedule_evictions are not the ones the metadata was written from")

    def _raise_fallback_fault(self, which: str) -> None:
        """Bit 1 of the flag word: the single launch that redoes a call on the general pipeline gave
        up a wait (ten seconds without progress: a device fault, not contention -- its phases wait
        for work, not for co-residency) and overwrote that call's outputs with the schedule that
        evicts nothing.  The reference always returns a valid schedule (metrics.py:441-847); the
        closest a faulting device allows is to say so, and to keep to the launch chain afterwards."""
        self._fb_fault = True
        self._fb_backoff = self._fb_penalty = 0
        raise RuntimeError(
            f"schedule_evictions: the on-device fallback of {which} call gave up a wait (device fault); "
            "that call's outputs were replaced by an empty schedule (nothing evicted). Further calls "
            "use the digit-round launch chain only.")

    @staticmethod
    def _describe_plan(plan: int, reason: int, backoff: bool) -> str:
        """Human-readable form of kvc_schedule_evictions_plan_reason (include/kvc_mi355x.h)."""
        why = _lib.WHY
        if plan == 1:
            tail = " (fallback: gated launch chain, coupled_batch)" if (reason >> 16) & 0xFF else ""
            return "small_eviction" + tail
        small = why.get(reason & 0xFF, "?")
        if backoff and small == "forced_path":
            small = "backoff_after_fallback"
        if plan == 2:
            return f"bracket (small_eviction: {small})"
        return f"general (small_eviction: {small}; bracket: {why.get((reason >> 8) & 0xFF, '?')})"

    def last_schedule_path(self) -> str:
        """Which schedule produced the last ``schedule_evictions`` result (synchronises; tests and
        bench.py): "general", "small_eviction" or "bracket" -- the latter two with "+fallback" when
        the schedule could not finish exactly and the general pipeline behind it redid the work."""
        ws, off, plan = self.last_schedule
        if not plan:
            return "general"
        name = {1: "small_eviction", 2: "bracket"}[plan]
        flag = int(ws[off:off + 4].view(torch.int32).item())
        if flag & 2:        # the single-launch fallback gave up a wait (device fault): the outputs evict nothing
            return name + "+fallback+wait_timeout"
        return name if flag == 0 else name + "+fallback"

    def profile_schedule_evictions(self):
        """reference metrics.py:277-335: peak extra device memory of one
        ``schedule_evictions`` call over ``max_kv_per_sort`` slots."""
        assert self.num_blocks is None, "cannot profile after initialization"
        bs = self.block_size
        sort_blocks = (self.max_kv_per_sort + bs - 1) // bs
        total_heads = self.num_layers * self.num_kv_heads
        blocks_per_head = (sort_blocks + total_heads - 1) // total_heads
        total_blocks = blocks_per_head * total_heads
        dev = self.device
        self.init_kv_metadata(total_blocks)
        self.metrics.zero_()
        self.token_positions.zero_()
        self.seq_index_by_block[:] = 0
        self.head_index_by_block[:] = (torch.arange(self.num_kv_heads)
                                       .repeat_interleave(blocks_per_head)
                                       .repeat(self.num_layers).to(dev))
        self.layer_index_by_block[:] = (torch.arange(self.num_layers)
                                        .repeat_interleave(self.num_kv_heads * blocks_per_head)
                                        .to(dev))
        self.logical_block_num_by_block[:] = (torch.arange(blocks_per_head)
                                              .repeat(total_heads).to(dev))
        context_lens = torch.full((self.num_layers, 1, self.num_kv_heads), bs * blocks_per_head,
                                  dtype=torch.int, device=dev)
        hanging = torch.full((1, self.num_layers, self.num_kv_heads), bs, dtype=torch.int,
                             device=dev)
        offs = (torch.arange(total_heads, dtype=torch.int, device=dev) * (bs * blocks_per_head)
                ).reshape(1, self.num_layers, self.num_kv_heads)
        torch.cuda.synchronize(dev)
        torch.cuda.reset_peak_memory_stats(dev)
        init_mem = torch.cuda.max_memory_allocated(dev)
        self.schedule_evictions([0], [1], [total_blocks], context_lens, hanging, offs, [32],
                                profile=True)
        torch.cuda.synchronize(dev)
        final_mem = torch.cuda.max_memory_allocated(dev)
        self.clear_kv_metadata()
        return final_mem - init_mem
