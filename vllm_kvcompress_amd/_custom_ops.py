"""KV-Compress section of the reference's ``vllm/_custom_ops.py``, on MI355X.

Same function names, positional order, in-place/out-parameter conventions and error
behaviour as the reference wrappers (``vllm/_custom_ops.py:641-658, 1065-1086,
1158-1179, 1220-1256``); each one calls the HIP kernels of ``libkvc_mi355x.so`` through
its C ABI on the current HIP stream.  Index tensors are int32 and are reinterpreted
without a dtype check exactly like the reference kernels do
(``csrc/kvcompress_eviction_kernels.cu:527-542``) -- except that here a wrong dtype
raises instead of silently reading garbage.
"""
from __future__ import annotations

import os
import weakref
from typing import Optional

import torch

from . import _lib
from ._lib import MAX_INT  # noqa: F401  (re-exported like the reference constant)

_WORKSPACES = {}


def _stream(t: torch.Tensor) -> int:
    # (the raw handle straight from the binding: torch.cuda.current_stream() builds a Stream object per call, and with a
    # device that has no index it walks through torch.cuda.is_available() -- an os.getenv -- every time)
    index = t.device.index
    return torch._C._cuda_getCurrentRawStream(index if index is not None else torch._C._cuda_getDevice())


class on_device:
    """``with torch.cuda.device(d):`` for launches through the C ABI, without that context manager's bookkeeping when
    ``d`` is the current device already -- one process per GPU: always -- (a few microseconds in front of the first launch
    of every op, on a host path that an idle device waits for)"""
    __slots__ = ("idx", "prev")

    def __init__(self, device):
        if isinstance(device, int):
            self.idx = device
        else:
            idx = device.index if isinstance(device, torch.device) else torch.device(device).index
            self.idx = idx if idx is not None else torch._C._cuda_getDevice()

    def __enter__(self):
        self.prev = torch._C._cuda_getDevice()
        if self.prev != self.idx:
            torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev != self.idx:
            torch.cuda.set_device(self.prev)
        return False


def _written(*tensors: torch.Tensor) -> None:
    """The kernels write through raw pointers, which torch's version counters do not see; consumers that
    ask "has anybody written to this tensor since?" (CompressionMetrics' harvest-ahead lists, the tracked
    move table) get the answer they would get after an in-place torch op."""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


def _version_of(t: torch.Tensor) -> Optional[int]:
    """a tensor's version counter, or None for a tensor made under torch.inference_mode() (vLLM's workers run
    under it): those do not keep one, so nothing can be said about who wrote to them -- every "untouched since?"
    question about them is answered with no"""
    return None if t.is_inference() else t._version


def _same_version(recorded: Optional[int], t: torch.Tensor) -> bool:
    return recorded is not None and not t.is_inference() and recorded == t._version


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _require(t: torch.Tensor, name: str, dtype=None) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on a HIP device (no CPU fallback exists)")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")


def workspace(device: torch.device, nbytes: int, tag: str) -> torch.Tensor:
    """Persistent per-(device, stream, tag) scratch buffer, grown geometrically; the C ABI never
    allocates (include/kvc_mi355x.h).  Keyed by the current stream so that calls enqueued on
    different streams never share scratch; a buffer that is outgrown stays referenced by the
    caching allocator's stream ordering until the kernels already enqueued on it have run
    (``record_stream``)."""
    index = device.index if device.index is not None else torch._C._cuda_getDevice()
    key = (index, torch._C._cuda_getCurrentRawStream(index), tag)
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            buf.record_stream(torch.cuda.current_stream(index))
        buf = torch.empty(max(int(nbytes * 1.25), 4096), dtype=torch.uint8,
                          device=torch.device("cuda", index))
        _WORKSPACES[key] = buf
    return buf


def _compiled_binding() -> bool:
    """True if the dispatcher ops come from libkvc_torch.so (torch_ops.register(): its C++ kernels
    keep a scratch cache and an attention schedule of their own)"""
    from . import torch_ops
    return torch_ops._REGISTERED == "compiled"


def reserve_workspace(device: torch.device, nbytes: int, tag: str) -> None:
    """Size a scratch buffer ahead of time (engine start-up / before HIP-graph capture) so that no
    op ever allocates while serving -- the cache of this module and, when the compiled dispatcher
    binding is the one registered, its cache as well (``_kvc_mi355x::reserve_workspace``)."""
    workspace(device, nbytes, tag)
    if _compiled_binding():
        like = torch.empty(0, dtype=torch.uint8, device=device)
        with on_device(device):
            torch.ops._kvc_mi355x.reserve_workspace(like, int(nbytes), tag)


def _contig(t: torch.Tensor, keep: list) -> torch.Tensor:
    """contiguous view of a read-only input; a temporary made here is kept alive in ``keep``
    until the launch that reads it through a raw pointer has been enqueued"""
    c = t.contiguous()
    keep.append(c)
    return c


# ---------------------------------------------------------------------------------------
def count_block_evictions(
    evicted_block_count: torch.Tensor,
    evicted_logical_indices: torch.Tensor,
    evicted_kv_offsets: torch.Tensor,
    hanging_token_count: torch.Tensor,
    block_size: int,
    null_value: int,
    evicted_blocks_per_seq: Optional[torch.Tensor] = None,
) -> None:
    """reference vllm/_custom_ops.py:1065-1086"""
    lib = _lib.load()
    for n, t in (("evicted_block_count", evicted_block_count),
                 ("evicted_logical_indices", evicted_logical_indices),
                 ("evicted_kv_offsets", evicted_kv_offsets),
                 ("hanging_token_count", hanging_token_count)):
        _require(t, n, torch.int32)
    if not evicted_logical_indices.is_contiguous() or not evicted_block_count.is_contiguous():
        raise RuntimeError("count_block_evictions: evicted_logical_indices / evicted_block_count "
                           "must be contiguous (written in place)")
    eli = evicted_logical_indices
    offs = evicted_kv_offsets.contiguous()
    hang = hanging_token_count.contiguous()
    with on_device(eli.device):
        _lib.check(lib.kvc_count_block_evictions(
            evicted_block_count.data_ptr(), eli.data_ptr(), offs.data_ptr(), hang.data_ptr(),
            evicted_block_count.numel(), eli.numel(), int(block_size), int(null_value),
            _stream(eli)))
    _written(evicted_block_count, eli)


# ---- move tables whose contents the package knows -------------------------------------------
# The reference clears the whole [max_kv_per_compression, 2] table in front of every
# schedule_t1_cache_moves (vllm/_custom_ops.py:1168) although that table is the scheduler's own
# persistent workspace (vllm/kvcompress/scheduler.py:74-86): all zeros except the rows the previous
# call wrote.  A table registered here carries a device-side dirty map (one bit per block_size rows);
# schedule_cache_moves then clears only the marked rows -- the table ends up exactly as after the
# reference's fill_(0) + op.  Anything else that writes to the tensor through torch bumps its
# version counter, which sends the next call back to the full fill.
class _TrackedTable:
    __slots__ = ("ref", "ptr", "dirty_map", "block_size", "version")

    def __init__(self, t):
        self.ref, self.ptr = weakref.ref(t), t.data_ptr()
        self.dirty_map, self.block_size, self.version = None, 0, -1


_TRACKED_TABLES = {}
_LAST_PLAN = {}      # (device index, stream) -> provenance of the plan the last schedule_cache_moves left there


def track_move_table(table: torch.Tensor) -> torch.Tensor:
    """Declare ``table`` ([rows, 2] int32, contiguous) a persistent move workspace that only
    ``schedule_cache_moves`` writes to -- what ``CompressionScheduler.cache_move_indices`` is
    (reference scheduler.py:74-86).  Returns the tensor itself."""
    _require(table, "table", torch.int32)
    if table.dim() != 2 or table.shape[1] != 2 or not table.is_contiguous():
        raise RuntimeError("track_move_table: expected a contiguous [rows, 2] int32 tensor")
    for key in [k for k, r in _TRACKED_TABLES.items() if r.ref() is None]:
        del _TRACKED_TABLES[key]
    _TRACKED_TABLES[id(table)] = _TrackedTable(table)
    return table


def _tracked(table: torch.Tensor):
    rec = _TRACKED_TABLES.get(id(table))
    if rec is None or rec.ref() is not table or rec.ptr != table.data_ptr():
        return None
    return rec


# The fork never calls track_move_table: its scheduler just passes the same persistent tensor to schedule_cache_moves
# step after step (reference scheduler.py:74-86, 513-522).  A table seen by schedule_cache_moves is therefore
# registered by the call itself (the first call pays the reference's full fill_(0), every later one clears only what
# the call before wrote) -- the guards are track_move_table's: the very tensor object, its storage pointer and its
# version counter (a tensor without one, e.g. made under torch.inference_mode(), is filled whole every call).
# KVC_AUTO_TRACK_MOVE_TABLE=0: only tables registered explicitly.  What a registration costs: a dirty map of
# kvc_cache_moves_dirty_map_bytes(rows, bs) (1 bit per block_size rows).
AUTO_TRACK_MOVE_TABLE = os.environ.get("KVC_AUTO_TRACK_MOVE_TABLE", "1") not in ("", "0")


def _auto_track(table: torch.Tensor):
    if (not AUTO_TRACK_MOVE_TABLE or table.dim() != 2 or table.shape[1] != 2 or not table.is_contiguous()
            or _version_of(table) is None or torch.cuda.is_current_stream_capturing()):
        return None
    track_move_table(table)
    return _tracked(table)


def schedule_cache_moves(
    out_cache_moves_indices: torch.Tensor,
    out_cache_moves_count: torch.Tensor,
    evicted_logical_indices: torch.Tensor,
    evicted_kv_count: torch.Tensor,
    evicted_kv_offsets: torch.Tensor,
    block_tables: torch.Tensor,
    context_lens: torch.Tensor,
    block_size: int,
) -> None:
    """reference vllm/_custom_ops.py:1158-1179 (zero fill of the workspace fused in)"""
    _schedule_t1_cache_moves(out_cache_moves_indices, out_cache_moves_count,
                             evicted_logical_indices, evicted_kv_count, evicted_kv_offsets,
                             block_tables, context_lens, block_size, zero_fill=True)


def _schedule_t1_cache_moves(cache_moves_idx, cache_moves_count, evicted_logical_indices,
                             evicted_kv_count, evicted_kv_offsets, block_tables, context_lens,
                             block_size, zero_fill):
    lib = _lib.load()
    for n, t in (("cache_moves_idx", cache_moves_idx), ("cache_moves_count", cache_moves_count),
                 ("evicted_logical_indices", evicted_logical_indices),
                 ("evicted_kv_count", evicted_kv_count),
                 ("evicted_kv_offsets", evicted_kv_offsets), ("block_tables", block_tables),
                 ("context_lens", context_lens)):
        _require(t, n, torch.int32)
    if not cache_moves_idx.is_contiguous() or not cache_moves_count.is_contiguous():
        raise RuntimeError("schedule_cache_moves: output tensors must be contiguous")
    num_seqs, num_layers, num_kv_heads = evicted_kv_count.shape
    keep = []
    dev = cache_moves_idx.device
    rows, bs = cache_moves_idx.shape[0], int(block_size)
    mode, dmap, dmap_bytes = (1 if zero_fill else 0), None, 0
    rec = _tracked(cache_moves_idx)
    if rec is None and zero_fill:
        rec = _auto_track(cache_moves_idx)
    if rec is not None and not zero_fill:
        rec.version = None                # the bare op leaves rows behind that the map does not know: full fill next time
        rec = None
    if rec is not None and bs >= 1:
        dmap_bytes = int(lib.kvc_cache_moves_dirty_map_bytes(rows, bs))
        known = (rec.dirty_map is not None and rec.block_size == bs and rec.dirty_map.numel() >= dmap_bytes
                 and _same_version(rec.version, cache_moves_idx))
        if known:
            mode = 2                      # zero everywhere but where the map says: clear just that
        else:                             # first use, or somebody else wrote to the table: the full fill
            if rec.dirty_map is None or rec.dirty_map.numel() < dmap_bytes:
                rec.dirty_map = torch.zeros(dmap_bytes, dtype=torch.uint8, device=dev)
            else:
                rec.dirty_map.zero_()
            rec.block_size = bs
        dmap = rec.dirty_map
    plan = workspace(dev, int(lib.kvc_cache_moves_plan_bytes()), "cache_moves_plan")
    with on_device(dev):
        _lib.check(lib.kvc_schedule_t1_cache_moves_ex(
            cache_moves_idx.data_ptr(), rows, cache_moves_count.data_ptr(),
            _contig(evicted_logical_indices, keep).data_ptr(),
            _contig(evicted_kv_count, keep).data_ptr(),
            _contig(evicted_kv_offsets, keep).data_ptr(),
            _contig(block_tables, keep).data_ptr(), _contig(context_lens, keep).data_ptr(),
            num_seqs, num_layers, num_kv_heads, block_tables.shape[3], bs,
            mode, _ptr(dmap), dmap_bytes, plan.data_ptr(), _stream(cache_moves_idx)))
    _written(cache_moves_idx, cache_moves_count)      # (before the map and the plan remember the versions they vouch for)
    if rec is not None:
        rec.version = _version_of(cache_moves_idx)
    # the plan belongs to exactly these three tensors as they are now (execute_cache_moves checks)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    _LAST_PLAN[(index, _stream(cache_moves_idx))] = (
        plan, tuple((weakref.ref(t), _version_of(t)) for t in (cache_moves_idx, cache_moves_count, evicted_kv_offsets)),
        num_seqs * num_layers * num_kv_heads, bs)


def _plan_of(k_cache, cmi, cmc, offs, total_heads, block_size):
    """the plan schedule_cache_moves left behind for exactly (cmi, cmc, offs) on this stream, or None"""
    dev = k_cache.device
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    rec = _LAST_PLAN.get((index, _stream(k_cache)))
    if rec is None or rec[2] != total_heads or rec[3] != block_size:
        return None
    for (ref, version), t in zip(rec[1], (cmi, cmc, offs)):
        if ref() is not t or not _same_version(version, t):
            return None
    return rec[0]


def execute_cache_moves(
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    kv_metrics: torch.Tensor,
    kv_position: torch.Tensor,
    cache_moves_indices: torch.Tensor,
    cache_moves_count: torch.Tensor,
    evicted_kv_offsets: torch.Tensor,
    blocks_per_head: int,
    threads_per_head: int,
) -> None:
    """reference vllm/_custom_ops.py:1220-1256.  ``blocks_per_head`` / ``threads_per_head``
    are launch hints of the reference CUDA kernel; kept in the signature, unused."""
    lib = _lib.load()
    _require(k_cache, "k_cache")
    _require(v_cache, "v_cache")
    _require(kv_metrics, "kv_metrics", torch.float32)
    for n, t in (("kv_position", kv_position), ("cache_moves_indices", cache_moves_indices),
                 ("cache_moves_count", cache_moves_count),
                 ("evicted_kv_offsets", evicted_kv_offsets)):
        _require(t, n, torch.int32)
    if k_cache.dim() != 4 or v_cache.dim() != 3:
        raise RuntimeError("execute_cache_moves: k_cache must be [NB, hd/x, bs, x] and "
                           "v_cache [NB, hd, bs]")
    for n, t in (("k_cache", k_cache), ("v_cache", v_cache), ("kv_metrics", kv_metrics),
                 ("kv_position", kv_position)):
        if not t.is_contiguous():
            raise RuntimeError(f"execute_cache_moves: {n} must be contiguous (mutated in place)")
    _execute_cache_moves(k_cache, v_cache, kv_metrics, kv_position, cache_moves_indices,
                         cache_moves_count, evicted_kv_offsets, "both")


def _execute_cache_moves(k_cache, v_cache, kv_metrics, kv_position, cache_moves_indices,
                         cache_moves_count, evicted_kv_offsets, half: str) -> None:
    """``half``: "both" (the op), or "plan" / "apply" -- the op's two halves
    (kvc_execute_cache_moves_plan / _apply), which bench.py separates to put events around the
    data kernel on the caller's own stream."""
    lib = _lib.load()
    num_blocks, head_size, block_size = v_cache.shape
    vec = k_cache.shape[3]
    cmi = cache_moves_indices.contiguous()
    cmc = cache_moves_count.contiguous()
    offs = evicted_kv_offsets.contiguous()
    total_heads = cmc.numel()
    # a list that schedule_cache_moves made on this stream and nobody touched since brings its plan
    # along: ONE launch (no planning pass, no claim table); any other list plans for itself
    plan = _plan_of(k_cache, cache_moves_indices, cache_moves_count, evicted_kv_offsets, total_heads, block_size)
    if half != "plan":
        _written(k_cache, v_cache, kv_metrics, kv_position)
    if _lib.block_layout_id() == _lib.LAYOUTS["slot_major"]:
        # a slot is two contiguous runs of hd * e bytes: the kernel copies exactly the moved bytes (no block images,
        # no claim table); without a plan of the list's own, one small launch makes one
        ws = workspace(k_cache.device, int(lib.kvc_cache_moves_plan_bytes()), "execute_cache_moves_slot_major")
        with on_device(k_cache.device):
            if plan is None and half in ("plan", "both"):
                _lib.check(lib.kvc_execute_cache_moves_slot_major_plan(cmc.data_ptr(), total_heads, ws.data_ptr(),
                                                                       ws.numel(), _stream(k_cache)))
            if half != "plan":
                _lib.check(lib.kvc_execute_cache_moves_slot_major(
                    k_cache.data_ptr(), v_cache.data_ptr(), kv_metrics.data_ptr(), kv_position.data_ptr(),
                    cmi.data_ptr(), cmc.data_ptr(), offs.data_ptr(), total_heads, num_blocks, block_size, head_size,
                    k_cache.element_size(), (plan if plan is not None else ws).data_ptr(), ws.data_ptr(), ws.numel(),
                    _stream(k_cache)))
        return
    if plan is not None:
        if half != "plan":
            with on_device(k_cache.device):
                _lib.check(lib.kvc_execute_cache_moves_planned(
                    k_cache.data_ptr(), v_cache.data_ptr(), kv_metrics.data_ptr(), kv_position.data_ptr(),
                    cmi.data_ptr(), cmc.data_ptr(), offs.data_ptr(), total_heads, num_blocks, block_size,
                    head_size, k_cache.element_size(), vec, plan.data_ptr(), _stream(k_cache)))
        return
    ws_bytes = lib.kvc_execute_cache_moves_workspace_bytes(total_heads, num_blocks)
    ws = workspace(k_cache.device, ws_bytes, "execute_cache_moves")
    shape = (total_heads, num_blocks, block_size, head_size, k_cache.element_size(), vec,
             ws.data_ptr(), ws.numel(), _stream(k_cache))
    with on_device(k_cache.device):
        if half == "plan":
            _lib.check(lib.kvc_execute_cache_moves_plan(cmi.data_ptr(), cmc.data_ptr(),
                                                        offs.data_ptr(), *shape))
        else:
            fn = lib.kvc_execute_cache_moves if half == "both" else lib.kvc_execute_cache_moves_apply
            _lib.check(fn(k_cache.data_ptr(), v_cache.data_ptr(), kv_metrics.data_ptr(),
                          kv_position.data_ptr(), cmi.data_ptr(), cmc.data_ptr(), offs.data_ptr(),
                          *shape))


def reshape_and_cache_kvc(
    key: torch.Tensor,
    value: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    kv_metrics: torch.Tensor,
    slot_mapping: torch.Tensor,
    kv_metric_head_bias: torch.Tensor,
    kv_cache_dtype: str,
    k_scale: float,
    v_scale: float,
) -> None:
    """reference vllm/_custom_ops.py:641-658; ``kv_cache_dtype`` in {"auto", "fp8",
    "fp8_e4m3", "fp8_e5m2"} (csrc/quantization/fp8/nvidia/quant_utils.cuh:525-566)"""
    lib = _lib.load()
    for n, t in (("key", key), ("value", value), ("key_cache", key_cache),
                 ("value_cache", value_cache)):
        _require(t, n)
    _require(kv_metrics, "kv_metrics", torch.float32)
    _require(slot_mapping, "slot_mapping", torch.int64)
    _require(kv_metric_head_bias, "kv_metric_head_bias", torch.float32)
    if key.stride(2) != 1 or key.stride(1) != key.shape[2] or \
            value.stride(2) != 1 or value.stride(1) != value.shape[2]:
        raise RuntimeError("reshape_and_cache_kvc: key/value must be dense in their last two dims")
    num_tokens, num_heads, head_size = key.shape
    block_size = key_cache.shape[2]
    sm = slot_mapping.contiguous()
    hb = kv_metric_head_bias.contiguous()
    _written(key_cache, value_cache, kv_metrics)
    if kv_cache_dtype == "auto":
        if key.dtype != key_cache.dtype or value.dtype != value_cache.dtype:
            raise RuntimeError("reshape_and_cache_kvc: kv_cache_dtype 'auto' needs cache dtype == "
                               "key/value dtype")
        with on_device(key.device):
            _lib.check(lib.kvc_reshape_and_cache_layout(
                key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
                kv_metrics.data_ptr(), sm.data_ptr(), hb.data_ptr(), num_tokens, num_heads,
                head_size, block_size, key.element_size(), key.stride(0), value.stride(0),
                _lib.block_layout_id(), _stream(key)))
        return
    kinds = {"fp8": 0, "fp8_e4m3": 0, "fp8_e5m2": 1}
    if kv_cache_dtype not in kinds:
        raise RuntimeError(f"Unsupported data type of kv cache: {kv_cache_dtype}")
    srcs = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}
    if key.dtype not in srcs or value.dtype != key.dtype:
        raise RuntimeError(f"Unsupported input type of kv cache: {key.dtype}")
    if key_cache.element_size() != 1 or value_cache.element_size() != 1:
        raise RuntimeError("reshape_and_cache_kvc: an fp8 kv cache must have 1-byte elements")
    with on_device(key.device):
        _lib.check(lib.kvc_reshape_and_cache_fp8_layout(
            key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
            kv_metrics.data_ptr(), sm.data_ptr(), hb.data_ptr(), num_tokens, num_heads, head_size,
            block_size, srcs[key.dtype], kinds[kv_cache_dtype], key.stride(0), value.stride(0),
            float(k_scale), float(v_scale), _lib.block_layout_id(), _stream(key)))


V1_DEAD_MESSAGE = ("schedule_cache_evictions (V1) is dead code in the reference; use "
                   "vllm_kvcompress_amd.kvcompress.metrics.CompressionMetrics.schedule_evictions")


def schedule_cache_evictions(*args, **kwargs):
    """reference vllm/_custom_ops.py:935-1006: the V1 CUDA scheduler.  It is dead code in
    the reference (vllm/kvcompress/scheduler.py:285 ``if False:``) and its wrapper
    dispatches to a namespace where the op is not registered; the live
    path is ``CompressionMetrics.schedule_evictions``.  The dispatcher schemas
    ``_C_kvc_ops::schedule_cache_evictions`` / ``truncate_cache_evictions``
    (csrc/torch_bindings.cpp:374-394) are registered by ``torch_ops.register()`` and raise the
    same message."""
    raise NotImplementedError(V1_DEAD_MESSAGE)


# ---------------------------------------------------------------------------------------
# F3: decode attention with KV-metric output      reference vllm/_custom_ops.py:135-201
_ATTN_PARTITION = 512           # reference vllm/attention/ops/paged_attn.py:_PARTITION_SIZE
_ATTENTION_SCHEDULE = 0         # kvc_attention_params.schedule of every call (0 = automatic)


def set_attention_schedule(schedule: int) -> None:
    """0 automatic (default), 1 always partitioned (the reference's v2 shape: per-partition kernel
    + reduce + metric rescale), 2 single pass whenever the context fits in LDS.  Passed per call
    in ``kvc_attention_params.schedule``; the library holds no state."""
    global _ATTENTION_SCHEDULE
    if schedule not in (0, 1, 2):
        raise ValueError("attention schedule must be 0, 1 or 2")
    _ATTENTION_SCHEDULE = int(schedule)
    if _compiled_binding():       # the C++ kernels of libkvc_torch.so read their own copy
        torch.ops._kvc_mi355x.set_attention_schedule(int(schedule))


def _paged_attention_kvc(out, kv_metric_out, exp_sum, max_logits, tmp_out, tmp_kv_metric_out,
                         query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                         context_lens, kv_position, last_position, kv_metric_buffer_len,
                         block_size, max_context_len, alibi_slopes, kv_cache_dtype, k_scale,
                         v_scale, record_kv_metrics, fused_metrics=None, use_l2=True, harvest=None, layer=0) -> None:
    lib = _lib.load()
    for name, t in (("out", out), ("query", query), ("key_cache", key_cache),
                    ("value_cache", value_cache)):
        _require(t, name)
    if fused_metrics is None:
        _require(kv_metric_out, "kv_metric_out", torch.float32)
    else:
        _require(fused_metrics, "metrics", torch.float32)
        if not fused_metrics.is_contiguous():
            raise RuntimeError("paged_attention_kvc: metrics must be contiguous (updated in place)")
    for name, t in (("block_tables", block_tables), ("context_lens", context_lens),
                    ("kv_position", kv_position), ("last_position", last_position),
                    ("kv_metric_buffer_len", kv_metric_buffer_len)):
        _require(t, name, torch.int32)
    kvds = {"auto": 0, "fp8": 1, "fp8_e4m3": 1, "fp8_e5m2": 2}
    if kv_cache_dtype not in kvds:
        raise RuntimeError(f"Unsupported data type of kv cache: {kv_cache_dtype}")
    dtypes = {torch.float16: 0, torch.bfloat16: 1}
    if query.dtype not in dtypes:
        raise RuntimeError(f"Unsupported data type: {query.dtype}")
    if out.dtype != query.dtype:
        raise RuntimeError("paged_attention_kvc: query and output must share a dtype")
    if kv_cache_dtype == "auto":
        if key_cache.dtype != query.dtype or value_cache.dtype != query.dtype:
            raise RuntimeError("paged_attention_kvc: an \"auto\" cache has the query's dtype")
    elif key_cache.element_size() != 1 or value_cache.element_size() != 1:
        raise RuntimeError("paged_attention_kvc: an fp8 kv cache must have 1-byte elements")
    num_seqs, num_heads, head_size = query.shape
    if query.stride(1) != head_size or query.stride(2) != 1:
        raise RuntimeError("paged_attention_kvc: query must be contiguous in (head, dim)")
    if not out.is_contiguous():
        raise RuntimeError("paged_attention_kvc: out must be contiguous")
    keep = []
    bt = _contig(block_tables, keep)
    p = _lib.KvcAttentionParams()
    p.out, p.kv_metric_out = out.data_ptr(), _ptr(kv_metric_out)
    p.fused_metrics, p.fused_use_l2 = _ptr(fused_metrics), int(bool(use_l2))
    p.exp_sums, p.max_logits = _ptr(exp_sum), _ptr(max_logits)
    p.tmp_out, p.tmp_kv_metric_out = _ptr(tmp_out), _ptr(tmp_kv_metric_out)
    p.query, p.key_cache, p.value_cache = query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr()
    p.block_tables = bt.data_ptr()
    p.context_lens = _contig(context_lens, keep).data_ptr()
    p.kv_position = _contig(kv_position, keep).data_ptr()
    p.last_position = _contig(last_position, keep).data_ptr()
    p.kv_metric_buffer_len = _contig(kv_metric_buffer_len, keep).data_ptr()
    p.alibi_slopes = None if alibi_slopes is None else _contig(alibi_slopes.float(), keep).data_ptr()
    p.q_stride, p.kv_block_stride = query.stride(0), key_cache.stride(0)
    p.scale, p.k_scale, p.v_scale = float(scale), float(k_scale), float(v_scale)
    p.num_seqs, p.num_heads, p.num_kv_heads = num_seqs, num_heads, int(num_kv_heads)
    p.head_size, p.block_size = head_size, int(block_size)
    p.max_num_blocks_per_seq = bt.shape[-1]
    p.max_context_len = int(max_context_len)
    p.dtype, p.kv_cache_dtype = dtypes[query.dtype], kvds[kv_cache_dtype]
    p.record_kv_metrics = int(bool(record_kv_metrics))
    p.schedule = _ATTENTION_SCHEDULE
    p.block_layout = _lib.block_layout_id()
    if harvest is not None:
        # the lists of the next schedule call, made by this launch's epilogue (CompressionMetrics.begin_attention_harvest)
        harvest.check_call(query, int(num_kv_heads), int(block_size), int(layer), _stream(query))
        p.harvest_buf = harvest.buf.data_ptr()
        p.harvest_seq_slot = harvest.seq_slot.data_ptr()
        p.harvest_seq_positions = harvest.seq_positions.data_ptr()
        p.harvest_num_protected = harvest.num_protected.data_ptr()
        p.harvest_num_seqs, p.harvest_layer = harvest.num_seqs, int(layer)
        p.harvest_num_layers, p.harvest_num_sinks = harvest.num_layers, harvest.num_sinks
    with on_device(query.device):
        _lib.check(lib.kvc_paged_attention_decode(p, _stream(query)))
    if harvest is not None:
        harvest.layers_done.add(int(layer))
    _written(out, kv_metric_out if record_kv_metrics else None)


def paged_attention_kvc_v1(out, kv_metric_out, query, key_cache, value_cache, num_kv_heads: int,
                           scale: float, block_tables, context_lens, kv_position, last_position,
                           kv_metric_buffer_len, block_size: int, max_context_len: int,
                           alibi_slopes, kv_cache_dtype: str, k_scale: float, v_scale: float,
                           record_kv_metrics: bool) -> None:
    """reference vllm/_custom_ops.py:135-163.  The reference's v1 keeps a whole context in
    one workgroup's shared memory; here long contexts at small batch are partitioned, so the
    small per-partition buffers the v1 signature does not carry (exp sums, max logits, partial
    outputs) come from the wrapper's scratch (``reserve_attention_scratch`` sizes it ahead of
    time).  The unnormalised per-partition weights need no buffer of their own: they are written
    to ``kv_metric_out`` and rescaled in place (the rescale kernel reads and writes the same
    ``slot * qpk + q`` element)."""
    exp_sum, max_logits, tmp_out = _partition_scratch(query, num_kv_heads, max_context_len, "attn_v1")
    tmp_metric = kv_metric_out if (record_kv_metrics and exp_sum is not None) else None
    _paged_attention_kvc(out, kv_metric_out, exp_sum, max_logits, tmp_out, tmp_metric, query,
                         key_cache, value_cache, num_kv_heads, scale, block_tables, context_lens,
                         kv_position, last_position, kv_metric_buffer_len, block_size,
                         max_context_len, alibi_slopes, kv_cache_dtype, k_scale, v_scale,
                         record_kv_metrics)


def _partition_scratch_bytes(num_seqs, num_heads, head_size, elem_bytes, max_context_len):
    parts = (int(max_context_len) + _ATTN_PARTITION - 1) // _ATTN_PARTITION
    n = num_seqs * num_heads * parts
    return n, n * 8 + n * head_size * elem_bytes + 256


def reserve_attention_scratch(device, num_seqs: int, num_heads: int, head_size: int,
                              elem_bytes: int, max_context_len: int) -> None:
    """Pre-size the partition scratch of ``paged_attention_kvc_v1`` / ``_fused_metrics`` for the
    largest batch the engine will run (call at start-up, next to the cache allocation), so the
    ops never allocate while serving or inside a HIP-graph capture."""
    _, nbytes = _partition_scratch_bytes(num_seqs, num_heads, head_size, elem_bytes, max_context_len)
    for tag in ("attn_v1", "attn_fused"):
        reserve_workspace(torch.device(device), nbytes, tag)


def _partition_scratch(query, num_kv_heads, max_context_len, tag):
    num_seqs, num_heads, head_size = query.shape
    parts = (int(max_context_len) + _ATTN_PARTITION - 1) // _ATTN_PARTITION
    if parts <= 1 or not _lib.load().kvc_paged_attention_decode_uses_partitions_in(
            num_seqs, num_heads, int(num_kv_heads), head_size, int(max_context_len),
            _ATTENTION_SCHEDULE, _lib.block_layout_id()):
        return None, None, None            # one kernel finishes the call: no scratch at all
    n, nbytes = _partition_scratch_bytes(num_seqs, num_heads, head_size, query.element_size(),
                                         max_context_len)
    buf = workspace(query.device, nbytes, tag)
    exp_sum = buf[:n * 4].view(torch.float32)
    max_logits = buf[n * 4:n * 8].view(torch.float32)
    tmp_out = buf[n * 8:n * 8 + n * head_size * query.element_size()].view(query.dtype)
    return exp_sum, max_logits, tmp_out


def paged_attention_kvc_fused_metrics(out, metrics, query, key_cache, value_cache,
                                      num_kv_heads: int, scale: float, block_tables, context_lens,
                                      kv_position, last_position, kv_metric_buffer_len,
                                      block_size: int, max_context_len: int, alibi_slopes,
                                      kv_cache_dtype: str, k_scale: float, v_scale: float,
                                      use_l2: bool = True,
                                      temp_metrics: Optional[torch.Tensor] = None,
                                      harvest=None, layer: int = 0) -> None:
    """Extension (the reference lists it as a to-do, vllm/kvcompress/README.md:32,49): the
    attention of ``paged_attention_kvc_v1`` that adds ``sum_q p^2`` (or ``sum_q p``) of every
    key inside the metric window straight into ``metrics [num_blocks, block_size]`` -- what
    ``kv_metric_out`` + ``CompressionMetrics.aggregate_decode`` + ``clear_temp_metrics`` do
    together (metrics.py:429-439, 337-342), bit for bit, without the [NB, bs, qpk] round trip.
    When the call takes the partitioned schedule (long contexts at small batch) the
    unnormalised weights need a ``[num_blocks, block_size, qpk]`` float32 scratch: pass
    ``CompressionMetrics.temp_metrics`` (which this path otherwise leaves unused) as
    ``temp_metrics``; the op never allocates a buffer of that size itself.

    ``harvest`` / ``layer``: the handle of ``CompressionMetrics.begin_attention_harvest`` and the layer
    this call computes -- the epilogue then also lists, per head, the keys that fall below the pivots the
    last ``schedule_evictions`` left behind, and the ``schedule_evictions`` that follows
    ``end_attention_harvest`` runs on those lists: a decode step of continual compression without a sweep
    of the metric store (kvc_attention_harvest_begin, include/kvc_mi355x.h)."""
    es, ml, to = _partition_scratch(query, num_kv_heads, max_context_len, "attn_fused")
    tm = None
    if es is not None:
        qpk = query.shape[1] // int(num_kv_heads)
        if temp_metrics is None:
            raise RuntimeError("paged_attention_kvc_fused_metrics: this shape takes the partitioned "
                               "schedule, which needs temp_metrics [num_blocks, block_size, qpk] "
                               "(CompressionMetrics.temp_metrics)")
        _require(temp_metrics, "temp_metrics", torch.float32)
        if not temp_metrics.is_contiguous() or temp_metrics.numel() < metrics.numel() * qpk:
            raise RuntimeError("paged_attention_kvc_fused_metrics: temp_metrics must be contiguous "
                               "with at least num_blocks * block_size * qpk elements")
        tm = temp_metrics
    _written(metrics)
    _paged_attention_kvc(out, None, es, ml, to, tm, query, key_cache, value_cache, num_kv_heads,
                         scale, block_tables, context_lens, kv_position, last_position,
                         kv_metric_buffer_len, block_size, max_context_len, alibi_slopes,
                         kv_cache_dtype, k_scale, v_scale, True, fused_metrics=metrics,
                         use_l2=use_l2, harvest=harvest, layer=layer)


def paged_attention_kvc_v2(out, kv_metric_out, exp_sum, max_logits, tmp_out, tmp_kv_metric_out,
                           query, key_cache, value_cache, num_kv_heads: int, scale: float,
                           block_tables, context_lens, kv_position, last_position,
                           kv_metric_buffer_len, block_size: int, max_context_len: int,
                           alibi_slopes, kv_cache_dtype: str, k_scale: float, v_scale: float,
                           record_kv_metrics: bool) -> None:
    """reference vllm/_custom_ops.py:166-201 (caller-provided partition buffers,
    vllm/attention/ops/paged_attn.py:367-379)."""
    _require(exp_sum, "exp_sum", torch.float32)
    _require(max_logits, "max_logits", torch.float32)
    _paged_attention_kvc(out, kv_metric_out, exp_sum, max_logits, tmp_out, tmp_kv_metric_out,
                         query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                         context_lens, kv_position, last_position, kv_metric_buffer_len,
                         block_size, max_context_len, alibi_slopes, kv_cache_dtype, k_scale,
                         v_scale, record_kv_metrics)
