"""ctypes wrapper of oracle/kvc_oracle.c (TEST INFRASTRUCTURE ONLY -- see kvc_oracle.py)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libkvc_oracle.so")
_lib = None


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def set_threads(n: int) -> None:
    """host threads of the head loops (1 = scalar port; results do not depend on it)"""
    _load().orc_set_threads(ctypes.c_int32(int(n)))


def max_threads() -> int:
    return int(_load().orc_max_threads())


def _p(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def count_block_evictions(evicted_block_count, evicted_logical_indices, evicted_kv_offsets,
                          hanging_token_count, block_size, null_value):
    _load().orc_count_block_evictions(
        _p(evicted_block_count), _p(evicted_logical_indices), _p(evicted_kv_offsets),
        _p(hanging_token_count), ctypes.c_int32(evicted_block_count.size),
        ctypes.c_int64(evicted_logical_indices.size), ctypes.c_int32(block_size),
        ctypes.c_int32(null_value))


def schedule_cache_moves(out_idx, out_count, evicted, ekc, offs, block_tables, context_lens,
                         block_size, zero_fill=True):
    B, L, H = ekc.shape
    _load().orc_schedule_t1_cache_moves(
        _p(out_idx), ctypes.c_int64(out_idx.shape[0]), _p(out_count), _p(evicted), _p(ekc),
        _p(offs), _p(block_tables), _p(context_lens), ctypes.c_int32(B), ctypes.c_int32(L),
        ctypes.c_int32(H), ctypes.c_int32(block_tables.shape[3]), ctypes.c_int32(block_size),
        ctypes.c_int32(1 if zero_fill else 0))


def execute_cache_moves(k_cache, v_cache, kv_metrics, kv_position, moves, count, offs):
    nb, hd, bs = v_cache.shape
    x = k_cache.shape[3]
    _load().orc_execute_cache_moves(
        _p(k_cache), _p(v_cache), _p(kv_metrics), _p(kv_position), _p(moves), _p(count), _p(offs),
        ctypes.c_int32(count.size), ctypes.c_int32(bs), ctypes.c_int32(hd),
        ctypes.c_int32(k_cache.itemsize), ctypes.c_int32(x))
