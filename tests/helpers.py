"""Shared helpers for the parity tests (oracle side + fixture loading)."""
from __future__ import annotations

import hashlib
import os

import numpy as np

from oracle import kvc_oracle as orc
from vllm_kvcompress_amd.harness import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}


def sched_kwargs(g):
    kw = dict(
        metrics=g["metrics"], token_positions=g["token_positions"],
        seq_index_by_block=g["seq_index_by_block"],
        layer_index_by_block=g["layer_index_by_block"],
        head_index_by_block=g["head_index_by_block"],
        logical_block_num_by_block=g["logical_block_num_by_block"],
        block_size=int(g["block_size"]), num_layers=int(g["num_layers"]),
        num_kv_heads=int(g["num_kv_heads"]),
        seq_indices=[int(s) for s in g["seq_indices"]], seq_positions=g["seq_positions"],
        evicted_blocks_per_seq=g["evicted_blocks_per_seq"],
        context_lens=g["context_lens"], hanging_token_count=g["hanging_token_count"],
        evicted_kv_offsets=g["evicted_kv_offsets"], num_protected=g["protected"],
        use_average=bool(int(g["use_average"])), num_sinks=int(g["num_sinks"]),
    )
    if "bias" in g:
        kw.update(bias=g["bias"], position_bins=g["position_bins"],
                  bias_weight=float(g["bias_weight"]))
    if "uniform_evict" in g and int(g["uniform_evict"]):
        kw.update(uniform_evict=True)
    return kw


def golden_caches(g):
    return synth.make_caches_u16(int(g["cache_seed"]), int(g["num_blocks"]),
                                 int(g["head_size"]), int(g["block_size"]))


def sha(a: np.ndarray) -> np.ndarray:
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(),
                         dtype=np.uint8)


def oracle_pipeline(st: synth.PagedState, evicted_blocks, k_cache=None, v_cache=None,
                    mode="reference", **kw):
    """schedule_evictions -> schedule_cache_moves -> execute_cache_moves on the oracle."""
    eli, ekc, ebc = orc.schedule_evictions(
        metrics=st.metrics, token_positions=st.token_positions,
        seq_index_by_block=st.seq_index_by_block,
        layer_index_by_block=st.layer_index_by_block,
        head_index_by_block=st.head_index_by_block,
        logical_block_num_by_block=st.logical_block_num_by_block,
        block_size=st.block_size, num_layers=st.num_layers, num_kv_heads=st.num_kv_heads,
        seq_indices=st.seq_indices, seq_positions=st.seq_positions,
        evicted_blocks_per_seq=evicted_blocks, context_lens=st.context_lens,
        hanging_token_count=st.hanging_token_count,
        evicted_kv_offsets=st.evicted_kv_offsets, num_protected=st.protected,
        mode=mode, **kw)
    N = st.total_slots
    cmi = np.zeros((N, 2), dtype=np.int32)
    cmc = np.zeros(ekc.shape, dtype=np.int32)
    orc.schedule_cache_moves(cmi, cmc, eli, ekc, st.evicted_kv_offsets, st.block_tables,
                             st.context_lens, st.block_size)
    out = dict(eli=eli, ekc=ekc, ebc=ebc, cmi=cmi, cmc=cmc)
    if k_cache is not None:
        k2, v2 = k_cache.copy(), v_cache.copy()
        m2, p2 = st.metrics.copy(), st.token_positions.copy()
        orc.execute_cache_moves(k2, v2, m2, p2, cmi, cmc, st.evicted_kv_offsets)
        out.update(k=k2, v=v2, metrics=m2, positions=p2)
    return out


def reference_prefill_metrics_numpy(g):
    """the oracle's NumPy restatement of the reference loop (flash_attn.py:1120-1211) on a
    golden case of oracle/gen_golden_aggregate.py"""
    bf16 = "dtype" in g and str(g["dtype"]) == "bf16"
    if bf16:
        conv = lambda b: (b.view(np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)
    else:
        conv = lambda b: b.view(np.float16).astype(np.float32)
    q, k = conv(g["q"]), conv(g["k"])
    hd = q.shape[2]
    return orc.naive_kvc_attention(
        q, k, g["prompt_lens"], hd ** -0.5, g["buffer_len"], n_observed=int(g["n_observed"]),
        max_observed_block_size=int(g["block"]), use_l2=bool(int(g["use_l2"])),
        use_average=bool(int(g["use_average"])), use_maxpool=bool(int(g["use_maxpool"])),
        logit_round="bf16" if bf16 else "f16")
