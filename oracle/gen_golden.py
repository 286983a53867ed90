#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own Python on seeded inputs.

Runs only in the build container (needs /root/reference).  It imports the
reference's ``vllm/_custom_ops.py`` (``ref_schedule_t1_cache_moves``,
``ref_execute_cache_moves``) and ``vllm/kvcompress/metrics.py``
(``CompressionMetrics``) under a stub parent package, following the recipe in
SURVEY.md Appendix A, and writes inputs + outputs as small ``.npz`` fixtures to
``tests/golden/``.  Nothing from the reference is copied: the fixtures are data.

The reference's ``count_block_evictions`` is a CUDA-only op; the recipe replaces
it by the oracle's restatement of ``count_block_evictions_kernel`` -- this is
the one step of the golden pipeline that is NOT independent of the oracle (it
is pinned separately by hand-checkable cases in tests/test_oracle_golden.py).

usage: python oracle/gen_golden.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import contextlib
import hashlib
import importlib
import io
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import kvc_oracle as orc                      # noqa: E402
from vllm_kvcompress_amd.harness import synth              # noqa: E402

REF = "/root/reference"


def import_reference():
    import torch
    sys.dont_write_bytecode = True
    pkg = types.ModuleType("vllm")
    pkg.__path__ = [os.path.join(REF, "vllm")]
    sys.modules["vllm"] = pkg
    torch.cuda.memory_allocated = lambda *a, **k: 0       # debug prints in metrics.py
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        ops = importlib.import_module("vllm._custom_ops")
        met = importlib.import_module("vllm.kvcompress.metrics")

    def cpu_count_block_evictions(evicted_block_count, evicted_logical_indices,
                                  evicted_kv_offsets, hanging_token_count, block_size,
                                  null_value, evicted_blocks_per_seq=None):
        ebc = evicted_block_count.numpy()
        eli = evicted_logical_indices.numpy()
        orc.count_block_evictions(ebc, eli, evicted_kv_offsets.numpy(),
                                  hanging_token_count.numpy(), block_size, null_value)
    met.count_block_evictions = cpu_count_block_evictions
    return ops, met


def run_reference_schedule(met, st: synth.PagedState, evicted_blocks, *, use_average=False,
                           num_sinks=0, bias=None, position_bins=None, bias_weight=0.0, uniform_evict=False):
    import torch
    L, H, bs = st.num_layers, st.num_kv_heads, st.block_size
    with contextlib.redirect_stdout(io.StringIO()):
        cm = met.CompressionMetrics(bs, L, H, 1, 10 ** 9, None, float(bias_weight), device="cpu",
                                    use_average=use_average, num_attention_sinks=num_sinks)
        cm.init_kv_metadata(st.num_blocks)
    if bias is not None:
        cm.kv_metric_head_bias = met.KVHeadBias(torch.from_numpy(bias.copy()),
                                                torch.from_numpy(position_bins.copy()))
    cm.metrics[:] = torch.from_numpy(st.metrics)
    cm.token_positions[:] = torch.from_numpy(st.token_positions)
    cm.seq_index_by_block[:] = torch.from_numpy(st.seq_index_by_block)
    cm.layer_index_by_block[:] = torch.from_numpy(st.layer_index_by_block)
    cm.head_index_by_block[:] = torch.from_numpy(st.head_index_by_block)
    cm.logical_block_num_by_block[:] = torch.from_numpy(st.logical_block_num_by_block)
    with contextlib.redirect_stdout(io.StringIO()):
        eli, ekc, ebc = cm.schedule_evictions(
            list(st.seq_indices),
            torch.from_numpy(st.seq_positions.copy()),
            torch.tensor(list(evicted_blocks), dtype=torch.int32),
            torch.from_numpy(st.context_lens.copy()),
            torch.from_numpy(st.hanging_token_count.copy()),
            torch.from_numpy(st.evicted_kv_offsets.copy()),
            list(st.protected),
            uniform_evict=uniform_evict,
        )
    return eli.numpy().astype(np.int32), ekc.numpy().astype(np.int32), ebc.numpy().astype(np.int32)


def run_reference_moves(ops, st, eli, ekc, k_cache, v_cache):
    import torch
    N = st.total_slots
    cmi = torch.zeros((N, 2), dtype=torch.int32)
    cmc = torch.zeros(ekc.shape, dtype=torch.int32)
    with contextlib.redirect_stdout(io.StringIO()):
        ops.ref_schedule_t1_cache_moves(
            cmi, cmc, torch.from_numpy(eli.copy()), torch.from_numpy(ekc.copy()),
            torch.from_numpy(st.evicted_kv_offsets.copy()),
            torch.from_numpy(st.block_tables.copy()),
            torch.from_numpy(st.context_lens.copy()), st.block_size)
        k = torch.from_numpy(k_cache.copy())
        v = torch.from_numpy(v_cache.copy())
        m = torch.from_numpy(st.metrics.copy())
        p = torch.from_numpy(st.token_positions.copy())
        ops.ref_execute_cache_moves(k, v, m, p, cmi, cmc,
                                    torch.from_numpy(st.evicted_kv_offsets.copy()), 1, 1)
    return cmi.numpy(), cmc.numpy(), k.numpy(), v.numpy(), m.numpy(), p.numpy()


def state_arrays(st: synth.PagedState) -> dict:
    return dict(
        block_size=np.int32(st.block_size), num_layers=np.int32(st.num_layers),
        num_kv_heads=np.int32(st.num_kv_heads), num_blocks=np.int32(st.num_blocks),
        metrics=st.metrics, token_positions=st.token_positions,
        seq_index_by_block=st.seq_index_by_block, layer_index_by_block=st.layer_index_by_block,
        head_index_by_block=st.head_index_by_block,
        logical_block_num_by_block=st.logical_block_num_by_block,
        context_lens=st.context_lens, block_tables=st.block_tables,
        hanging_token_count=st.hanging_token_count, evicted_kv_offsets=st.evicted_kv_offsets,
        seq_indices=np.asarray(st.seq_indices, dtype=np.int32), seq_positions=st.seq_positions,
        protected=np.asarray(st.protected, dtype=np.int32),
    )


def case_grid():
    """(name, make_state kwargs, eviction spec, schedule kwargs)."""
    cases = []
    # block-size sweep, single sequence, uncompressed prefill state
    for bs in (1, 2, 4, 16, 32):
        cases.append((f"b1_bs{bs}_prefill", dict(num_layers=2, num_kv_heads=2, block_size=bs,
                      seq_lens=[5 * bs + 3 if bs > 1 else 23], seed=bs, protected=1), "mid", {}))
    # multi-sequence (exercises the reference's batch>1 quirk), ragged lens
    for seed, prot in ((0, 1), (1, 3), (2, 4)):
        cases.append((f"b3_bs4_prefill_s{seed}", dict(num_layers=2, num_kv_heads=2, block_size=4,
                      seq_lens=[21, 9, 30], seed=seed, protected=prot), "mixed", {}))
    # second-compression states
    for seed in (0, 1, 2):
        cases.append((f"b2_bs4_compressed_s{seed}", dict(num_layers=3, num_kv_heads=2, block_size=4,
                      seq_lens=[41, 37], seed=10 + seed, protected=2, compressed=True), "mid", {}))
    cases.append(("b3_bs16_compressed", dict(num_layers=2, num_kv_heads=4, block_size=16,
                  seq_lens=[130, 97, 200], seed=5, protected=[1, 15, 16], compressed=True),
                  "mixed", {}))
    # protected window variants and eviction extremes
    cases.append(("b1_bs4_prot_big", dict(num_layers=2, num_kv_heads=2, block_size=4,
                  seq_lens=[33], seed=3, protected=1000), "all", {}))
    cases.append(("b1_bs4_evict0", dict(num_layers=2, num_kv_heads=2, block_size=4,
                  seq_lens=[33], seed=4, protected=3), "zero", {}))
    cases.append(("b1_bs4_evict1", dict(num_layers=2, num_kv_heads=2, block_size=4,
                  seq_lens=[33], seed=5, protected=4), "one", {}))
    cases.append(("b1_bs4_evict_allfinite", dict(num_layers=2, num_kv_heads=2, block_size=4,
                  seq_lens=[33], seed=6, protected=5), "all", {}))
    cases.append(("b1_bs4_evict_over", dict(num_layers=2, num_kv_heads=2, block_size=4,
                  seq_lens=[33], seed=7, protected=5), "over", {}))
    cases.append(("b3_bs4_evict_over", dict(num_layers=2, num_kv_heads=2, block_size=4,
                  seq_lens=[33, 18, 25], seed=13, protected=[5, 2, 7]), "over", {}))
    # average / sinks / bias
    cases.append(("b2_bs4_avg", dict(num_layers=2, num_kv_heads=2, block_size=4,
                  seq_lens=[29, 18], seed=8, protected=2), "mid", dict(use_average=True)))
    cases.append(("b2_bs4_sinks", dict(num_layers=2, num_kv_heads=2, block_size=4,
                  seq_lens=[29, 18], seed=9, protected=2), "mid", dict(num_sinks=3)))
    cases.append(("b2_bs4_bias", dict(num_layers=2, num_kv_heads=2, block_size=4,
                  seq_lens=[29, 18], seed=11, protected=2), "mid", dict(bias=True)))
    # a medium case closer to the production shape (bs16, hd128 caches in fp16)
    cases.append(("b2_bs16_hd128", dict(num_layers=2, num_kv_heads=2, block_size=16,
                  seq_lens=[150, 91], seed=12, protected=32), "mid", dict(hd=128)))
    cases.append(("b2_bs16_L4H8_med", dict(num_layers=4, num_kv_heads=8, block_size=16,
                  seq_lens=[300, 171], seed=14, protected=32), "mid", {}))
    # continual-compression steady state (every head at the cap + 1 appended token, about one
    # block per head freed): what the small-eviction schedule of the HIP path is chosen for
    cases.append(("b2_bs16_steady256", dict(num_layers=2, num_kv_heads=2, block_size=16,
                  seq_lens=[768, 768], seed=21, protected=17, steady_cap=256), "cap256", {}))
    cases.append(("b3_bs32_steady512", dict(num_layers=2, num_kv_heads=2, block_size=32,
                  seq_lens=[1536] * 3, seed=22, protected=33, steady_cap=512), "cap512", {}))
    cases.append(("b1_bs8_steady128", dict(num_layers=3, num_kv_heads=2, block_size=8,
                  seq_lens=[384], seed=23, protected=9, steady_cap=128), "cap128", {}))
    # bulk evictions big enough for the HIP path's bracket schedule to be the automatic choice
    # (>= 64 Ki candidate slots per sequence, heads of >= 64 blocks): one sequence, two sequences
    # under the batch > 1 rule with uneven asks, a second compression at bs 32
    cases.append(("b1_bs16_bulk64k", dict(num_layers=4, num_kv_heads=8, block_size=16,
                  seq_lens=[2100], seed=31, protected=32), "mid", dict(hd=16)))
    cases.append(("b2_bs16_bulk64k_rule", dict(num_layers=4, num_kv_heads=8, block_size=16,
                  seq_lens=[2100, 2500], seed=32, protected=[32, 17]), "uneven", dict(hd=16)))
    cases.append(("b1_bs32_bulk64k_compressed", dict(num_layers=2, num_kv_heads=8, block_size=32,
                  seq_lens=[9000], seed=33, protected=40, compressed=True), "mid", dict(hd=16)))
    # the reference's other selection rule, uniform_evict (metrics.py:639-666: the same number of chunks
    # from every head; heads of equal length, i.e. prefill states)
    cases.append(("b1_bs4_uniform", dict(num_layers=2, num_kv_heads=2, block_size=4,
                  seq_lens=[41], seed=41, protected=3), "mid", dict(uniform=True)))
    cases.append(("b2_bs16_uniform", dict(num_layers=2, num_kv_heads=4, block_size=16,
                  seq_lens=[300, 171], seed=42, protected=[32, 5]), "uneven", dict(uniform=True, hd=16)))
    cases.append(("b1_bs16_uniform_bulk64k", dict(num_layers=4, num_kv_heads=8, block_size=16,
                  seq_lens=[2100], seed=43, protected=32), "mid", dict(uniform=True, hd=16)))
    return cases


def eviction_spec(st, spec, rng):
    bs = st.block_size
    if spec.startswith("cap"):       # the scheduler's own count (Appendix B of SURVEY.md, harness/synth.py)
        cap = int(spec[3:])
        return [synth.evict_block_count(context_lens_lh=st.context_lens[:, b, :], seq_len=3 * cap,
                                        block_size=bs, protected_window_size=st.protected[b],
                                        max_cache_tokens=cap) for b in range(st.num_seqs)]
    nblk = ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)    # [B]
    TH = st.num_layers * st.num_kv_heads
    out = []
    for b in range(st.num_seqs):
        limit = max(int(nblk[b]) - (st.protected[b] + bs - 1) // bs * TH, 0)
        if spec == "zero":
            k = 0
        elif spec == "one":
            k = min(1, limit)
        elif spec == "all":
            k = limit
        elif spec == "over":
            k = int(nblk[b])   # more than the finite-threshold chunks
        elif spec == "mid":
            k = limit // 2
        elif spec == "mixed":
            k = [limit // 2, 0, limit][b % 3]
        elif spec == "uneven":
            k = [limit // 2, limit // 3, 2 * limit // 3][b % 3]
        else:
            raise ValueError(spec)
        out.append(int(k))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default="", help="write only the cases whose name contains this")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    ops, met = import_reference()
    rng = np.random.default_rng(1234)
    for name, mk, spec, extra in case_grid():
        if args.only not in name:
            continue
        st = synth.make_state(**mk)
        evicted = eviction_spec(st, spec, rng)
        sched_kw = {}
        bias = bins = None
        if extra.get("bias"):
            bins = np.array([0, 4, 11], dtype=np.int32)
            bias = np.random.default_rng(99).normal(size=(st.num_layers, st.num_kv_heads, 3)) \
                .astype(np.float32) * 40
            sched_kw.update(bias=bias, position_bins=bins, bias_weight=0.5)
        if extra.get("use_average"):
            sched_kw["use_average"] = True
        if extra.get("num_sinks"):
            sched_kw["num_sinks"] = extra["num_sinks"]
        if extra.get("uniform"):
            sched_kw["uniform_evict"] = True
        eli, ekc, ebc = run_reference_schedule(met, st, evicted, **sched_kw)
        # caches: tiny head dim for the small cases; every element distinct is not
        # needed for bitwise parity -- random 16-bit patterns
        hd = extra.get("hd", 8)
        cache_seed = 1000 + st.num_blocks
        k_cache, v_cache = synth.make_caches_u16(cache_seed, st.num_blocks, hd, st.block_size)
        cmi, cmc, k2, v2, m2, p2 = run_reference_moves(ops, st, eli, ekc, k_cache, v_cache)
        arrs = state_arrays(st)
        arrs.update(
            evicted_blocks_per_seq=np.asarray(evicted, dtype=np.int32),
            use_average=np.int32(bool(extra.get("use_average"))),
            num_sinks=np.int32(extra.get("num_sinks", 0)),
            bias_weight=np.float32(sched_kw.get("bias_weight", 0.0)),
            uniform_evict=np.int32(bool(extra.get("uniform"))),
            ref_evicted_logical_indices=eli, ref_evicted_kv_count=ekc,
            ref_evicted_block_count=ebc,
            cache_seed=np.int64(cache_seed), head_size=np.int32(hd),
            ref_cache_moves_idx=cmi, ref_cache_moves_count=cmc,
            ref_metrics=m2, ref_positions=p2,
            ref_k_sha256=np.frombuffer(hashlib.sha256(k2.tobytes()).digest(), dtype=np.uint8),
            ref_v_sha256=np.frombuffer(hashlib.sha256(v2.tobytes()).digest(), dtype=np.uint8),
        )
        if hd <= 8:      # small caches are stored verbatim, big ones by digest only
            arrs.update(ref_k_cache=k2, ref_v_cache=v2)
        if bias is not None:
            arrs.update(bias=bias, position_bins=bins)
        path = os.path.join(args.out, f"{name}.npz")
        np.savez_compressed(path, **arrs)
        print(f"{name}: N={st.total_slots} evict={evicted} moves={int(cmc.sum())} "
              f"-> {os.path.getsize(path)} B")


if __name__ == "__main__":
    main()
