#!/usr/bin/env python3
"""Hot-path benchmark: KV slots evicted+compacted per second on MI355X.

A "step" is one full pass of the eviction/compaction hot path over one compression
batch of synthetic paged-cache state resident in HBM:

    S1  CompressionMetrics.schedule_evictions   (A3, includes A4's count)
    S2  schedule_cache_moves                     (A5)
    S3  execute_cache_moves                      (A6, the K/V compaction)

Default workload = BASELINE.json configs[1]: Llama-3-8B shape (32 layers, 8 KV heads,
hd 128), 32k-token cache, block_size 16, batch 1, compress_once to half the cache
(max_cache_tokens = T/2), fp16 K/V, tie-free permutation metrics.  Scheduling always
reads the pristine metric store; compaction writes working copies, so every step does
identical work (the moves never touch their own sources).

Multi GPU (launched by torch.distributed.run): sequences are sharded, every rank owns a
private cache and runs the same per-rank workload (weak scaling); the only collective is
the reduction of the throughput scalars.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=8)
    ap.add_argument("--head-size", type=int, default=128)
    ap.add_argument("--block-size", type=int, default=16)
    ap.add_argument("--seq-len", type=int, default=32768, help="cached tokens per sequence")
    ap.add_argument("--batch", type=int, default=1, help="sequences per GPU")
    ap.add_argument("--keep", type=float, default=0.5, help="max_cache_tokens / seq_len")
    ap.add_argument("--protected", type=int, default=32)
    ap.add_argument("--metric-shape", default="perm", choices=["perm", "decay", "oldest"])
    ap.add_argument("--mode", default="per_sequence", choices=["per_sequence", "reference"])
    ap.add_argument("--kv-dtype", default="fp16", choices=["fp16", "fp8"],
                    help="cache element type (fp8 = 1-byte elements, K vectors of 16)")
    ap.add_argument("--steady-cap", type=int, default=0,
                    help="continual-compression steady state: every head holds this many survivors + 1 "
                         "appended token and is compressed back to the cap (max_cache_tokens)")
    ap.add_argument("--contiguous-blocks", action="store_true",
                    help="physical blocks in allocation order (fresh prefill) instead of shuffled")
    ap.add_argument("--lean", action="store_true",
                    help="extension: no MAX_INT padding / key-scratch clear in schedule_evictions and no "
                         "zero fill of the move workspace (outputs a consumer reads are unchanged)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-adjacent", action="store_true",
                    help="skip the decode-attention (F3) side measurement")
    ap.add_argument("--traffic-json", default=os.path.join(REPO, "profiles", "traffic.json"),
                    help="PMC-derived HBM bytes per launch of the compaction kernel, if collected")
    return ap.parse_args()


def alg_bytes_per_move(head_size: int, elem_bytes: int) -> int:
    """K and V, read+write, + metric r/w + position r/w + move pair read."""
    return 4 * head_size * elem_bytes + 24


def build_workload(args, seed, device):
    import torch
    from vllm_kvcompress_amd.harness import device as hdev
    from vllm_kvcompress_amd.harness import synth
    L, H, bs, hd = args.layers, args.kv_heads, args.block_size, args.head_size
    # seq_len counts the freshly sampled token whose KV is not cached yet
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs,
                          seq_lens=[args.seq_len + 1] * args.batch, seed=seed,
                          protected=args.protected, metric_shape=args.metric_shape,
                          spare_block_frac=0.02, shuffle_blocks=not args.contiguous_blocks,
                          steady_cap=args.steady_cap or None)
    cap = args.steady_cap if args.steady_cap else int(args.seq_len * args.keep)
    evicted = [synth.evict_block_count(context_lens_lh=st.context_lens[:, b, :],
                                       seq_len=args.seq_len + 1, block_size=bs,
                                       protected_window_size=args.protected, max_cache_tokens=cap)
               for b in range(args.batch)]
    ds = hdev.upload(st, device, num_queries_per_kv=1, mode=args.mode)
    ds.cm.lean_outputs = bool(args.lean)
    g = torch.Generator(device=device)
    g.manual_seed(1234 + seed)
    if args.kv_dtype == "fp8":
        kv = torch.randint(0, 256, (2, st.num_blocks, bs * hd), dtype=torch.uint8, device=device,
                           generator=g)
    else:
        kv = torch.randint(-32768, 32767, (2, st.num_blocks, bs * hd), dtype=torch.int16,
                           device=device, generator=g).view(torch.float16)
    k_cache, v_cache = hdev.split_kv_cache(kv, hd)
    return st, ds, evicted, k_cache, v_cache


def cpu_baseline(args):
    """The oracle (a port of the reference algorithm) on the host, one core: one full S1+S2+S3
    pass over ONE sequence of the bench's own shape and cache length (the default workload
    itself: ~5-10 s of CPU work), capped at 32k tokens for bigger configurations."""
    from oracle import kvc_oracle as orc
    from oracle import kvc_oracle_c as orc_c
    from vllm_kvcompress_amd.harness import synth
    L, H, bs, hd = args.layers, args.kv_heads, args.block_size, args.head_size
    T = min(args.seq_len, 32768)
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=[T + 1], seed=0,
                          protected=args.protected, spare_block_frac=0.02, metric_shape=args.metric_shape)
    evicted = [synth.evict_block_count(context_lens_lh=st.context_lens[:, 0, :], seq_len=T + 1,
                                       block_size=bs, protected_window_size=args.protected,
                                       max_cache_tokens=int(T * args.keep))]
    # cache contents do not matter for the copy loop's time: one random 32 MiB chunk, tiled
    e = 1 if args.kv_dtype == "fp8" else 2
    n = st.num_blocks * hd * bs * e
    chunk = np.random.default_rng(0).integers(0, 256, size=1 << 25, dtype=np.uint8)
    kflat = np.tile(chunk, n // chunk.size + 1)[:n]
    vflat = np.tile(chunk[::-1], n // chunk.size + 1)[:n]
    x = 16 // e
    dt = np.uint8 if e == 1 else np.uint16
    k = np.ascontiguousarray(kflat.view(dt).reshape(st.num_blocks, hd // x, bs, x))
    v = np.ascontiguousarray(vflat.view(dt).reshape(st.num_blocks, hd, bs))
    m, p = st.metrics.copy(), st.token_positions.copy()
    t0 = time.perf_counter()
    eli, ekc, ebc = orc.schedule_evictions(
        metrics=st.metrics, token_positions=st.token_positions,
        seq_index_by_block=st.seq_index_by_block, layer_index_by_block=st.layer_index_by_block,
        head_index_by_block=st.head_index_by_block,
        logical_block_num_by_block=st.logical_block_num_by_block, block_size=bs, num_layers=L,
        num_kv_heads=H, seq_indices=st.seq_indices, seq_positions=st.seq_positions,
        evicted_blocks_per_seq=evicted, context_lens=st.context_lens,
        hanging_token_count=st.hanging_token_count, evicted_kv_offsets=st.evicted_kv_offsets,
        num_protected=st.protected, mode="reference")
    t1 = time.perf_counter()
    cmi = np.zeros((st.total_slots, 2), np.int32)
    cmc = np.zeros(ekc.shape, np.int32)
    orc_c.schedule_cache_moves(cmi, cmc, eli, ekc, st.evicted_kv_offsets,
                               np.ascontiguousarray(st.block_tables),
                               np.ascontiguousarray(st.context_lens), bs)
    t2 = time.perf_counter()
    orc_c.execute_cache_moves(k, v, m, p, cmi, cmc, st.evicted_kv_offsets)
    t3 = time.perf_counter()
    units = int(ekc.sum()) + int(cmc.sum())
    same = T == args.seq_len and args.batch == 1 and not args.steady_cap
    return {
        "value": units / (t3 - t0), "unit": "KV slots/s", "cores": 1, "kind": "port",
        "sample": ("the bench workload itself" if same else "one sequence of the bench's shape")
                  + f": L{L} H{H} hd{hd}, {T}-token cache, bs{bs}, B=1, keep={args.keep}, "
                  f"{args.metric_shape} metrics - one S1+S2+S3 pass of the oracle (NumPy "
                  f"schedule_evictions + C move / compaction loops), {units} slots in {t3 - t0:.2f} s",
        "stage_seconds": {"S1_schedule": t1 - t0, "S2_moves": t2 - t1, "S3_compact": t3 - t2},
        "host_cpus": os.cpu_count(),
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from vllm_kvcompress_amd import _custom_ops as ops
    import vllm_kvcompress_amd
    vllm_kvcompress_amd.load()      # fail loudly if the HIP extension is missing

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # one rank per GPU; KVC_BENCH_BACKEND=gloo (test hook) lets several ranks share the one
    # GPU of a single-GPU box to exercise the N > 1 code path
    backend = os.environ.get("KVC_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend=backend)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    st, ds, evicted, k_cache, v_cache = build_workload(args, seed=rank, device=device)
    bs = args.block_size
    N = st.total_slots
    work_metrics = ds.cm.metrics.clone()
    work_pos = ds.cm.token_positions.clone()
    cmi = torch.empty((N, 2), dtype=torch.int32, device=device)
    cmc = torch.empty((st.num_seqs, st.num_layers, st.num_kv_heads), dtype=torch.int32, device=device)
    evicted_t = torch.tensor(evicted, dtype=torch.int32, device=device)
    seq_idx = list(st.seq_indices)
    prot = list(st.protected)

    ev = lambda: torch.cuda.Event(enable_timing=True)
    # marks: S1 start, S2 start, S3 start (plan half), data kernel start, end.  All events are
    # recorded on the stream the kernels are launched on (torch's current stream).
    marks = [[ev() for _ in range(5)] for _ in range(args.steps)]
    out = {}

    def step(i=None):
        if i is not None: marks[i][0].record()
        eli, ekc, ebc = ds.cm.schedule_evictions(seq_idx, ds.seq_positions, evicted_t,
                                                 ds.context_lens, ds.hanging_token_count,
                                                 ds.evicted_kv_offsets, prot, total_slots=N)
        if i is not None: marks[i][1].record()
        if args.lean:
            ops._schedule_t1_cache_moves(cmi, cmc, eli, ekc, ds.evicted_kv_offsets, ds.block_tables,
                                         ds.context_lens, bs, zero_fill=False)
        else:
            ops.schedule_cache_moves(cmi, cmc, eli, ekc, ds.evicted_kv_offsets, ds.block_tables,
                                     ds.context_lens, bs)
        if i is not None: marks[i][2].record()
        # execute_cache_moves = its two halves (kvc_execute_cache_moves_plan / _apply), called
        # separately so that the data kernel is bracketed by events of its own
        ops._execute_cache_moves(k_cache, v_cache, work_metrics, work_pos, cmi, cmc,
                                 ds.evicted_kv_offsets, "plan")
        if i is not None: marks[i][3].record()
        ops._execute_cache_moves(k_cache, v_cache, work_metrics, work_pos, cmi, cmc,
                                 ds.evicted_kv_offsets, "apply")
        if i is not None: marks[i][4].record()
        out["ekc"], out["ebc"] = ekc, ebc

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms = sum(m[3].elapsed_time(m[4]) for m in marks) / args.steps

    evicted_slots = int(out["ekc"].sum().item())
    moved_slots = int(cmc.sum().item())
    freed_blocks = int(out["ebc"].sum().item())
    if args.mode == "per_sequence" or args.batch == 1:   # the reference's batch>1 quirk frees fewer
        assert freed_blocks == sum(evicted), (freed_blocks, sum(evicted))
    s1 = sum(m[0].elapsed_time(m[1]) for m in marks) / args.steps
    s2 = sum(m[1].elapsed_time(m[2]) for m in marks) / args.steps
    s3 = sum(m[2].elapsed_time(m[4]) for m in marks) / args.steps

    # whole-job rate = units of all ranks / slowest rank's time; the only collective of the
    # path (RCCL all_gather of two scalars per rank)
    from vllm_kvcompress_amd.harness import dist as hdist
    units_local = float((evicted_slots + moved_slots) * args.steps)
    red = hdist.reduce_throughput(units_local, elapsed, device=device)
    units, elapsed = red["units"], red["seconds"]
    per_rank = None
    if world > 1:
        per_rank = [{"units": u, "seconds": sec} for u, sec in
                    zip(red["per_rank_units"], red["per_rank_seconds"])]

    if rank == 0:
        e = 1 if args.kv_dtype == "fp8" else 2
        bpm = alg_bytes_per_move(args.head_size, e)
        alg_bytes = moved_slots * bpm + 8 * st.total_heads
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        default_workload = (args.layers, args.kv_heads, args.head_size, bs, args.seq_len, args.batch,
                            args.keep, args.protected, args.metric_shape, args.kv_dtype,
                            args.steady_cap, args.contiguous_blocks) == (
                                32, 8, 128, 16, 32768, 1, 0.5, 32, "perm", "fp16", 0, False)
        # the committed PMC figure belongs to the default workload only
        if default_workload and os.path.exists(args.traffic_json):
            try:
                traffic = json.load(open(args.traffic_json)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "KV slots evicted+compacted/sec and HBM GB/s, Llama-3-8B 32k cache blk16",
            "value": units / elapsed,
            "unit": "KV slots/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8",
            "dtype_detail": f"compaction copies bytes ({args.kv_dtype} K/V, {e}-byte elements); scheduling "
                            "compares float32 metrics and computes int32 indices",
            "data": "synthetic (seeded paged cache, tie-free permutation metrics, random K/V bits)",
            "config": {
                "workload": f"{'Llama-3-8B' if args.layers == 32 else 'Llama-3-70B' if args.layers == 80 else 'custom'} shape L{args.layers} H{args.kv_heads} hd{args.head_size}, "
                            f"{args.seq_len}-token cache, block_size {bs}, batch {args.batch}/GPU, "
                            f"{args.kv_dtype} K/V, "
                            + (f"continual steady state cap={args.steady_cap}+1 token, " if args.steady_cap
                               else f"compress_once keep={args.keep}, ")
                            + f"protected_window={args.protected}, "
                            f"metrics={args.metric_shape}, schedule mode={args.mode}"
                            + (", lean outputs (extension)" if args.lean else "") + ", physical blocks "
                            f"{'in allocation order' if args.contiguous_blocks else 'shuffled'}",
                "candidate_slots": N, "evicted_slots": evicted_slots, "moved_slots": moved_slots,
                "freed_blocks": freed_blocks,
            },
            "stages_ms": {"S1_schedule_evictions": s1, "S2_schedule_moves": s2, "S3_execute_moves": s3},
            "stage_rates": {
                "S1_candidate_slots_per_s": N / (s1 * 1e-3),
                "S2_moves_per_s": moved_slots / (s2 * 1e-3),
                "S3_moved_slots_per_s": moved_slots / (s3 * 1e-3),
            },
            "roofline": {
                "kernel": f"kvc::compact_runs_kernel<{args.head_size},{bs},{e}> (execute_cache_moves)",
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_move": bpm,
                "avg_launch_ms": kernel_ms,
                "timing": "HIP events on the launch stream around the compaction kernel "
                          "(kvc_execute_cache_moves_apply), every timed step",
                "traffic_frac_of_peak": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
            },
        }
        if per_rank:
            res["per_rank"] = per_rank
        if world == 1 and not args.no_adjacent:
            # the producer of the metrics (row F3), one layer step at the continual-compression
            # shape; not part of `value`
            from vllm_kvcompress_amd.harness.attention_bench import run as attn_run
            res["adjacent_decode_attention"] = [
                attn_run(64, 4097, Hq=4 * args.kv_heads, Hkv=args.kv_heads, hd=args.head_size,
                         bs=args.block_size if args.block_size in (16, 32) else 16, iters=10, record=r)
                for r in (True, False)]
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
