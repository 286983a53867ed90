#!/usr/bin/env python3
"""Hot-path benchmark: KV slots evicted+compacted per second on MI355X.

A "step" is one full pass of the eviction/compaction hot path over one compression
batch of synthetic paged-cache state resident in HBM:

    S1  CompressionMetrics.schedule_evictions   (A3, includes A4's count)
    S2  schedule_cache_moves                     (A5)
    S3  execute_cache_moves                      (A6, the K/V compaction)

Default workload = BASELINE.json configs[1]: Llama-3-8B shape (32 layers, 8 KV heads,
hd 128), 32k-token cache, block_size 16, batch 1, compress_once to half the cache
(max_cache_tokens = T/2), fp16 K/V, tie-free permutation metrics.  ``--config c3|c3i|c4|c5``
select the other BASELINE configurations (c3: 256 resident sequences in the continual
steady state; c3i: its initial phase, a wave of 16 sequences compressed 32k -> 4k tokens; c4: Llama-3-70B shape, 16k tokens, 32 sequences per GPU; c5: fp8, bs 32, 64k).
Scheduling always reads the pristine metric store; compaction writes working copies, so
every step does identical work (the moves never touch their own sources).

Multi GPU: ``python bench.py --gpus N`` launches N ranks itself (one per GPU, RCCL) when it
is not already running under torch.distributed.run; sequences are sharded, every rank owns
a private cache (weak scaling: the same per-rank workload; ``--scaling strong`` splits a
fixed batch); the only collective is the reduction of the throughput scalars.

Prints ONE compact JSON line (rank 0, < 4 KB, the last line of stdout): the contract's keys, `roofline`,
`cpu_baseline`, the parity verdict.  The legs that do not fit it (other configurations, engine-sized cache, call
forms, S0 stages, adjacent attention) go to ``--detail-json`` (default bench_detail.json), named in the line.
The timed step calls the path exactly as the fork's scheduler does (``--call-form fork``).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

CONFIGS = {     # BASELINE.json configs[1..4]; c2 is the configuration the metric is quoted on
    "c2": {},
    "c3": {"batch": 256, "steady_cap": 4096},           # phase ii: all 256 sequences resident at the cap
    "c3i": {"batch": 16, "keep": 0.125},               # phase i: a wave of 16 sequences, 32k -> 4k tokens
    "c4": {"layers": 80, "seq_len": 16384, "batch": 32},
    "c5": {"kv_dtype": "fp8", "block_size": 32, "seq_len": 65536},
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS),
                    help="BASELINE.json configuration (explicit shape flags override it)")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--kv-heads", type=int, default=8)
    ap.add_argument("--head-size", type=int, default=128)
    ap.add_argument("--block-size", type=int, default=None)
    ap.add_argument("--seq-len", type=int, default=None, help="cached tokens per sequence")
    ap.add_argument("--batch", type=int, default=None, help="sequences per GPU (weak scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="strong: --batch is the whole job's sequence count, split over the GPUs")
    ap.add_argument("--keep", type=float, default=None, help="max_cache_tokens / seq_len (default 0.5)")
    ap.add_argument("--protected", type=int, default=32)
    ap.add_argument("--metric-shape", default="perm", choices=["perm", "decay", "oldest"])
    ap.add_argument("--mode", default="per_sequence", choices=["per_sequence", "reference"])
    ap.add_argument("--kv-dtype", default=None, choices=["fp16", "fp8"],
                    help="cache element type (fp8 = 1-byte elements, K vectors of 16)")
    ap.add_argument("--steady-cap", type=int, default=None,
                    help="continual-compression steady state: every head holds this many survivors + 1 "
                         "appended token and is compressed back to the cap (max_cache_tokens)")
    ap.add_argument("--spare-blocks", type=float, default=0.02,
                    help="free blocks in the cache as a fraction of the allocated ones (an engine sizes its "
                         "cache to fill HBM: e.g. 30 puts the sequence's blocks into a 31x larger cache)")
    ap.add_argument("--contiguous-blocks", action="store_true",
                    help="physical blocks in allocation order (fresh prefill) instead of shuffled")
    ap.add_argument("--lean", action="store_true",
                    help="extension: no MAX_INT padding / key-scratch clear in schedule_evictions and no "
                         "zero fill of the move workspace (outputs a consumer reads are unchanged)")
    ap.add_argument("--pass-block-tables", action="store_true",
                    help="hand BlockState.block_tables to schedule_evictions (optional argument: bulk evictions of a batch "
                         "that is sparse in its cache build their keys through it; the streaming schedule ignores it)")
    ap.add_argument("--call-form", default="fork", choices=["fork", "hinted"],
                    help="how the timed step calls the path.  fork (default, the headline): exactly the fork's scheduler "
                         "(reference vllm/kvcompress/scheduler.py:74-86, 245-260, 491-529) -- evicted_blocks_per_seq and the "
                         "last token positions as FRESH device int tensors, protected windows as a tuple, no total_slots=, "
                         "no block_tables=, a plain persistent move workspace, a fresh cache_moves_count.  hinted: host list "
                         "of counts + total_slots= (the method never waits for the device) and a workspace registered with "
                         "track_move_table (`stages_ms_hinted` of the default line)")
    ap.add_argument("--block-layout", default="reference", choices=["reference", "slot_major"],
                    help="how the bytes inside a cache block are laid out (include/kvc_mi355x.h KVC_LAYOUT_*).  reference "
                         "(default, the headline): the fork's K [hd/x][bs][x] / V [hd][bs].  slot_major: K [bs][hd] / V [bs][hd], "
                         "the package's opt-in MI355X-native layout (KVC_BLOCK_LAYOUT=slot_major) -- the default run measures "
                         "it in a child process and reports it as `roofline_native_layout`")
    ap.add_argument("--no-native-layout", action="store_true", help="skip the slot-major leg of the default run")
    ap.add_argument("--headline-only", action="store_true", default=None,
                    help="only the timed loop, the parity gate and (N = 1) roofline + cpu_baseline: none of the other legs "
                         "(other configs, engine-sized cache, S0 stages, call forms, adjacent attention, native layout).  "
                         "The default for --gpus N > 1")
    ap.add_argument("--detail-json", default=os.path.join(REPO, "bench_detail.json"),
                    help="where everything that does not fit the compact line goes (named in the line as `detail`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-gate", action="store_true",
                    help="skip the oracle comparison of the timed workload's results (profiling passes that "
                         "bench.py starts for its own counters use this; a reported line never does)")
    ap.add_argument("--no-adjacent", action="store_true",
                    help="skip the decode-attention (F3) side measurement")
    ap.add_argument("--no-s0", action="store_true", help="skip the S0 (metric aggregation) stage timings")
    ap.add_argument("--no-engine-cache", action="store_true",
                    help="skip the second measurement of the same step in an HBM-filling cache")
    ap.add_argument("--engine-cache-frac", type=float, default=0.80,
                    help="fraction of the free HBM that second cache takes")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two short rocprofv3 --pmc passes that measure roofline.traffic (the "
                         "committed figure of profiles/traffic.json is reported instead)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short runs of BASELINE configs[4] / configs[2] the default line carries")
    ap.add_argument("--no-probe", action="store_true",
                    help="skip the access-pattern ceiling probe (tools/libkvc_probe.so)")
    ap.add_argument("--traffic-json", default=os.path.join(REPO, "profiles", "traffic.json"),
                    help="PMC-derived HBM bytes per launch of the compaction kernel, if collected")
    args = ap.parse_args(argv)
    preset = CONFIGS[args.config]
    for key, default in (("layers", 32), ("block_size", 16), ("seq_len", 32768), ("batch", 1),
                         ("kv_dtype", "fp16"), ("steady_cap", 0), ("keep", 0.5)):
        if getattr(args, key) is None:
            setattr(args, key, preset.get(key, default))
    return args


# --------------------------------------------------------------------------- launch
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    """``python bench.py --gpus N`` outside torch.distributed.run: start N ranks on this node
    (one per GPU, rendezvous on 127.0.0.1) running this same command line, relay rank 0's line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env, cwd=REPO).returncode


def alg_bytes_per_move(head_size: int, elem_bytes: int) -> int:
    """K and V, read+write, + metric r/w + position r/w + move pair read."""
    return 4 * head_size * elem_bytes + 24


def build_workload(args, seed, device, batch=None):
    import torch
    from vllm_kvcompress_amd.harness import device as hdev
    from vllm_kvcompress_amd.harness import synth
    L, H, bs, hd = args.layers, args.kv_heads, args.block_size, args.head_size
    batch = args.batch if batch is None else batch
    # seq_len counts the freshly sampled token whose KV is not cached yet
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs,
                          seq_lens=[args.seq_len + 1] * batch, seed=seed,
                          protected=args.protected, metric_shape=args.metric_shape,
                          spare_block_frac=args.spare_blocks, shuffle_blocks=not args.contiguous_blocks,
                          steady_cap=args.steady_cap or None)
    cap = args.steady_cap if args.steady_cap else int(args.seq_len * args.keep)
    evicted = [synth.evict_block_count(context_lens_lh=st.context_lens[:, b, :],
                                       seq_len=args.seq_len + 1, block_size=bs,
                                       protected_window_size=args.protected, max_cache_tokens=cap)
               for b in range(batch)]
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


def build_workload_that_fits(args, seed, device, batch):
    """the requested per-GPU batch, or the largest power-of-two fraction of it that fits in HBM
    next to the workspaces (said in config.workload)"""
    import torch
    while True:
        try:
            return (batch,) + build_workload(args, seed, device, batch)
        except torch.OutOfMemoryError:
            torch.cuda.empty_cache()
            if batch <= 1:
                raise
            batch //= 2


# --------------------------------------------------------------------------- parity gate
PARITY_ORACLE_MAX_SLOTS = 1 << 25       # the oracle's schedule takes seconds up to here (C2: 8.4 M, C5: 16.8 M)
PARITY_HOST_KV_MAX_BYTES = 9 << 30      # K/V that may be copied to the host for the oracle's compaction


def parity_snapshot(k_cache, v_cache, device_check_rows=None):
    """The cache as it is BEFORE the first step, for the gate below: a host copy when it is small
    enough for the oracle's compaction to run on it, otherwise nothing (the gate then checks the
    compaction on the device: every moved slot equals its source, read before the step)."""
    nbytes = 2 * k_cache.numel() * k_cache.element_size()
    if nbytes > PARITY_HOST_KV_MAX_BYTES:
        return None
    import torch
    it = torch.uint8 if k_cache.element_size() == 1 else torch.int16
    return k_cache.view(it).cpu().numpy(), v_cache.view(it).cpu().numpy()


def parity_gate(args, st, ds, evicted, mode, gpu, snap, k_cache, v_cache, wm, wp, lean=False):
    """no number is reported for results that differ from the reference's.
    The oracle (oracle/: NumPy restatement of schedule_evictions pinned to reference-generated
    vectors, C restatement of the serial move / compaction kernels) runs on the SAME seeded state
    the timed steps ran on; evicted indices, counts, the move list and -- when the cache fits on the
    host -- the compacted K / V / metrics / positions must be bit-identical
    (the reference's twin-equality pattern, tests/kernels/test_kvcompress_eviction.py:900-901, 967,
    1007-1008).  The checker is never inside the timed region.  Returns the `parity_checked` object;
    `bit_exact` False makes bench.py exit non-zero without printing a line."""
    import hashlib
    import torch
    from oracle import kvc_oracle as orc
    from oracle import kvc_oracle_c as orc_c
    N = st.total_slots
    if N > PARITY_ORACLE_MAX_SLOTS:
        if st.num_seqs > 1 and not lean:
            return parity_gate_sampled(args, st, evicted, gpu, k_cache, v_cache, wm, wp, mode=mode)
        return {"bit_exact": None, "skipped": f"{N} candidate slots: the oracle's sorts take minutes at this size "
                                              "(this shape is parity-tested at oracle sizes in tests/)"}
    t0 = time.perf_counter()
    bs = st.block_size
    eli, ekc, ebc = orc.schedule_evictions(
        metrics=st.metrics, token_positions=st.token_positions, seq_index_by_block=st.seq_index_by_block,
        layer_index_by_block=st.layer_index_by_block, head_index_by_block=st.head_index_by_block,
        logical_block_num_by_block=st.logical_block_num_by_block, block_size=bs, num_layers=st.num_layers,
        num_kv_heads=st.num_kv_heads, seq_indices=st.seq_indices, seq_positions=st.seq_positions,
        evicted_blocks_per_seq=evicted, context_lens=st.context_lens,
        hanging_token_count=st.hanging_token_count, evicted_kv_offsets=st.evicted_kv_offsets,
        num_protected=st.protected, mode=mode)
    cmi = np.zeros((N, 2), np.int32)
    cmc = np.zeros(ekc.shape, np.int32)
    orc_c.set_threads(min(os.cpu_count() or 1, orc_c.max_threads()))
    orc_c.schedule_cache_moves(cmi, cmc, eli, ekc, st.evicted_kv_offsets, np.ascontiguousarray(st.block_tables),
                               np.ascontiguousarray(st.context_lens), bs)
    offs = st.evicted_kv_offsets.reshape(-1).astype(np.int64)

    def defined(counts):        # rows a consumer reads: [off_g, off_g + count_g) of every head
        c = counts.reshape(-1).astype(np.int64)
        return np.repeat(offs - (np.cumsum(c) - c), c) + np.arange(int(c.sum()))

    compared, bad = [], []

    def cmp(name, got, want, rows=None):
        got = got.cpu().numpy() if hasattr(got, "cpu") else got
        if rows is not None:
            got, want = got[rows], want[rows]
        compared.append(name)
        if got.shape != want.shape or not np.array_equal(got, want):
            bad.append(name)

    cmp("evicted_kv_count", gpu["ekc"], ekc)
    cmp("evicted_block_count", gpu["ebc"], ebc)
    cmp("evicted_logical_indices", gpu["eli"], eli, defined(ekc) if lean else None)
    cmp("cache_moves_count", gpu["cmc"], cmc)
    cmp("cache_moves_idx", gpu["cmi"], cmi, defined(cmc) if lean else None)
    out = {"workload": "the timed workload itself (same seeded state, same eviction counts)",
           "oracle": "oracle/kvc_oracle.py schedule_evictions + oracle/kvc_oracle.c schedule_t1_cache_moves / "
                     "execute_cache_moves", "mode": mode}
    m, p = st.metrics.copy(), st.token_positions.copy()
    if snap is not None:
        k, v = snap
        orc_c.execute_cache_moves(k, v, m, p, cmi, cmc, st.evicted_kv_offsets)
        it = torch.uint8 if k_cache.element_size() == 1 else torch.int16
        gk, gv = k_cache.view(it).cpu().numpy(), v_cache.view(it).cpu().numpy()
        cmp("k_cache", gk, k)
        cmp("v_cache", gv, v)
        h = hashlib.sha256()
        h.update(gk.data)
        h.update(gv.data)
        out["kv_sha256"] = h.hexdigest()
        h = hashlib.sha256()
        h.update(k.data)
        h.update(v.data)
        out["kv_sha256_oracle"] = h.hexdigest()
        if out["kv_sha256"] != out["kv_sha256_oracle"] and "k_cache" not in bad and "v_cache" not in bad:
            bad.append("kv_sha256")
        out["kv_bytes"] = int(gk.nbytes + gv.nbytes)
        del gk, gv
    else:
        # a cache that fills the HBM: the compaction is checked where it ran -- metrics / positions (which
        # travel with every slot) against the oracle's, K / V rows of every moved slot against their
        # sources (sources are never written: reading them after the steps is reading them before)
        rows = torch.from_numpy(defined(cmc)).to(gpu["cmi"].device)
        mv = gpu["cmi"][rows].long()
        m2, p2 = torch.from_numpy(st.metrics).to(mv.device), torch.from_numpy(st.token_positions).to(mv.device)
        m2.view(-1)[mv[:, 0]] = m2.view(-1)[mv[:, 1]]
        p2.view(-1)[mv[:, 0]] = p2.view(-1)[mv[:, 1]]
        m, p = m2.cpu().numpy(), p2.cpu().numpy()
        ok_kv = True
        it = torch.uint8 if k_cache.element_size() == 1 else torch.int16      # (bits, not floats: random bits hold NaNs)
        kb, vb = k_cache.view(it), v_cache.view(it)
        for lo in range(0, mv.shape[0], 1 << 20):
            d, s_ = mv[lo:lo + (1 << 20), 0], mv[lo:lo + (1 << 20), 1]
            ok_kv &= bool(torch.equal(kb[d // bs, :, d % bs, :], kb[s_ // bs, :, s_ % bs, :]))
            ok_kv &= bool(torch.equal(vb[d // bs, :, d % bs], vb[s_ // bs, :, s_ % bs]))
        compared.append("k_cache/v_cache rows of every moved slot == their sources (on the device)")
        if not ok_kv:
            bad.append("k_cache/v_cache moved rows")
    cmp("kv_metrics", wm, m)
    cmp("kv_position", wp, p)
    out.update({"bit_exact": not bad, "compared": compared, "mismatched": bad,
                "seconds": time.perf_counter() - t0})
    return out


def parity_gate_sampled(args, st, evicted, gpu, k_cache, v_cache, wm, wp, num_sampled=8, schedule_only=False,
                        mode="per_sequence"):
    """The gate for a batch too large for the oracle's sorts (configs[2]: 256 sequences, 270 M slots).
    In per_sequence mode every sequence is scheduled as if alone, so the oracle runs on a sub-batch of
    `num_sampled` sequences spread over the batch and every output of THEIR heads -- evicted indices,
    counts, move rows -- must equal the corresponding piece of the full batch's outputs bit for bit;
    the compaction of the WHOLE batch is checked on the device (K / V rows of every moved slot equal
    their sources; metrics / positions against torch's own scatter of the move list).

    mode "reference" (the fork's default: the batch > 1 rule of metrics.py:709-729 couples the sequences):
    the oracle's two-stage form (oracle/kvc_oracle.py, pinned to the reference's b2_* / b3_* fixtures and to the
    imported reference) -- stage 1 counts every sequence's finite thresholds over the WHOLE batch (no sort),
    stage 2 is the batch rule on those counts: the number of chunks EVERY sequence really frees, checked
    against the device's evicted_block_count of every sequence; stage 3 is the per-sequence schedule with
    those counts, run for the sampled sequences as above."""
    import torch
    from oracle import kvc_oracle as orc
    from oracle import kvc_oracle_c as orc_c
    from vllm_kvcompress_amd.harness import synth
    t0 = time.perf_counter()
    bs, L, H, B = st.block_size, st.num_layers, st.num_kv_heads, st.num_seqs
    num_sampled = max(1, min(num_sampled, PARITY_ORACLE_MAX_SLOTS // max(1, st.total_slots // B)))   # (seconds of oracle)
    sel = sorted(set(int(x) for x in np.linspace(0, B - 1, num_sampled).round()))
    ctx = np.ascontiguousarray(st.context_lens[:, sel, :])
    offs_s = synth.kv_offsets(ctx, bs)
    hang_s = synth.hanging_tokens(ctx.transpose(1, 0, 2), bs)
    coupled = None
    if mode == "reference":
        fin, nblk = orc.finite_threshold_chunks(
            metrics=st.metrics, token_positions=st.token_positions, seq_index_by_block=st.seq_index_by_block,
            layer_index_by_block=st.layer_index_by_block, head_index_by_block=st.head_index_by_block,
            logical_block_num_by_block=st.logical_block_num_by_block, block_size=bs, num_layers=L, num_kv_heads=H,
            seq_indices=st.seq_indices, seq_positions=st.seq_positions, context_lens=st.context_lens,
            hanging_token_count=st.hanging_token_count, num_protected=st.protected)
        keff = orc.coupled_eviction_counts(evicted, nblk, fin, "reference")
        got_blocks = gpu["ebc"].reshape(B, -1).sum(1).cpu().numpy().astype(np.int64)
        coupled = {"sequences": B, "asked_blocks": int(sum(evicted)), "freed_blocks_by_the_batch_rule": int(keff.sum()),
                   "sequences_that_free_less_than_asked": int((keff < np.asarray(evicted)).sum()),
                   "every_sequence_frees_what_the_rule_says": bool(np.array_equal(got_blocks, keff))}
        evicted = [int(x) for x in keff]
    eli, ekc, ebc = orc.schedule_evictions(
        metrics=st.metrics, token_positions=st.token_positions, seq_index_by_block=st.seq_index_by_block,
        layer_index_by_block=st.layer_index_by_block, head_index_by_block=st.head_index_by_block,
        logical_block_num_by_block=st.logical_block_num_by_block, block_size=bs, num_layers=L, num_kv_heads=H,
        seq_indices=[st.seq_indices[i] for i in sel], seq_positions=np.ascontiguousarray(st.seq_positions[sel]),
        evicted_blocks_per_seq=[evicted[i] for i in sel], context_lens=ctx, hanging_token_count=hang_s,
        evicted_kv_offsets=offs_s, num_protected=[st.protected[i] for i in sel], mode="per_sequence")
    n_s = int(((ctx.astype(np.int64) + bs - 1) // bs).sum()) * bs
    cmi = np.zeros((n_s, 2), np.int32)
    cmc = np.zeros(ekc.shape, np.int32)
    orc_c.set_threads(min(os.cpu_count() or 1, orc_c.max_threads()))
    orc_c.schedule_cache_moves(cmi, cmc, eli, ekc, offs_s, np.ascontiguousarray(st.block_tables[:, sel]), ctx, bs)
    dev = gpu["eli"].device
    compared, bad = [], []

    def cmp(name, got, want):
        compared.append(name)
        if tuple(got.shape) != tuple(want.shape) or not np.array_equal(got, want):
            bad.append(name)

    idx = torch.tensor(sel, device=dev)
    if coupled is not None:
        compared.append("evicted blocks of every sequence == the batch rule on the whole batch's finite-threshold counts")
        if not coupled["every_sequence_frees_what_the_rule_says"]:
            bad.append("evicted blocks per sequence (batch rule)")
    cmp("evicted_kv_count", gpu["ekc"][idx].cpu().numpy(), ekc)
    cmp("evicted_block_count", gpu["ebc"][idx].cpu().numpy(), ebc)
    cmp("cache_moves_count", gpu["cmc"][idx].cpu().numpy(), cmc)
    # the heads' segments: [off, off + nblk * bs) in the full batch's lists against the sub-batch's
    lens = (((ctx.astype(np.int64) + bs - 1) // bs) * bs).transpose(1, 0, 2).reshape(-1)          # (b, l, h) order
    off_full = st.evicted_kv_offsets[sel].reshape(-1).astype(np.int64)
    off_sub = offs_s.reshape(-1).astype(np.int64)
    start = np.cumsum(lens) - lens
    within = np.arange(int(lens.sum())) - np.repeat(start, lens)
    rows_full = torch.from_numpy(np.repeat(off_full, lens) + within).to(dev)
    rows_sub = np.repeat(off_sub, lens) + within
    cmp("evicted_logical_indices", gpu["eli"][rows_full].cpu().numpy(), eli[rows_sub])
    cmp("cache_moves_idx", gpu["cmi"][rows_full].cpu().numpy(), cmi[rows_sub])
    if schedule_only:
        return {"mode": mode, "bit_exact": not bad, "sampled_sequences": sel, "compared": compared,
                "mismatched": bad, "seconds": time.perf_counter() - t0, **({"batch_rule": coupled} if coupled else {})}
    # the compaction, all sequences, on the device
    cnt = gpu["cmc"].reshape(-1).long()
    o = torch.from_numpy(st.evicted_kv_offsets.reshape(-1).astype(np.int64)).to(dev)
    st_ = torch.cumsum(cnt, 0) - cnt
    mrows = torch.repeat_interleave(o - st_, cnt) + torch.arange(int(cnt.sum()), device=dev)
    mv = gpu["cmi"][mrows].long()
    it = torch.uint8 if k_cache.element_size() == 1 else torch.int16
    kb, vb = k_cache.view(it), v_cache.view(it)
    ok_kv = True
    for lo in range(0, mv.shape[0], 1 << 20):
        d, s_ = mv[lo:lo + (1 << 20), 0], mv[lo:lo + (1 << 20), 1]
        ok_kv &= bool(torch.equal(kb[d // bs, :, d % bs, :], kb[s_ // bs, :, s_ % bs, :]))
        ok_kv &= bool(torch.equal(vb[d // bs, :, d % bs], vb[s_ // bs, :, s_ % bs]))
    compared.append("k_cache/v_cache rows of every moved slot == their sources (on the device)")
    if not ok_kv:
        bad.append("k_cache/v_cache moved rows")
    m2, p2 = torch.from_numpy(st.metrics).to(dev), torch.from_numpy(st.token_positions).to(dev)
    m2.view(-1)[mv[:, 0]] = m2.view(-1)[mv[:, 1]]
    p2.view(-1)[mv[:, 0]] = p2.view(-1)[mv[:, 1]]
    compared += ["kv_metrics", "kv_position"]
    if not torch.equal(m2, wm): bad.append("kv_metrics")
    if not torch.equal(p2, wp): bad.append("kv_position")
    return {"workload": "the timed workload itself", "mode": mode, "bit_exact": not bad,
            "sampled_sequences": sel, **({"batch_rule": coupled} if coupled else {}),
            "what": (f"the oracle on a sub-batch of {len(sel)} of the {B} sequences (per_sequence: a sequence's schedule does not "
                     "depend on the others): every output of their heads against the full batch's; the compaction of all "
                     "sequences checked on the device") if coupled is None else (
                     f"the oracle's two-stage form of the reference's batch > 1 rule: finite-threshold counts of all {B} sequences "
                     "-> the rule as arithmetic -> blocks freed per sequence, ALL sequences against the device's; the per-sequence "
                     f"schedule with those counts for {len(sel)} sampled sequences, every output of their heads against the full "
                     "batch's; the compaction of all sequences checked on the device"),
            "compared": compared, "mismatched": bad, "seconds": time.perf_counter() - t0}


# --------------------------------------------------------------------------- roofline helpers
def s1_call_forms(ds, st, evicted, cmi, cmc, k_cache, v_cache, wm, wp, bs, steps=10, warmup=3, ref_counts=None):
    """S1 through the call the FORK makes -- ``evicted_blocks_per_seq`` a fresh device int tensor, the last token
    positions a fresh device tensor, the protected windows a tuple, no ``total_slots`` (reference
    vllm/kvcompress/scheduler.py:245-260, 491-499) -- next to the list form the other figures of this file use
    (host list of counts + ``total_slots=``: the method never waits for the device).  The tensor form costs one
    launch and one wait (CompressionMetrics._batch_summary_enqueue / _read) and must take the same schedule.
    Both forms are timed the same way twice: with HIP events inside the full step loop (S2 and S3 enqueued behind
    every call, as for ``stages_ms``; the tensor form's wait drains the previous step first, so what the events see
    on top of the kernels is the host's path from the wait to the first launch), and as host wall-clock from an
    idle device to the results (``torch.cuda.synchronize()`` on both sides: what an engine that has just read back
    its sampled tokens sees)."""
    import gc
    import time
    import torch
    from vllm_kvcompress_amd import _custom_ops as ops
    cm, dev = ds.cm, ds.cm.device
    N = st.total_slots
    seq_idx, prot = list(st.seq_indices), list(st.protected)
    seq_lens = [int(x) + 1 for x in st.seq_positions]
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def fork_args():
        return (seq_idx, torch.tensor(seq_lens, dtype=torch.int, device=dev) - 1,
                torch.tensor(evicted, dtype=torch.int, device=dev), ds.context_lens, ds.hanging_token_count,
                ds.evicted_kv_offsets, tuple(prot))

    def call(form, args):
        if form == "reference_call_form":
            return cm.schedule_evictions(*args)
        return cm.schedule_evictions(seq_idx, ds.seq_positions, evicted, ds.context_lens, ds.hanging_token_count,
                                     ds.evicted_kv_offsets, prot, total_slots=N)

    out = {}
    gc.collect()
    gc.disable()
    for form in ("list_form", "reference_call_form"):
        marks = [[ev(), ev()] for _ in range(steps)]
        wall = []
        eli = ekc = ebc = None
        for i in range(-warmup, steps):
            del eli, ekc, ebc
            args = fork_args()                     # (the scheduler's own work, outside the call that is timed)
            if i >= 0: marks[i][0].record()
            eli, ekc, ebc = call(form, args)
            if i >= 0: marks[i][1].record()
            ops.schedule_cache_moves(cmi, cmc, eli, ekc, ds.evicted_kv_offsets, ds.block_tables, ds.context_lens, bs)
            ops.execute_cache_moves(k_cache, v_cache, wm, wp, cmi, cmc, ds.evicted_kv_offsets, 1, 16)
        torch.cuda.synchronize()
        rec = {"ms": sum(a.elapsed_time(b) for a, b in marks) / steps, "schedule": cm.last_schedule_reason}
        if ref_counts is not None:
            rec["same_counts"] = bool(torch.equal(ekc, ref_counts))
        for i in range(-warmup, steps):
            del eli, ekc, ebc
            args = fork_args()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eli, ekc, ebc = call(form, args)
            torch.cuda.synchronize()
            if i >= 0: wall.append(time.perf_counter() - t0)
        rec["wall_from_idle_ms"] = 1e3 * sorted(wall)[len(wall) // 2]
        out[form] = rec
        del eli, ekc, ebc
    gc.enable()
    out["delta_ms"] = out["reference_call_form"]["ms"] - out["list_form"]["ms"]
    out["delta_wall_from_idle_ms"] = out["reference_call_form"]["wall_from_idle_ms"] - out["list_form"]["wall_from_idle_ms"]
    out["what"] = ("schedule_evictions as the fork calls it (device int tensor of counts, fresh position tensor, no total_slots; "
                   "scheduler.py:245-260, 491-499) against the list form with total_slots=; ms: HIP events in the full step loop, "
                   f"{steps} steps; wall_from_idle_ms: median host wall-clock, device idle at entry, results complete at exit")
    return out


def traffic_floor(cmi, cmc, offs, bs, block_bytes):
    """HBM bytes the cache LAYOUT forces for this move list: a destination block is
    rewritten whole (K + V images) and, unless every slot of it is overwritten, read whole first;
    every source block is read whole once; + the 8 B metric / position pair of the same slots and
    the move list itself."""
    import torch
    cnt = cmc.reshape(-1).long()
    total = int(cnt.sum())
    if total == 0:
        return {"bytes": 0, "dst_blocks": 0, "dst_blocks_fully_overwritten": 0, "src_blocks": 0}
    o = offs.reshape(-1).long()
    start = torch.cumsum(cnt, 0) - cnt
    rows = torch.repeat_interleave(o - start, cnt) + torch.arange(total, device=cmi.device)
    dst = cmi[rows, 0].long() // bs
    src = cmi[rows, 1].long() // bs
    _, per_dst = torch.unique(dst, return_counts=True)
    d = int(per_dst.numel())
    full = int((per_dst == bs).sum())
    s = int(torch.unique(src).numel())
    img = 2 * block_bytes + 8 * bs                       # K + V images + metric / position rows
    return {"bytes": (d - full) * img + d * img + s * img + total * 8,
            "dst_blocks": d, "dst_blocks_fully_overwritten": full, "src_blocks": s}


PATTERN_CEILING_TYPICAL_GBPS = 5100.0     # random 4 KiB images inside one 64 GiB region (6100 across regions)


def frac_ceiling(alg_bytes, floor, ceiling, frac):
    """What `frac` (algorithmic bytes / time / 8 TB/s) CAN reach for this move list in the reference's
    cache layout: the layout forces `traffic_floor_bytes` through HBM for `algorithmic_bytes` of
    payload (one V element per 32 B sector, one K piece per 16 B: a block that receives one slot is
    read and rewritten whole), and randomly placed block images stream at the pattern ceiling, not
    at the 8 TB/s of a linear stream.  alg_frac_ceiling = alg / floor x pattern ceiling / peak."""
    if not floor or not floor.get("bytes"):
        return {"alg_frac_ceiling": None, "frac_of_ceiling": None}
    rmw = floor["dst_blocks_fully_overwritten"] * 2 < floor["dst_blocks"]
    if ceiling:
        pc, src = ceiling["rmw_2R1W" if rmw else "copy_1R1W"], "measured on this run's cache (pattern_ceiling_GBps)"
    else:
        pc, src = PATTERN_CEILING_TYPICAL_GBPS, "typical figure (no probe in this run)"
    c = alg_bytes / floor["bytes"] * pc / HBM_PEAK_GBPS
    return {"alg_frac_ceiling": c, "frac_of_ceiling": frac / c if c > 0 else None,
            "layout_amplification": floor["bytes"] / alg_bytes if alg_bytes else None,
            "alg_frac_ceiling_pattern_GBps": pc, "alg_frac_ceiling_source": src}


def pattern_ceiling(k_cache, v_cache, block_bytes, iters=6):
    """What THIS box sustains for the compaction kernel's bare access pattern (tools/kvc_probe.hip:
    random block images, one round trip per run, no logic), on the bench's own cache buffers.
    The same binary differs by 20 % between MI355X boxes (profiles/r2_compact_variants.md)."""
    import ctypes
    import torch
    path = os.path.join(REPO, "tools", "libkvc_probe.so")
    if block_bytes not in (4096, 8192) or not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    fn = lib.kvc_probe_block_stream
    fn.restype = ctypes.c_int32
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                   ctypes.c_int32, ctypes.c_void_p]
    nb = int(k_cache.shape[0])                      # block ids from the whole cache, like the sequence's own
    nruns = min(nb // 2, 1 << 19)
    runs = torch.randperm(nb, device=k_cache.device)[:2 * nruns].to(torch.int32).reshape(nruns, 2).contiguous()
    stream = torch.cuda.current_stream(k_cache.device).cuda_stream
    out = {}
    for name, rd, per_run in (("rmw_2R1W", 1, 6 * block_bytes), ("copy_1R1W", 0, 4 * block_bytes)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(2 + iters):
            if i == 2:
                a.record()
            if fn(k_cache.data_ptr(), v_cache.data_ptr(), runs.data_ptr(), nruns, block_bytes, rd, stream):
                return None
        b.record()
        torch.cuda.synchronize()
        out[name] = per_run * nruns / (a.elapsed_time(b) / iters * 1e-3) / 1e9
    out["what"] = (f"tools/kvc_probe.hip on this GPU, {nruns} runs over random {block_bytes}-byte images of the cache's {nb} blocks: "
                   "rmw = read destination + source K + V images, write destination "
                   "(random evictions); copy = read source, write destination (clustered evictions)")
    return out


def s0_stages(args, device):
    """Stage S0 (metric aggregation, rows A2a / A2b / A2c) at the bench shape, COLD: each call works
    on the next of several buffer sets whose total exceeds twice the 256 MiB Infinity Cache, so
    the rates are HBM rates (round 1's single-buffer numbers were cache numbers)."""
    import torch
    import vllm_kvcompress_amd
    from vllm_kvcompress_amd.kvcompress.prefill import accumulate_prefill_tile
    lib = vllm_kvcompress_amd.load()
    L, H, bs, qpk = args.layers, args.kv_heads, args.block_size, 4
    T = args.seq_len                # (configs[4]: 65 536 keys, its own size)
    slots = L * H * (T // bs) * bs
    stream = torch.cuda.current_stream(device).cuda_stream
    set_bytes = slots * (4 * qpk + 4)
    nsets = max(2, -(-(640 << 20) // set_bytes))
    temps = [torch.rand((slots, qpk), device=device) for _ in range(nsets)]
    mets = [torch.zeros(slots, device=device) for _ in range(nsets)]

    def timed(fn, iters):
        for i in range(nsets):
            fn(i)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(iters):
            fn(i)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters

    out, iters = {}, 3 * nsets
    for name, clear, bps in (("S0_aggregate_decode", 0, 4 * qpk + 8), ("S0_aggregate_decode_fused_clear", 1, 8 * qpk + 8)):
        ms = timed(lambda i: lib.kvc_aggregate_decode(mets[i % nsets].data_ptr(), temps[i % nsets].data_ptr(),
                                                      slots, qpk, 1, clear, stream), iters)
        out[name] = {"ms": ms, "slots": slots, "bytes_per_slot": bps, "GBps": slots * bps / ms / 1e6,
                     "frac_of_hbm_peak": slots * bps / ms / 1e6 / HBM_PEAK_GBPS,
                     "buffer_sets": nsets, "working_set_MiB": nsets * set_bytes >> 20}
    # aggregate_prefill: the T tokens of one layer scattered into a cold metric store
    pm = [torch.rand((T, H * qpk), device=device) for _ in range(nsets)]
    sm = torch.randperm(slots, device=device)[:T * H].reshape(T, H).contiguous()
    ms = timed(lambda i: lib.kvc_aggregate_prefill(mets[i % nsets].data_ptr(), pm[i % nsets].data_ptr(),
                                                   sm.data_ptr(), T, H, qpk, stream), iters)
    # 64 B sectors: a scattered 4 B read-modify-write moves a whole sector each way
    out["S0_aggregate_prefill"] = {"ms": ms, "tokens": T, "algorithmic_GBps": T * H * (4 * qpk + 16) / ms / 1e6,
                                   "note": "T tokens x H heads of one layer; 8 B slot index + 16 B of "
                                           "weights read, one scattered 4 B metric updated per (token, head)"}
    del temps, mets, pm, sm
    torch.cuda.empty_cache()
    # prefill epilogue: one query block of softmax probabilities (Hq x qb x K floats, 4 GiB at the
    # bench shape: far beyond any cache)
    Hq, qb, K = H * qpk, 1024, T
    probs = torch.rand((Hq, qb, K), device=device)
    outm = torch.zeros((K, Hq), device=device)
    for _ in range(2):
        accumulate_prefill_tile(outm, probs, K - qb, 0, True, False, True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        accumulate_prefill_tile(outm, probs, K - qb, 0, True, False, True)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 4
    nbytes = Hq * qb * K * 4 + Hq * K * 12
    out["S0_prefill_epilogue"] = {"ms": ms, "tile": f"Hq{Hq} x qb{qb} x K{K} float32", "GBps": nbytes / ms / 1e6,
                                  "frac_of_hbm_peak": nbytes / ms / 1e6 / HBM_PEAK_GBPS}
    del probs, outm
    torch.cuda.empty_cache()
    return out


def workload_flags(args):
    """the command-line flags that reproduce this run's workload (for the PMC passes)"""
    f = ["--config", args.config, "--layers", str(args.layers), "--kv-heads", str(args.kv_heads),
         "--head-size", str(args.head_size), "--block-size", str(args.block_size), "--seq-len", str(args.seq_len),
         "--batch", str(args.batch), "--keep", str(args.keep), "--protected", str(args.protected),
         "--metric-shape", args.metric_shape, "--mode", args.mode, "--kv-dtype", args.kv_dtype]
    if args.steady_cap:
        f += ["--steady-cap", str(args.steady_cap)]
    if args.contiguous_blocks:
        f.append("--contiguous-blocks")
    if args.lean:
        f.append("--lean")
    f += ["--block-layout", args.block_layout, "--call-form", args.call_form]
    return f


def live_pmc_traffic(extra_flags, timeout=240, kernel="compact_runs_kernel"):
    """HBM bytes per launch of the compaction kernel from the TCC counters, measured NOW: two short
    runs of this same script under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate
    passes, no trace domain, as MI355X_MICROARCH.md prescribes; FETCH doubled for a wide streaming
    kernel on gfx950).  None if rocprofv3 is missing or a pass fails -- the committed figure of
    profiles/ is used then."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--headline-only", "--no-cpu-baseline",
           "--no-probe", "--no-live-traffic", "--no-parity-gate", "--detail-json", ""] + list(extra_flags)
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            try:
                r = subprocess.run([exe, "--pmc", counter, "-d", out, "--output-format", "csv", "--"] + cmd,
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            except (subprocess.TimeoutExpired, OSError):
                return None
            files = glob.glob(os.path.join(out, "*", "*_counter_collection.csv"))
            if r.returncode != 0 or not files:
                return None
            v = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                 if row["Counter_Name"] == counter and kernel in row["Kernel_Name"]]
            if not v:
                return None
            vals[counter] = sum(v) / len(v)
    return {"hbm_bytes_per_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
            "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"]}


def measure_workload(a2, seed, device, steps, warmup, probe):
    """One workload, timed like the main loop (HIP events on the launch stream around every stage
    and around the compaction kernel): per-stage ms, whole-step rate, the compaction kernel's
    roofline numbers.  Used for the engine-sized cache and for the other BASELINE configurations
    the default run reports next to its headline.  Returns None if the workload does not fit."""
    import torch
    from vllm_kvcompress_amd import _custom_ops as ops
    e = 1 if a2.kv_dtype == "fp8" else 2
    bs, hd = a2.block_size, a2.head_size
    block_bytes = hd * bs * e
    try:
        st, ds, evicted, k_cache, v_cache = build_workload(a2, seed, device)
    except torch.OutOfMemoryError:
        torch.cuda.empty_cache()
        return None
    N = st.total_slots
    wm, wp = ds.cm.metrics.clone(), ds.cm.token_positions.clone()
    # the fork's own call form, like the headline (main()): a plain persistent move workspace (reference
    # scheduler.py:74-86), fresh device tensors of counts and positions and a fresh cache_moves_count per step
    # (:245-260, 508-512), no total_slots=
    cmi = torch.empty((N, 2), dtype=torch.int32, device=device)
    seq_idx, prot = list(st.seq_indices), list(st.protected)
    prot_t, seq_lens = tuple(prot), [int(x) + 1 for x in st.seq_positions]
    BLH = (st.num_seqs, st.num_layers, st.num_kv_heads)
    snap = parity_snapshot(k_cache, v_cache) if N <= PARITY_ORACLE_MAX_SLOTS else None
    ev = lambda: torch.cuda.Event(enable_timing=True)
    marks = [[ev() for _ in range(5)] for _ in range(steps)]
    out = {}
    import gc
    gc.collect()
    gc.disable()                       # (see main(): the collector's pauses are the host's, not the path's)
    eli = ekc = ebc = None
    for i in range(-warmup, steps):
        rec = i >= 0
        out.clear()
        del eli, ekc, ebc              # (the previous step's results go out of scope, as in the engine's loop)
        k_t = torch.tensor(evicted, dtype=torch.int, device=device)
        pos_t = torch.tensor(seq_lens, dtype=torch.int, device=device) - 1
        if rec: marks[i][0].record()
        eli, ekc, ebc = ds.cm.schedule_evictions(seq_idx, pos_t, k_t, ds.context_lens,
                                                 ds.hanging_token_count, ds.evicted_kv_offsets, prot_t)
        if rec: marks[i][1].record()
        cmc = torch.empty(BLH, dtype=torch.int32, device=device)
        ops.schedule_cache_moves(cmi, cmc, eli, ekc, ds.evicted_kv_offsets, ds.block_tables, ds.context_lens, bs)
        if rec: marks[i][2].record()
        ops._execute_cache_moves(k_cache, v_cache, wm, wp, cmi, cmc, ds.evicted_kv_offsets, "plan")
        if rec: marks[i][3].record()
        ops._execute_cache_moves(k_cache, v_cache, wm, wp, cmi, cmc, ds.evicted_kv_offsets, "apply")
        if rec: marks[i][4].record()
        out["ekc"], out["ebc"], out["eli"] = ekc, ebc, eli
    torch.cuda.synchronize()
    gc.enable()
    parity = parity_gate(a2, st, ds, evicted, a2.mode, dict(eli=out["eli"], ekc=out["ekc"], ebc=out["ebc"], cmi=cmi, cmc=cmc),
                         snap, k_cache, v_cache, wm, wp, lean=bool(a2.lean))
    del snap
    if parity["bit_exact"] is False:
        raise SystemExit(f"bench.py: PARITY GATE FAILED ({a2.config}): {json.dumps(parity)}")
    ms = lambda a, b: sum(m[a].elapsed_time(m[b]) for m in marks) / steps
    kernel_ms, step_ms = ms(3, 4), ms(0, 4)
    moved, evs = int(cmc.sum().item()), int(out["ekc"].sum().item())
    alg = moved * alg_bytes_per_move(hd, e) + 8 * st.total_heads
    floor = traffic_floor(cmi, cmc, ds.evicted_kv_offsets, bs, block_bytes)
    s1 = ms(0, 1)
    res = {
        "cache_blocks": st.num_blocks, "cache_GiB": 2 * st.num_blocks * block_bytes / 2 ** 30,
        "candidate_slots": N, "evicted_slots": evs, "moved_slots": moved,
        "ms_per_step": step_ms, "value": (evs + moved) / (step_ms * 1e-3),
        "stages_ms": {"S1_schedule_evictions": s1, "S2_schedule_moves": ms(1, 2), "S3_execute_moves": ms(2, 4)},
        "S1_schedule": ds.cm.last_schedule_path(),
        # (a "[pivots: the call before]" here: the small-eviction schedule did not sample -- on this bench's static store
        # the pivots of the call before are exact; in an engine they are one decode step of attention old, which
        # profiles/r6_harvest_soak.txt runs for 400 steps of an evolving state without a pass that listed too little)
        "S1_schedule_reason": ds.cm.last_schedule_reason,
        # S1 against ITS lower bound (12.75 B per candidate slot)
        "S1_lower_bound_GBps": N * 12.75 / (s1 * 1e-3) / 1e9,
        "S1_frac_of_hbm_peak_at_lower_bound": N * 12.75 / (s1 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "roofline": {"avg_launch_ms": kernel_ms, "achieved": alg / (kernel_ms * 1e-3) / 1e9,
                     "frac": alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "floor_GBps": floor["bytes"] / (kernel_ms * 1e-3) / 1e9,
                     "frac_of_floor": floor["bytes"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "pattern_ceiling_GBps": pattern_ceiling(k_cache, v_cache, block_bytes) if probe else None},
        "timing": f"{steps} steps after {warmup} warm-up steps, HIP events on the launch stream",
        "call_form": "fork (fresh device tensors of counts / positions, no total_slots=, plain move workspace); "
                     "S1_call_forms has the hinted form next to it",
        "parity_checked": parity,
    }
    res["roofline"].update(frac_ceiling(alg, floor, res["roofline"]["pattern_ceiling_GBps"], res["roofline"]["frac"]))
    if st.total_slots < st.num_blocks * bs // 2 and not ds.cm.last_schedule_path().startswith("small_eviction"):
        # the batch is sparse in its cache: the same steps once more with BlockState.block_tables handed
        # to schedule_evictions (optional argument: the key pass then goes through the tables instead
        # of sweeping every block's metadata) -- S1 only, the other stages are untouched by it
        n2 = min(steps, 10)
        m2 = [[ev(), ev()] for _ in range(n2)]
        for i in range(-2, n2):
            if i >= 0: m2[i][0].record()
            eli, ekc, ebc = ds.cm.schedule_evictions(seq_idx, ds.seq_positions, evicted, ds.context_lens,
                                                     ds.hanging_token_count, ds.evicted_kv_offsets, prot, total_slots=N,
                                                     block_tables=ds.block_tables)
            if i >= 0: m2[i][1].record()
            ops.schedule_cache_moves(cmi, cmc, eli, ekc, ds.evicted_kv_offsets, ds.block_tables, ds.context_lens, bs)
            ops._execute_cache_moves(k_cache, v_cache, wm, wp, cmi, cmc, ds.evicted_kv_offsets, "plan")
            ops._execute_cache_moves(k_cache, v_cache, wm, wp, cmi, cmc, ds.evicted_kv_offsets, "apply")
        torch.cuda.synchronize()
        same = bool(torch.equal(ekc, out["ekc"]))
        res["S1_with_block_tables"] = {
            "ms": sum(a.elapsed_time(b) for a, b in m2) / n2, "steps": n2, "used": bool(ds.cm.last_used_block_tables),
            "same_counts": same,
            "what": "schedule_evictions(..., block_tables=BlockState.block_tables): an extension of the reference's "
                    "signature (INTEGRATION.md); every other figure of this entry is measured without it"}
    if res["S1_schedule"] == "small_eviction" and ds.cm.last_pivot_memory_used:
        # the figures above ran on the pivots of the call before (pivot memory; exact on this bench's static store, one
        # decode step of attention old in an engine): the same steps with every call sampling the store for its pivots
        n2 = min(steps, 10)
        m2 = [[ev(), ev()] for _ in range(n2)]
        ds.cm.pivot_memory = False
        eli = ekc = ebc = None
        for i in range(-2, n2):
            del eli, ekc, ebc
            if i >= 0: m2[i][0].record()
            eli, ekc, ebc = ds.cm.schedule_evictions(seq_idx, ds.seq_positions, evicted, ds.context_lens,
                                                     ds.hanging_token_count, ds.evicted_kv_offsets, prot, total_slots=N)
            if i >= 0: m2[i][1].record()
        torch.cuda.synchronize()
        ds.cm.pivot_memory = True
        res["S1_sampled_pivots"] = {
            "ms": sum(a.elapsed_time(b) for a, b in m2) / n2, "steps": n2, "same_counts": bool(torch.equal(ekc, out["ekc"])),
            "what": "S1 with CompressionMetrics.pivot_memory = False: every call samples the store for its pivots "
                    "(sampling pass + pivot kernel, twice the candidates in the collecting pass)"}
        del eli, ekc, ebc
    if not a2.lean:
        forms = s1_call_forms(ds, st, evicted, cmi, cmc, k_cache, v_cache, wm, wp, bs, steps=min(steps, 10), ref_counts=out["ekc"])
        res["S1_reference_call_form_ms"] = forms["reference_call_form"]["ms"]
        res["S1_call_forms"] = forms
    if a2.steady_cap and res["S1_schedule"] == "small_eviction" and not a2.lean:
        del wm, wp
        res["decode_step"] = decode_step_compare(a2, st, ds, evicted, k_cache, v_cache, cmi, cmc, device,
                                                 steps=min(steps, 12), warmup=3)
        if res["decode_step"].get("parity_checked", {}).get("bit_exact") is False:
            raise SystemExit(f"bench.py: PARITY GATE FAILED ({a2.config}, decode step): {json.dumps(res['decode_step'])}")
        wm = wp = None
    del k_cache, v_cache, ds, cmi, wm, wp
    torch.cuda.empty_cache()
    return res


S0_PMC_FILES = ("r6_decode_step_pmc.json", "r5_decode_step_pmc.json", "r4_decode_step_pmc.json")     # newest first


def _committed_s0_traffic(kernel_tag):
    """(HBM bytes per launch of the aggregation kernel, the file it comes from) from the committed PMC collection --
    a separately profiled run of tools/decode_step.py (tools/collect_decode_step_pmc.sh), not a measurement of this run"""
    for name in S0_PMC_FILES:
        try:
            d = json.load(open(os.path.join(REPO, "profiles", name)))
            for kname, v in d.get("aggregation", {}).items():
                if kernel_tag in kname:
                    return v["hbm_bytes_per_launch"], name
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def decode_step_compare(a2, st, ds, evicted, k_cache, v_cache, cmi, cmc, device, steps, warmup):
    """The WHOLE decode step of the continual steady state -- S0 aggregate_decode (reference metrics.py:429-439: the
    whole store, every step; 4 * qpk + 8 B per slot) + S1 + S2 + S3 -- timed twice from the same state with the same
    attention mass: as the TWO SWEEPS of the store the reference's flow implies (aggregate_decode, then
    schedule_evictions' own collecting pass) and HARVEST-AHEAD (CompressionMetrics.aggregate_decode_and_harvest: the
    sums are looked at while they pass through registers, the schedule call does not stream the store again;
    include/kvc_mi355x.h ABI version 5).  Both variants must leave the same store and the same schedule; the final
    state is checked against the oracle on a sample of the sequences.  The state is static apart from the metrics
    (nobody frees the evicted blocks between bench steps), so the harvest's pivots also cover the keys an engine
    would have freed the step before: the harvest measured here is ~1.4 x as long as an engine's."""
    import gc
    import torch
    from vllm_kvcompress_amd import _custom_ops as ops
    cm = ds.cm
    qpk, N, bs = 4, st.total_slots, st.block_size
    slots = st.num_blocks * bs
    g = torch.Generator(device=device)
    g.manual_seed(99)
    try:
        # attention mass per (slot, query head): sum_q t^2 ~ 3 per step, next to metrics that are a permutation of
        # 0 .. slots-per-sequence: a key near the pivot drifts by a few ranks per step
        temp = torch.rand((st.num_blocks, bs, qpk), dtype=torch.float32, device=device, generator=g) * 1.5
        m0 = cm.metrics.clone()
        wm, wp = cm.metrics.clone(), cm.token_positions.clone()
    except torch.OutOfMemoryError:
        torch.cuda.empty_cache()
        return {"skipped": "does not fit"}
    saved = (cm.num_queries_per_kv, cm._temp_metrics, cm.harvest_ahead)
    saved_spec = cm.speculative_harvest
    cm.num_queries_per_kv, cm._temp_metrics = qpk, temp
    seq_idx, prot = list(st.seq_indices), list(st.protected)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    out, keep = {}, {}
    # "fork_flow": the fork's flow with nothing changed -- aggregate_decode() at the end of an iteration, schedule_evictions
    # at the start of the next -- where aggregate_decode() harvests for the call it predicts (the last call's batch, one token
    # further on: CompressionMetrics.speculative_harvest, the default).  A decode step advances every sequence by a token, so
    # this variant's schedule calls get positions that advance by one per step (the other two keep the static store's).
    pos_steps = [ds.seq_positions + j for j in range(steps + warmup + 1)]
    for variant in ("two_sweeps", "fork_flow", "harvest_ahead"):       # (harvest_ahead last: the gate below checks its outputs)
        cm.metrics.copy_(m0)
        cm.harvest_ahead = True if variant == "harvest_ahead" else (None if variant == "fork_flow" else False)
        cm.speculative_harvest = variant == "fork_flow"
        cm._hv = cm._hv_lists = None
        misses0, used = cm.harvest_misses, 0
        marks = [[ev() for _ in range(5)] for _ in range(steps)]
        gc.collect()
        gc.disable()
        eli = ekc = ebc = None
        for i in range(-warmup, steps):
            rec = i >= 0
            del eli, ekc, ebc
            pos_now = pos_steps[i + warmup] if variant == "fork_flow" else ds.seq_positions
            if rec: marks[i][0].record()
            if variant == "harvest_ahead":
                cm.aggregate_decode_and_harvest(seq_idx, ds.seq_positions, prot, ds.context_lens, total_slots=N, fuse_clear=False)
            else:
                cm.aggregate_decode(fuse_clear=False)
            if rec: marks[i][1].record()
            eli, ekc, ebc = cm.schedule_evictions(seq_idx, pos_now, evicted, ds.context_lens,
                                                  ds.hanging_token_count, ds.evicted_kv_offsets, prot, total_slots=N)
            used += int(rec and cm.last_harvest_used)
            if rec: marks[i][2].record()
            ops.schedule_cache_moves(cmi, cmc, eli, ekc, ds.evicted_kv_offsets, ds.block_tables, ds.context_lens, bs)
            if rec: marks[i][3].record()
            ops.execute_cache_moves(k_cache, v_cache, wm, wp, cmi, cmc, ds.evicted_kv_offsets, 1, 16)
            if rec: marks[i][4].record()
        torch.cuda.synchronize()
        gc.enable()
        ms = lambda a, b: sum(m[a].elapsed_time(m[b]) for m in marks) / steps
        out[variant] = {
            "ms_per_step": ms(0, 4),
            "stages_ms": {"S0_aggregate_decode": ms(0, 1), "S1_schedule_evictions": ms(1, 2), "S2_schedule_moves": ms(2, 3),
                          "S3_execute_moves": ms(3, 4)},
            "S0_GBps": slots * (4 * qpk + 8) / (ms(0, 1) * 1e-3) / 1e9,
            "S0_roofline": {"kernel": "kvc::aggregate_harvest_kernel" if variant != "two_sweeps" else "kvc::aggregate_decode_q4_kernel",
                            "bound": "hbm", "achieved": slots * (4 * qpk + 8) / (ms(0, 1) * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                            "unit": "GB/s", "frac": slots * (4 * qpk + 8) / (ms(0, 1) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                            "algorithmic_bytes_per_launch": slots * (4 * qpk + 8),
                            "traffic": _committed_s0_traffic("aggregate_harvest" if variant != "two_sweeps" else "aggregate_decode_q4")[0],
                            "traffic_source": f"profiles/{_committed_s0_traffic('aggregate_harvest' if variant != 'two_sweeps' else 'aggregate_decode_q4')[1]} "
                                              "(a separately profiled run of tools/decode_step.py, NOT a measurement of this run: rocprofv3 --pmc "
                                              "FETCH_SIZE / WRITE_SIZE in separate passes; 2 x FETCH + WRITE)",
                            "timing": "the S0 stage: HIP events on the launch stream around the call (one kernel + the counters' fill)"},
            "S1_schedule": cm.last_schedule_path(), "harvested_steps": used, "harvest_misses": cm.harvest_misses - misses0}
        if variant == "fork_flow":
            # its own verdict: the last step's schedule against the oracle's schedule of the store, at the last step's positions
            import copy as _copy2
            st3 = _copy2.copy(st)
            st3.metrics = cm.metrics.cpu().numpy()
            st3.seq_positions = pos_steps[steps + warmup - 1].cpu().numpy()
            pf = parity_gate_sampled(a2, st3, evicted, dict(eli=eli, ekc=ekc, ebc=ebc, cmi=cmi, cmc=cmc), k_cache, v_cache,
                                     None, None, schedule_only=True, mode=a2.mode)
            out[variant]["parity_checked"] = pf
            out[variant]["what"] = ("aggregate_decode() then schedule_evictions, as the fork calls them; aggregate_decode() harvests for the "
                                    "call it predicts (positions + 1 per step), the schedule call verifies on the device")
            continue
        keep[variant] = (cm.metrics.clone(), eli.clone(), ekc.clone(), ebc.clone(), cmc.clone(),
                         torch.cat([cmi[o:o + int(c)] for o, c in zip(st.evicted_kv_offsets.reshape(-1)[:64].tolist(),
                                                                      cmc.reshape(-1)[:64].tolist())]) if N else cmi[:0])
    a, b = keep["two_sweeps"], keep["harvest_ahead"]
    names = ("metrics", "evicted_logical_indices", "evicted_kv_count", "evicted_block_count", "cache_moves_count",
             "cache_moves_idx (first 64 heads)")
    differ = [n for n, x, y in zip(names, a, b) if not torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x,
                                                                   y.view(torch.int32) if y.dtype == torch.float32 else y)]
    # the oracle on the final state (the harvested variant ran last: its outputs are the live ones)
    import copy as _copy
    st2 = _copy.copy(st)
    st2.metrics = cm.metrics.cpu().numpy()
    parity = parity_gate_sampled(a2, st2, evicted, dict(eli=eli, ekc=ekc, ebc=ebc, cmi=cmi, cmc=cmc), k_cache, v_cache,
                                 None, None, schedule_only=True, mode=a2.mode)
    parity["bit_exact"] = bool(parity["bit_exact"]) and not differ
    parity["variants_agree"] = not differ
    parity["variants_differ_in"] = differ
    parity["what"] = ("both variants from the same store with the same attention mass: the stores and the last step's outputs must "
                      "be equal; the last step of the harvested variant against the oracle's schedule of that store ("
                      + ("a sample of the sequences" if a2.mode == "per_sequence" else
                         "the reference's batch > 1 rule in the oracle's two-stage form: every sequence's freed blocks, and the full "
                         "schedule of a sample of the sequences")
                      + "; S0's sums against the oracle: tests/test_gpu_harvest.py, tests/test_gpu_parity.py)")
    if out.get("fork_flow", {}).get("parity_checked", {}).get("bit_exact") is False:
        parity["bit_exact"] = False
        parity["fork_flow_mismatched"] = out["fork_flow"]["parity_checked"].get("mismatched")
    cm.metrics.copy_(m0)
    cm.num_queries_per_kv, cm._temp_metrics, cm.harvest_ahead = saved
    cm.speculative_harvest = saved_spec
    cm._hv = cm._hv_lists = None
    if a2.block_size in (16, 32) and a2.head_size in (64, 128):
        del wm, wp, keep
        wm = wp = keep = None
        out["fused_attention"] = decode_step_fused_attention(a2, st, ds, evicted, k_cache, v_cache, cmi, cmc, temp, m0, device,
                                                            steps, warmup)
        fa = out["fused_attention"]
        if isinstance(fa.get("parity_checked"), dict) and fa["parity_checked"].get("bit_exact") is False:
            parity["bit_exact"] = False
            parity["fused_attention_mismatched"] = fa["parity_checked"].get("mismatched")
    out["saved_ms_per_step"] = out["two_sweeps"]["ms_per_step"] - out["harvest_ahead"]["ms_per_step"]
    out["what"] = ("S0 + S1 + S2 + S3 per decode step; num_queries_per_kv 4, aggregate_decode without the fused clear "
                   f"(24 B per slot over {slots} slots); {steps} steps after {warmup} warm-up steps")
    out["parity_checked"] = parity
    del temp, m0, wm, wp, keep
    torch.cuda.empty_cache()
    return out


def decode_step_fused_attention(a2, st, ds, evicted, k_cache, v_cache, cmi, cmc, temp, m0, device, steps, warmup):
    """The decode step WITHOUT a sweep of the metric store (the reference author's to-do, vllm/kvcompress/README.md:32, 49):
    the L launches of the fused-metric decode attention add the step's weights straight into the store (no temp_metrics,
    no aggregate_decode) and their epilogues make the candidate lists of the schedule call that follows
    (CompressionMetrics.begin_attention_harvest; include/kvc_mi355x.h ABI version 6) -- S1 then runs on lists, S0 does not
    exist.  Timed: the attention of one step in three forms (weights to temp_metrics as the reference's kernel writes
    them | fused into the store | fused + harvest) and S1 + S2 + S3 behind the third.  K / V are overwritten with N(0, 0.5)
    values (the timed workload's caches hold random bits: NaNs) and the store is scaled so that a step's attention moves
    a key near the pivot by about as many ranks as the synthetic mass of the other variants does (sum_q p^2 of a
    softmax over 4 k keys is ~2.4e-7; the stores of this file are permutations of 0 .. slots-per-sequence)."""
    import gc
    import torch
    from vllm_kvcompress_amd import _custom_ops as ops
    cm = ds.cm
    L, H, bs, hd, qpk, B, N = st.num_layers, st.num_kv_heads, st.block_size, a2.head_size, 4, st.num_seqs, st.total_slots
    if k_cache.element_size() != 2:
        return {"skipped": "fp16 caches only"}
    try:
        k_cache.normal_(0.0, 0.5)
        v_cache.normal_(0.0, 0.5)
        q = [(torch.randn((B, H * qpk, hd), device=device) * 0.8).to(torch.float16) for _ in range(2)]
        outb = torch.empty_like(q[0])
        bt = [ds.block_tables[l].contiguous() for l in range(L)]
        ctx = [ds.context_lens[l].contiguous() for l in range(L)]
        wm, wp = cm.metrics.clone(), cm.token_positions.clone()
    except torch.OutOfMemoryError:
        torch.cuda.empty_cache()
        return {"skipped": "does not fit"}
    scale_store = 1.0e-7
    last = ds.seq_positions - 1                            # the token being processed (the schedule call's position - 1)
    buf = torch.zeros((B,), dtype=torch.int32, device=device)
    max_ctx = int(ds.context_lens.max().item())
    seq_idx, prot = list(st.seq_indices), list(st.protected)
    saved = (cm.num_queries_per_kv, cm._temp_metrics, cm.harvest_ahead)
    cm.num_queries_per_kv, cm._temp_metrics, cm.harvest_ahead = qpk, temp, True
    cm._hv = cm._hv_lists = None
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def attention(form, h=None):
        for l in range(L):
            args = (q[l & 1], k_cache, v_cache, H, hd ** -0.5, bt[l], ctx[l], cm.token_positions, last, buf, bs, max_ctx, None,
                    "auto", 1.0, 1.0)
            if form == "record":
                ops.paged_attention_kvc_v1(outb, temp, *args, True)
            else:
                ops.paged_attention_kvc_fused_metrics(outb, cm.metrics, *args, use_l2=True, temp_metrics=temp, harvest=h, layer=l)

    res = {}
    # ---- the attention alone, three forms
    forms = {}
    for form in ("record", "fused", "fused_harvest"):
        cm.metrics.copy_(m0).mul_(scale_store)
        cm.schedule_evictions(seq_idx, ds.seq_positions, evicted, ds.context_lens, ds.hanging_token_count,
                              ds.evicted_kv_offsets, prot, total_slots=N)         # (leaves the pivots)
        n = 4
        marks = [[ev(), ev()] for _ in range(n)]
        for i in range(-1, n):
            h = None
            if form == "fused_harvest":
                h = cm.begin_attention_harvest(seq_idx, ds.seq_positions, prot, ds.context_lens, total_slots=N)
                if h is None:
                    break
            if i >= 0: marks[i][0].record()
            attention(form, h)
            if i >= 0: marks[i][1].record()
            if h is not None:
                cm.end_attention_harvest(h)
        torch.cuda.synchronize()
        if form == "fused_harvest" and h is None:
            forms[form] = None
        else:
            forms[form] = sum(a.elapsed_time(b) for a, b in marks) / n
    res["attention_ms_per_step"] = forms
    kv_bytes = float(sum(int(c.sum().item()) for c in ctx)) * 2 * hd * 2
    res["attention_GBps_kv_only"] = {k: (kv_bytes / (v * 1e-3) / 1e9 if v else None) for k, v in forms.items()}
    if forms["fused_harvest"] and forms["record"]:
        res["harvest_cost_vs_record_kv_metrics"] = forms["fused_harvest"] / forms["record"] - 1.0
    # ---- the whole step
    cm.metrics.copy_(m0).mul_(scale_store)
    cm._hv = cm._hv_lists = None
    misses0, used = cm.harvest_misses, 0
    marks = [[ev() for _ in range(5)] for _ in range(steps)]
    gc.collect()
    gc.disable()
    eli = ekc = ebc = None
    for i in range(-warmup, steps):
        rec = i >= 0
        del eli, ekc, ebc
        h = cm.begin_attention_harvest(seq_idx, ds.seq_positions, prot, ds.context_lens, total_slots=N)
        if rec: marks[i][0].record()
        attention("fused_harvest", h)
        if rec: marks[i][1].record()
        cm.end_attention_harvest(h)
        eli, ekc, ebc = cm.schedule_evictions(seq_idx, ds.seq_positions, evicted, ds.context_lens,
                                              ds.hanging_token_count, ds.evicted_kv_offsets, prot, total_slots=N)
        used += int(rec and cm.last_harvest_used)
        if rec: marks[i][2].record()
        ops.schedule_cache_moves(cmi, cmc, eli, ekc, ds.evicted_kv_offsets, ds.block_tables, ds.context_lens, bs)
        if rec: marks[i][3].record()
        ops.execute_cache_moves(k_cache, v_cache, wm, wp, cmi, cmc, ds.evicted_kv_offsets, 1, 16)
        if rec: marks[i][4].record()
    torch.cuda.synchronize()
    gc.enable()
    ms = lambda a, b: sum(m[a].elapsed_time(m[b]) for m in marks) / steps
    res.update({
        "ms_per_step_without_attention": ms(1, 4),
        "stages_ms": {"attention_L_launches": ms(0, 1), "S0_aggregate_decode": 0.0, "S1_schedule_evictions": ms(1, 2),
                      "S2_schedule_moves": ms(2, 3), "S3_execute_moves": ms(3, 4)},
        "S1_schedule": cm.last_schedule_path(), "S1_schedule_reason": cm.last_schedule_reason,
        "steps_on_the_epilogues_lists": used, "harvest_misses": cm.harvest_misses - misses0,
        "what": f"{L} launches of paged_attention_kvc_fused_metrics(harvest=) per step over {B} sequences (qpk 4, hd {hd}, "
                f"contexts of {max_ctx}), then S1 + S2 + S3; no aggregate_decode launch, no collecting pass; {steps} steps after "
                f"{warmup} warm-up steps"})
    # the oracle on the final state
    import copy as _copy
    st2 = _copy.copy(st)
    st2.metrics = cm.metrics.cpu().numpy()
    parity = parity_gate_sampled(a2, st2, evicted, dict(eli=eli, ekc=ekc, ebc=ebc, cmi=cmi, cmc=cmc), k_cache, v_cache,
                                 None, None, schedule_only=True, mode=a2.mode)
    parity["what"] = ("the last step's schedule (run on the lists the attention's epilogues made) against the oracle's schedule "
                      "of the store the attention left, on a sample of the sequences; the sums themselves and an evolving "
                      "400-step soak against the reference flow: tests/test_gpu_attention_harvest.py, tools/soak_attention_harvest.py")
    res["parity_checked"] = parity
    cm.metrics.copy_(m0)
    cm.num_queries_per_kv, cm._temp_metrics, cm.harvest_ahead = saved
    cm._hv = cm._hv_lists = None
    del wm, wp, q, outb
    return res


def engine_sized_cache_run(args, rank, device, steps=20, warmup=3):
    """The SAME step with the sequence's blocks inside a cache sized the way an engine sizes it: to
    the GPU's memory (vLLM's gpu_memory_utilization) instead of to the sequence.  Candidate, evicted
    slots are identical to the main run and the moved ones differ by a fraction of a percent (another
    seed's metrics); what changes is the number of (free) blocks in the cache tensor.  Reported next to the main line because random 4 KiB block traffic runs at
    5.1 or 6.1 TB/s depending on where the blocks lie in HBM."""
    import copy
    import torch
    e = 1 if args.kv_dtype == "fp8" else 2
    bs, hd = args.block_size, args.head_size
    block_bytes = hd * bs * e
    free, _ = torch.cuda.mem_get_info()
    own = args.layers * args.kv_heads * args.batch * (args.seq_len // bs + 1)
    nb_target = int(args.engine_cache_frac * free / (2 * block_bytes + 8 * bs + 16))
    if nb_target < 8 * own:
        return None
    a2 = copy.copy(args)
    a2.spare_blocks = nb_target / own - 1.0
    res = measure_workload(a2, rank + 1000, device, steps, warmup, not args.no_probe)
    if res is None:
        return None
    res = {"what": f"the same step, the sequence's blocks scattered over a cache that fills {args.engine_cache_frac:.0%} of the free HBM "
                   "(as an engine sizes it) instead of one sized to the sequence", **res}
    # PMC traffic of this placement: measured now (two short rocprofv3 passes over the same step in a
    # cache of the same size), else the committed figure of a separately profiled run in a cache of
    # 61 x the sequence's blocks (tools/collect_profiles.sh: bench.py --spare-blocks 60)
    live = None if args.no_live_traffic else live_pmc_traffic(
        workload_flags(args) + ["--spare-blocks", f"{a2.spare_blocks:.3f}"], timeout=300)
    tj_path = os.path.join(REPO, "profiles", "traffic_engine.json")
    if live is not None:
        t = live["hbm_bytes_per_launch"]
        res["roofline"]["traffic"] = t
        res["roofline"]["traffic_frac_of_peak"] = t / (res["roofline"]["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
        res["roofline"]["traffic_source"] = (
            "measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over 3 steps of "
            f"`bench.py --spare-blocks {a2.spare_blocks:.3f}` (the same step in a cache of the same size; 2 x FETCH + WRITE, "
            f"FETCH_SIZE {live['FETCH_SIZE_KB']:.0f} KB, WRITE_SIZE {live['WRITE_SIZE_KB']:.0f} KB per launch)")
    elif os.path.exists(tj_path):
        try:
            tj = json.load(open(tj_path))
            t = tj.get("hbm_bytes_per_launch")
            res["roofline"]["traffic"] = t
            res["roofline"]["traffic_frac_of_peak"] = t / (res["roofline"]["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
            res["roofline"]["traffic_source"] = (
                f"profiles/traffic_engine.json ({tj.get('tag', '?')}; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
                "passes over `bench.py --spare-blocks 60`, the same step in a 222 GiB cache; 2 x FETCH + WRITE; a "
                "separately profiled run, not a measurement of this one)")
        except Exception:
            pass
    return res


OTHER_CONFIGS = (("c5", 10, 2), ("c3", 10, 2))


def other_configs_run(args, rank, device):
    """BASELINE configs[4] (fp8, bs 32, 64k) and configs[2] (256 resident sequences in the continual
    steady state -- or the largest power-of-two fraction that fits) with a few timed steps each, so
    that the default line carries driver-run numbers for them: per-stage ms, the compaction
    kernel's frac / frac_of_floor, S1 against its own lower bound."""
    import copy
    import torch
    out = []
    for name, steps, warmup in OTHER_CONFIGS:
        a2 = copy.copy(args)
        for key, default in (("layers", 32), ("block_size", 16), ("seq_len", 32768), ("batch", 1),
                             ("kv_dtype", "fp16"), ("steady_cap", 0), ("keep", 0.5)):
            setattr(a2, key, CONFIGS[name].get(key, default))
        a2.config = name
        batch = a2.batch
        res = None
        while res is None and batch >= 1:
            a2.batch = batch
            # K/V + metric store + move workspace + schedule workspace, roughly, against the free memory
            e = 1 if a2.kv_dtype == "fp8" else 2
            per_seq = a2.layers * a2.kv_heads * ((a2.steady_cap or a2.seq_len) // a2.block_size + 1)
            need = per_seq * batch * (2 * a2.head_size * a2.block_size * e + a2.block_size * 36 + 64)
            free, _ = torch.cuda.mem_get_info()
            if need < 0.9 * free:
                res = measure_workload(a2, rank + 2000, device, steps, warmup, False)
            if res is None:
                batch //= 2
        if res is None:
            out.append({"config": name, "skipped": "does not fit"})
            continue
        out.append({"config": name,
                    "workload": f"L{a2.layers} H{a2.kv_heads} hd{a2.head_size}, {a2.seq_len}-token cache, bs {a2.block_size}, "
                                f"batch {batch}" + (f" (asked for {CONFIGS[name].get('batch', 1)})" if batch != CONFIGS[name].get("batch", 1) else "")
                                + f", {a2.kv_dtype} K/V, "
                                + (f"continual steady state cap={a2.steady_cap}+1 token" if a2.steady_cap else f"compress_once keep={a2.keep}")
                                + f", mode={a2.mode}", **res})
    return out


# --------------------------------------------------------------------------- CPU baseline
def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args):
    """The reference's scheduler path on the host cores of this box: S1 in
    the reference's own formulation -- six global torch.sort calls + the per-sequence host loop,
    oracle/kvc_oracle_torch.py -- with torch.set_num_threads(all cores); S2 / S3 = the C
    restatement of the serial kernels, head loop under OpenMP on all cores.  ONE sequence of the
    bench's shape (the default workload itself), 1 warm-up + 5 timed passes per stage, medians."""
    import torch
    from oracle import kvc_oracle as orc
    from oracle import kvc_oracle_c as orc_c
    from oracle import kvc_oracle_torch as orc_t
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
    kw = dict(metrics=st.metrics, token_positions=st.token_positions,
              seq_index_by_block=st.seq_index_by_block, layer_index_by_block=st.layer_index_by_block,
              head_index_by_block=st.head_index_by_block,
              logical_block_num_by_block=st.logical_block_num_by_block, block_size=bs, num_layers=L,
              num_kv_heads=H, seq_indices=st.seq_indices, seq_positions=st.seq_positions,
              evicted_blocks_per_seq=evicted, context_lens=st.context_lens,
              hanging_token_count=st.hanging_token_count, evicted_kv_offsets=st.evicted_kv_offsets,
              num_protected=st.protected, mode="reference")
    host_cpus = os.cpu_count() or 1
    prev_threads = torch.get_num_threads()
    # S1's thread count: all host cores as BASELINE.md asks, unless a moderate count is faster
    # (torch's 1-D sort does not scale; with 256 threads every small op pays the fan-out) -- one
    # probe pass each (they double as the warm-up), the timed passes use the faster setting
    scan = {}
    for nt in sorted({host_cpus, min(host_cpus, 16)}, reverse=True):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        orc_t.schedule_evictions(**kw)
        scan[nt] = time.perf_counter() - t0
    cores = min(scan, key=scan.get)
    torch.set_num_threads(cores)
    orc_c.set_threads(min(host_cpus, orc_c.max_threads()))
    bt = np.ascontiguousarray(st.block_tables)
    cl = np.ascontiguousarray(st.context_lens)

    def med(fn, iters=5, warm=True):
        if warm:
            fn()
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), r

    s1, (eli, ekc, ebc) = med(lambda: orc_t.schedule_evictions(**kw), warm=False)
    cmi = np.zeros((st.total_slots, 2), np.int32)
    cmc = np.zeros(ekc.shape, np.int32)
    s2, _ = med(lambda: orc_c.schedule_cache_moves(cmi, cmc, eli, ekc, st.evicted_kv_offsets, bt, cl, bs))
    s3, _ = med(lambda: orc_c.execute_cache_moves(k, v, m, p, cmi, cmc, st.evicted_kv_offsets))
    units = int(ekc.sum()) + int(cmc.sum())
    torch_threads = torch.get_num_threads()
    # the round-1 figure for continuity: the same pass on ONE core (NumPy lexsort port + serial C)
    orc_c.set_threads(1)
    torch.set_num_threads(prev_threads)
    t0 = time.perf_counter()
    eli1, ekc1, _ = orc.schedule_evictions(**kw)
    t1 = time.perf_counter()
    orc_c.schedule_cache_moves(cmi, cmc, eli1, ekc1, st.evicted_kv_offsets, bt, cl, bs)
    orc_c.execute_cache_moves(k, v, m, p, cmi, cmc, st.evicted_kv_offsets)
    t2 = time.perf_counter()
    same = T == args.seq_len and args.batch == 1 and not args.steady_cap
    total = s1 + s2 + s3
    return {
        "value": units / total, "unit": "KV slots/s",
        # threads actually used, per stage (S1 = torch's intra-op threads, S2 / S3 = OpenMP); `cores` = the most
        # any stage used
        "cores": max(cores, min(host_cpus, orc_c.max_threads())),
        "threads_by_stage": {"S1_schedule": cores, "S2_moves": min(host_cpus, orc_c.max_threads()),
                             "S3_compact": min(host_cpus, orc_c.max_threads())},
        "kind": "port",
        "host_cpus": host_cpus, "torch_threads": torch_threads,
        "openmp_threads": min(host_cpus, orc_c.max_threads()),
        "S1_seconds_by_torch_threads": {str(k): v for k, v in scan.items()},
        "cpu_model": _cpu_model(),
        "sample": ("the bench workload itself" if same else "one sequence of the bench's shape")
                  + f": L{L} H{H} hd{hd}, {T}-token cache, bs{bs}, B=1, keep={args.keep}, "
                  f"{args.metric_shape} metrics; S1 = the reference's six-sort torch formulation "
                  "(oracle/kvc_oracle_torch.py), S2/S3 = the C restatement of the serial kernels with the "
                  f"head loop on all cores; 1 warm-up + 5 passes per stage, medians; S1 ran on {cores} torch threads "
                  f"(the faster of all {host_cpus} and 16), S2 / S3 on {min(host_cpus, orc_c.max_threads())} OpenMP "
                  f"threads (`threads_by_stage`); {units} slots per pass",
        "stage_seconds": {"S1_schedule": s1, "S2_moves": s2, "S3_compact": s3},
        "single_core_port": {"value": units / (t2 - t0), "cores": 1,
                             "stage_seconds": {"S1_schedule_numpy_lexsort": t1 - t0, "S2_S3_serial_C": t2 - t1},
                             "note": "one pass, no warm-up (the round-1 baseline)"},
    }


def native_layout_run(args, timeout=900):
    """The same workload with slot-major cache blocks (KVC_BLOCK_LAYOUT=slot_major: K [bs][hd], V [bs][hd] inside every
    block; tensor shapes, op signatures, slot numbers unchanged), measured by this same script in a child process
    (`--block-layout slot_major --headline-only`): the compaction kernel's roofline object, the step's stage times and the
    parity verdict (the compacted cache, permuted back, against the oracle).  None if the child fails."""
    import copy
    a2 = copy.copy(args)
    a2.block_layout = "slot_major"
    cmd = ([sys.executable, os.path.abspath(__file__), "--steps", str(min(args.steps, 20)), "--warmup", "3",
            "--headline-only", "--no-cpu-baseline", "--no-probe", "--detail-json", "", "--spare-blocks", str(args.spare_blocks)]
           + workload_flags(a2) + (["--no-live-traffic"] if args.no_live_traffic else []))
    try:
        r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=timeout)
    except (subprocess.TimeoutExpired, OSError) as e:
        return {"failed": str(e)[:200]}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"failed": (r.stderr or r.stdout)[-400:]}
    d = json.loads(lines[-1])
    out = dict(d["roofline"])
    out.update({"value": d["value"], "ms_per_step": d["ms_per_step"], "stages_ms": d["stages_ms"],
                "parity_checked": d["parity_checked"], "steps": d["steps"],
                "what": "python bench.py --block-layout slot_major: the package's opt-in in-block layout "
                        "(KVC_BLOCK_LAYOUT=slot_major), same workload, same call form, same oracle"})
    return out


# --------------------------------------------------------------------------- output
LINE_MAX_BYTES = 4096           # the driver parses the LAST stdout line; round 5's 22 KB line was not parsed


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def _round(x, sig=6):
    """floats at 6 significant digits, recursively (the line is for parsers and readers, the detail file keeps all)"""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _round(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round(v, sig) for v in x]
    return x


def compact_line(res, detail_path):
    """The ONE line the driver parses: the contract's keys, `roofline`, `cpu_baseline`, the parity verdict and the
    stage times -- everything else (other configs, the engine-sized cache, call forms, S0 stages, the adjacent
    attention, prose) lives in the detail file this line names."""
    line = _pick(res, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                       "scaling", "vs_baseline", "dtype", "data"))
    line["config"] = _pick(res["config"], ("workload", "candidate_slots", "evicted_slots", "moved_slots", "freed_blocks"))
    line["call_form"] = res.get("call_form", "").split(":")[0]
    line["stages_ms"] = res["stages_ms"]
    if "stages_ms_hinted" in res:
        line["stages_ms_hinted"] = res["stages_ms_hinted"]
    line["S1_schedule"] = res.get("S1_schedule")
    r = res["roofline"]
    line["roofline"] = _pick(r, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                 "algorithmic_bytes_per_launch", "avg_launch_ms", "layout_amplification",
                                 "alg_frac_ceiling", "frac_of_ceiling", "traffic_frac_of_peak"))
    if line["roofline"].get("traffic_source"):
        line["roofline"]["traffic_source"] = line["roofline"]["traffic_source"].split(":")[0].split(" (")[0][:120]
    if res.get("roofline_native_layout"):
        line["roofline_native_layout"] = res["roofline_native_layout"]
    c = res.get("cpu_baseline")
    if c:
        line["cpu_baseline"] = _pick(c, ("value", "unit", "cores", "kind", "cpu_model", "host_cpus", "threads_by_stage",
                                         "stage_seconds"))
        line["cpu_baseline"]["sample"] = c["sample"].split(";")[0][:200]
    pc = res.get("parity_checked")
    line["parity_checked"] = None if pc is None else _pick(pc, ("bit_exact", "mismatched", "skipped", "kv_bytes"))
    if isinstance(line["parity_checked"], dict) and "skipped" in line["parity_checked"]:
        line["parity_checked"]["skipped"] = line["parity_checked"]["skipped"][:100]
    oc = res.get("other_configs")
    if oc:      # one number per configuration; the legs themselves are in the detail file
        line["other_configs"] = {
            v["config"]: (_pick(v, ("value", "ms_per_step", "skipped"))
                          | {"frac": (v.get("roofline") or {}).get("frac"),
                             "bit_exact": (v.get("parity_checked") or {}).get("bit_exact")})
            for v in oc}
    if "per_rank" in res:
        line["per_rank"] = res["per_rank"]
    line["detail"] = os.path.relpath(detail_path, REPO) if detail_path else None
    line = _round(line)
    txt = json.dumps(line, separators=(",", ":"))
    for drop in ("other_configs", "stages_ms_hinted", "per_rank"):      # (never the contract's keys)
        if len(txt) < LINE_MAX_BYTES:
            break
        if drop == "per_rank" and "per_rank" in line and len(line["per_rank"]) <= 8:
            continue
        line.pop(drop, None)
        txt = json.dumps(line, separators=(",", ":"))
    assert len(txt) < LINE_MAX_BYTES, f"bench line is {len(txt)} bytes"
    return txt


def emit(res, detail_path):
    """detail file first (everything), then the compact line as the last thing on stdout"""
    if detail_path:
        try:
            with open(detail_path, "w") as f:
                json.dump(res, f, indent=1)
                f.write("\n")
        except OSError as e:
            print(f"bench.py: could not write {detail_path}: {e}", file=sys.stderr)
            detail_path = None
    sys.stdout.flush()
    print(compact_line(res, detail_path), flush=True)


# --------------------------------------------------------------------------- main
def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch
    import torch.distributed as dist
    from vllm_kvcompress_amd import _custom_ops as ops
    import vllm_kvcompress_amd
    vllm_kvcompress_amd.load()      # fail loudly if the HIP extension is missing

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # one rank per GPU; KVC_BENCH_BACKEND=gloo (test hook) lets several ranks share the one
    # GPU of a single-GPU box to exercise the N > 1 code path
    backend = os.environ.get("KVC_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"bench.py --gpus {world}: this node has {torch.cuda.device_count()} GPU(s) "
                         "(one rank per GPU over RCCL; KVC_BENCH_BACKEND=gloo shares a GPU for tests)")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend=backend)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # N > 1: the driver's scaling run is the timed loop + the all_gather, nothing else
    headline_only = bool(args.headline_only) if args.headline_only is not None else (world > 1 or args.block_layout != "reference")
    extras = world == 1 and not headline_only

    # sequences of this rank: weak = --batch each; strong = --batch split by harness.dist
    from vllm_kvcompress_amd.harness import dist as hdist
    if args.scaling == "strong":
        want_batch = len(hdist.shard_sequences([1.0] * args.batch, world)[rank])
        if want_batch == 0:
            raise SystemExit(f"--scaling strong: {args.batch} sequence(s) cannot feed {world} ranks")
    else:
        want_batch = args.batch
    batch, st, ds, evicted, k_cache, v_cache = build_workload_that_fits(args, rank, device, want_batch)
    bs = args.block_size
    N = st.total_slots
    work_metrics = ds.cm.metrics.clone()
    work_pos = ds.cm.token_positions.clone()
    # the cache as it is before the first step (the reference's layout), for the parity gate behind the timed region
    snap = parity_snapshot(k_cache, v_cache) if (rank == 0 and N <= PARITY_ORACLE_MAX_SLOTS and not args.no_parity_gate) else None
    slot_major = args.block_layout == "slot_major"
    if slot_major:
        # the package's opt-in layout: the same cache contents, every block re-laid-out K [bs][hd] / V [bs][hd]; the
        # three ops that interpret a block follow the package's switch
        from vllm_kvcompress_amd import layout as kvc_layout
        vllm_kvcompress_amd.set_block_layout("slot_major")
        kvc_layout.convert_block_layout(k_cache, v_cache, "reference", "slot_major")
    # the move workspace, held the way CompressionScheduler holds its own (reference scheduler.py:74-86: ONE persistent
    # [max_kv_per_compression, 2] int table, torch.empty).  --call-form hinted registers it (track_move_table: the wrapper's
    # per-call fill_(0) then clears only what the previous call wrote); the fork's form leaves it a plain tensor
    fork = args.call_form == "fork"
    cmi = torch.empty((N, 2), dtype=torch.int32, device=device)
    if not fork:
        cmi = ops.track_move_table(cmi)
    seq_idx = list(st.seq_indices)
    prot = list(st.protected)
    prot_t = tuple(prot)
    seq_lens = [int(x) + 1 for x in st.seq_positions]
    BLH = (st.num_seqs, st.num_layers, st.num_kv_heads)

    ev = lambda: torch.cuda.Event(enable_timing=True)
    # marks: S1 start, S2 start, S3 start (plan half), data kernel start, end.  All events are
    # recorded on the stream the kernels are launched on (torch's current stream).
    marks = [[ev() for _ in range(5)] for _ in range(args.steps)]
    out = {}

    host_t = []

    def step(i=None, fork=fork, cmi=cmi, marks=marks):
        out.clear()                    # (the previous step's results go out of scope, as in the engine's loop)
        if fork:
            # what CompressionScheduler._schedule_compression builds anew every call (scheduler.py:245-260): part of
            # the step's wall time, outside the S1 marks (they bracket the method the fork calls)
            k_t = torch.tensor(evicted, dtype=torch.int, device=device)
            pos_t = torch.tensor(seq_lens, dtype=torch.int, device=device) - 1
        if i is not None: marks[i][0].record()
        host_t.append(time.perf_counter())
        if fork:
            eli, ekc, ebc = ds.cm.schedule_evictions(seq_idx, pos_t, k_t, ds.context_lens, ds.hanging_token_count,
                                                     ds.evicted_kv_offsets, prot_t)          # (scheduler.py:491-499)
        else:
            eli, ekc, ebc = ds.cm.schedule_evictions(seq_idx, ds.seq_positions, evicted,
                                                     ds.context_lens, ds.hanging_token_count,
                                                     ds.evicted_kv_offsets, prot, total_slots=N,
                                                     block_tables=ds.block_tables if args.pass_block_tables else None)
        host_t.append(time.perf_counter())
        if i is not None: marks[i][1].record()
        cmc = torch.empty(BLH, dtype=torch.int32, device=device)                              # (scheduler.py:508-512)
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
        out["ekc"], out["ebc"], out["eli"], out["cmc"], out["cmi"] = ekc, ebc, eli, cmc, cmi

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # (the interpreter's cyclic collector stays out of the timed region: with torch imported a full
    # collection is a 35-40 ms host pause, which at 0.2 ms per step lands inside one of the steps and
    # reads as that step's first stage -- it is the host's, not the path's)
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    if os.environ.get("KVC_BENCH_DUMP_STEPS"):
        import faulthandler
        for i in range(args.steps):
            faulthandler.dump_traceback_later(0.005, exit=False)      # where the host is when a step takes > 5 ms
            step(i)
            faulthandler.cancel_dump_traceback_later()
    else:
        for i in range(args.steps):
            step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    gc.enable()
    elapsed = t1 - t0
    kernel_ms = sum(m[3].elapsed_time(m[4]) for m in marks) / args.steps

    evicted_slots = int(out["ekc"].sum().item())
    cmc, ekc_ref = out["cmc"], out["ekc"]
    moved_slots = int(cmc.sum().item())
    freed_blocks = int(out["ebc"].sum().item())
    if args.mode == "per_sequence" or batch == 1:   # the reference's batch>1 quirk frees fewer
        assert freed_blocks == sum(evicted), (freed_blocks, sum(evicted))
    if os.environ.get("KVC_BENCH_DUMP_STEPS"):       # per-step stage times (diagnostics: a one-off stall shows here)
        for i, m in enumerate(marks):
            h = host_t[2 * (args.warmup + i):2 * (args.warmup + i) + 2]
            print(f"step {i}: S1 {m[0].elapsed_time(m[1]):.3f} S2 {m[1].elapsed_time(m[2]):.3f} "
                  f"S3 {m[2].elapsed_time(m[4]):.3f} ms; host in schedule_evictions {1e3 * (h[1] - h[0]):.3f} ms, "
                  f"at {1e3 * (h[0] - host_t[2 * args.warmup]):.2f}", file=sys.stderr)
    s1 = sum(m[0].elapsed_time(m[1]) for m in marks) / args.steps
    s2 = sum(m[1].elapsed_time(m[2]) for m in marks) / args.steps
    s3 = sum(m[2].elapsed_time(m[4]) for m in marks) / args.steps

    # whole-job rate = units of all ranks / slowest rank's time; the only collective of the
    # path (RCCL all_gather of two scalars per rank)
    units_local = float((evicted_slots + moved_slots) * args.steps)
    red = hdist.reduce_throughput(units_local, elapsed, device=device)
    units, elapsed = red["units"], red["seconds"]
    per_rank = None
    if world > 1:
        per_rank = [{"units": u, "seconds": sec} for u, sec in
                    zip(red["per_rank_units"], red["per_rank_seconds"])]

    parity = None
    if rank == 0 and not args.no_parity_gate:
        # parity gates before any number is reported
        if slot_major:       # (the oracle computes in the reference's layout: the compacted cache goes back through the permutation)
            kvc_layout.convert_block_layout(k_cache, v_cache, "slot_major", "reference")
        parity = parity_gate(args, st, ds, evicted, args.mode, dict(eli=out["eli"], ekc=out["ekc"], ebc=out["ebc"], cmi=cmi, cmc=cmc),
                             snap, k_cache, v_cache, work_metrics, work_pos, lean=bool(args.lean))
        del snap
        if parity["bit_exact"] is False:
            print(f"bench.py: PARITY GATE FAILED, no number reported: {json.dumps(parity)}", file=sys.stderr, flush=True)
            if world > 1:
                dist.destroy_process_group()
            sys.exit(3)
        if slot_major:
            parity["layout"] = ("the compacted slot-major cache, permuted back to the reference's layout "
                                "(vllm_kvcompress_amd.layout.convert_block_layout), against the oracle's")
            kvc_layout.convert_block_layout(k_cache, v_cache, "reference", "slot_major")
    if rank == 0:
        e = 1 if args.kv_dtype == "fp8" else 2
        block_bytes = args.head_size * bs * e
        bpm = alg_bytes_per_move(args.head_size, e)
        alg_bytes = moved_slots * bpm + 8 * st.total_heads
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        floor = traffic_floor(cmi, cmc, ds.evicted_kv_offsets, bs, block_bytes)
        if slot_major:       # a move is two contiguous copies: the layout forces nothing beyond the algorithmic bytes
            floor = {"bytes": alg_bytes, "dst_blocks": floor["dst_blocks"], "dst_blocks_fully_overwritten": floor["dst_blocks"],
                     "src_blocks": floor["src_blocks"], "note": "slot-major blocks: floor == algorithmic bytes"}
        floor_gbps = floor["bytes"] / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        default_workload = (args.layers, args.kv_heads, args.head_size, bs, args.seq_len, batch,
                            args.keep, args.protected, args.metric_shape, args.kv_dtype,
                            args.steady_cap, args.contiguous_blocks, args.spare_blocks) == (
                                32, 8, 128, 16, 32768, 1, 0.5, 32, "perm", "fp16", 0, False, 0.02)
        # the committed PMC figure belongs to the default workload only; it is a separately
        # profiled run of the same kernel and workload, not a measurement of this run
        live = (live_pmc_traffic(workload_flags(args) + ["--spare-blocks", str(args.spare_blocks)],
                                 kernel="compact_slots_kernel" if slot_major else "compact_runs_kernel")
                if (world == 1 and not args.no_live_traffic) else None)
        if live is not None:
            traffic = live["hbm_bytes_per_launch"]
            traffic_source = ("measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over 3 "
                              "steps of this same workload (2 x FETCH + WRITE; FETCH_SIZE "
                              f"{live['FETCH_SIZE_KB']:.0f} KB, WRITE_SIZE {live['WRITE_SIZE_KB']:.0f} KB per launch)")
        elif default_workload and os.path.exists(args.traffic_json):
            try:
                tj = json.load(open(args.traffic_json))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = (f"{os.path.relpath(args.traffic_json, REPO)} "
                                  f"({tj.get('tag', '?')}; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
                                  "separate passes over this same command, 2 x FETCH + WRITE)")
            except Exception:
                traffic = None
        ceiling = None if (world > 1 or args.no_probe) else pattern_ceiling(k_cache, v_cache, block_bytes)
        model = "Llama-3-8B" if args.layers == 32 else "Llama-3-70B" if args.layers == 80 else "custom"
        res = {
            "metric": "KV slots evicted+compacted/sec and HBM GB/s, Llama-3-8B 32k cache blk16",
            "value": units / elapsed,
            "unit": "KV slots/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u8",
            "dtype_detail": f"compaction copies bytes ({args.kv_dtype} K/V, {e}-byte elements); scheduling "
                            "compares float32 metrics and computes int32 indices",
            "data": "synthetic (seeded paged cache, tie-free permutation metrics, random K/V bits)",
            "config": {
                "workload": f"{args.config}: {model} shape L{args.layers} H{args.kv_heads} hd{args.head_size}, "
                            f"{args.seq_len}-token cache, block_size {bs}, batch {batch}/GPU"
                            + (f" (asked for {want_batch}: the largest that fits in HBM)" if batch != want_batch else "")
                            + (f" ({args.batch} sequences split over {world} GPUs)" if args.scaling == "strong" else "")
                            + f", {args.kv_dtype} K/V, "
                            + (f"continual steady state cap={args.steady_cap}+1 token, " if args.steady_cap
                               else f"compress_once keep={args.keep}, ")
                            + f"protected_window={args.protected}, "
                            f"metrics={args.metric_shape}, schedule mode={args.mode}"
                            + (", slot-major cache blocks (KVC_BLOCK_LAYOUT=slot_major)" if slot_major else "")
                            + (", lean outputs (extension)" if args.lean else "")
                            + (", block_tables passed to schedule_evictions (read only when the batch is sparse in its cache)" if args.pass_block_tables else "")
                            + ", physical blocks "
                            f"{'in allocation order' if args.contiguous_blocks else 'shuffled'}"
                            + (f" inside a cache of {st.num_blocks} blocks" if args.spare_blocks != 0.02 else ""),
                "candidate_slots": N, "evicted_slots": evicted_slots, "moved_slots": moved_slots,
                "freed_blocks": freed_blocks,
            },
            "parity_checked": parity,
            "stages_ms": {"S1_schedule_evictions": s1, "S2_schedule_moves": s2, "S3_execute_moves": s3},
            "S1_schedule": ds.cm.last_schedule_path(),
            "S1_schedule_reason": ds.cm.last_schedule_reason,
            "stage_rates": {
                "S1_candidate_slots_per_s": N / (s1 * 1e-3),
                "S2_moves_per_s": moved_slots / (s2 * 1e-3),
                "S3_moved_slots_per_s": moved_slots / (s3 * 1e-3),
            },
            "roofline": {
                "kernel": (f"kvc::compact_slots_kernel<{args.head_size * e // 16}> (execute_cache_moves, slot-major blocks)"
                           if slot_major else f"kvc::compact_runs_kernel<{args.head_size},{bs},{e},4> (execute_cache_moves)"),
                "block_layout": args.block_layout,
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_move": bpm,
                "avg_launch_ms": kernel_ms,
                "timing": "HIP events on the launch stream around the compaction kernel "
                          "(kvc_execute_cache_moves_apply), every timed step",
                "traffic_frac_of_peak": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                # what the K [NB,hd/x,bs,x] / V [NB,hd,bs] layout forces through HBM for this move list
                "traffic_floor_bytes": floor["bytes"], "traffic_floor": floor,
                "floor_GBps": floor_gbps, "frac_of_floor": floor_gbps / HBM_PEAK_GBPS,
                "pattern_ceiling_GBps": ceiling,
                "floor_frac_of_pattern_ceiling": (
                    floor_gbps / (ceiling["rmw_2R1W"] if floor["dst_blocks_fully_overwritten"] * 2 < floor["dst_blocks"]
                                  else ceiling["copy_1R1W"]) if ceiling else None),
                **frac_ceiling(alg_bytes, floor, ceiling, achieved / HBM_PEAK_GBPS),
            },
        }
        if per_rank:
            res["per_rank"] = per_rank
        res["call_form"] = (
            "fork: schedule_evictions(slot_indices, FRESH device int tensor of last token positions, FRESH device int tensor "
            "of evicted_blocks_per_seq, context_lens, hanging_token_count, evicted_kv_offsets, tuple of protected windows) -- "
            "no total_slots=, no block_tables= -- then schedule_cache_moves on a plain persistent [N, 2] workspace with a fresh "
            "cache_moves_count, then execute_cache_moves (reference scheduler.py:74-86, 245-260, 491-529)" if fork else
            "hinted: host list of counts + total_slots=, move workspace registered with track_move_table")
        if extras and fork and not args.lean:
            # the same step through the hinted form (host list + total_slots=: the method never waits; tracked table)
            hsteps = min(args.steps, 20)
            hmarks = [[ev() for _ in range(5)] for _ in range(hsteps)]
            hcmi = ops.track_move_table(torch.empty((N, 2), dtype=torch.int32, device=device))
            for i in range(-3, hsteps):
                step(i if i >= 0 else None, fork=False, cmi=hcmi, marks=hmarks)
            torch.cuda.synchronize()
            hs = [sum(m[a].elapsed_time(m[b]) for m in hmarks) / hsteps for a, b in ((0, 1), (1, 2), (2, 4))]
            res["stages_ms_hinted"] = {"S1_schedule_evictions": hs[0], "S2_schedule_moves": hs[1], "S3_execute_moves": hs[2],
                                       "same_counts": bool(torch.equal(out["ekc"], ekc_ref))}
            del hcmi
        if extras and not slot_major and not args.no_native_layout and not args.steady_cap:
            res["roofline_native_layout"] = native_layout_run(args)
        if extras and not args.lean:
            forms = s1_call_forms(ds, st, evicted, cmi, cmc, k_cache, v_cache, work_metrics, work_pos, bs,
                                  steps=min(args.steps, 10), ref_counts=ekc_ref)
            res["S1_reference_call_form_ms"] = forms["reference_call_form"]["ms"]
            res["S1_call_forms"] = forms
        if extras and not args.no_engine_cache and not args.steady_cap and args.spare_blocks == 0.02:
            res["engine_sized_cache"] = engine_sized_cache_run(args, rank, device)
        if extras and not args.no_s0:
            del cmi, work_metrics, work_pos
            out.clear()
            res["stages_ms_S0"] = s0_stages(args, device)
        if extras and default_workload and not args.no_other_configs:
            # (the main workload's K/V are not needed any more: the big configurations want the HBM)
            del k_cache, v_cache
            ds.cm.metrics = ds.cm.token_positions = None
            torch.cuda.empty_cache()
            res["other_configs"] = other_configs_run(args, rank, device)
        if extras and not args.no_adjacent:
            # the producer of the metrics (row F3), one layer step at the continual-compression
            # shape; not part of `value`
            from vllm_kvcompress_amd.harness.attention_bench import run as attn_run
            res["adjacent_decode_attention"] = [
                attn_run(64, 4097, Hq=4 * args.kv_heads, Hkv=args.kv_heads, hd=args.head_size,
                         bs=args.block_size if args.block_size in (16, 32) else 16, iters=10, record=r)
                for r in (True, False)]
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args)
        emit(res, args.detail_json)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
