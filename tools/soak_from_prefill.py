#!/usr/bin/env python3
"""Soak of the all-device engine loop FROM PREFILL (vllm_kvcompress_amd/harness/engine_device.py): sequences arrive
(kvc_add_sequence + reshape_and_cache + aggregate_prefill), decode and are compressed back to a cap every iteration
(schedule_evictions in the fork's call form -> schedule_cache_moves -> execute_cache_moves -> free_compressed_blocks ->
append_slots -> reshape_and_cache -> aggregate_decode), leave and are replaced -- no NumPy state on the device side.
The oracle's engine (oracle/engine_oracle.py, the NumPy / C restatements) runs in lockstep; every `--check-every`
iterations ALL state is compared bit for bit, and the schedule outputs of every compression are.

    python tools/soak_from_prefill.py --steps 400 [--layout slot_major] [--mode reference]
prints one JSON line (committed under profiles/)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--check-every", type=int, default=10)
    ap.add_argument("--layout", default="reference", choices=["reference", "slot_major"])
    ap.add_argument("--mode", default="per_sequence", choices=["per_sequence", "reference"])
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import torch
    from oracle.engine_oracle import OracleEngine
    from vllm_kvcompress_amd import _lib
    from vllm_kvcompress_amd.harness.engine_device import DeviceEngine
    from vllm_kvcompress_amd.layout import convert_block_layout
    DEV = "cuda:0"
    L, H, hd, bs, qpk, cap, prot = 4, 8, 128, 16, 4, 256, 32
    S, M = 8, 64
    NB = 6 * L * H * (cap // bs + 24) + 64
    rng = np.random.default_rng(args.seed)
    _lib.set_block_layout(args.layout)
    dev = DeviceEngine(num_layers=L, num_kv_heads=H, head_size=hd, block_size=bs, num_blocks=NB, max_num_seqs=S,
                       max_blocks_per_head=M, num_queries_per_kv=qpk, mode=args.mode, protected_window=prot,
                       max_cache_tokens=cap)
    o = OracleEngine(L, H, hd, bs, NB, S, M, qpk, prot, cap, args.mode, fast=True)
    stats = dict(steps=0, arrivals=0, departures=0, compressions=0, state_checks=0, evicted_blocks=0, moved_slots=0,
                 harvested_calls=0, redone_calls=0, schedules={})

    def check(where):
        if args.layout == "slot_major":
            convert_block_layout(dev.k_cache, dev.v_cache, "slot_major", "reference")
        cm = dev.cm
        live = np.arange(M)[None, None, None, :] < ((o.ctx + bs - 1) // bs)[..., None]
        alloc = o.seq >= 0
        pairs = [("context_lens", dev.context_lens.cpu().numpy(), o.ctx), ("free_mask", dev.free_mask.cpu().numpy(), o.free),
                 ("metrics", cm.metrics.cpu().numpy(), o.metrics), ("seq_index", cm.seq_index_by_block.cpu().numpy(), o.seq),
                 ("K", dev.k_cache.view(torch.int16).cpu().numpy(), o.k.view(np.int16)),
                 ("V", dev.v_cache.view(torch.int16).cpu().numpy(), o.v.view(np.int16)),
                 ("block_tables", dev.block_tables.cpu().numpy()[live], o.bt[live]),
                 ("positions", cm.token_positions.cpu().numpy()[alloc], o.pos[alloc]),
                 ("lbn", cm.logical_block_num_by_block.cpu().numpy()[alloc], o.lbn[alloc])]
        for name, got, want in pairs:
            if not np.array_equal(got, want):
                raise SystemExit(f"soak_from_prefill: {where}: {name} differs from the oracle's")
        if args.layout == "slot_major":
            convert_block_layout(dev.k_cache, dev.v_cache, "reference", "slot_major")
        stats["state_checks"] += 1

    def arrive(slot, T):
        key = rng.standard_normal((L, T, H, hd)).astype(np.float16)
        val = rng.standard_normal((L, T, H, hd)).astype(np.float16)
        pm = rng.random((L, T, H * qpk)).astype(np.float32)
        sm = o.add_sequence(slot, key, val, pm)
        dev.add_sequence(slot, torch.from_numpy(key).to(DEV), torch.from_numpy(val).to(DEV), torch.from_numpy(pm).to(DEV))
        if not np.array_equal(dev.last["slot_mapping"].cpu().numpy(), sm):
            raise SystemExit(f"soak_from_prefill: prefill slot mapping of slot {slot} differs")
        stats["arrivals"] += 1

    t0 = time.time()
    for s in range(4):
        arrive(s, int(rng.integers(cap // 2, cap + 5 * bs)))
    check("after the first prefills")
    for it in range(args.steps):
        if it % 37 == 36 and len(o.slots) > 2:               # a sequence finishes ...
            s = o.slots[int(rng.integers(len(o.slots)))]
            dev.remove_sequence(s)
            o.remove_sequence(s)
            stats["departures"] += 1
        if it % 29 == 28 and len(o.slots) < 6:               # ... another one arrives in a free slot
            s = min(set(range(S)) - set(o.slots))
            arrive(s, int(rng.integers(bs, cap + 8 * bs)))
        r_o, r_d = o.compress(), dev.compress()
        if (r_o is None) != (r_d is None):
            raise SystemExit(f"soak_from_prefill: iteration {it}: one engine compressed, the other did not")
        if r_o is not None:
            stats["compressions"] += 1
            for k in ("eli", "ekc", "ebc", "cmc", "cmi", "freed"):
                if not np.array_equal(r_d[k].cpu().numpy(), r_o[k]):
                    raise SystemExit(f"soak_from_prefill: iteration {it}: {k} differs from the oracle's")
            stats["evicted_blocks"] += int(r_o["ebc"].sum())
            stats["moved_slots"] += int(r_o["cmc"].sum())
            stats["harvested_calls"] += bool(dev.cm.last_harvest_used)
            path = dev.cm.last_schedule_path()
            stats["schedules"][path] = stats["schedules"].get(path, 0) + 1
        B = len(o.slots)
        key = rng.standard_normal((L, B, H, hd)).astype(np.float16)
        val = rng.standard_normal((L, B, H, hd)).astype(np.float16)
        temp = rng.random((NB, bs, qpk)).astype(np.float32)
        if o.decode(key, val, temp) != dev.decode(torch.from_numpy(key).to(DEV), torch.from_numpy(val).to(DEV),
                                                  torch.from_numpy(temp).to(DEV)):
            raise SystemExit(f"soak_from_prefill: iteration {it}: allocation counts differ")
        stats["steps"] += 1
        if it % args.check_every == args.check_every - 1:
            check(f"iteration {it}")
    check("end")
    stats["redone_calls"] = int(dev.cm.harvest_misses)
    stats.update(layout=args.layout, mode=args.mode, seconds=round(time.time() - t0, 1), result="identical to the oracle's engine",
                 shape=f"L{L} H{H} hd{hd} bs{bs} qpk{qpk}, cap {cap}, protected {prot}, {NB} blocks, <= 6 resident sequences")
    print(json.dumps(stats))


if __name__ == "__main__":
    main()
