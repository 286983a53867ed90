#!/usr/bin/env python3
"""The zero-sweep decode step over many iterations of an EVOLVING block state, all of it on the device: two engines
stepped with the same queries and K/V (tests/test_gpu_attention_harvest.py::AttnEngine) -- one in the reference's flow
(attention -> temp_metrics, aggregate_decode, schedule_evictions' own pass), one whose fused-metric attention folds the
weights into the store and makes the next schedule call's lists in its epilogue (no aggregation, no collecting pass) --
must hold the same state after every step, and every `check`-th step the second one's move list is compared with the
ORACLE's schedule of the store it ran on.
    python tools/soak_attention_harvest.py [steps] [seqs] [layers] [cap] [check]"""
import copy
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.test_gpu_attention_harvest import AttnEngine, _oracle_schedule_of, _same  # noqa: E402
from vllm_kvcompress_amd.harness import synth  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    cap = int(sys.argv[4]) if len(sys.argv) > 4 else 512
    check = int(sys.argv[5]) if len(sys.argv) > 5 else 10
    H, bs, qpk, hd = 8, 16, 4, 128
    seq_lens = [cap + 300 + 37 * i for i in range(B)]
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=4, protected=bs + 1,
                          spare_block_frac=0.5, steady_cap=cap)
    a = AttnEngine(copy.deepcopy(st), seq_lens, cap, fused=False, qpk=qpk, hd=hd)
    b = AttnEngine(copy.deepcopy(st), seq_lens, cap, fused=True, qpk=qpk, hd=hd)
    sel = list(range(B))
    oracle_checked = 0
    for it in range(steps):
        want = ost = None
        if it % check == 0:
            want, ost = _oracle_schedule_of(b, sel)
        ra, rb = a.compress(sel), b.compress(sel)
        _same({k: v for k, v in ra.items() if k in ("cmc", "cmi")}, rb, f"step {it} (schedule)")
        if want is not None:
            rows = np.concatenate([np.arange(o, o + c) for o, c in zip(ost.evicted_kv_offsets.reshape(-1), want["cmc"].reshape(-1))]
                                  + [np.zeros(0, np.int64)]).astype(np.int64)
            if not (np.array_equal(rb["cmc"].cpu().numpy(), want["cmc"])
                    and np.array_equal(rb["cmi"].cpu().numpy()[rows], want["cmi"][rows])):
                raise SystemExit(f"step {it}: the schedule differs from the oracle's")
            oracle_checked += 1
        a.append(); b.append()
        a.forward(sel); b.forward(sel)
        _same(a.state(), b.state(), f"step {it}")
    print(json.dumps({"steps": steps, "sequences": B, "heads_per_sequence": L * H, "cap": cap, "candidate_slots": st.total_slots,
                      "steps_offered_lists": b.offered, "steps_on_the_epilogues_lists": b.used,
                      "harvest_misses": b.cm.harvest_misses, "steps_without_fallback": sum(p == "small_eviction" for p in b.paths),
                      "oracle_checked_steps": oracle_checked, "identical_state_every_step": True,
                      "reference_flow": {"steps_on_remembered_pivots": sum(1 for _ in a.paths), "misses": a.cm.harvest_misses}}))


if __name__ == "__main__":
    main()
