#!/usr/bin/env python3
"""Harvest-ahead over many decode steps of an EVOLVING block state, all of it on the device: two engines stepped
with the same attention mass -- one in the reference's order (aggregate_decode at the end of an iteration,
schedule_evictions' own pass at the start of the next), one that leaves the aggregate to the scheduler
(CompressionScheduler.schedule_compression(aggregate_decode=True): the harvesting pass) -- must hold the same
state after every step.  Prints how many steps ran on harvested lists and how many of those fell short -- and,
for the engine in the reference's order, how many collecting passes took their pivots from the call before
(pivot memory) and how many of those listed too little.
    python tools/soak_harvest.py [steps] [seqs] [layers] [cap] [mass]
mass: "uniform" | "peaky" (a few keys per head take most of a step's attention, the rest next to nothing)"""
import copy
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from tests.test_gpu_harvest import DEV, _Engine  # noqa: E402
from vllm_kvcompress_amd.harness import synth  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    cap = int(sys.argv[4]) if len(sys.argv) > 4 else 512
    mass = sys.argv[5] if len(sys.argv) > 5 else "uniform"
    H, bs, qpk = 8, 16, 4
    seq_lens = [cap + 300 + 37 * i for i in range(B)]
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=4, protected=bs + 1,
                          spare_block_frac=0.5, steady_cap=cap)
    a = _Engine(copy.deepcopy(st), seq_lens, cap, qpk, deferred=False)
    b = _Engine(copy.deepcopy(st), seq_lens, cap, qpk, deferred=True)
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    used = paths = remembered = paths_a = 0
    sel = list(range(B))
    for it in range(steps):
        temp = torch.rand((st.num_blocks, bs, qpk), device=DEV, generator=g)
        if mass == "peaky":
            temp = temp ** 24 * 30.0 + 1e-4 * torch.rand((st.num_blocks, bs, qpk), device=DEV, generator=g)
        ra, rb = a.step(temp, sel), b.step(temp, sel)
        for key in ("metrics", "cmc", "cmi", "ctx", "seq", "after", "pos"):
            x, y = ra[key], rb[key]
            if x.dtype == torch.float32:
                x, y = x.view(torch.int32), y.view(torch.int32)
            if not torch.equal(x, y):
                raise SystemExit(f"step {it}: {key} differs")
        used += bool(rb["used"])
        paths += b.cm.last_schedule_path() == "small_eviction"
        remembered += bool(a.cm.last_pivot_memory_used)
        paths_a += a.cm.last_schedule_path() == "small_eviction"
    print(json.dumps({"steps": steps, "sequences": B, "heads_per_sequence": L * H, "cap": cap, "attention_mass": mass,
                      "candidate_slots": st.total_slots, "harvested_steps": used, "harvest_misses": b.cm.harvest_misses,
                      "harvest_widen_at_end": b.cm.harvest_widen, "steps_without_fallback": paths,
                      "reference_order": {"steps_on_remembered_pivots": remembered, "misses": a.cm.harvest_misses,
                                          "steps_without_fallback": paths_a},
                      "identical_state_every_step": True}))


if __name__ == "__main__":
    main()
