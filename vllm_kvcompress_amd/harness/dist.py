"""Multi-GPU layout of the hot path: shard sequences, reduce throughput scalars.

The per-sequence eviction schedule, its move schedule and the compaction touch only the
sequence's own blocks, so a batch is sharded by sequence across
the GPUs of a node with NO data-path collective; ``torch.distributed`` (backend "nccl" =
RCCL on ROCm, "gloo" in the CPU tests) is used only to agree on the wall time and to add
up the processed units.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.distributed as dist


def shard_sequences(seq_costs: Sequence[float], world: int) -> List[List[int]]:
    """Assign sequences to ranks balancing total cost (KV bytes / slots): longest first,
    each to the currently lightest rank; ties broken by rank index.  Deterministic, and a
    pure function of ``seq_costs`` so every rank computes the same assignment."""
    order = sorted(range(len(seq_costs)), key=lambda i: (-float(seq_costs[i]), i))
    load = [0.0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        shards[r].append(i)
        load[r] += float(seq_costs[i])
    return [sorted(s) for s in shards]


def reduce_throughput(units: float, seconds: float, device=None) -> Dict[str, object]:
    """Whole-job throughput: (sum of units over ranks) / (max seconds over ranks).
    Works without an initialised process group (single process)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return {"units": float(units), "seconds": float(seconds), "per_rank_units": [float(units)],
                "per_rank_seconds": [float(seconds)], "value": float(units) / float(seconds)}
    world = dist.get_world_size()
    t = torch.tensor([float(units), float(seconds)], dtype=torch.float64, device=device)
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    per_units = [float(g[0]) for g in gathered]
    per_secs = [float(g[1]) for g in gathered]
    total, worst = sum(per_units), max(per_secs)
    return {"units": total, "seconds": worst, "per_rank_units": per_units,
            "per_rank_seconds": per_secs, "value": total / worst}
