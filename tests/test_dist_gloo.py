"""N>1 path on CPU: two gloo ranks shard a batch by sequence, each runs the (oracle) hot
path on its shard only, and the results -- gathered with torch.distributed -- must equal
the whole-batch run in per_sequence mode (sharding must not change any sequence's
schedule), with throughput reduced as sum(units)/max(seconds)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import kvc_oracle as orc
from tests.helpers import oracle_pipeline
from vllm_kvcompress_amd.harness import dist as hdist
from vllm_kvcompress_amd.harness import synth

SEQ_LENS = [70, 33, 121, 50, 18]
L, H, BS = 2, 2, 4


def _seq_state(i):
    return synth.make_state(num_layers=L, num_kv_heads=H, block_size=BS, seq_lens=[SEQ_LENS[i]],
                            seed=100 + i, protected=3)


def _evict(st):
    nblk = ((st.context_lens.astype(np.int64) + BS - 1) // BS).sum()
    return [int(max(nblk - L * H, 0) // 2)]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shards = hdist.shard_sequences([s * L * H for s in SEQ_LENS], world)
    mine = shards[rank]
    res = {}
    units = 0
    for i in mine:
        st = _seq_state(i)
        out = oracle_pipeline(st, _evict(st), mode="per_sequence")
        res[i] = (out["eli"], out["ekc"], out["cmi"], out["cmc"])
        units += int(out["ekc"].sum()) + int(out["cmc"].sum())
    red = hdist.reduce_throughput(units, 1.0 + rank)           # fake, rank-dependent seconds
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        q.put((shards, red, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_whole_batch():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    shards, red, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every sequence on exactly one rank
    assert sorted(sum(shards, [])) == list(range(len(SEQ_LENS)))
    merged = {}
    for part in gathered:
        merged.update(part)
    assert sorted(merged) == list(range(len(SEQ_LENS)))
    total_units = 0
    for i in range(len(SEQ_LENS)):
        st = _seq_state(i)
        want = oracle_pipeline(st, _evict(st), mode="per_sequence")
        for got, key in zip(merged[i], ("eli", "ekc", "cmi", "cmc")):
            np.testing.assert_array_equal(got, want[key])
        total_units += int(want["ekc"].sum()) + int(want["cmc"].sum())
    assert red["units"] == total_units
    assert red["seconds"] == 2.0 and red["per_rank_seconds"] == [1.0, 2.0]
    assert abs(red["value"] - total_units / 2.0) < 1e-9


def test_per_sequence_batch_equals_individual_runs():
    """the semantic that makes sharding legal: a B>1 call in per_sequence mode gives each
    sequence exactly the schedule of a B=1 reference call on it"""
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=BS, seq_lens=[41, 23, 60],
                          seed=7, protected=[2, 5, 3])
    nblk = ((st.context_lens.astype(np.int64) + BS - 1) // BS).sum(0).sum(-1)
    evicted = [int(n) // 2 for n in nblk]
    whole = oracle_pipeline(st, evicted, mode="per_sequence")
    for b in range(3):
        kw = dict(metrics=st.metrics, token_positions=st.token_positions,
                  seq_index_by_block=st.seq_index_by_block,
                  layer_index_by_block=st.layer_index_by_block,
                  head_index_by_block=st.head_index_by_block,
                  logical_block_num_by_block=st.logical_block_num_by_block, block_size=BS,
                  num_layers=L, num_kv_heads=H, seq_indices=[b],
                  seq_positions=st.seq_positions[b:b + 1], evicted_blocks_per_seq=[evicted[b]],
                  context_lens=np.ascontiguousarray(st.context_lens[:, b:b + 1, :]),
                  hanging_token_count=np.ascontiguousarray(st.hanging_token_count[b:b + 1]),
                  evicted_kv_offsets=np.ascontiguousarray(
                      st.evicted_kv_offsets[b:b + 1] - st.evicted_kv_offsets[b, 0, 0]),
                  num_protected=[st.protected[b]])
        eli, ekc, ebc = orc.schedule_evictions(**kw, mode="reference")
        np.testing.assert_array_equal(ekc, whole["ekc"][b:b + 1])
        np.testing.assert_array_equal(ebc, whole["ebc"][b:b + 1])
        lo = int(st.evicted_kv_offsets[b, 0, 0])
        np.testing.assert_array_equal(eli, whole["eli"][lo:lo + eli.shape[0]])


def test_shard_sequences_balances():
    shards = hdist.shard_sequences([10, 1, 1, 1, 7, 3], 2)
    loads = [sum([10, 1, 1, 1, 7, 3][i] for i in s) for s in shards]
    assert sorted(sum(shards, [])) == list(range(6))
    assert abs(loads[0] - loads[1]) <= 1
    assert hdist.shard_sequences([5, 5, 5, 5], 4) == [[0], [1], [2], [3]]


# ---- eight ranks: BASELINE configs[3] is 256 sequences over the 8 GPUs of a node ------------------------------------
SEQ8 = 256            # configs[3]'s batch; ragged toy lengths so that the shards are not trivially equal
LENS8 = [9 + (7 * i) % 23 for i in range(SEQ8)]


def _seq_state8(i):
    return synth.make_state(num_layers=1, num_kv_heads=2, block_size=BS, seq_lens=[LENS8[i]], seed=500 + i, protected=2)


def _worker8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shards = hdist.shard_sequences([float(n) for n in LENS8], world)
    res, units = {}, 0
    for i in shards[rank]:
        st = _seq_state8(i)
        nblk = int(((st.context_lens.astype(np.int64) + BS - 1) // BS).sum())
        out = oracle_pipeline(st, [max(nblk - 2, 0) // 2], mode="per_sequence")
        res[i] = (out["ekc"], out["cmc"])
        units += int(out["ekc"].sum()) + int(out["cmc"].sum())
    red = hdist.reduce_throughput(units, 1.0 + 0.25 * rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        q.put((shards, red, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_shard_256_sequences():
    """configs[3]'s layout -- 256 sequences sharded by sequence over the 8 GPUs of one node, no data-path collective,
    one all_gather of two scalars -- with eight gloo ranks: every sequence on exactly one rank, 32 per rank, balanced
    cost, every sequence's schedule equal to its stand-alone run, throughput = sum(units) / max(seconds)"""
    world = 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    shards, red, gathered = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(sum(shards, [])) == list(range(SEQ8)) and [len(x) for x in shards] == [32] * 8
    loads = [sum(LENS8[i] for i in x) for x in shards]
    assert max(loads) - min(loads) <= max(LENS8)
    merged = {}
    for part in gathered:
        assert not (set(part) & set(merged))
        merged.update(part)
    assert sorted(merged) == list(range(SEQ8))
    total = 0
    for i in range(0, SEQ8, 17):                      # (a sample against stand-alone runs; the totals cover the rest)
        st = _seq_state8(i)
        nblk = int(((st.context_lens.astype(np.int64) + BS - 1) // BS).sum())
        want = oracle_pipeline(st, [max(nblk - 2, 0) // 2], mode="per_sequence")
        np.testing.assert_array_equal(merged[i][0], want["ekc"])
        np.testing.assert_array_equal(merged[i][1], want["cmc"])
    total = sum(int(a.sum()) + int(b.sum()) for a, b in merged.values())
    assert red["units"] == total and len(red["per_rank_seconds"]) == 8
    assert red["seconds"] == 1.0 + 0.25 * 7 and abs(red["value"] - total / red["seconds"]) < 1e-9
