"""The engine loop of vllm_kvcompress_amd/harness/engine_device.py on the oracle's NumPy restatements -- prefill
allocation, cache writes, aggregation, schedule, moves, compaction, block frees, decode appends -- each pinned to
fixtures generated from the reference's own code (tests/test_oracle_golden.py).  Test infrastructure: driven in
lockstep with the device engine by tests/test_gpu_engine_from_prefill.py and tools/soak_from_prefill.py.
``fast=True`` runs the serial move / compaction kernels through the C restatement (oracle/kvc_oracle.c)."""
import numpy as np

from oracle import kvc_oracle as orc
from vllm_kvcompress_amd.harness import synth


class OracleEngine:
    """the same loop on the oracle's NumPy restatements"""

    def __init__(self, L, H, hd, bs, NB, S, M, qpk, protected, cap, mode, fast=False):
        self.fast = fast
        self.L, self.H, self.hd, self.bs, self.NB, self.qpk = L, H, hd, bs, NB, qpk
        self.protected, self.cap, self.mode = protected, cap, mode
        self.bt = np.zeros((L, S, H, M), np.int32)
        self.ctx = np.zeros((L, S, H), np.int32)
        self.free = np.ones(NB, bool)
        self.k = np.zeros((NB, hd // 8, bs, 8), np.float16)
        self.v = np.zeros((NB, hd, bs), np.float16)
        self.metrics = np.zeros((NB, bs), np.float32)
        self.pos = np.zeros((NB, bs), np.int32)
        self.seq = np.full(NB, -1, np.int32)
        self.lay, self.head, self.lbn = (np.zeros(NB, np.int32) for _ in range(3))
        self.slots, self.seq_len = [], {}

    def add_sequence(self, slot, key, value, prefill_metrics):
        T = key.shape[1]
        sm = orc.add_sequence(self.bt, self.ctx, slot, T, self.free, self.seq, self.lay, self.head, self.lbn, self.pos, self.bs)
        zero = np.zeros(self.H, np.float32)
        for l in range(self.L):
            orc.reshape_and_cache_kvc(key[l], value[l], self.k, self.v, self.metrics, sm[l].reshape(-1), zero)
            orc.aggregate_prefill(self.metrics, prefill_metrics[l], sm[l], self.H)
        self.slots = sorted(self.slots + [slot])
        self.seq_len[slot] = T + 1
        return sm

    def compress(self):
        bs, L, H = self.bs, self.L, self.H
        slots = list(self.slots)
        ctx = np.ascontiguousarray(self.ctx[:, slots])
        evicted = [synth.evict_block_count(context_lens_lh=ctx[:, b, :], seq_len=self.seq_len[s], block_size=bs,
                                           protected_window_size=self.protected, max_cache_tokens=self.cap)
                   for b, s in enumerate(slots)]
        if not any(evicted):
            return None
        hang = synth.hanging_tokens(ctx.transpose(1, 0, 2), bs)
        offs = synth.kv_offsets(ctx, bs)
        N = int(((ctx.astype(np.int64) + bs - 1) // bs).sum()) * bs
        seq_pos = np.asarray([self.seq_len[s] - 1 for s in slots], np.int32)
        eli, ekc, ebc = orc.schedule_evictions(
            metrics=self.metrics, token_positions=self.pos, seq_index_by_block=self.seq, layer_index_by_block=self.lay,
            head_index_by_block=self.head, logical_block_num_by_block=self.lbn, block_size=bs, num_layers=L,
            num_kv_heads=H, seq_indices=slots, seq_positions=seq_pos, evicted_blocks_per_seq=evicted, context_lens=ctx,
            hanging_token_count=hang, evicted_kv_offsets=offs, num_protected=[self.protected] * len(slots), mode=self.mode)
        cmi = np.zeros((N, 2), np.int32)
        cmc = np.zeros(ekc.shape, np.int32)
        if self.fast:
            from oracle import kvc_oracle_c as orc_c
            orc_c.schedule_cache_moves(cmi, cmc, eli, ekc, offs, np.ascontiguousarray(self.bt[:, slots]), ctx, bs)
            orc_c.execute_cache_moves(self.k.view(np.uint16), self.v.view(np.uint16), self.metrics, self.pos, cmi, cmc, offs)
        else:
            orc.schedule_cache_moves(cmi, cmc, eli, ekc, offs, np.ascontiguousarray(self.bt[:, slots]), ctx, bs)
            orc.execute_cache_moves(self.k, self.v, self.metrics, self.pos, cmi, cmc, offs)
        freed = orc.free_compressed_blocks(self.bt, self.ctx, slots, ebc, self.seq, bs, self.free)
        return dict(evicted=evicted, eli=eli, ekc=ekc, ebc=ebc, cmi=cmi, cmc=cmc, freed=freed, N=N)

    def remove_sequence(self, slot):
        M = self.bt.shape[3]
        blocks = self.bt[:, slot][np.arange(M)[None, None, :] < ((self.ctx[:, slot] + self.bs - 1) // self.bs)[..., None]]
        self.free[blocks] = True
        self.seq[blocks] = -1
        self.ctx[:, slot] = 0
        self.slots.remove(slot)
        del self.seq_len[slot]

    def decode(self, key, value, temp):
        slots = list(self.slots)
        last_pos = [self.seq_len[s] - 1 for s in slots]
        n = orc.append_slots(self.bt, self.ctx, slots, last_pos, self.free, self.seq, self.lay, self.head, self.lbn,
                             self.pos, self.bs, write_token_position=True)
        c1 = self.ctx[:, slots] - 1
        sm = np.take_along_axis(self.bt[:, slots], (c1 // self.bs)[..., None], axis=3)[..., 0].astype(np.int64) * self.bs + c1 % self.bs
        zero = np.zeros(self.H, np.float32)
        for l in range(self.L):
            orc.reshape_and_cache_kvc(key[l], value[l], self.k, self.v, self.metrics, sm[l].reshape(-1), zero)
        orc.aggregate_decode(self.metrics, temp, use_l2=True)
        for s in slots:
            self.seq_len[s] += 1
        return n
