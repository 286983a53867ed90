"""schedule_evictions in the reference's OWN formulation, on torch CPU tensors.

TEST INFRASTRUCTURE ONLY (see kvc_oracle.py): used by ``bench.py``'s ``cpu_baseline`` leg --
"the reference's CPU scheduler path timed on the host cores" (BASELINE.json) -- and by
``tests/test_oracle_golden.py``, which pins it to the golden vectors.  Nothing under
``vllm_kvcompress_amd/`` imports it.

``oracle/kvc_oracle.py`` restates WHAT the reference computes (one lexsort per step, canonical
tie order).  This file restates HOW it computes it, step for step, because that is what a host
core count can be quoted against: the six device-wide ``torch.sort`` calls over all candidate
slots / chunks and the per-sequence host loop of ``vllm/kvcompress/metrics.py:441-847``:

    :465-544  mask the candidate blocks, average / bias, +inf outside the evictable range
    :562-570  sort all slots by metric, then stably by (seq, layer, head)          [2 sorts]
    :583-596  one threshold per block-sized chunk (the hanging-token-th metric of the chunk)
    :669-680  sort the chunk thresholds by value, then stably by sequence          [2 sorts]
    :704-729  host loop over the sequences: the first k chunks of each are evicted, minus the
              +inf ones (counted from position 0: the batch > 1 quirk)
    :755      scatter the keep mask back to (seq, layer, head, metric) order
    :773-792  count_block_evictions (serial kernel -> the C restatement) and the KV counts
    :822-834  sort the logical indices, then stably by head                         [2 sorts]

``torch.sort`` on floats is not stable, so with tied metrics the evicted set may differ from the
canonical one (exactly as the reference's does); the fixtures are tie-free.
"""
from __future__ import annotations

import numpy as np
import torch

from . import kvc_oracle_c as orc_c

MAX_INT = 2147483000      # vllm/kvcompress/metrics.py:12


def schedule_evictions(*, metrics, token_positions, seq_index_by_block, layer_index_by_block,
                       head_index_by_block, logical_block_num_by_block, block_size, num_layers,
                       num_kv_heads, seq_indices, seq_positions, evicted_blocks_per_seq,
                       context_lens, hanging_token_count, evicted_kv_offsets, num_protected,
                       use_average=False, num_sinks=0, mode="reference", uniform_evict=False):
    """NumPy arrays in (the layouts of harness/synth.PagedState), NumPy arrays out:
    (evicted_logical_indices [N] i32, evicted_kv_count [B,L,H] i32, evicted_block_count [B,L,H] i32).
    Bias is not restated here (the bench never uses it; kvc_oracle.py covers it)."""
    bs, L, H = int(block_size), int(num_layers), int(num_kv_heads)
    t = torch.from_numpy
    seq_indices = [int(s) for s in seq_indices]
    B = len(seq_indices)
    sib = t(np.ascontiguousarray(seq_index_by_block)).long()
    top = max(seq_indices)
    slot_of_seq = torch.full((top + 1,), -1, dtype=torch.long)
    slot_of_seq[torch.tensor(seq_indices)] = torch.arange(B)
    seq_pos_all = torch.zeros(top + 1, dtype=torch.int32)
    prot_all = torch.zeros(top + 1, dtype=torch.int32)
    seq_pos_all[torch.tensor(seq_indices)] = t(np.asarray(seq_positions, dtype=np.int32).reshape(-1))
    prot_all[torch.tensor(seq_indices)] = t(np.asarray(num_protected, dtype=np.int32).reshape(-1))

    # ---- candidate blocks in ascending physical order                       :465-493
    in_batch = (sib >= 0) & (sib <= top)
    in_batch &= slot_of_seq[sib.clamp(0, top)] >= 0
    m = t(np.ascontiguousarray(metrics))[in_batch].reshape(-1).clone()
    pos = t(np.ascontiguousarray(token_positions))[in_batch]
    seq_of = sib[in_batch]
    lay = t(np.ascontiguousarray(layer_index_by_block))[in_batch].long()
    head = t(np.ascontiguousarray(head_index_by_block))[in_batch].long()
    lbn = t(np.ascontiguousarray(logical_block_num_by_block))[in_batch]
    lam = lbn[:, None] * bs + torch.arange(bs, dtype=torch.int32)[None]
    if use_average:                                                            # :495-501
        m /= (seq_pos_all[seq_of][:, None] - pos).reshape(-1)
    slh = slot_of_seq[seq_of] * (L * H) + lay * H + head                       # :520-524
    ctx = t(np.ascontiguousarray(context_lens)).transpose(0, 1).reshape(-1)[slh]
    max_in_range = (seq_pos_all - prot_all)[seq_of]
    in_range = ((lbn < (ctx + bs - 1) // bs)[:, None] & (pos <= max_in_range[:, None])
                & (pos >= num_sinks))                                          # :539-543
    m[~in_range.reshape(-1)] = float("inf")

    # ---- 1. (head, metric) order: value sort, then stable head sort          :562-570
    sm, order = m.sort()
    s_slh, by_head = slh.repeat_interleave(bs)[order].sort(stable=True)
    order = order[by_head]
    sm = sm[by_head]
    # ---- 2. chunk thresholds                                                 :583-596
    chunk_slh = s_slh.view(-1, bs)[:, 0]
    hang = t(np.ascontiguousarray(hanging_token_count)).reshape(-1)[chunk_slh].long()
    thr = sm.view(-1, bs).gather(1, (hang - 1)[:, None]).squeeze(1)
    # ---- 3. per-sequence selection                                           :604-755
    s_lam = lam.reshape(-1)[order].view(-1, bs).clone()
    blocks_per_seq = ((t(np.ascontiguousarray(context_lens)).long() + bs - 1) // bs).sum(0).sum(-1)
    if uniform_evict:                                                          # :639-666 (no chunk sorts)
        offset = 0
        for i, k in enumerate(int(x) for x in np.asarray(evicted_blocks_per_seq).reshape(-1)):
            end = offset + int(blocks_per_seq[i])
            cur = s_lam[offset:end].reshape(L * H, -1, bs)                    # (heads of equal length)
            cur[:, max(k, 0) // (L * H):] = MAX_INT
            s_lam[offset:end] = cur.view(-1, bs)
            offset = end
        evicted_blocks_per_seq = []
    v_thr, by_thr = thr.sort()                                                 # :669
    chunk_seq = seq_of[order.view(-1, bs)[:, 0] // bs][by_thr]                 # :670-677
    _, by_seq = chunk_seq.sort(stable=True)                                    # :678
    seq_thr = v_thr[by_seq]
    seq_chunks = by_thr[by_seq]
    seq_lam = s_lam[seq_chunks]
    offset = 0
    for i, k in enumerate(int(x) for x in np.asarray(evicted_blocks_per_seq).reshape(-1)):   # :704-729
        end = offset + int(blocks_per_seq[i])
        un = offset + k
        lo = 0 if mode == "reference" else offset
        un -= int((seq_thr[lo:un] == float("inf")).sum())
        seq_lam[un:end] = MAX_INT
        offset = end
    if not uniform_evict:
        s_lam[seq_chunks] = seq_lam                                            # :755
    flat = s_lam.reshape(-1).numpy()
    # ---- count_block_evictions + KV counts                                   :773-792
    ebc = np.empty_like(evicted_kv_offsets, dtype=np.int32)
    orc_c.count_block_evictions(ebc, flat, np.ascontiguousarray(evicted_kv_offsets),
                                np.ascontiguousarray(hanging_token_count), bs, MAX_INT)
    ekc = np.where(ebc > 0, (ebc - 1) * bs + hanging_token_count, 0).astype(np.int32)
    # ---- 4. ascending logical index per head                                 :822-834
    v_lam, by_lam = torch.from_numpy(flat).sort()
    _, by_head2 = s_slh[by_lam].sort(stable=True)
    return v_lam[by_head2].numpy().astype(np.int32), ekc, ebc
