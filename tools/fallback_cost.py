"""What a call that falls back costs on the small-eviction schedule (the single-launch general
pipeline of kvc_schedule.hip section 8) against the general pipeline run directly, and the phase
times workgroup 0 of the fallback kernel leaves in the workspace.  GPU box: python tools/fallback_cost.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from vllm_kvcompress_amd.harness import device as hdev, synth
DEV = "cuda:0"
def run(B, path, skew):
    st = synth.make_state(num_layers=32, num_kv_heads=8, block_size=16, seq_lens=[32769] * B, seed=1, protected=32, steady_cap=4096, spare_block_frac=0.02)
    ev = [synth.evict_block_count(context_lens_lh=st.context_lens[:, b, :], seq_len=32769, block_size=16, protected_window_size=32, max_cache_tokens=4096) for b in range(B)]
    if skew:
        blk = np.nonzero((st.layer_index_by_block == 0) & (st.head_index_by_block == 0) & (st.seq_index_by_block == 0))[0]
        st.metrics[blk] -= np.float32(1e8)
        ev[0] = 64
    ds = hdev.upload(st, DEV, mode="per_sequence")
    ds.cm.schedule_path = path
    args = (list(st.seq_indices), ds.seq_positions, ev, ds.context_lens, ds.hanging_token_count, ds.evicted_kv_offsets, list(st.protected))
    for _ in range(3): out = ds.cm.schedule_evictions(*args, total_slots=st.total_slots)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): out = ds.cm.schedule_evictions(*args, total_slots=st.total_slots)
    b.record(); torch.cuda.synchronize()
    ws, off, _ = ds.cm.last_schedule
    stamps = ws[off + 128:off + 128 + 72].view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    ph = [(int(stamps[k + 1]) - int(stamps[k])) & 0xFFFFFFFF for k in range(1, 14) if stamps[k + 1]]
    global PH
    PH = [round(x / 100.0, 1) for x in ph]      # us
    return a.elapsed_time(b) / 10, ds.cm.last_schedule_path(), out
for B in (16, 64):
    t2, how2, o2 = run(B, 2, True)
    PH2 = PH
    t1, how1, o1 = run(B, 1, True)
    t0, how0, _ = run(B, 2, False)
    same = all(torch.equal(x, y) for x, y in zip(o1, o2))
    print("fallback phases (us): keys | 4 x (hist | scan+pick) :", PH2)
    print(B, "skewed: path2", round(t2, 3), how2, "| path1", round(t1, 3), how1, "| equal", same, "| unskewed path2", round(t0, 3), how0)
