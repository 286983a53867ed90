"""The fork's call form (device tensor of counts, no N; reference vllm/kvcompress/scheduler.py:245-260, 491-499) WITHOUT a
wait in front of the launches (ABI version 8, kvc_schedule_params.total_slots_dev; CompressionMetrics.deferred_n): the
schedule is enqueued on an upper bound of N right behind the summary launch that leaves the true N on the device.  Same
results as the oracle whichever way a call goes -- the waiting way (first call of a batch size), the deferred way (bracket
and digit rounds), and a deferred call whose bound did not hold (voided on the device, repeated)."""
import numpy as np
import pytest
import torch

from tests.helpers import oracle_pipeline
from vllm_kvcompress_amd.harness import device as hdev, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KEYS = ("eli", "ekc", "ebc")


def _blocks(st):
    bs = st.block_size
    return ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)


def _fork_call(ds, st, evicted):
    """exactly what the fork passes: slot indices as a list, a fresh position tensor, the counts as a device int tensor,
    the protected windows as a tuple, no total_slots"""
    got = ds.cm.schedule_evictions(list(st.seq_indices), ds.seq_positions.clone(),
                                   torch.tensor(evicted, dtype=torch.int, device=DEV), ds.context_lens,
                                   ds.hanging_token_count, ds.evicted_kv_offsets, tuple(st.protected))
    return dict(zip(KEYS, (t.cpu().numpy() for t in got))), ds.cm.last_schedule_path()


@pytest.mark.parametrize("mode,seq_lens,expect", [
    ("per_sequence", [8200], "bracket"),             # bulk eviction of one long sequence
    ("per_sequence", [8200, 6100], "bracket"),       # ... of two (sequences that do not couple)
    ("reference", [8200], "bracket"),                # the reference's mode with one sequence
    ("per_sequence", [500], "general"),              # a small batch: the digit rounds
    ("per_sequence", [700, 300, 450], "general"),
])
def test_deferred_calls_equal_the_oracle(mode, seq_lens, expect):
    st = synth.make_state(num_layers=4, num_kv_heads=4, block_size=16, seq_lens=seq_lens, seed=21, protected=32)
    ds = hdev.upload(st, DEV, mode=mode)
    ds.cm.schedule_path = 0
    assert ds.cm.deferred_n
    for frac in (0.5, 0.5, 0.3, 0.7, 0.5):
        evicted = [int(n * frac) for n in _blocks(st)]
        want = oracle_pipeline(st, evicted, mode=mode)
        got, how = _fork_call(ds, st, evicted)
        assert how == expect, (how, ds.cm.last_schedule_reason)
        assert got["eli"].shape[0] == st.total_slots
        for key in KEYS:
            np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} frac {frac} call {ds.cm.deferred_calls}")
    # the first call waited (it learnt the bound), the rest did not
    assert ds.cm.deferred_calls == 4 and ds.cm.deferred_voided == 0
    assert "N on the device" in ds.cm.last_schedule_reason


def test_a_bound_that_does_not_hold_voids_the_call_and_it_is_repeated():
    st = synth.make_state(num_layers=4, num_kv_heads=4, block_size=16, seq_lens=[8200], seed=22, protected=32)
    ds = hdev.upload(st, DEV, mode="per_sequence")
    evicted = [int(n * 0.5) for n in _blocks(st)]
    want = oracle_pipeline(st, evicted, mode="per_sequence")
    got, _ = _fork_call(ds, st, evicted)              # learns the bound
    B = len(st.seq_indices)
    assert ds.cm._dn_bound[B] >= st.total_slots
    ds.cm._dn_bound[B] = 1 << 16                      # ... which some later, larger batch exceeds (a multiple of the block size)
    assert st.total_slots > 1 << 16
    got, how = _fork_call(ds, st, evicted)
    assert ds.cm.deferred_calls == 1 and ds.cm.deferred_voided == 1
    assert how == "bracket" and "N on the device" not in ds.cm.last_schedule_reason
    for key in KEYS:
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)
    assert ds.cm._dn_bound[B] >= st.total_slots       # raised: the next call goes the deferred way again
    got, how = _fork_call(ds, st, evicted)
    assert ds.cm.deferred_calls == 2 and ds.cm.deferred_voided == 1
    for key in KEYS:
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)


def test_small_eviction_batches_and_coupled_batches_keep_waiting():
    """the small-eviction schedule is chosen from host-side counts, the reference's batch > 1 rule couples the sequences:
    neither goes the deferred way"""
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=16, seq_lens=[1100, 900], seed=23, protected=32)
    for mode, evicted in (("per_sequence", [3, 2]), ("reference", [int(n * 0.5) for n in _blocks(st)])):
        ds = hdev.upload(st, DEV, mode=mode)
        want = oracle_pipeline(st, evicted, mode=mode)
        for _ in range(3):
            got, how = _fork_call(ds, st, evicted)
            for key in KEYS:
                np.testing.assert_array_equal(got[key], want[key], err_msg=f"{mode} {key} ({how})")
        assert ds.cm.deferred_calls == 0, mode


def test_switched_off():
    st = synth.make_state(num_layers=4, num_kv_heads=4, block_size=16, seq_lens=[8200], seed=24, protected=32)
    ds = hdev.upload(st, DEV, mode="per_sequence")
    ds.cm.deferred_n = False
    evicted = [int(n * 0.5) for n in _blocks(st)]
    want = oracle_pipeline(st, evicted, mode="per_sequence")
    for _ in range(3):
        got, _ = _fork_call(ds, st, evicted)
        for key in KEYS:
            np.testing.assert_array_equal(got[key], want[key], err_msg=key)
    assert ds.cm.deferred_calls == 0
