"""schedule_evictions' third schedule (kvc_schedule_params.schedule_path 4, chosen by itself for bulk
evictions): T* from a bracket around a quantile of a sample of the
keys, one counting / collecting pass, per-head sorted lists -- instead of four digit rounds over all
the keys.  Exact or not at all: when the bracket misses (lists run over, T* not among the listed
thresholds) a device flag is raised and the digit rounds behind it redo the work.  The oracle's
result bit for bit either way; these tests also pin WHICH schedule produced it."""
import numpy as np
import pytest
import torch

from tests.helpers import oracle_pipeline
from vllm_kvcompress_amd.harness import device as hdev
from vllm_kvcompress_amd.harness import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KEYS = ("eli", "ekc", "ebc", "cmi", "cmc")


def _run(st, evicted, path, mode="per_sequence", lean=False):
    ds = hdev.upload(st, DEV, mode=mode)
    ds.cm.schedule_path = path
    ds.cm.lean_outputs = lean
    eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, st, evicted)
    out = dict(eli=eli.cpu().numpy(), ekc=ekc.cpu().numpy(), ebc=ebc.cpu().numpy(),
               cmi=cmi.cpu().numpy(), cmc=cmc.cpu().numpy())
    return out, ds.cm.last_schedule_path()


def _blocks(st):
    bs = st.block_size
    return ((st.context_lens.astype(np.int64) + bs - 1) // bs).sum(0).sum(-1)


@pytest.mark.parametrize("mode", ["per_sequence", "reference"])
@pytest.mark.parametrize("frac", [0.03, 0.3, 0.5, 0.9, 1.0])
@pytest.mark.parametrize("L,H,bs,seq_lens,compressed,shape", [
    (4, 8, 16, [2100], False, "perm"),
    (4, 8, 16, [2100, 1500], False, "decay"),
    (2, 8, 32, [4200], True, "perm"),
    (3, 4, 8, [1100, 900, 1300], True, "perm"),
    (8, 8, 16, [1040], False, "oldest"),
])
def test_bracket_schedule_equals_the_oracle(L, H, bs, seq_lens, compressed, shape, frac, mode):
    """(mode "reference" with more than one sequence: the batch > 1 rule, k' of every sequence from
    the key pass's counts before the brackets are placed)"""
    st = synth.make_state(num_layers=L, num_kv_heads=H, block_size=bs, seq_lens=seq_lens, seed=11,
                          protected=[bs + 3 + 5 * i for i in range(len(seq_lens))], compressed=compressed,
                          metric_shape=shape)
    evicted = [int(n * frac) for n in _blocks(st)]
    if mode == "reference" and len(seq_lens) > 1:
        evicted = [int(e * (0.6 + 0.2 * i)) for i, e in enumerate(evicted)]      # uneven asks: later sequences un-evict earlier ones
    want = oracle_pipeline(st, evicted, mode=mode)
    got, how = _run(st, evicted, 4, mode=mode)
    assert how.startswith("bracket"), how
    for key in KEYS:
        np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} {how}")


def test_bracket_finishes_on_its_own_on_bulk_evictions():
    """heads of 8 k slots, half of them evicted: no fallback"""
    for mode, seq_lens in (("reference", [8200]), ("per_sequence", [8200, 6100])):
        st = synth.make_state(num_layers=4, num_kv_heads=4, block_size=16, seq_lens=seq_lens, seed=3, protected=32)
        evicted = [int(n * 0.5) for n in _blocks(st)]
        want = oracle_pipeline(st, evicted, mode=mode)
        got, how = _run(st, evicted, 4, mode=mode)
        assert how == "bracket", how
        for key in KEYS:
            np.testing.assert_array_equal(got[key], want[key], err_msg=key)


def test_automatic_choice():
    """64 Ki slots per sequence and 64 blocks per head on: the bracket; smaller calls: the digit rounds"""
    big = synth.make_state(num_layers=4, num_kv_heads=8, block_size=16, seq_lens=[2100], seed=1, protected=32)
    two = synth.make_state(num_layers=4, num_kv_heads=8, block_size=16, seq_lens=[2100, 2500], seed=1, protected=32)
    small = synth.make_state(num_layers=4, num_kv_heads=8, block_size=16, seq_lens=[500], seed=1, protected=32)
    for st, mode, expect in ((big, "reference", "bracket"), (two, "per_sequence", "bracket"),
                             (two, "reference", "bracket"), (small, "reference", "general")):
        evicted = [int(n * 0.5) for n in _blocks(st)]
        want = oracle_pipeline(st, evicted, mode=mode)
        got, how = _run(st, evicted, 0, mode=mode)
        assert how == expect, (how, expect)
        for key in KEYS:
            np.testing.assert_array_equal(got[key], want[key], err_msg=key)


def test_forced_general_path_never_takes_the_bracket():
    st = synth.make_state(num_layers=4, num_kv_heads=8, block_size=16, seq_lens=[2100], seed=1, protected=32)
    evicted = [int(n * 0.5) for n in _blocks(st)]
    _, how = _run(st, evicted, 1)
    assert how == "general"


@pytest.mark.parametrize("ties", [1, 2, 3, 30])
def test_metric_ties(ties):
    """a few distinct metric values: every key of T*'s value lies inside the bracket -- the lists
    run over (fallback) or hold them all (ties handed out in (head, chunk) order); exact either way"""
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=16, seq_lens=[4100, 3000], seed=9, protected=17,
                          tie_levels=ties)
    evicted = [int(n * 0.4) for n in _blocks(st)]
    want = oracle_pipeline(st, evicted, mode="per_sequence")
    got, how = _run(st, evicted, 4)
    assert how.startswith("bracket"), how
    for key in KEYS:
        np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} {how}")


def test_tiny_heads_overflow_their_lists_and_fall_back():
    st = synth.make_state(num_layers=1, num_kv_heads=2, block_size=16, seq_lens=[300], seed=2, protected=16)
    evicted = [int(n * 0.5) for n in _blocks(st)]
    want = oracle_pipeline(st, evicted, mode="reference")
    got, how = _run(st, evicted, 4, mode="reference")
    assert how in ("bracket", "bracket+fallback"), how
    for key in KEYS:
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)


def test_zero_and_overasked_sequences():
    """a sequence that evicts nothing next to one that is asked for more chunks than it has
    finite-threshold ones (k' = all of them), against the digit rounds"""
    st = synth.make_state(num_layers=4, num_kv_heads=4, block_size=16, seq_lens=[4100, 4100, 2000], seed=4,
                          protected=[40, 20, 17])
    nb = _blocks(st)
    for evicted in ([0, int(nb[1]) + 50, int(nb[2] * 0.5)], [int(nb[0]), 0, 0], [0, 0, 0]):
        a, how_a = _run(st, evicted, 4)
        b, how_b = _run(st, evicted, 1)
        assert how_a.startswith("bracket") and how_b == "general"
        for key in ("eli", "ekc", "ebc"):
            np.testing.assert_array_equal(a[key], b[key], err_msg=f"{key} {evicted} {how_a}")


def test_batch_rule_with_zero_and_uneven_asks_against_the_digit_rounds():
    """the reference's batch > 1 rule (k' of a sequence depends on the inf-threshold chunks of the
    ones in front and on the asks of the ones behind), bracket against digit rounds"""
    st = synth.make_state(num_layers=4, num_kv_heads=4, block_size=16, seq_lens=[4100, 3000, 4100, 2000], seed=14,
                          protected=[40, 20, 17, 33])
    nb = _blocks(st)
    for evicted in ([int(nb[0] * 0.5), 0, int(nb[2] * 0.9), int(nb[3] * 0.1)], [int(n) for n in nb], [0, 0, 0, int(nb[3] * 0.5)],
                    [int(nb[0] * 0.2), int(nb[1] * 0.2), int(nb[2] * 0.2), 3]):
        a, how_a = _run(st, evicted, 4, mode="reference")
        b, how_b = _run(st, evicted, 1, mode="reference")
        assert how_a.startswith("bracket") and how_b == "general"
        for key in ("eli", "ekc", "ebc"):
            np.testing.assert_array_equal(a[key], b[key], err_msg=f"{key} {evicted} {how_a}")


def test_batch_rule_with_a_chunk_nobody_claims():
    """a logical block without a physical one (metadata detached; outside what the reference
    defines): its keys were never written and never counted -- the counting pass meets them,
    raises the flag, and the digit rounds give their answer: one behaviour whatever the schedule"""
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=16, seq_lens=[4100, 3000], seed=15, protected=32)
    victim = int(st.block_tables[1, 1, 2, 3])
    st.seq_index_by_block[victim] = -1
    evicted = [int(n * 0.5) for n in _blocks(st)]
    a, how_a = _run(st, evicted, 4, mode="reference")
    b, how_b = _run(st, evicted, 1, mode="reference")
    assert how_a == "bracket+fallback" and how_b == "general", how_a
    for key in KEYS:
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    # sequences that do not couple: the key pass (which no longer has a 0xFF fill in front of it) counts the
    # logical blocks it finds, bracket_kernel sees one missing -> the fallback builds keys and holes anew: the same answer
    c, how_c = _run(st, evicted, 4, mode="per_sequence")
    d, _ = _run(st, evicted, 1, mode="per_sequence")
    assert how_c == "bracket+fallback", how_c
    for key in KEYS:
        np.testing.assert_array_equal(c[key], d[key], err_msg=key)
    # ... and the call after it starts from clean counters again
    st2 = synth.make_state(num_layers=2, num_kv_heads=4, block_size=16, seq_lens=[4100, 3000], seed=15, protected=32)
    e, how_e = _run(st2, evicted, 4, mode="per_sequence")
    f, _ = _run(st2, evicted, 1, mode="per_sequence")
    assert how_e == "bracket", how_e
    c, d = e, f
    for key in KEYS:
        np.testing.assert_array_equal(c[key], d[key], err_msg=key)


def test_a_skewed_head_absorbs_the_eviction():
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=16, seq_lens=[8200], seed=6, protected=32)
    blocks = np.nonzero((st.layer_index_by_block == 1) & (st.head_index_by_block == 2))[0]
    st.metrics[blocks] -= np.float32(1e7)
    evicted = [int(_blocks(st)[0] * 0.1)]
    want = oracle_pipeline(st, evicted, mode="reference")
    got, how = _run(st, evicted, 4, mode="reference")
    assert how.startswith("bracket"), how
    for key in KEYS:
        np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} {how}")


def test_lean_outputs():
    st = synth.make_state(num_layers=4, num_kv_heads=4, block_size=16, seq_lens=[4100], seed=8, protected=32)
    evicted = [int(_blocks(st)[0] * 0.5)]
    want = oracle_pipeline(st, evicted, mode="reference")
    got, how = _run(st, evicted, 4, mode="reference", lean=True)
    assert how == "bracket", how
    offs = st.evicted_kv_offsets.reshape(-1)
    for g in range(offs.size):          # lean: only the first evicted_kv_count entries of a head are defined
        n = int(got["ekc"].reshape(-1)[g])
        np.testing.assert_array_equal(got["eli"][offs[g]:offs[g] + n], want["eli"][offs[g]:offs[g] + n])
    for key in ("ekc", "ebc", "cmi", "cmc"):
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)


def test_a_raised_flag_sends_the_next_calls_to_the_digit_rounds():
    """the host policy of the small-eviction schedule holds for the bracket as well: three metric
    values, a third of the keys equal to T* -> the lists run over on every call"""
    st = synth.make_state(num_layers=4, num_kv_heads=8, block_size=16, seq_lens=[2100], seed=5, protected=32,
                          tie_levels=3)
    evicted = [int(_blocks(st)[0] * 0.5)]
    want = oracle_pipeline(st, evicted, mode="reference")
    ds = hdev.upload(st, DEV, mode="reference")
    ds.cm.schedule_path = 0
    args = (list(st.seq_indices), ds.seq_positions, evicted, ds.context_lens, ds.hanging_token_count,
            ds.evicted_kv_offsets, list(st.protected))
    hows = []
    for _ in range(6):
        out = ds.cm.schedule_evictions(*args, total_slots=st.total_slots)
        hows.append(ds.cm.last_schedule_path())      # (synchronises: the flag copy has landed by the next call)
        for got, key in zip(out, ("eli", "ekc", "ebc")):
            np.testing.assert_array_equal(got.cpu().numpy(), want[key], err_msg=key)
    assert hows == ["bracket+fallback", "general", "bracket+fallback", "general", "general", "bracket+fallback"], hows


def test_config2_size_against_the_digit_rounds():
    """BASELINE configs[1] (32 layers x 8 heads x 32 768 slots, keep half): the bracket finishes on
    its own and equals the digit rounds (which the suite pins to the oracle at oracle sizes)"""
    st = synth.make_state(num_layers=32, num_kv_heads=8, block_size=16, seq_lens=[32769], seed=0, protected=32)
    evicted = [synth.evict_block_count(context_lens_lh=st.context_lens[:, 0, :], seq_len=32769, block_size=16,
                                       protected_window_size=32, max_cache_tokens=16384)]
    ds = hdev.upload(st, DEV, mode="reference")
    args = (list(st.seq_indices), ds.seq_positions, evicted, ds.context_lens, ds.hanging_token_count,
            ds.evicted_kv_offsets, list(st.protected))
    ds.cm.schedule_path = 1
    want = [t.clone() for t in ds.cm.schedule_evictions(*args, total_slots=st.total_slots)]
    ds.cm.schedule_path = 0
    got = ds.cm.schedule_evictions(*args, total_slots=st.total_slots)
    assert ds.cm.last_schedule_path() == "bracket"
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # every freed chunk's threshold is at most every kept chunk's, over the whole sequence
    ekc = got[1].cpu().numpy().reshape(-1)
    assert int(got[2].sum().item()) == evicted[0]
    assert (ekc[ekc > 0] % 16 == st.hanging_token_count.reshape(-1)[ekc > 0] % 16).all()
