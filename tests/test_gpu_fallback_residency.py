"""The single launch that redoes a small-eviction / bracket call on the general pipeline
(kvc_schedule.hip section 8) must give the oracle's schedule WHATEVER part of its grid is resident:
its phases wait for work (claimed virtual workgroups), not for workgroups.  Forced here two ways --
a grid far larger than the device can hold at once (kvc_schedule_params.fallback_grid) and compute
units held by a spinning kernel on another stream -- on states that raise the flag.  Either the
oracle's result or an exception, never a different schedule with rc 0; the reference always returns
a valid schedule (vllm/kvcompress/metrics.py:441-847)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from tests.helpers import oracle_pipeline
from vllm_kvcompress_amd.harness import device as hdev
from vllm_kvcompress_amd.harness import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KEYS = ("eli", "ekc", "ebc", "cmi", "cmc")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OVERSIZED = 40000            # workgroups; an MI355X holds at most 8 x 256 of this kernel at once


def _run(st, evicted, path, mode, grid=0, strict=False, before=None):
    ds = hdev.upload(st, DEV, mode=mode)
    ds.cm.schedule_path = path
    ds.cm.fallback_grid = grid
    ds.cm.strict_fallback = strict
    if before is not None:
        before()
    eli, ekc, ebc, cmi, cmc = hdev.schedule(ds, st, evicted)
    out = dict(eli=eli.cpu().numpy(), ekc=ekc.cpu().numpy(), ebc=ebc.cpu().numpy(),
               cmi=cmi.cpu().numpy(), cmc=cmc.cpu().numpy())
    return out, ds.cm.last_schedule_path(), ds.cm


def _skewed_small(mode, L=2):
    """one head absorbs the whole eviction: more chunks than a record holds -> flag"""
    B = 1 if mode == "per_sequence" else 2
    cap = 1024
    st = synth.make_state(num_layers=L, num_kv_heads=4, block_size=16, seq_lens=[3 * cap] * B, seed=5,
                          protected=17, steady_cap=cap, spare_block_frac=0.05)
    blocks = np.nonzero((st.layer_index_by_block == 1) & (st.head_index_by_block == 2)
                        & (st.seq_index_by_block == 0))[0]
    st.metrics[blocks] -= np.float32(1e7)
    return st, [24] + [1] * (B - 1)


def _tied_bulk():
    """three metric values, a third of the keys equal to T*: the bracket's lists run over -> flag"""
    st = synth.make_state(num_layers=4, num_kv_heads=8, block_size=16, seq_lens=[2100], seed=5, protected=32,
                          tie_levels=3)
    nblk = ((st.context_lens.astype(np.int64) + 15) // 16).sum()
    return st, [int(nblk * 0.5)]


def _detached_bulk():
    """the batch > 1 rule with a chunk nobody claims: the counting pass raises the flag and the
    coupled phases (totals | k' | pick) run inside the single launch"""
    st = synth.make_state(num_layers=2, num_kv_heads=4, block_size=16, seq_lens=[4100, 3000], seed=15, protected=32)
    st.seq_index_by_block[int(st.block_tables[1, 1, 2, 3])] = -1
    nb = ((st.context_lens.astype(np.int64) + 15) // 16).sum(0).sum(-1)
    return st, [int(n * 0.5) for n in nb]


CASES = [
    ("small_per_sequence", lambda: _skewed_small("per_sequence"), 2, "per_sequence", "small_eviction+fallback"),
    ("small_reference_b2", lambda: _skewed_small("reference"), 2, "reference", "small_eviction+fallback"),
    ("small_auto", lambda: _skewed_small("per_sequence", L=4), 0, "per_sequence", "small_eviction+fallback"),
    ("bracket_ties", _tied_bulk, 4, "reference", "bracket+fallback"),
    ("bracket_coupled_detached", _detached_bulk, 4, "reference", "bracket+fallback"),
]


def _check(got, want, tag):
    for key in KEYS:
        np.testing.assert_array_equal(got[key], want[key], err_msg=f"{key} {tag}")


@pytest.mark.parametrize("name,make,path,mode,how_want", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("grid", [1, 7, OVERSIZED])
def test_fallback_result_does_not_depend_on_how_much_of_the_grid_is_resident(name, make, path, mode, how_want, grid):
    st, evicted = make()
    if name == "bracket_coupled_detached":
        # (a detached block is outside what the reference defines: the digit rounds are the yardstick)
        want, how_w, _ = _run(st, evicted, 1, mode)
        assert how_w == "general"
    else:
        want = oracle_pipeline(st, evicted, mode=mode)
    got, how, cm = _run(st, evicted, path, mode, grid=grid, strict=True)
    assert how == how_want, how
    _check(got, want, f"{name} grid={grid}")


def _probe():
    path = os.path.join(REPO, "tools", "libkvc_probe.so")
    if not os.path.exists(path):
        pytest.skip("tools/libkvc_probe.so not built")
    lib = ctypes.CDLL(path)
    if not hasattr(lib, "kvc_probe_occupy"):
        pytest.skip("tools/libkvc_probe.so predates kvc_probe_occupy")
    lib.kvc_probe_occupy.restype = ctypes.c_int32
    lib.kvc_probe_occupy.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]
    return lib


@pytest.mark.parametrize("name,make,path,mode,how_want", CASES, ids=[c[0] for c in CASES])
def test_fallback_while_another_stream_holds_the_compute_units(name, make, path, mode, how_want):
    """2 x 64 KiB of LDS per CU spinning for 40 ms on a side stream: the fallback's default grid
    (three workgroups per CU) cannot be resident at once while that runs"""
    lib = _probe()
    st, evicted = make()
    if name == "bracket_coupled_detached":
        want, _, _ = _run(st, evicted, 1, mode)
    else:
        want = oracle_pipeline(st, evicted, mode=mode)
    side = torch.cuda.Stream(device=DEV)
    cus = torch.cuda.get_device_properties(DEV).multi_processor_count

    def occupy():
        torch.cuda.synchronize()
        assert lib.kvc_probe_occupy(2 * cus, 65536, 40000, ctypes.c_void_p(side.cuda_stream)) == 0

    got, how, cm = _run(st, evicted, path, mode, strict=True, before=occupy)
    torch.cuda.synchronize()
    assert how == how_want, how
    _check(got, want, f"{name} next to a spinning kernel")


def test_two_fallbacks_at_once_on_two_streams():
    """two CompressionMetrics, two streams, both calls fall back: with a barrier over workgroups
    two half-resident grids wait for each other; here both finish and both are exact"""
    st, evicted = _skewed_small("per_sequence")
    want = oracle_pipeline(st, evicted, mode="per_sequence")
    streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    states = []
    for s in streams:
        with torch.cuda.stream(s):
            ds = hdev.upload(st, DEV, mode="per_sequence")
            ds.cm.schedule_path = 2
            ds.cm.fallback_grid = 1536         # twice what the occupancy query sizes: the two do not fit together
            states.append(ds)
    torch.cuda.synchronize()
    outs = []
    for rep in range(3):
        outs.clear()
        for s, ds in zip(streams, states):
            with torch.cuda.stream(s):
                outs.append(hdev.schedule(ds, st, evicted))
        torch.cuda.synchronize()
        for ds, out in zip(states, outs):
            assert ds.cm.last_schedule_path() == "small_eviction+fallback"
            got = dict(zip(KEYS, (t.cpu().numpy() for t in out)))
            _check(got, want, f"rep {rep}")


def test_a_timed_out_wait_is_an_error_not_a_schedule():
    """bit 1 of the flag word (a wait given up: device fault) -> RuntimeError from the host, at the
    call in strict mode, one call later otherwise; afterwards only the launch chain is used"""
    st, evicted = _skewed_small("per_sequence")
    ds = hdev.upload(st, DEV, mode="per_sequence")
    ds.cm.schedule_path = 0
    args = (list(st.seq_indices), ds.seq_positions, [1], ds.context_lens, ds.hanging_token_count,
            ds.evicted_kv_offsets, list(st.protected))
    ds.cm.schedule_evictions(*args, total_slots=st.total_slots)
    assert ds.cm.last_schedule_path() == "small_eviction"
    torch.cuda.synchronize()
    slot, _, _, ticket = ds.cm._fb_inflight[-1]
    # what the call's last launch would have left in its page-locked word after a fault: (ticket << 8) | flag word
    ds.cm._fb_pin[slot] = (ticket << 8) | 2
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="gave up a wait"):
        ds.cm.schedule_evictions(*args, total_slots=st.total_slots)
    out = ds.cm.schedule_evictions(*args, total_slots=st.total_slots)
    assert ds.cm.last_schedule_path() == "general"
    want = oracle_pipeline(st, [1], mode="per_sequence")
    for got, key in zip(out, ("eli", "ekc", "ebc")):
        np.testing.assert_array_equal(got.cpu().numpy(), want[key], err_msg=key)
