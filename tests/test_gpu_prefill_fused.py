"""F4 parity: the fused prefill metric collector (kvc_prefill_metric_fused) against
 * the reference's own `_naive_kvc_attention` output (golden, fp16 / bf16 inputs): the
   reference's logits are fp16-rounded einsum outputs whose last bit depends on the GEMM's
   summation order, so 5e-3 relative (the bar of the existing A2c device test);
 * an exact-arithmetic check: small-integer q / k make every logit exactly representable,
   so every implementation sees identical logits and the fp32 pipeline must agree to 2e-5;
 * the in-repo unfused path (library GEMM + softmax + HIP epilogue) at a BASELINE-sized
   tile, and size-independent properties at full size."""
import numpy as np
import pytest
import torch

from tests.helpers import load_golden, reference_prefill_metrics_numpy

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(g, fn):
    tdt = torch.bfloat16 if ("dtype" in g and str(g["dtype"]) == "bf16") else torch.float16
    q = torch.from_numpy(g["q"].view(np.int16).copy()).view(tdt).to(DEV)
    k = torch.from_numpy(g["k"].view(np.int16).copy()).view(tdt).to(DEV)
    hd = q.shape[2]
    _, got = fn(q, k, None, [int(x) for x in g["prompt_lens"]], hd ** -0.5,
                torch.from_numpy(g["buffer_len"]), n_observed=int(g["n_observed"]),
                max_observed_block_size=int(g["block"]), use_l2=bool(int(g["use_l2"])),
                use_average=bool(int(g["use_average"])), use_maxpool=bool(int(g["use_maxpool"])))
    return got.cpu().numpy()


@pytest.mark.parametrize("case", range(5))
def test_fused_collector_matches_reference_output(case):
    from vllm_kvcompress_amd.kvcompress.prefill import fused_kvc_attention
    g = load_golden(f"agg_prefill_fused_{case}")
    got = _run(g, fused_kvc_attention)
    tol = 5e-3 if str(g["dtype"]) == "f16" else 4e-2          # bf16 logits: 8-bit mantissa
    np.testing.assert_allclose(got, g["ref_kv_metric_output"], rtol=tol, atol=1e-4)


@pytest.mark.parametrize("cfg", [
    # lens, n_observed, block, buffer, Hq, Hk, hd, flags(l2, avg, pool)
    ([300], 300, 128, [0], 4, 4, 64, (True, False, True)),
    ([129, 64], 1000, 64, [4, 0], 8, 2, 128, (True, True, True)),        # GQA keys, avg
    ([513], 200, 200, [17], 4, 1, 128, (False, False, False)),           # MQA keys, L1
    ([70], 70, 4096, [0], 2, 2, 64, (True, False, False)),               # one block
])
def test_fused_collector_exact_logits(cfg):
    """integer-valued q, k in [-2, 2]: |q.k| <= 4 hd is exact in fp16 and in fp32, so the
    rounding of the logits cannot differ between implementations"""
    from vllm_kvcompress_amd.kvcompress.prefill import fused_kvc_attention
    lens, n_obs, blk, buf, Hq, Hk, hd, (l2, avg, pool) = cfg
    rng = np.random.default_rng(len(lens) * 1000 + hd + Hq)
    T = sum(lens)
    q = rng.integers(-2, 3, size=(T, Hq, hd)).astype(np.float16)
    kk = rng.integers(-2, 3, size=(T, Hk, hd)).astype(np.float16)
    k_rep = np.repeat(kk, Hq // Hk, axis=1)                                # what the engine passes
    g = dict(q=q.view(np.int16), k=k_rep.view(np.int16), prompt_lens=np.asarray(lens, np.int32),
             buffer_len=np.asarray(buf, np.int32), n_observed=np.int32(n_obs), block=np.int32(blk),
             use_l2=np.int32(l2), use_average=np.int32(avg), use_maxpool=np.int32(pool))
    want = reference_prefill_metrics_numpy(g)
    _, got = fused_kvc_attention(torch.from_numpy(q).to(DEV), torch.from_numpy(kk).to(DEV), None, lens,
                                 hd ** -0.5, torch.tensor(buf, dtype=torch.int32), n_observed=n_obs,
                                 max_observed_block_size=blk, use_l2=l2, use_average=avg,
                                 use_maxpool=pool)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-5, atol=1e-7)
    # repeated keys give the same result as the un-repeated ones
    _, got2 = fused_kvc_attention(torch.from_numpy(q).to(DEV), torch.from_numpy(k_rep).to(DEV), None,
                                  lens, hd ** -0.5, torch.tensor(buf, dtype=torch.int32),
                                  n_observed=n_obs, max_observed_block_size=blk, use_l2=l2,
                                  use_average=avg, use_maxpool=pool)
    assert torch.equal(got, got2)


@pytest.mark.parametrize("cfg", [
    ([300], 300, 128, [0], 4, 4, 64, (True, False, True)),
    ([129, 64], 1000, 64, [4, 0], 8, 2, 128, (True, True, True)),
    ([513], 200, 200, [17], 4, 1, 128, (False, False, False)),
    ([200, 333], 150, 64, [0, 9], 8, 8, 128, (True, False, False)),
])
def test_fused_collector_exact_logits_bf16(cfg):
    """the same with bfloat16 inputs: integers in [-2, 2] are bf16 values, q.k is an integer of
    magnitude far below 256 (exact in bf16 as well as in fp32), so the logits every implementation
    rounds to bf16 are the same numbers -- the fp32 softmax / column-sum pipeline must then agree to
    2e-5 like the fp16 one, which the 4e-2 of the reference-output fixtures (bf16 logits of random
    inputs: 8 bits of mantissa, summation-order dependent) cannot show"""
    from vllm_kvcompress_amd.kvcompress.prefill import fused_kvc_attention
    lens, n_obs, blk, buf, Hq, Hk, hd, (l2, avg, pool) = cfg
    rng = np.random.default_rng(len(lens) * 77 + hd + Hq)
    T = sum(lens)
    qf = rng.integers(-2, 3, size=(T, Hq, hd)).astype(np.float32)
    kf = rng.integers(-2, 3, size=(T, Hk, hd)).astype(np.float32)
    assert float(np.abs(np.einsum("thd,shd->hts", qf[:64], np.repeat(kf, Hq // Hk, axis=1)[:64])).max()) < 256
    bits = lambda a: (a.view(np.uint32) >> np.uint32(16)).astype(np.uint16).view(np.int16)   # exact: small integers
    k_rep = np.repeat(kf, Hq // Hk, axis=1)
    g = dict(q=bits(qf), k=bits(k_rep), prompt_lens=np.asarray(lens, np.int32), dtype=np.asarray("bf16"),
             buffer_len=np.asarray(buf, np.int32), n_observed=np.int32(n_obs), block=np.int32(blk),
             use_l2=np.int32(l2), use_average=np.int32(avg), use_maxpool=np.int32(pool))
    want = reference_prefill_metrics_numpy(g)
    qt = torch.from_numpy(qf).to(DEV).to(torch.bfloat16)
    kt = torch.from_numpy(kf).to(DEV).to(torch.bfloat16)
    _, got = fused_kvc_attention(qt, kt, None, lens, hd ** -0.5, torch.tensor(buf, dtype=torch.int32),
                                 n_observed=n_obs, max_observed_block_size=blk, use_l2=l2, use_average=avg,
                                 use_maxpool=pool)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-5, atol=1e-7)


def test_fused_collector_vs_unfused_path_and_properties():
    """K = 8192 keys, all queries observed in blocks of 1024 (config-5 style, scaled to what
    the unfused path can materialise): fused vs library-GEMM path, and properties: with L1
    metrics, no pooling and buffer 0 every query row distributes exactly 1 over the keys, so
    the metrics of a sequence sum to the number of observed queries."""
    from vllm_kvcompress_amd.kvcompress.prefill import fused_kvc_attention, naive_kvc_attention
    torch.manual_seed(0)
    K, Hq, Hk, hd = 8192, 8, 2, 128
    q = (torch.randn(K, Hq, hd, device=DEV) * 0.7).half()
    kk = (torch.randn(K, Hk, hd, device=DEV) * 0.7).half()
    k_rep = kk.repeat_interleave(Hq // Hk, dim=1)
    buf = torch.zeros(1, dtype=torch.int32)
    _, fused = fused_kvc_attention(q, kk, None, [K], hd ** -0.5, buf, n_observed=K,
                                   max_observed_block_size=1024)
    _, naive = naive_kvc_attention(q, k_rep, None, [K], hd ** -0.5, buf, n_observed=K,
                                   max_observed_block_size=1024)
    np.testing.assert_allclose(fused.cpu().numpy(), naive.cpu().numpy(), rtol=1e-2, atol=1e-4)
    _, l1 = fused_kvc_attention(q, kk, None, [K], hd ** -0.5, buf, n_observed=K,
                                max_observed_block_size=1024, use_l2=False, use_maxpool=False)
    sums = l1.double().sum(0).cpu().numpy()
    np.testing.assert_allclose(sums, np.full(Hq, float(K)), rtol=1e-4)
    assert bool((l1 >= 0).all())
