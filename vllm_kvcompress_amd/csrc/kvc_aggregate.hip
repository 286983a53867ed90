// A2a/A2b/A2c metric aggregation and A7 reshape_and_cache for gfx950.
// All four are streaming, HBM-bound passes: wide coalesced loads, no LDS, float32
// arithmetic with explicit rounding (no FMA contraction) so results are bit-identical to
// the sequential float32 restatement in oracle/kvc_oracle.py.
#include "kvc_common.h"
#include "../../include/kvc_mi355x.h"

namespace kvc {

// ------------------------------------------------------------------ A2a
// metrics[s] += sum_q f(temp[s,q]),  optionally temp := 0 (clear_temp_metrics fused)
// reference: vllm/kvcompress/metrics.py:429-439 and :337-342
template <int QPK>
__global__ __launch_bounds__(256) void aggregate_decode_kernel(float* __restrict__ metrics,
                                                               float* __restrict__ temp,
                                                               int64_t num_slots, int qpk_rt,
                                                               int use_l2, int clear_temp) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < num_slots; s += stride) {
    float acc = 0.0f;
    if constexpr (QPK == 4) {
      float4 t = reinterpret_cast<const float4*>(temp)[s];
      if (use_l2) { t.x = __fmul_rn(t.x, t.x); t.y = __fmul_rn(t.y, t.y); t.z = __fmul_rn(t.z, t.z); t.w = __fmul_rn(t.w, t.w); }
      acc = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.0f, t.x), t.y), t.z), t.w);
      if (clear_temp) reinterpret_cast<float4*>(temp)[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const int qpk = QPK > 0 ? QPK : qpk_rt;
      for (int q = 0; q < qpk; ++q) {
        float t = temp[s * qpk + q];
        if (use_l2) t = __fmul_rn(t, t);
        acc = __fadd_rn(acc, t);
        if (clear_temp) temp[s * qpk + q] = 0.0f;
      }
    }
    metrics[s] = __fadd_rn(metrics[s], acc);
  }
}

// ------------------------------------------------------------------ A2b
// reference: vllm/kvcompress/metrics.py:396-427
__global__ __launch_bounds__(256) void aggregate_prefill_kernel(float* __restrict__ metrics,
                                                                const float* __restrict__ pm,
                                                                const int64_t* __restrict__ slots,
                                                                int64_t n, int qpk) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (token, kv head)
  if (i >= n) return;
  float acc = 0.0f;
  for (int q = 0; q < qpk; ++q) acc = __fadd_rn(acc, pm[i * qpk + q]);
  const int64_t s = slots[i];
  if (s >= 0) metrics[s] = __fadd_rn(metrics[s], acc);
}

// ------------------------------------------------------------------ A2c
// step 1: masked column sums of the (optionally squared) probability tile
// reference: vllm/attention/backends/flash_attn.py:1189-1201
__global__ __launch_bounds__(256) void epilogue_colsum_kernel(float* __restrict__ colsum,
                                                              const float* __restrict__ probs,
                                                              int Hq, int qb, int K, int q_offset,
                                                              int buffer_len, int use_l2,
                                                              int use_average) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int h = blockIdx.y;
  if (k >= K) return;
  // tril(diagonal = q_offset - buffer_len): key k counts for query row q iff k - q <= diag
  int qmin = k - (q_offset - buffer_len);
  qmin = qmin < 0 ? 0 : qmin;
  const float* col = probs + ((int64_t)h * qb) * K + k;
  float acc = 0.0f;
  int q = qmin;
  for (; q + 8 <= qb; q += 8) {                  // 8 independent loads in flight, summed in order
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = col[(int64_t)(q + u) * K];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __fadd_rn(acc, use_l2 ? __fmul_rn(t[u], t[u]) : t[u]);
  }
  for (; q < qb; ++q) {
    float v = col[(int64_t)q * K];
    if (use_l2) v = __fmul_rn(v, v);
    acc = __fadd_rn(acc, v);
  }
  if (use_average) acc = __fmul_rn(acc, __fdiv_rn((float)(k + 1), (float)qb));
  colsum[(int64_t)h * K + k] = acc;
}

// step 2: max_pool1d(kernel 7, pad 3, stride 1) over keys, accumulate into out[K,Hq]
// reference: flash_attn.py:1204-1210 and the accumulation at :1161
__global__ __launch_bounds__(256) void epilogue_pool_kernel(float* __restrict__ out_kh,
                                                            const float* __restrict__ colsum,
                                                            int Hq, int K, int use_maxpool) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (k, h), h fastest
  if (i >= (int64_t)K * Hq) return;
  const int k = (int)(i / Hq), h = (int)(i % Hq);
  const float* row = colsum + (int64_t)h * K;
  float v = row[k];
  if (use_maxpool) {
    const int lo = k - 3 < 0 ? 0 : k - 3, hi = k + 3 >= K ? K - 1 : k + 3;
    v = row[lo];
    for (int j = lo + 1; j <= hi; ++j) v = fmaxf(v, row[j]);
  }
  out_kh[i] = __fadd_rn(out_kh[i], v);
}

// ------------------------------------------------------------------ A7
// reference: csrc/kvcompress_cache_kernels.cu:27-89 ("auto" dtype: byte-identical store)
template <int E>
__global__ __launch_bounds__(512) void reshape_and_cache_kernel(
    const uint8_t* __restrict__ key, const uint8_t* __restrict__ value,
    uint8_t* __restrict__ key_cache, uint8_t* __restrict__ value_cache,
    float* __restrict__ kv_metrics, const int64_t* __restrict__ slot_mapping,
    const float* __restrict__ bias, int num_heads, int head_size, int bs, int64_t key_stride,
    int64_t value_stride) {
  constexpr int X = 16 / E;                            // elements per 16 B K vector
  const int64_t token = blockIdx.x;
  const int n = num_heads * head_size;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int head = i / head_size, d = i % head_size;
    const int64_t slot = slot_mapping[token * num_heads + head];
    if (slot < 0) continue;                            // padding token
    if (d == 0) kv_metrics[slot] = bias[head];
    const int64_t blk = slot / bs;
    const int off = (int)(slot % bs);
    const int64_t block_bytes = (int64_t)head_size * bs * E;
    // V: one element
    const uint8_t* vs = value + (token * value_stride + i) * E;
    uint8_t* vd = value_cache + blk * block_bytes + ((int64_t)d * bs + off) * E;
    if constexpr (E == 1) *vd = *vs;
    else if constexpr (E == 2) *reinterpret_cast<uint16_t*>(vd) = *reinterpret_cast<const uint16_t*>(vs);
    else *reinterpret_cast<uint32_t*>(vd) = *reinterpret_cast<const uint32_t*>(vs);
    // K: the first lane of every X-element group stores the whole vector
    if (d % X == 0) {
      const uint8_t* ks = key + (token * key_stride + i) * E;
      uint8_t* kd = key_cache + blk * block_bytes + (((int64_t)(d / X)) * bs + off) * 16;
      if ((reinterpret_cast<uintptr_t>(ks) & 15) == 0) {
        *reinterpret_cast<uint4*>(kd) = *reinterpret_cast<const uint4*>(ks);
      } else {
        for (int b = 0; b < 16; ++b) kd[b] = ks[b];
      }
    }
  }
}

}  // namespace kvc

extern "C" int kvc_aggregate_decode(float* metrics, float* temp_metrics, int64_t num_slots,
                                    int32_t num_queries_per_kv, int32_t use_l2,
                                    int32_t clear_temp, kvc_stream_t stream) {
  using namespace kvc;
  if (num_queries_per_kv < 1) return fail_invalid("aggregate_decode: num_queries_per_kv < 1");
  if (num_slots <= 0) return KVC_OK;
  const int64_t want = (num_slots + 255) / 256;
  const unsigned grid = (unsigned)(want < 256 * 16 ? want : 256 * 16);
  hipStream_t s = (hipStream_t)stream;
  if (num_queries_per_kv == 4)
    hipLaunchKernelGGL(aggregate_decode_kernel<4>, dim3(grid), dim3(256), 0, s, metrics, temp_metrics,
                       num_slots, 4, use_l2, clear_temp);
  else
    hipLaunchKernelGGL(aggregate_decode_kernel<0>, dim3(grid), dim3(256), 0, s, metrics, temp_metrics,
                       num_slots, num_queries_per_kv, use_l2, clear_temp);
  return check_launch("aggregate_decode");
}

extern "C" int kvc_aggregate_prefill(float* metrics, const float* prefill_metrics,
                                     const int64_t* slot_mapping, int64_t num_tokens,
                                     int32_t num_kv_heads, int32_t num_queries_per_kv,
                                     kvc_stream_t stream) {
  using namespace kvc;
  const int64_t n = num_tokens * num_kv_heads;
  if (n <= 0) return KVC_OK;
  hipLaunchKernelGGL(aggregate_prefill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, metrics, prefill_metrics, slot_mapping, n,
                     num_queries_per_kv);
  return check_launch("aggregate_prefill");
}

extern "C" size_t kvc_prefill_metric_epilogue_workspace_bytes(int32_t num_q_heads, int32_t num_keys) {
  return (size_t)num_q_heads * (size_t)num_keys * sizeof(float);
}

extern "C" int kvc_prefill_metric_epilogue(float* out_kh, const float* probs_hqk,
                                           int32_t num_q_heads, int32_t q_block, int32_t num_keys,
                                           int32_t q_offset, int32_t buffer_len, int32_t use_l2,
                                           int32_t use_average, int32_t use_maxpool,
                                           void* workspace, size_t workspace_bytes,
                                           kvc_stream_t stream) {
  using namespace kvc;
  if (num_q_heads <= 0 || num_keys <= 0 || q_block <= 0) return KVC_OK;
  if (workspace_bytes < kvc_prefill_metric_epilogue_workspace_bytes(num_q_heads, num_keys))
    return fail_invalid("prefill_metric_epilogue: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* colsum = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(epilogue_colsum_kernel, dim3((num_keys + 255) / 256, num_q_heads), dim3(256), 0,
                     s, colsum, probs_hqk, num_q_heads, q_block, num_keys, q_offset, buffer_len,
                     use_l2, use_average);
  const int64_t n = (int64_t)num_keys * num_q_heads;
  hipLaunchKernelGGL(epilogue_pool_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out_kh,
                     colsum, num_q_heads, num_keys, use_maxpool);
  return check_launch("prefill_metric_epilogue");
}

extern "C" int kvc_reshape_and_cache(const void* key, const void* value, void* key_cache,
                                     void* value_cache, float* kv_metrics,
                                     const int64_t* slot_mapping, const float* kv_metric_head_bias,
                                     int64_t num_tokens, int32_t num_heads, int32_t head_size,
                                     int32_t block_size, int32_t elem_bytes, int64_t key_stride,
                                     int64_t value_stride, kvc_stream_t stream) {
  using namespace kvc;
  if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4)
    return fail_invalid("Unsupported cache element size: " + std::to_string(elem_bytes));
  if (head_size % (16 / elem_bytes) != 0)
    return fail_invalid("Unsupported head size: " + std::to_string(head_size));
  if (num_tokens <= 0) return KVC_OK;
  const int n = num_heads * head_size;
  const int threads = n < 512 ? ((n + 63) / 64 * 64) : 512;
  hipStream_t s = (hipStream_t)stream;
#define KVC_RC(E)                                                                                   \
  hipLaunchKernelGGL(reshape_and_cache_kernel<E>, dim3((unsigned)num_tokens), dim3(threads), 0, s,   \
                     (const uint8_t*)key, (const uint8_t*)value, (uint8_t*)key_cache,               \
                     (uint8_t*)value_cache, kv_metrics, slot_mapping, kv_metric_head_bias,           \
                     num_heads, head_size, block_size, key_stride, value_stride)
  if (elem_bytes == 1) KVC_RC(1); else if (elem_bytes == 2) KVC_RC(2); else KVC_RC(4);
#undef KVC_RC
  return check_launch("kvcompress_reshape_and_cache");
}
