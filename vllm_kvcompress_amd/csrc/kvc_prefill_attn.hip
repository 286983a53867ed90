// F4: fused prefill metric collector.
//   replaces  _naive_kvc_attention / _naive_kvc_masked_attention
//             (vllm/attention/backends/flash_attn.py:1120-1211)
// For one sequence and one block of observed queries the reference materialises
// P = softmax(scale * Q K^T + causal) as fp32 [Hq, qb, K] (8 GiB at Hq 32, qb 1024, K 65536)
// plus four temporaries of the same size, then reduces it to [Hq, K] column sums.  Here P
// never exists in memory: two matrix-core passes over S = Q K^T,
//   1. per query row: log-sum-exp over the causal keys (online max / sum),
//   2. per key column: sum over the block's queries of P (or P^2) under the metric-window
//      mask, P recomputed as exp(S - lse),
// followed by the existing pool + accumulate step.  Both passes hold one operand tile of
// 64 rows in registers per wave (queries in pass 1, keys in pass 2) and stream the other.
// v_mfma_f32_32x32x16 with M = keys, N = queries, K = head dims; the contraction order is
// free, so a lane's operand is one 16-byte piece of a token's head vector.
// Numerics follow the reference: logits are rounded to the input type before the softmax
// (its einsum returns fp16 / bf16, flash_attn.py:1189), everything after is fp32.
#include "kvc_common.h"
#include "../../include/kvc_mi355x.h"

#include <math.h>
#include <type_traits>
#ifndef KVC_PF_RB
#define KVC_PF_RB 1
#endif
#ifndef KVC_PF_RBQ
#define KVC_PF_RBQ 2
#endif

namespace kvc {

typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t pu32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mma32;
template <> struct Mma32<_Float16> {
  using V8 = pf16x8;
  static __device__ __forceinline__ f32x16 mma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma32<__bf16> {
  using V8 = pbf16x8;
  static __device__ __forceinline__ f32x16 mma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

struct PfArgs {
  const void* q;        // first query row of the block, [nq, Hq, hd], token stride q_stride
  const void* k;        // first key of the sequence, [K, Hk, hd], token stride k_stride
  float* lse;           // [Hq, lse_stride]  (log2 domain), entry [h, r] for query row r of `q`
  int64_t lse_stride;
  float* colsum;        // [Hq, K]
  int64_t q_stride, k_stride;
  float scale;
  int32_t Hq, Hk, nq, K, q_offset, buffer_len, use_l2, use_average;
};

constexpr float LOG2E = 1.4426950408889634f;

// 64 rows x HD of one head as MFMA operand fragments: frag[blk][s] holds, for row
// (32 blk + lane % 32), head dims 16 s + 8 (lane / 32) .. + 7
template <typename T, int HD>
__device__ __forceinline__ void load_rows(typename Mma32<T>::V8 (&frag)[2][HD / 16], const T* base,
                                          int64_t stride, int row0, int nrows, int lane) {
  using V8 = typename Mma32<T>::V8;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int r = row0 + 32 * b + (lane & 31);
    const bool ok = r < nrows;
    const T* p = base + (int64_t)(ok ? r : 0) * stride + 8 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < HD / 16; ++s) {
      pu32x4 raw = {0u, 0u, 0u, 0u};
      if (ok) raw = *reinterpret_cast<const pu32x4*>(p + 16 * s);
      frag[b][s] = __builtin_bit_cast(V8, raw);
    }
  }
}

// 32 resident rows x HD: frag[s] holds, for row (row0 + lane % 32), dims 16 s + 8 (lane / 32) .. + 7
template <typename T, int HD>
__device__ __forceinline__ void load_rows32(typename Mma32<T>::V8 (&frag)[HD / 16], const T* base,
                                            int64_t stride, int row0, int nrows, int lane) {
  using V8 = typename Mma32<T>::V8;
  const int r = row0 + (lane & 31);
  const bool ok = r < nrows;
  const T* p = base + (int64_t)(ok ? r : 0) * stride + 8 * (lane >> 5);
#pragma unroll
  for (int s = 0; s < HD / 16; ++s) {
    pu32x4 raw = {0u, 0u, 0u, 0u};
    if (ok) raw = *reinterpret_cast<const pu32x4*>(p + 16 * s);
    frag[s] = __builtin_bit_cast(V8, raw);
  }
}

// The streamed operand: 32 rows x HD per step, shared by the four waves of a workgroup.
// Reading it straight from global memory costs 64 different cache lines per wave load
// instruction (every lane another token row) and makes the kernel L1-line-rate bound
// (measured: 2.3x off the VALU bound).  So the workgroup fetches the tile once, coalesced
// (16 consecutive lanes = one 256-byte row), and parks it in LDS in operand order:
// 16-byte unit (p, row) at index p * 33 + row, p = dim / 8 -- the pad makes the transposing
// write 2-way instead of 16-way conflicted, the fragment reads are conflict free.
template <typename T, int HD, int THREADS = 256>
struct StreamTile {
  static constexpr int PP = HD / 8;                 // 16-byte pieces per row
  static constexpr int CH = 32 * PP / THREADS;      // chunks per thread
  static constexpr int UNITS = PP * 33;
  pu32x4 regs[CH];
  __device__ __forceinline__ void fetch(const T* base, int64_t stride, int row0, int nrows, int tid) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int c = tid + THREADS * j, row = c / PP, p = c % PP;
      regs[j] = pu32x4{0u, 0u, 0u, 0u};
      if (row0 + row < nrows)
        regs[j] = *reinterpret_cast<const pu32x4*>(base + (int64_t)(row0 + row) * stride + 8 * p);
    }
  }
  __device__ __forceinline__ void park(pu32x4* lds, int tid) const {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int c = tid + THREADS * j, row = c / PP, p = c % PP;
      lds[p * 33 + row] = regs[j];
    }
  }
  // fragment of k-step s for lane (row = lane % 32, half = lane / 32): dims 16 s + 8 half ..
  static __device__ __forceinline__ typename Mma32<T>::V8 frag(const pu32x4* lds, int s, int lane) {
    return __builtin_bit_cast(typename Mma32<T>::V8, lds[(2 * s + (lane >> 5)) * 33 + (lane & 31)]);
  }
};

// pass 1: lse[h, r] = log2 sum_k 2^(t[r,k]) over keys k <= q_offset + r, t = logit * log2(e).
// A wave keeps its 64 queries in registers; the workgroup streams the keys through LDS.
template <typename T, int HD, int RB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void prefill_lse_kernel(PfArgs a) {
  using M = Mma32<T>;
  using V8 = typename M::V8;
  using ST = StreamTile<T, HD>;
  constexpr int KS = HD / 16;
  __shared__ __attribute__((aligned(16))) pu32x4 tile[2][ST::UNITS];
  const int h = blockIdx.y, hk = h / (a.Hq / a.Hk);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  constexpr int WQ = 32 * RB;                               // queries per wave (RB resident 32-row blocks)
  const int rwg = blockIdx.x * (4 * WQ);                    // the workgroup's first query row
  const int r0 = rwg + w * WQ;                              // this wave's first query row
  const bool active = r0 < a.nq;
  const T* qb = reinterpret_cast<const T*>(a.q) + (int64_t)h * HD;
  const T* kb = reinterpret_cast<const T*>(a.k) + (int64_t)hk * HD;
  V8 bq[RB][KS];
#pragma unroll
  for (int nb = 0; nb < RB; ++nb) load_rows32<T, HD>(bq[nb], qb, a.q_stride, r0 + 32 * nb, a.nq, lane);
  const float sc = a.scale * LOG2E;
  float m[RB], l[RB];
#pragma unroll
  for (int nb = 0; nb < RB; ++nb) { m[nb] = -INFINITY; l[nb] = 0.0f; }
  const int col = lane & 31, half = lane >> 5;
  const int kend = min(a.K, a.q_offset + min(a.nq, r0 + WQ));          // this wave's causal limit
  const int kend_wg = min(a.K, a.q_offset + min(a.nq, rwg + 4 * WQ));   // the workgroup's
  ST st;
  st.fetch(kb, a.k_stride, 0, a.K, tid);
  st.park(tile[0], tid);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < kend_wg; k0 += 32, buf ^= 1) {
    const bool more = k0 + 32 < kend_wg;
    if (more) st.fetch(kb, a.k_stride, k0 + 32, a.K, tid);
    if (active && k0 < kend) {
      const pu32x4* tl = tile[buf];
      const bool diag = k0 + 31 > a.q_offset + r0 || k0 + 32 > a.K;   // wave-uniform: needs masking
      // two copies of the body so that the mask arithmetic exists only on the diagonal tiles
      auto body = [&](auto diag_tag) {
        constexpr bool DIAG = decltype(diag_tag)::value;
        V8 ak[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) ak[s] = ST::frag(tl, s, lane);
#pragma unroll
        for (int nb = 0; nb < RB; ++nb) {
          f32x16 c = f32x16{0.f};
#pragma unroll
          for (int s = 0; s < KS; ++s) c = M::mma(ak[s], bq[nb][s], c);
          const int pq = a.q_offset + r0 + 32 * nb + col;  // this lane's query position
          float tmax = -INFINITY;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            float t = (float)(T)c[i] * sc;                  // logits are rounded to T (:1189)
            if constexpr (DIAG) {
              const int key = k0 + (i & 3) + 8 * (i >> 2) + 4 * half;
              if (key > pq || key >= a.K) t = -INFINITY;
            }
            c[i] = t;
            tmax = fmaxf(tmax, t);
          }
          tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
          const float mn = fmaxf(m[nb], tmax);
          if (mn != -INFINITY) {
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += __builtin_amdgcn_exp2f(c[i] - mn);
            sum += __shfl_xor(sum, 32, 64);
            l[nb] = l[nb] * __builtin_amdgcn_exp2f(m[nb] - mn) + sum;
            m[nb] = mn;
          }
        }
      };
      if (diag) body(std::true_type{}); else body(std::false_type{});
    }
    if (more) st.park(tile[buf ^ 1], tid);
    __syncthreads();
  }
  if (active && half == 0) {
#pragma unroll
    for (int nb = 0; nb < RB; ++nb) {
      const int r = r0 + 32 * nb + col;
      if (r < a.nq) a.lse[(int64_t)h * a.lse_stride + r] = m[nb] + __builtin_amdgcn_logf(l[nb]);
    }
  }
}

// pass 2: colsum[h, k] = sum over the block's query rows r with k + buffer_len <= q_offset + r
// of P[r, k] (or its square), P = 2^(t - lse).  A wave keeps its 64 keys in registers; the
// workgroup streams the queries through LDS.
template <typename T, int HD, bool L2, int RB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void prefill_colsum_kernel(PfArgs a) {
  using M = Mma32<T>;
  using V8 = typename M::V8;
  using ST = StreamTile<T, HD>;
  constexpr int KS = HD / 16;
  __shared__ __attribute__((aligned(16))) pu32x4 tile[2][ST::UNITS];
  const int h = blockIdx.y, hk = h / (a.Hq / a.Hk);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  constexpr int WK = 32 * RB;                               // keys per wave (RB resident 32-row blocks)
  const int kwg = blockIdx.x * (4 * WK);                    // the workgroup's first key
  const int k0 = kwg + w * WK;                              // this wave's first key
  const bool active = k0 < a.K;
  const T* qb = reinterpret_cast<const T*>(a.q) + (int64_t)h * HD;
  const T* kb = reinterpret_cast<const T*>(a.k) + (int64_t)hk * HD;
  V8 ak[RB][KS];
#pragma unroll
  for (int mb = 0; mb < RB; ++mb) load_rows32<T, HD>(ak[mb], kb, a.k_stride, k0 + 32 * mb, a.K, lane);
  const float sc = a.scale * LOG2E;
  const int col = lane & 31, half = lane >> 5;
  f32x16 acc[RB];
#pragma unroll
  for (int mb = 0; mb < RB; ++mb) acc[mb] = f32x16{0.f};
  // first query row that can see the first key of the wave / workgroup through the window
  auto first_row = [&](int key) { const int r = key + a.buffer_len - a.q_offset; return r < 0 ? 0 : r / 32 * 32; };
  const int rs = first_row(k0), rs_wg = first_row(kwg);
  const float* lse_h = a.lse + (int64_t)h * a.lse_stride;
  ST st;
  if (rs_wg < a.nq) {
    st.fetch(qb, a.q_stride, rs_wg, a.nq, tid);
    st.park(tile[0], tid);
  }
  __syncthreads();
  int buf = 0;
  for (int r0 = rs_wg; r0 < a.nq; r0 += 32, buf ^= 1) {
    const bool more = r0 + 32 < a.nq;
    if (more) st.fetch(qb, a.q_stride, r0 + 32, a.nq, tid);
    if (active && r0 >= rs) {
      const pu32x4* tl = tile[buf];
      const int r = r0 + col;
      const float ls = r < a.nq ? lse_h[r] : 0.0f;
      // every (key, query) pair of the tile inside the window and inside the block?
      const bool edge = k0 + WK - 1 + a.buffer_len > a.q_offset + r0 || r0 + 32 > a.nq || k0 + WK > a.K;
      const int pq = a.q_offset + r;
      auto body = [&](auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        V8 bq[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) bq[s] = ST::frag(tl, s, lane);
#pragma unroll
        for (int mb = 0; mb < RB; ++mb) {
          f32x16 c = f32x16{0.f};
#pragma unroll
          for (int s = 0; s < KS; ++s) c = M::mma(ak[mb][s], bq[s], c);
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            float p = __builtin_amdgcn_exp2f(__builtin_fmaf((float)(T)c[i], sc, -ls));
            if constexpr (EDGE) {
              const int key = k0 + 32 * mb + (i & 3) + 8 * (i >> 2) + 4 * half;
              if (key + a.buffer_len > pq || r >= a.nq || key >= a.K) p = 0.0f;
            }
            if constexpr (L2) acc[mb][i] = __builtin_fmaf(p, p, acc[mb][i]);
            else acc[mb][i] += p;
          }
        }
      };
      if (edge) body(std::true_type{}); else body(std::false_type{});
    }
    if (more) st.park(tile[buf ^ 1], tid);
    __syncthreads();
  }
  if (!active) return;
  // sum over the 32 query columns held by the lanes of each half
#pragma unroll
  for (int mb = 0; mb < RB; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = acc[mb][i];
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) v += __shfl_xor(v, d, 64);
      acc[mb][i] = v;
    }
  // lane `col` of each half writes the keys (i = col % 16, mb = col / 16) of its half
  if (col < 16 * RB) {
    const int mb = col >> 4, i = col & 15;
    float v = 0.0f;
#pragma unroll
    for (int mm = 0; mm < RB; ++mm)
#pragma unroll
      for (int ii = 0; ii < 16; ++ii)
        if (mm == mb && ii == i) v = acc[mm][ii];
    const int key = k0 + 32 * mb + (i & 3) + 8 * (i >> 2) + 4 * half;
    if (key < a.K) {
      if (a.use_average) v = __fmul_rn(v, __fdiv_rn((float)(key + 1), (float)a.nq));   // :1196-1203
      a.colsum[(int64_t)h * a.K + key] = v;
    }
  }
}

// pass 1 once for ALL observed rows (it does not depend on the block structure and one query
// block alone cannot fill the chip), then per query block: pass 2 + pool/accumulate
template <typename T, int HD>
static int launch_prefill(PfArgs a, float* out_kh, int n_obs, int q_block, int use_maxpool, hipStream_t s) {
  const T* q0 = reinterpret_cast<const T*>(a.q);
  float* lse0 = a.lse;
  const int off0 = a.q_offset;
  a.nq = n_obs;
  a.lse_stride = n_obs;
  constexpr int RBQ = KVC_PF_RBQ;
  hipLaunchKernelGGL((prefill_lse_kernel<T, HD, RBQ>), dim3((n_obs + 128 * RBQ - 1) / (128 * RBQ), a.Hq),
                     dim3(256), 0, s, a);
  for (int l = 0; l < n_obs; l += q_block) {
    a.q = q0 + (int64_t)l * a.q_stride;
    a.lse = lse0 + l;
    a.nq = n_obs - l < q_block ? n_obs - l : q_block;
    a.q_offset = off0 + l;
    constexpr int RB = KVC_PF_RB;
    const dim3 grid((a.K + 128 * RB - 1) / (128 * RB), a.Hq);
    if (a.use_l2)
      hipLaunchKernelGGL((prefill_colsum_kernel<T, HD, true, RB>), grid, dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((prefill_colsum_kernel<T, HD, false, RB>), grid, dim3(256), 0, s, a);
    const int rc = launch_epilogue_pool(out_kh, a.colsum, a.Hq, a.K, use_maxpool, s);
    if (rc != KVC_OK) return rc;
  }
  return check_launch("prefill_metric_fused");
}

}  // namespace kvc

extern "C" size_t kvc_prefill_metric_fused_workspace_bytes(int32_t num_q_heads, int32_t num_observed,
                                                           int32_t num_keys) {
  return ((size_t)num_q_heads * (size_t)num_keys + (size_t)num_q_heads * (size_t)num_observed) * sizeof(float);
}

extern "C" int kvc_prefill_metric_fused(float* out_kh, const void* query, const void* key,
                                        int32_t num_q_heads, int32_t num_k_heads, int32_t head_size,
                                        int32_t num_observed, int32_t q_block, int32_t num_keys,
                                        int32_t q_offset, int32_t buffer_len, int64_t q_stride,
                                        int64_t k_stride, float scale, int32_t dtype, int32_t use_l2,
                                        int32_t use_average, int32_t use_maxpool, void* workspace,
                                        size_t workspace_bytes, kvc_stream_t stream) {
  using namespace kvc;
  if (num_q_heads <= 0 || num_keys <= 0 || num_observed <= 0) return KVC_OK;
  if (q_block <= 0) return fail_invalid("prefill_metric_fused: q_block must be positive");
  if (num_k_heads < 1 || num_q_heads % num_k_heads != 0)
    return fail_invalid("prefill_metric_fused: query heads must be a multiple of key heads");
  if (dtype != 0 && dtype != 1) return fail_invalid("Unsupported data type of query");
  if (workspace_bytes < kvc_prefill_metric_fused_workspace_bytes(num_q_heads, num_observed, num_keys))
    return fail_invalid("prefill_metric_fused: workspace too small");
  if ((reinterpret_cast<uintptr_t>(query) & 15) || (reinterpret_cast<uintptr_t>(key) & 15) ||
      (q_stride % 8) || (k_stride % 8))
    return fail_invalid("prefill_metric_fused: query / key rows must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  PfArgs a;
  a.q = query; a.k = key;
  a.colsum = reinterpret_cast<float*>(workspace);
  a.lse = a.colsum + (size_t)num_q_heads * num_keys;
  a.lse_stride = num_observed;
  a.q_stride = q_stride; a.k_stride = k_stride; a.scale = scale;
  a.Hq = num_q_heads; a.Hk = num_k_heads; a.nq = num_observed; a.K = num_keys;
  a.q_offset = q_offset; a.buffer_len = buffer_len; a.use_l2 = use_l2; a.use_average = use_average;
  if (head_size == 128)
    return dtype == 0 ? launch_prefill<_Float16, 128>(a, out_kh, num_observed, q_block, use_maxpool, s)
                      : launch_prefill<__bf16, 128>(a, out_kh, num_observed, q_block, use_maxpool, s);
  if (head_size == 64)
    return dtype == 0 ? launch_prefill<_Float16, 64>(a, out_kh, num_observed, q_block, use_maxpool, s)
                      : launch_prefill<__bf16, 64>(a, out_kh, num_observed, q_block, use_maxpool, s);
  return fail_invalid("Unsupported head size: " + std::to_string(head_size));
}
