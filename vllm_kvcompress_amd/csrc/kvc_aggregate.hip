// A2a/A2b/A2c metric aggregation and A7 reshape_and_cache for gfx950.
// All four are streaming, HBM-bound passes: wide coalesced loads, no LDS, float32
// arithmetic with explicit rounding (no FMA contraction) so results are bit-identical to
// the sequential float32 restatement in oracle/kvc_oracle.py.
#include "kvc_common.h"
#include "../../include/kvc_mi355x.h"
#include <hip/hip_fp16.h>

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
  int64_t s0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if constexpr (QPK == 4) {
    // four slots per thread and trip, a grid stride apart (every wave instruction stays one
    // contiguous 1 KiB / 256 B piece), all eight loads requested before the first is used: the
    // pass is a pure stream (24 B per slot) and what it lacks with one slot per trip is bytes in
    // flight: 5.1 -> 5.7 TB/s at 8.4 M slots, cold
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int U = 4;
    // (with the fused clear the pass is a mixed read / write stream, which runs at its own ceiling
    // -- 4.9 of the ~5.3 TB/s such streams reach here -- one slot per trip; unrolled it lost 8 %)
    for (; !clear_temp && s0 + (U - 1) * stride < num_slots; s0 += U * stride) {
      f32x4 t[U];
      float m[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        t[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(temp) + (s0 + u * stride));
        m[u] = metrics[s0 + u * stride];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        f32x4 v = t[u];
        if (use_l2) { v.x = __fmul_rn(v.x, v.x); v.y = __fmul_rn(v.y, v.y); v.z = __fmul_rn(v.z, v.z); v.w = __fmul_rn(v.w, v.w); }
        const float acc = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.0f, v.x), v.y), v.z), v.w);
        metrics[s0 + u * stride] = __fadd_rn(m[u], acc);
      }
    }
  }
  for (int64_t s = s0; s < num_slots; s += stride) {
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

// qpk == 4 without the fused clear -- the pass of every decode step -- in the two forms that measured best
// on stores of every size (tools/agg_bw.hip; 24 B per slot):
//   TILE   a workgroup takes contiguous tiles of 8 x 256 slots (eight adjacent rows in flight per
//          wave), tiles a grid stride apart, up to 16 Ki workgroups: 5.65 TB/s at 275 M slots (the
//          four-rows-a-grid-stride-apart loop above at 4 Ki workgroups: 5.37), 7.1 against 6.0 at 67 M,
//          where part of the store stays in the 256 MiB Infinity Cache between passes;
//   STREAM for a store several times that cache (>= 1 GiB of metrics: config 3): the grid-stride loop
//          with the metrics loaded and stored non-temporally -- nothing of them can stay anywhere --
//          at 16 Ki workgroups: 5.89 TB/s at 275 M slots (1.120 ms against 1.229); on smaller stores it
//          loses (5.1 against 6.0 at 67 M: what it pushes out is what the next pass wants).  With the
//          fused clear (40 B per slot; the zeros stored non-temporally as well) 2.069 ms against the
//          2.285 of one slot per trip at 4 Ki workgroups.
// The additions and their order are the loop's above.
__device__ __forceinline__ float row_sum4(float4 v, int use_l2) {
  if (use_l2) { v.x = __fmul_rn(v.x, v.x); v.y = __fmul_rn(v.y, v.y); v.z = __fmul_rn(v.z, v.z); v.w = __fmul_rn(v.w, v.w); }
  return __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.0f, v.x), v.y), v.z), v.w);
}
template <bool STREAM>
__global__ __launch_bounds__(256) void aggregate_decode_q4_kernel(float* __restrict__ metrics, float* __restrict__ temp,
                                                                  int64_t num_slots, int use_l2, int clear_temp) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int U = STREAM ? 4 : 8;
  f32x4* temp4 = reinterpret_cast<f32x4*>(temp);
  int64_t done;                                      // slots below this index are covered by the unrolled part
  if constexpr (STREAM) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t s0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    done = num_slots / (U * stride) * (U * stride);
    for (; s0 < done; s0 += U * stride) {
      f32x4 t[U];
      float m[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        t[u] = __builtin_nontemporal_load(temp4 + (s0 + u * stride));
        m[u] = __builtin_nontemporal_load(metrics + (s0 + u * stride));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        __builtin_nontemporal_store(__fadd_rn(m[u], row_sum4(make_float4(t[u].x, t[u].y, t[u].z, t[u].w), use_l2)),
                                    metrics + (s0 + u * stride));
        if (clear_temp) __builtin_nontemporal_store(f32x4{0.f, 0.f, 0.f, 0.f}, temp4 + (s0 + u * stride));
      }
    }
  } else {
    const int64_t tile = 256 * U, ntiles = num_slots / tile;
    done = ntiles * tile;
    for (int64_t ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
      const int64_t base = ti * tile + threadIdx.x;
      f32x4 t[U];
      float m[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        t[u] = __builtin_nontemporal_load(temp4 + (base + u * 256));
        m[u] = metrics[base + u * 256];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        metrics[base + u * 256] = __fadd_rn(m[u], row_sum4(make_float4(t[u].x, t[u].y, t[u].z, t[u].w), use_l2));
    }
  }
  for (int64_t s = done + (int64_t)blockIdx.x * 256 + threadIdx.x; s < num_slots; s += (int64_t)gridDim.x * 256) {
    metrics[s] = __fadd_rn(metrics[s], row_sum4(reinterpret_cast<const float4*>(temp)[s], use_l2));
    if (clear_temp) temp4[s] = f32x4{0.f, 0.f, 0.f, 0.f};
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

// The same sums, four adjacent key columns per thread (16 B loads) with the next batch of rows
// requested before the current one is added up.  Every column is still summed in ascending
// query order, so the result is bit-identical to the kernel above.  What decides the rate is how
// much of a query row one workgroup reads contiguously (rows are K x 4 B apart): on a 4 GiB tile
// (Hq 32, qb 1024, K 32768) 1 KiB per row and workgroup (the scalar kernel) 1.55 ms = 2.8 TB/s,
// 2 KiB 1.69 ms, 4 KiB 1.43 ms, 8 KiB 1.00 ms, 16 KiB 0.86 ms = 5.0 TB/s; streaming loads cost
// 10 %.  The host picks the largest workgroup that still gives one workgroup per CU.
// Needs K % 4 == 0 and 16 B aligned tile / workspace.
constexpr int EPI_ROWS = 4;
template <int THREADS>
__global__ __launch_bounds__(THREADS) void epilogue_colsum4_kernel(float* __restrict__ colsum,
                                                                   const float* __restrict__ probs,
                                                                   int Hq, int qb, int K, int q_offset,
                                                                   int buffer_len, int use_l2,
                                                                   int use_average) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int k = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int h = blockIdx.y;
  if (k >= K) return;
  const int diag = q_offset - buffer_len;
  int qmin[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { qmin[c] = k + c - diag; qmin[c] = qmin[c] < 0 ? 0 : qmin[c]; }
  const f32x4* col = reinterpret_cast<const f32x4*>(probs + ((int64_t)h * qb) * K + k);
  const int64_t rs = K / 4;                       // row stride in vectors
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  auto add_row = [&](const f32x4& t, int q) {
    const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // rows above a column's first counted query add nothing (+0 leaves the sum's bits alone)
      const float x = q >= qmin[c] ? (use_l2 ? __fmul_rn(v[c], v[c]) : v[c]) : 0.0f;
      acc[c] = __fadd_rn(acc[c], x);
    }
  };
  int q = qmin[0];
  f32x4 cur[EPI_ROWS], nxt[EPI_ROWS];
  auto load_rows = [&](f32x4 (&dst)[EPI_ROWS], int q0) {
#pragma unroll
    for (int u = 0; u < EPI_ROWS; ++u) dst[u] = col[(int64_t)(q0 + u) * rs];
  };
  auto add_rows = [&](const f32x4 (&src)[EPI_ROWS], int q0) {
#pragma unroll
    for (int u = 0; u < EPI_ROWS; ++u) add_row(src[u], q0 + u);
  };
  if (q + EPI_ROWS <= qb) {
    load_rows(cur, q);                            // cur = rows [q, q + R)
    // ping-pong between the two register sets (a copy from one to the other would wait for the
    // rows still in flight)
    while (q + 3 * EPI_ROWS <= qb) {
      load_rows(nxt, q + EPI_ROWS);
      add_rows(cur, q);
      load_rows(cur, q + 2 * EPI_ROWS);
      add_rows(nxt, q + EPI_ROWS);
      q += 2 * EPI_ROWS;
    }
    if (q + 2 * EPI_ROWS <= qb) {
      load_rows(nxt, q + EPI_ROWS);
      add_rows(cur, q);
      add_rows(nxt, q + EPI_ROWS);
      q += 2 * EPI_ROWS;
    } else {
      add_rows(cur, q);
      q += EPI_ROWS;
    }
  }
  for (; q < qb; ++q) add_row(col[(int64_t)q * rs], q);
  f32x4 o;
  float r[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
    r[c] = use_average ? __fmul_rn(acc[c], __fdiv_rn((float)(k + c + 1), (float)qb)) : acc[c];
  o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
  *reinterpret_cast<f32x4*>(colsum + (int64_t)h * K + k) = o;
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

// ------------------------------------------------------------------ A7 (block path)
// Prefill writes bs consecutive tokens of a head into one fresh block.  When the slots of
// tokens t0..t0+bs-1 are exactly one aligned block (s0 % bs == 0, s_i = s0 + i) the wave of
// the first token assembles the whole K and V block images (lane l holds the 16 B pieces
// l, l+64, ...) and streams them out 1 KiB per instruction instead of bs*hd scattered 2 B
// stores; every other (token, head) falls back to element-wise stores.
typedef uint32_t u32x4a __attribute__((ext_vector_type(4)));

template <int HD, int BS, int E>
__global__ __launch_bounds__(256) void reshape_and_cache_blocks_kernel(
    const uint8_t* __restrict__ key, const uint8_t* __restrict__ value,
    uint8_t* __restrict__ key_cache, uint8_t* __restrict__ value_cache,
    float* __restrict__ kv_metrics, const int64_t* __restrict__ slot_mapping,
    const float* __restrict__ bias, int64_t num_tokens, int num_heads, int64_t key_stride,
    int64_t value_stride) {
  constexpr int64_t BLOCK_BYTES = (int64_t)HD * BS * E;
  constexpr int NPL = (int)(BLOCK_BYTES / 16 / 64);
  constexpr int EP = 16 / E;                 // elements per 16 B piece
  constexpr int RB = BS * E, PR = RB / 16;   // V row bytes, pieces per V row
  constexpr int ROWB = HD * E;               // bytes of one token's head vector
  constexpr int PPR = ROWB / 16;             // 16 B pieces per token row
  __shared__ __attribute__((aligned(16))) uint8_t tile_s[4][BLOCK_BYTES];   // per wave: [BS tokens][HD*E]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t* tile = tile_s[wave];
  // lane <-> one (token, head) item; 64 consecutive items per wave
  const int64_t item = ((int64_t)blockIdx.x * 4 + wave) * 64 + lane;
  const bool live = item < num_tokens * num_heads;
  const int64_t t = live ? item / num_heads : 0;
  const int h = live ? (int)(item % num_heads) : 0;
  const int64_t slot = live ? slot_mapping[t * num_heads + h] : -1;
  // is the group of BS tokens this slot belongs to exactly one aligned block?
  bool aligned = false;
  if (slot >= 0) {
    const int o = (int)(slot % BS);
    const int64_t t0 = t - o;
    aligned = t0 >= 0 && t0 + BS <= num_tokens;
    for (int i = 0; aligned && i < BS; ++i) aligned = slot_mapping[(t0 + i) * num_heads + h] == slot - o + i;
  }
  unsigned long long starters = __ballot(aligned && slot % BS == 0);
  unsigned long long singles = __ballot(slot >= 0 && !aligned);
  // ---- whole blocks, one after the other, all 64 lanes on each -----------------------
  while (starters) {
    const int src_lane = __ffsll((long long)starters) - 1;
    starters &= starters - 1;
    const int64_t s0 = __shfl(slot, src_lane, 64);            // 64-bit shuffle (two halves)
    const int64_t it0 = ((int64_t)blockIdx.x * 4 + wave) * 64 + src_lane;
    const int64_t t0 = it0 / num_heads;
    const int hh = (int)(it0 % num_heads);
    const int64_t blk = s0 / BS;
    uint8_t* kd = key_cache + blk * BLOCK_BYTES;
    uint8_t* vd = value_cache + blk * BLOCK_BYTES;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {                            // K: piece p = r*BS + s  <- token s, chunk r
      const int p = i * 64 + lane, r = p / BS, s = p % BS;
      const u32x4a q = *reinterpret_cast<const u32x4a*>(key + ((t0 + s) * key_stride + (int64_t)hh * HD) * E + r * 16);
      __builtin_nontemporal_store(q, reinterpret_cast<u32x4a*>(kd + (int64_t)p * 16));
    }
    // V: stage the BS value rows (coalesced 16 B pieces), then every lane gathers the EP
    // elements of its output piece from LDS and streams the block image out
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int p = i * 64 + lane, s = p / PPR, c = p % PPR;
      *reinterpret_cast<u32x4a*>(tile + (int64_t)p * 16) =
          *reinterpret_cast<const u32x4a*>(value + ((t0 + s) * value_stride + (int64_t)hh * HD) * E + c * 16);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < NPL; ++i) {                            // piece p = row d, slots [pr*EP, +EP)
      const int p = i * 64 + lane, d = p / PR, pr = p % PR;
      uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < EP; ++j) {
        const uint8_t* src = tile + (pr * EP + j) * ROWB + d * E;
        uint32_t e;
        if constexpr (E == 1) e = *src;
        else if constexpr (E == 2) e = *reinterpret_cast<const uint16_t*>(src);
        else e = *reinterpret_cast<const uint32_t*>(src);
        w[j * E / 4] |= e << ((j * E % 4) * 8);
      }
      __builtin_nontemporal_store(u32x4a{w[0], w[1], w[2], w[3]}, reinterpret_cast<u32x4a*>(vd + (int64_t)p * 16));
    }
    if (lane < BS) kv_metrics[s0 + lane] = bias[hh];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                           // tile is reused by the next block
  }
  // ---- everything else: one (token, head) at a time, element-wise -----------------------
  while (singles) {
    const int src_lane = __ffsll((long long)singles) - 1;
    singles &= singles - 1;
    const int64_t s1 = __shfl(slot, src_lane, 64);
    const int64_t it1 = ((int64_t)blockIdx.x * 4 + wave) * 64 + src_lane;
    const int64_t t1 = it1 / num_heads;
    const int hh = (int)(it1 % num_heads);
    const int64_t blk = s1 / BS;
    const int o = (int)(s1 % BS);
    if (lane == 0) kv_metrics[s1] = bias[hh];
    for (int d = lane; d < HD; d += 64) {
      const uint8_t* vs = value + (t1 * value_stride + (int64_t)hh * HD + d) * E;
      uint8_t* vdp = value_cache + blk * BLOCK_BYTES + ((int64_t)d * BS + o) * E;
      if constexpr (E == 1) *vdp = *vs;
      else if constexpr (E == 2) *reinterpret_cast<uint16_t*>(vdp) = *reinterpret_cast<const uint16_t*>(vs);
      else *reinterpret_cast<uint32_t*>(vdp) = *reinterpret_cast<const uint32_t*>(vs);
    }
    for (int r = lane; r < PPR; r += 64)
      *reinterpret_cast<u32x4a*>(key_cache + blk * BLOCK_BYTES + ((int64_t)r * BS + o) * 16) =
          *reinterpret_cast<const u32x4a*>(key + (t1 * key_stride + (int64_t)hh * HD) * E + r * 16);
  }
}

// ------------------------------------------------------------------ A7 (slot-major blocks)
// KVC_LAYOUT_SLOT_MAJOR (include/kvc_mi355x.h): a (token, head)'s K and V rows go, unchanged, to the
// hd * e contiguous bytes of their slot -- one thread per 16 B piece of K and of V
__global__ __launch_bounds__(256) void reshape_and_cache_slots_kernel(
    const uint8_t* __restrict__ key, const uint8_t* __restrict__ value,
    uint8_t* __restrict__ key_cache, uint8_t* __restrict__ value_cache,
    float* __restrict__ kv_metrics, const int64_t* __restrict__ slot_mapping,
    const float* __restrict__ bias, int64_t items, int num_heads, int pieces, int64_t key_stride_bytes,
    int64_t value_stride_bytes, int aligned) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= items * pieces) return;
  const int64_t item = i / pieces;
  const int pc = (int)(i % pieces);
  const int64_t slot = slot_mapping[item];
  if (slot < 0) return;                                          // padding token
  const int64_t token = item / num_heads;
  const int head = (int)(item % num_heads);
  if (pc == 0) kv_metrics[slot] = bias[head];
  const int64_t sb = (int64_t)pieces * 16;
  const uint8_t* ks = key + token * key_stride_bytes + (int64_t)head * sb + pc * 16;
  const uint8_t* vs = value + token * value_stride_bytes + (int64_t)head * sb + pc * 16;
  uint8_t* kd = key_cache + slot * sb + pc * 16;
  uint8_t* vd = value_cache + slot * sb + pc * 16;
  if (aligned) {
    *reinterpret_cast<u32x4a*>(kd) = *reinterpret_cast<const u32x4a*>(ks);
    *reinterpret_cast<u32x4a*>(vd) = *reinterpret_cast<const u32x4a*>(vs);
  } else {
    for (int b = 0; b < 16; ++b) { kd[b] = ks[b]; vd[b] = vs[b]; }
  }
}

// ------------------------------------------------------------------ A7 (fp8 cache)
// fp8 = cvt(float(x) / scale) with round-to-nearest-even and saturation to the largest
// finite value, OCP e4m3fn / e5m2 -- what the reference computes with
// __nv_cvt_float_to_fp8(x / scale, __NV_SATFINITE, fp8_type)
// (csrc/quantization/fp8/nvidia/quant_utils.cuh:456-489).  Done in integer arithmetic so
// the result does not depend on the conversion mode bits of the hardware instruction.
template <int MBITS, int BIAS, int MAXCODE>
__device__ __forceinline__ uint32_t f32_to_fp8_satfinite(float f) {
  const uint32_t u = __float_as_uint(f);
  const uint32_t sign = (u >> 24) & 0x80u;
  const uint32_t a = u & 0x7FFFFFFFu;
  if (a > 0x7F800000u) return sign | 0x7Fu;                     // NaN
  constexpr int SHIFT = 23 - MBITS;
  constexpr uint32_t MIN_NORMAL = (uint32_t)(127 - BIAS + 1) << 23;
  uint32_t code;
  if (a < MIN_NORMAL) {
    // subnormal target: integer multiple of 2^(1-BIAS-MBITS), ties to even
    code = (uint32_t)__float2int_rn(__uint_as_float(a) * __uint_as_float((uint32_t)(127 + BIAS - 1 + MBITS) << 23));
  } else {
    const uint32_t r = a + ((a >> SHIFT) & 1u) + ((1u << (SHIFT - 1)) - 1u);   // RNE on the cut bits
    code = (r - ((uint32_t)(127 - BIAS) << 23)) >> SHIFT;
  }
  if (code > (uint32_t)MAXCODE || a >= 0x7F800000u) code = MAXCODE;            // saturate (also +-inf)
  return sign | code;
}

// SRC: 0 = fp16, 1 = bf16, 2 = fp32;  KIND: 0 = e4m3fn, 1 = e5m2
template <int SRC, int KIND>
__global__ __launch_bounds__(512) void reshape_and_cache_fp8_kernel(
    const uint8_t* __restrict__ key, const uint8_t* __restrict__ value,
    uint8_t* __restrict__ key_cache, uint8_t* __restrict__ value_cache,
    float* __restrict__ kv_metrics, const int64_t* __restrict__ slot_mapping,
    const float* __restrict__ bias, int num_heads, int head_size, int bs, int64_t key_stride,
    int64_t value_stride, float k_scale, float v_scale, int slot_major) {
  const int64_t token = blockIdx.x;
  const int n = num_heads * head_size;
  auto load = [&](const uint8_t* base, int64_t idx) -> float {
    if constexpr (SRC == 0) return __half2float(reinterpret_cast<const __half*>(base)[idx]);
    else if constexpr (SRC == 1) return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(base)[idx] << 16);
    else return reinterpret_cast<const float*>(base)[idx];
  };
  auto cvt = [&](float x, float scale) -> uint8_t {
    const float y = __fdiv_rn(x, scale);
    if constexpr (KIND == 0) return (uint8_t)f32_to_fp8_satfinite<3, 7, 0x7E>(y);
    else return (uint8_t)f32_to_fp8_satfinite<2, 15, 0x7B>(y);
  };
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int head = i / head_size, d = i % head_size;
    const int64_t slot = slot_mapping[token * num_heads + head];
    if (slot < 0) continue;
    if (d == 0) kv_metrics[slot] = bias[head];
    const int64_t blk = slot / bs;
    const int off = (int)(slot % bs);
    const int64_t block_bytes = (int64_t)head_size * bs;        // 1 byte per element
    if (slot_major) {                                           // KVC_LAYOUT_SLOT_MAJOR: [bs][hd] in both planes
      value_cache[slot * head_size + d] = cvt(load(value, token * value_stride + i), v_scale);
      key_cache[slot * head_size + d] = cvt(load(key, token * key_stride + i), k_scale);
      continue;
    }
    value_cache[blk * block_bytes + (int64_t)d * bs + off] = cvt(load(value, token * value_stride + i), v_scale);
    key_cache[blk * block_bytes + ((int64_t)(d / 16) * bs + off) * 16 + d % 16] = cvt(load(key, token * key_stride + i), k_scale);
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
  const bool big = num_slots >= (int64_t)1 << 28;                 // >= 1 GiB of metrics
  if (num_queries_per_kv == 4 && (big || (!clear_temp && num_slots >= (int64_t)1 << 20))) {
    const int64_t per_wg = big ? 256 * 4 : 256 * 8;
    const int64_t w2 = (num_slots + per_wg - 1) / per_wg;
    const unsigned g2 = (unsigned)(w2 < 16384 ? w2 : 16384);
    if (big)
      hipLaunchKernelGGL(aggregate_decode_q4_kernel<true>, dim3(g2), dim3(256), 0, s, metrics, temp_metrics, num_slots, use_l2, clear_temp);
    else
      hipLaunchKernelGGL(aggregate_decode_q4_kernel<false>, dim3(g2), dim3(256), 0, s, metrics, temp_metrics, num_slots, use_l2, 0);
  } else if (num_queries_per_kv == 4)
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

namespace kvc {
// shared with the fused collector (kvc_prefill_attn.hip): step 2 of the epilogue
int launch_epilogue_pool(float* out_kh, const float* colsum, int num_q_heads, int num_keys,
                         int use_maxpool, hipStream_t s) {
  const int64_t n = (int64_t)num_keys * num_q_heads;
  hipLaunchKernelGGL(epilogue_pool_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out_kh,
                     colsum, num_q_heads, num_keys, use_maxpool);
  return check_launch("prefill_metric_epilogue(pool)");
}
}  // namespace kvc

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
  if (num_keys % 4 == 0 && ((reinterpret_cast<uintptr_t>(probs_hqk) | reinterpret_cast<uintptr_t>(workspace)) & 15u) == 0) {
    const int cg = num_keys / 4;                        // column groups
    int threads = 1024;
    while (threads > 128 && (int64_t)((cg + threads - 1) / threads) * num_q_heads < 256) threads >>= 1;
    const dim3 grid((cg + threads - 1) / threads, num_q_heads);
#define KVC_EPI_LAUNCH(T)                                                                          \
    hipLaunchKernelGGL(epilogue_colsum4_kernel<T>, grid, dim3(T), 0, s, colsum, probs_hqk, num_q_heads, \
                       q_block, num_keys, q_offset, buffer_len, use_l2, use_average)
    if (threads == 1024) KVC_EPI_LAUNCH(1024);
    else if (threads == 512) KVC_EPI_LAUNCH(512);
    else if (threads == 256) KVC_EPI_LAUNCH(256);
    else KVC_EPI_LAUNCH(128);
#undef KVC_EPI_LAUNCH
  }
  else
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
  return kvc_reshape_and_cache_layout(key, value, key_cache, value_cache, kv_metrics, slot_mapping, kv_metric_head_bias,
                                      num_tokens, num_heads, head_size, block_size, elem_bytes, key_stride, value_stride,
                                      KVC_LAYOUT_REFERENCE, stream);
}

extern "C" int kvc_reshape_and_cache_layout(const void* key, const void* value, void* key_cache,
                                            void* value_cache, float* kv_metrics,
                                            const int64_t* slot_mapping, const float* kv_metric_head_bias,
                                            int64_t num_tokens, int32_t num_heads, int32_t head_size,
                                            int32_t block_size, int32_t elem_bytes, int64_t key_stride,
                                            int64_t value_stride, int32_t block_layout, kvc_stream_t stream) {
  using namespace kvc;
  if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4)
    return fail_invalid("Unsupported cache element size: " + std::to_string(elem_bytes));
  if (head_size % (16 / elem_bytes) != 0)
    return fail_invalid("Unsupported head size: " + std::to_string(head_size));
  if (block_layout != KVC_LAYOUT_REFERENCE && block_layout != KVC_LAYOUT_SLOT_MAJOR)
    return fail_invalid("Unsupported block layout: " + std::to_string(block_layout));
  if (num_tokens <= 0) return KVC_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t items = num_tokens * num_heads;
  if (block_layout == KVC_LAYOUT_SLOT_MAJOR) {
    const int pieces = head_size * elem_bytes / 16;
    const int aligned = (((uintptr_t)key | (uintptr_t)value | (uintptr_t)key_cache | (uintptr_t)value_cache) % 16 == 0) &&
                        ((key_stride * elem_bytes) % 16 == 0) && ((value_stride * elem_bytes) % 16 == 0);
    const int64_t threads = items * pieces;
    hipLaunchKernelGGL(reshape_and_cache_slots_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s,
                       (const uint8_t*)key, (const uint8_t*)value, (uint8_t*)key_cache, (uint8_t*)value_cache, kv_metrics,
                       slot_mapping, kv_metric_head_bias, items, num_heads, pieces, key_stride * elem_bytes,
                       value_stride * elem_bytes, aligned);
    return check_launch("kvcompress_reshape_and_cache (slot-major)");
  }
#define KVC_RCB(HD, BS, E)                                                                          \
  hipLaunchKernelGGL((reshape_and_cache_blocks_kernel<HD, BS, E>), dim3((unsigned)((items + 255) / 256)), \
                     dim3(256), 0, s, (const uint8_t*)key, (const uint8_t*)value, (uint8_t*)key_cache, \
                     (uint8_t*)value_cache, kv_metrics, slot_mapping, kv_metric_head_bias, num_tokens, \
                     num_heads, key_stride, value_stride)
  const bool key_ok = ((uintptr_t)key % 16 == 0) && ((key_stride * elem_bytes) % 16 == 0);
  if (key_ok && head_size == 128 && block_size == 16 && elem_bytes == 2) { KVC_RCB(128, 16, 2); return check_launch("kvcompress_reshape_and_cache"); }
  if (key_ok && head_size == 128 && block_size == 32 && elem_bytes == 2) { KVC_RCB(128, 32, 2); return check_launch("kvcompress_reshape_and_cache"); }
  if (key_ok && head_size == 64 && block_size == 16 && elem_bytes == 2) { KVC_RCB(64, 16, 2); return check_launch("kvcompress_reshape_and_cache"); }
  if (key_ok && head_size == 128 && block_size == 16 && elem_bytes == 4) { KVC_RCB(128, 16, 4); return check_launch("kvcompress_reshape_and_cache"); }
#undef KVC_RCB
  const int n = num_heads * head_size;
  const int threads = n < 512 ? ((n + 63) / 64 * 64) : 512;
#define KVC_RC(E)                                                                                   \
  hipLaunchKernelGGL(reshape_and_cache_kernel<E>, dim3((unsigned)num_tokens), dim3(threads), 0, s,   \
                     (const uint8_t*)key, (const uint8_t*)value, (uint8_t*)key_cache,               \
                     (uint8_t*)value_cache, kv_metrics, slot_mapping, kv_metric_head_bias,           \
                     num_heads, head_size, block_size, key_stride, value_stride)
  if (elem_bytes == 1) KVC_RC(1); else if (elem_bytes == 2) KVC_RC(2); else KVC_RC(4);
#undef KVC_RC
  return check_launch("kvcompress_reshape_and_cache");
}

extern "C" int kvc_reshape_and_cache_fp8(const void* key, const void* value, void* key_cache,
                                         void* value_cache, float* kv_metrics,
                                         const int64_t* slot_mapping, const float* kv_metric_head_bias,
                                         int64_t num_tokens, int32_t num_heads, int32_t head_size,
                                         int32_t block_size, int32_t src_dtype, int32_t fp8_kind,
                                         int64_t key_stride, int64_t value_stride, float k_scale,
                                         float v_scale, kvc_stream_t stream) {
  return kvc_reshape_and_cache_fp8_layout(key, value, key_cache, value_cache, kv_metrics, slot_mapping, kv_metric_head_bias,
                                          num_tokens, num_heads, head_size, block_size, src_dtype, fp8_kind, key_stride,
                                          value_stride, k_scale, v_scale, KVC_LAYOUT_REFERENCE, stream);
}

extern "C" int kvc_reshape_and_cache_fp8_layout(const void* key, const void* value, void* key_cache,
                                                void* value_cache, float* kv_metrics,
                                                const int64_t* slot_mapping, const float* kv_metric_head_bias,
                                                int64_t num_tokens, int32_t num_heads, int32_t head_size,
                                                int32_t block_size, int32_t src_dtype, int32_t fp8_kind,
                                                int64_t key_stride, int64_t value_stride, float k_scale,
                                                float v_scale, int32_t block_layout, kvc_stream_t stream) {
  using namespace kvc;
  if (block_layout != KVC_LAYOUT_REFERENCE && block_layout != KVC_LAYOUT_SLOT_MAJOR)
    return fail_invalid("Unsupported block layout: " + std::to_string(block_layout));
  const int slot_major = block_layout == KVC_LAYOUT_SLOT_MAJOR ? 1 : 0;
  if (src_dtype < 0 || src_dtype > 2) return fail_invalid("Unsupported input type of kv cache");
  if (fp8_kind < 0 || fp8_kind > 1) return fail_invalid("Unsupported data type of kv cache");
  if (head_size % 16 != 0) return fail_invalid("Unsupported head size: " + std::to_string(head_size));
  if (num_tokens <= 0) return KVC_OK;
  const int n = num_heads * head_size;
  const int threads = n < 512 ? ((n + 63) / 64 * 64) : 512;
  hipStream_t s = (hipStream_t)stream;
#define KVC_RC8(SRC, KIND)                                                                           \
  hipLaunchKernelGGL((reshape_and_cache_fp8_kernel<SRC, KIND>), dim3((unsigned)num_tokens),           \
                     dim3(threads), 0, s, (const uint8_t*)key, (const uint8_t*)value,                \
                     (uint8_t*)key_cache, (uint8_t*)value_cache, kv_metrics, slot_mapping,            \
                     kv_metric_head_bias, num_heads, head_size, block_size, key_stride, value_stride, \
                     k_scale, v_scale, slot_major)
  switch (src_dtype * 2 + fp8_kind) {
    case 0: KVC_RC8(0, 0); break;
    case 1: KVC_RC8(0, 1); break;
    case 2: KVC_RC8(1, 0); break;
    case 3: KVC_RC8(1, 1); break;
    case 4: KVC_RC8(2, 0); break;
    default: KVC_RC8(2, 1); break;
  }
#undef KVC_RC8
  return check_launch("kvcompress_reshape_and_cache(fp8)");
}
