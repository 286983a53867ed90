// F3: single-query ("decode") attention over the per-head paged cache with
// the per-key softmax weight written out as the KV-Compress metric.
//   replaces  torch.ops._C.kvcompress_paged_attention_v1 / _v2
//             (csrc/attention/kvcompress_attention_kernels.cu:97-455, 532-651, 686-1056)
//
// MI355X design (not the reference's thread-group dot products):
//  * one workgroup = one 512-token partition of ONE (sequence, KV head) and ALL the query heads
//    that share that KV head (GQA): K and V are read from HBM once per group of up to 16 query
//    heads, not once per query head;
//  * four waves, 128 tokens each.  QK^T runs on the matrix cores as
//    v_mfma_f32_16x16x32 with the 16 TOKENS of half a block as the M dimension, the (<= 16)
//    query heads as N and 32 head dims per instruction as K: a lane's A operand is exactly one
//    16-byte piece of the K cache ([hd/8][bs][8] layout: 8 dims of one token), so the wave
//    reads 1 KiB contiguous per instruction and nothing is transposed;
//  * softmax statistics per partition (max, sum) via two tiny LDS exchanges;
//  * P.V on the matrix cores too: M = 16 head dims, N = query heads, K = 32 tokens; a lane's
//    A operand is one 16-byte piece of the V cache ([hd][bs] layout: 8 tokens of one dim);
//    P comes back from LDS in the B-operand layout;
//  * partition results are combined by a small second kernel (same maths as the reference's
//    v2 reduce); heads that fit one partition are finished by the first kernel.
//  * KVC_LAYOUT_SLOT_MAJOR blocks (K [bs][hd], V [bs][hd]: include/kvc_mi355x.h) run through the same two MFMA products;
//    only the way the operands reach the lanes differs -- K tiles are read as the contiguous lines they are and parked in
//    a wave-private LDS tile (KSlots), the P.V operand is assembled inside the lane from whole token rows (PvSlots).
// The kernel is HBM-bound: 2*hd*e bytes per cached token and KV head.
#pragma once
#include "kvc_common.h"
#include "kvc_harvest_layout.h"
#include "../../include/kvc_mi355x.h"

#include <math.h>
#include <algorithm>
#include <type_traits>

#ifndef KVC_WHOLE_ATTR
#define KVC_WHOLE_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
#ifndef KVC_PF
#define KVC_PF 3
#endif
// experiment builds only (tools/, timing): 1 = slot-major K addressing with the reference P.V, 2 = reference K
// addressing with the slot-major P.V -- which half of the layout costs what.  Results are wrong for 1 and 2.
#ifndef KVC_SM_EXP
#define KVC_SM_EXP 0
#endif
// K and V are streamed exactly once per call: non-temporal loads (measured +8 % at batch 256)
#define KVC_LD(p) __builtin_nontemporal_load(p)

namespace kvc {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t au32x4 __attribute__((ext_vector_type(4)));

constexpr int ATT_PART = 512;            // tokens per workgroup (= the reference's partition)
constexpr int ATT_WAVES = 4;
constexpr int ATT_CHUNK = ATT_PART / ATT_WAVES;   // tokens per wave
constexpr int ATT_NSUB = ATT_CHUNK / 16;          // 16-token MFMA row tiles per wave
constexpr int ATT_NQ = 16;               // query heads per workgroup (MFMA N)

template <typename T> struct Mma;
template <> struct Mma<_Float16> {
  using V8 = f16x8;
  static __device__ __forceinline__ f32x4 mma(V8 a, V8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<__bf16> {
  using V8 = bf16x8;
  static __device__ __forceinline__ f32x4 mma(V8 a, V8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

// 8 consecutive cache elements -> MFMA operand, in two steps so that the (non-temporal) load
// can be issued far ahead of the conversion.  KVD 0: the cache holds T (16-byte load);
// KVD 1 / 2: OCP fp8 e4m3fn / e5m2 bytes (8-byte load), dequantised like the reference's
// fp8::scaled_convert (csrc/attention/kvcompress_attention_kernels.cu:229-236, 369-377):
// T(float(fp8) * scale); KVD 3 / 4: the same formats with scale == 1 (the engine default),
// where every fp8 value is exact in fp16 and e5m2 is simply the top byte of an fp16.
// K vectors of an fp8 cache hold x = 16 elements.
typedef uint32_t au32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T, int KVD>
struct KvFrag {
  using V8 = typename Mma<T>::V8;
  static constexpr int X = KVD == 0 ? 8 : 16;            // elements per K vector
  static constexpr bool E5M2 = KVD == 2 || KVD == 4;
  static constexpr bool UNIT = KVD == 3 || KVD == 4;
  struct Raw16 { au32x4 v; };
  struct Raw8 { au32x2 v; };
  using Raw = typename std::conditional<KVD == 0, Raw16, Raw8>::type;
  static __device__ __forceinline__ Raw zero() { Raw r; r.v = 0; return r; }
  static __device__ __forceinline__ Raw load(const void* base, int64_t elem) {
    Raw r;
    if constexpr (KVD == 0)
      r.v = KVC_LD(reinterpret_cast<const au32x4*>(reinterpret_cast<const T*>(base) + elem));
    else
      r.v = KVC_LD(reinterpret_cast<const au32x2*>(reinterpret_cast<const uint8_t*>(base) + elem));
    return r;
  }
  // blocks of fewer than 8 tokens (the reference instantiates block size 1,
  // kvcompress_attention_kernels.cu:797): the 8 tokens of a V fragment sit in different blocks,
  // one 2-byte element each.  pe[e] = element offset of token e's block (+ its slot), < 0: masked
  static __device__ __forceinline__ Raw gather(const void* base, const int64_t (&pe)[8], int64_t off) {
    static_assert(KVD == 0, "small blocks: unquantised caches only");
    const uint16_t* b = reinterpret_cast<const uint16_t*>(base);
    uint32_t h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = pe[e] >= 0 ? (uint32_t)b[pe[e] + off] : 0u;
    Raw r;
    r.v = au32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
    return r;
  }
  static __device__ __forceinline__ V8 convert(const Raw& r, float scale) {
    if constexpr (KVD == 0) {
      return __builtin_bit_cast(V8, r.v);
    } else if constexpr (KVD == 3 && __is_same(T, _Float16)) {
      // gfx950 converts two e4m3 bytes straight to packed fp16 (exact; the scale operand only
      // contributes its exponent, so this is the unit-scale path)
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      const h2 a0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(r.v[0], 1.0f, false);
      const h2 a1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(r.v[0], 1.0f, true);
      const h2 a2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(r.v[1], 1.0f, false);
      const h2 a3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(r.v[1], 1.0f, true);
      V8 v;
      v[0] = a0[0]; v[1] = a0[1]; v[2] = a1[0]; v[3] = a1[1];
      v[4] = a2[0]; v[5] = a2[1]; v[6] = a3[0]; v[7] = a3[1];
      return v;
    } else if constexpr (KVD == 4 && __is_same(T, _Float16)) {
      au32x4 o;
      o[0] = ((r.v[0] & 0xFFu) << 8) | ((r.v[0] & 0xFF00u) << 16);
      o[1] = ((r.v[0] >> 8) & 0xFF00u) | (r.v[0] & 0xFF000000u);
      o[2] = ((r.v[1] & 0xFFu) << 8) | ((r.v[1] & 0xFF00u) << 16);
      o[3] = ((r.v[1] >> 8) & 0xFF00u) | (r.v[1] & 0xFF000000u);
      return __builtin_bit_cast(V8, o);
    } else {
      f32x2 f[4];
      if constexpr (!E5M2) {
        f[0] = __builtin_amdgcn_cvt_pk_f32_fp8(r.v[0], false); f[1] = __builtin_amdgcn_cvt_pk_f32_fp8(r.v[0], true);
        f[2] = __builtin_amdgcn_cvt_pk_f32_fp8(r.v[1], false); f[3] = __builtin_amdgcn_cvt_pk_f32_fp8(r.v[1], true);
      } else {
        f[0] = __builtin_amdgcn_cvt_pk_f32_bf8(r.v[0], false); f[1] = __builtin_amdgcn_cvt_pk_f32_bf8(r.v[0], true);
        f[2] = __builtin_amdgcn_cvt_pk_f32_bf8(r.v[1], false); f[3] = __builtin_amdgcn_cvt_pk_f32_bf8(r.v[1], true);
      }
      V8 v;
      if constexpr (UNIT && __is_same(T, _Float16)) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const h2 h = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(f[e][0], f[e][1]));   // exact
          v[2 * e] = h[0]; v[2 * e + 1] = h[1];
        }
      } else if constexpr (UNIT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] = (T)f[e][0]; v[2 * e + 1] = (T)f[e][1]; }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] = (T)(f[e][0] * scale); v[2 * e + 1] = (T)(f[e][1] * scale); }
      }
      return v;
    }
  }
};

// Harvest in the fused-metric epilogue (the decode step without a sweep of the metric store; the reference author's
// to-do, vllm/kvcompress/README.md:32, 49, kernel side csrc/attention/kvcompress_attention_kernels.cu:297-313):
// where the epilogue folds a key's weights into metrics[slot] it has the new sum in a register and the key's
// position at hand -- a key of a sequence of the NEXT compression batch whose sum lies below that sequence's pivot
// (left in the harvest buffer by the previous schedule_evictions) is appended to its head's candidate list there,
// exactly as kvc_schedule_harvest.h's aggregation pass does in its position-lazy form.  After the L launches of a
// decode step (one per layer) the lists hold every evictable key below the pivots, and the schedule call runs on
// them: no aggregate_decode, no collecting pass.  Exactness never depends on the pivots: lists that fall short
// raise the schedule's flag (kvc_schedule_small.h).  A head whose metric window does not reach the eviction bound
// (kv_metric_buffer_len too large for the protected window) cannot list everything: its count is pushed over the
// record length, which raises the same flag.
struct AttnHarvest {
  const uint32_t* pivot;          // [B]
  uint32_t* cnt;                  // [G] candidates per head, G = B * L * Hkv, g = (b * L + l) * Hkv + h
  uint32_t* def;                  // [G] masked / non-finite slots of the head's blocks (what the full collecting pass counts:
                                  //     the reference's batch > 1 rule needs every head's count of evictable keys)
  unsigned long long* lists;      // [G, KREC] (key << 32 | physical slot)
  uint32_t* claimed;              // [CLAIM_SHARDS x 32] logical blocks the epilogues walked
  int32_t* seen_ctx;              // [G] the context length each head's list was made with
  const int32_t* seq_slot;        // [S] this call's sequence -> position in the compression batch, or -1
  const int32_t* seq_positions;   // [B] the schedule call's (position of the token sampled last, not cached yet)
  const int32_t* num_protected;   // [B]
  int32_t layer, num_layers, num_sinks;
};
struct HarvestCtx { uint32_t pivot; int g; int bound; };   // g < 0: this head is not harvested; pivot 0: nothing of it is listed

struct AttnArgs {
  void* out;                      // [S, Hq, hd] T
  float* kv_metric_out;           // [NB, bs, qpk]
  float* exp_sums;                // [S, Hq, max_parts]
  float* max_logits;              // [S, Hq, max_parts]
  void* tmp_out;                  // [S, Hq, max_parts, hd] T
  float* tmp_kv_metric_out;       // [NB, bs, qpk]
  float* fused_metrics;           // [NB, bs] or null: metrics[slot] += sum_q p^2 (or p) instead of kv_metric_out
  const void* q;                  // [S, Hq, hd] T, seq stride q_stride
  const void* k_cache;            // [NB, hd/8, bs, 8] T
  const void* v_cache;            // [NB, hd, bs] T
  const int32_t* block_tables;    // [S, Hkv, max_blocks]
  const int32_t* context_lens;    // [S, Hkv]
  const int32_t* kv_position;     // [NB, bs]
  const int32_t* last_position;   // [S]
  const int32_t* kv_metric_buffer_len;   // [S]
  const float* alibi_slopes;      // [Hq] or null
  int64_t q_stride, kv_block_stride;
  float scale, k_scale, v_scale;
  int32_t num_heads, num_kv_heads, max_blocks, max_parts, record, max_ctx, use_l2, schedule;
  int32_t layout;                 // KVC_LAYOUT_REFERENCE / KVC_LAYOUT_SLOT_MAJOR
  AttnHarvest hv;                 // hv.cnt == nullptr: no harvest
};

// ---------------------------------------------------------------- KVC_LAYOUT_SLOT_MAJOR: P.V
// In slot-major blocks (K [bs][hd], V [bs][hd]) a token's K row is still 16-byte pieces of 8 dims -- QK^T keeps its
// MFMA form with other addresses.  V is the other way round: the 16x16x32 A operand wants 8 TOKENS of one dim in a
// lane, a slot-major piece is consecutive DIMS of one token.  The operand is therefore assembled inside the lane:
// lane (c, g) owns EPL = hd / 16 consecutive dims (piece c of a token's row) and requests that piece of ITS 8 tokens
// t0 + 8 g .. + 7 -- an 8 x EPL matrix [token][dim] in its own registers, every wave instruction reading the whole
// hd * e contiguous bytes of four tokens.  MFMA e of the group takes column e of that matrix (8 tokens of dim
// EPL c + e: one v_perm_b32 per operand dword, nothing crosses lanes, nothing goes through the LDS) against the same P
// operand as the reference layout: accumulator e holds out(dim EPL (4 g + j) + e, query c) in element j.  As many
// MFMAs and bytes in flight as the reference layout's P.V; the dims are merely numbered differently, which only the
// final store sees.
template <typename T, int HD, int KVD>
struct PvSlots {
  using KF = KvFrag<T, KVD>;
  using V8 = typename Mma<T>::V8;
  static constexpr int EPL = HD / 16;                     // dims per lane (= MFMAs per 32-token group = output tiles)
  static constexpr int NB = EPL * (KVD == 0 ? 2 : 1);     // bytes per token and lane: 4, 8, 16, 32
  static constexpr int ND = NB / 4;                       // raw dwords
  static constexpr int NH = EPL / 2;                      // dwords of T pairs
  static_assert(HD == 64 || HD == 128 || HD == 256, "slot-major attention: head sizes 64, 128, 256");
  struct Group { uint32_t r[8][ND]; };
  // requests the lane's piece of tokens t0 + 8 g .. + 7 (t0 a multiple of 32: one block); tokens past the context: 0
  static __device__ __forceinline__ void load(Group& G, const AttnArgs& a, const int32_t* bt, int t0, int ctx, int c, int g,
                                              int64_t blk_stride, int bs) {
    const int tok = t0 + 8 * g;
    const int64_t phys = tok < ctx ? bt[tok / bs] : 0;
    const uint8_t* base = reinterpret_cast<const uint8_t*>(a.v_cache) +
                          ((phys * blk_stride + (int64_t)(tok % bs) * HD) * (KVD == 0 ? 2 : 1) + (int64_t)c * NB);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint8_t* p = base + (int64_t)i * HD * (KVD == 0 ? 2 : 1);
      const bool live = tok + i < ctx;
      if constexpr (ND == 1) {
        G.r[i][0] = live ? KVC_LD(reinterpret_cast<const uint32_t*>(p)) : 0u;
      } else if constexpr (ND == 2) {
        const au32x2 v = live ? KVC_LD(reinterpret_cast<const au32x2*>(p)) : au32x2{0u, 0u};
        G.r[i][0] = v[0]; G.r[i][1] = v[1];
      } else {
#pragma unroll
        for (int k = 0; k < ND / 4; ++k) {
          const au32x4 v = live ? KVC_LD(reinterpret_cast<const au32x4*>(p) + k) : au32x4{0u, 0u, 0u, 0u};
#pragma unroll
          for (int d = 0; d < 4; ++d) G.r[i][4 * k + d] = v[d];
        }
      }
    }
  }
  // O[e] += V^T(dims EPL c' + e of the 16 lanes c', 32 tokens) . P(32 tokens, 16 queries)
  static __device__ __forceinline__ void mma(const Group& G, V8 pb, f32x4 (&O)[EPL], float v_scale) {
    uint32_t h[8][NH];                                    // [token][pair of dims] as T
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (KVD == 0) {
#pragma unroll
        for (int d = 0; d < NH; ++d) h[i][d] = G.r[i][d];
      } else if constexpr (EPL == 4) {
        typename KF::Raw raw; raw.v = au32x2{G.r[i][0], 0u};
        const au32x4 v = __builtin_bit_cast(au32x4, KF::convert(raw, v_scale));
        h[i][0] = v[0]; h[i][1] = v[1];
      } else {
#pragma unroll
        for (int k = 0; k < EPL / 8; ++k) {
          typename KF::Raw raw; raw.v = au32x2{G.r[i][2 * k], G.r[i][2 * k + 1]};
          const au32x4 v = __builtin_bit_cast(au32x4, KF::convert(raw, v_scale));
#pragma unroll
          for (int d = 0; d < 4; ++d) h[i][4 * k + d] = v[d];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      au32x4 av;
#pragma unroll
      for (int tp = 0; tp < 4; ++tp)                      // tokens 2 tp (low half), 2 tp + 1 (high half) of dim e
        av[tp] = __builtin_amdgcn_perm(h[2 * tp + 1][e / 2], h[2 * tp][e / 2], (e & 1) ? 0x07060302u : 0x05040100u);
      O[e] = Mma<T>::mma(__builtin_bit_cast(V8, av), pb, O[e]);
    }
  }
  // row: the [hd] fp32 output row of query c; lane (c, g) holds dims EPL (4 g + j) + e in O[e][j]
  static __device__ __forceinline__ void store(float* row, const f32x4 (&O)[EPL], int g) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e4 = 0; e4 < EPL / 4; ++e4)
        *reinterpret_cast<f32x4*>(row + EPL * (4 * g + j) + 4 * e4) =
            f32x4{O[4 * e4][j], O[4 * e4 + 1][j], O[4 * e4 + 2][j], O[4 * e4 + 3][j]};
  }
};
// ---------------------------------------------------------------- KVC_LAYOUT_SLOT_MAJOR: K
// The QK^T operand is the same 8 dims of one token as in the reference layout, but the MFMA wants 16 TOKENS in 16
// consecutive lanes and slot-major tokens are a whole row (hd * e bytes) apart: read as operands, four neighbouring
// lanes ask for four different rows (16 bytes each) where the reference layout has them side by side -- measured
// +5..10 % (fp16) / +17 % (fp8) on the whole kernel.  The wave therefore reads a 16-token tile the way it lies in
// memory -- per instruction 8 tokens x 128 contiguous bytes (whole cache lines, neighbouring lanes on neighbouring
// bytes) -- parks 128 bytes per token at a time ("phase": 64 fp16 dims, 128 fp8 dims) in its own 2 KiB of LDS and
// takes the operands from there.  The 16-byte units of a parked row are XOR-swizzled by the token so that the lane
// groups of ds_read_b128 / ds_read_b64 / ds_write_b128 each cover all banks once.  Wave-private: no barrier, the LDS
// executes a wave's instructions in order.
constexpr int KSLOTS_STAGE = 2048;                        // LDS bytes per wave
template <typename T, int HD, int KVD>
struct KSlots {
  using KF = KvFrag<T, KVD>;
  static constexpr int ES = KVD == 0 ? 2 : 1;
  static constexpr int RB = HD * ES;                      // bytes per token row: 64 .. 512
  static constexpr int PB = RB < 128 ? RB : 128;          // bytes per token and phase
  static constexpr int NPH = RB / PB;                     // phases per tile
  static constexpr int NI = 16 * PB / 1024;               // wave loads per phase: 2 (1 for 64-byte rows)
  static constexpr int U = PB / 16;                       // 16-byte units per parked row: 8 (4)
  static constexpr int R = 256 / PB;                      // parked rows per 256-byte bank row: 2 (4)
  static constexpr int SPP = PB / (32 * ES);              // MFMA k-steps per phase
  static_assert(HD == 64 || HD == 128 || HD == 256, "slot-major attention: head sizes 64, 128, 256");
  static_assert(16 * PB <= KSLOTS_STAGE && NPH * SPP == HD / 32, "KSlots");
  static __device__ __forceinline__ int swz(int tok) { return (tok / R) % U; }
  struct Raw { au32x4 v[NPH][NI]; };
  // tile: element offset of the tile's first token row in the K cache
  static __device__ __forceinline__ void load(Raw& r, const void* k_cache, int64_t tile, int lane) {
    const uint8_t* base = reinterpret_cast<const uint8_t*>(k_cache) + tile * ES;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const int u = 64 * k + lane, tok = u / U, j = u % U;
        r.v[ph][k] = KVC_LD(reinterpret_cast<const au32x4*>(base + tok * RB + ph * PB + j * 16));
      }
  }
  static __device__ __forceinline__ void stage(const Raw& r, int ph, uint8_t* kst, int lane) {
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int u = 64 * k + lane, tok = u / U, j = u % U;
      *reinterpret_cast<au32x4*>(kst + tok * PB + ((j ^ swz(tok)) << 4)) = r.v[ph][k];
    }
  }
  // dims 32 (ph SPP + sp) + 8 g .. + 7 of token c, out of the parked phase ph
  static __device__ __forceinline__ typename KF::Raw frag(const uint8_t* kst, int sp, int c, int g) {
    const int off = (32 * sp + 8 * g) * ES;
    const uint8_t* p = kst + c * PB + (((off >> 4) ^ swz(c)) << 4) + (off & 15);
    typename KF::Raw r;
    if constexpr (KVD == 0) r.v = *reinterpret_cast<const au32x4*>(p);
    else r.v = *reinterpret_cast<const au32x2*>(p);
    return r;
  }
  // the tile's MFMA operands kf[HD / 32], through the wave's LDS tile
  template <int KS>
  static __device__ __forceinline__ void operands(const Raw& r, uint8_t* kst, typename KF::Raw (&kf)[KS], int lane) {
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      stage(r, ph, kst, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // (read by other lanes of this wave)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int sp = 0; sp < SPP; ++sp) kf[ph * SPP + sp] = frag(kst, sp, c, g);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // (and overwritten by the next phase)
      __builtin_amdgcn_wave_barrier();
    }
  }
};
template <int HD, int BS>
constexpr bool slot_major_shape() { return (HD == 64 || HD == 128 || HD == 256) && (BS == 16 || BS == 32); }

// once per (sequence, KV head) workgroup; `first`: one thread of the ONE workgroup that accounts for the head
__device__ __forceinline__ HarvestCtx harvest_ctx(const AttnArgs& a, int seq, int hk, int max_pos, int ctx, int bs, bool first) {
  HarvestCtx h{0u, -1, 0};
  if (a.hv.cnt == nullptr || a.fused_metrics == nullptr) return h;
  const int i = a.hv.seq_slot[seq];
  if (i < 0) return h;
  h.g = (i * a.hv.num_layers + a.hv.layer) * a.num_kv_heads + hk;
  h.pivot = a.hv.pivot[i];
  h.bound = a.hv.seq_positions[i] - a.hv.num_protected[i];
  if (first) {
    atomicAdd(&a.hv.claimed[((unsigned)(seq * a.num_kv_heads + hk) % (unsigned)CLAIM_SHARDS) * 32], (uint32_t)((ctx + bs - 1) / bs));
    if (max_pos < h.bound) atomicAdd(&a.hv.cnt[h.g], (uint32_t)KREC + 1u);
    a.hv.seen_ctx[h.g] = ctx;
    const uint32_t tail = (uint32_t)((ctx + bs - 1) / bs * bs - ctx);     // the last block's slots behind the context: masked
    if (tail) atomicAdd(&a.hv.def[h.g], tail);
  }
  return h;
}
// a head with nothing cached returns before harvest_ctx: its (empty) list still says which context length it was made
// with -- the schedule call compares seen_ctx with its own context_lens and would otherwise redo every step's call
__device__ __forceinline__ void harvest_empty_head(const AttnArgs& a, int seq, int hk) {
  if (a.hv.cnt == nullptr || a.fused_metrics == nullptr || !a.record) return;
  const int i = a.hv.seq_slot[seq];
  if (i >= 0) a.hv.seen_ctx[(i * a.hv.num_layers + a.hv.layer) * a.num_kv_heads + hk] = 0;
}
// a key inside the metric window with its new sum: listed if it is evictable and below the pivot.  Returns 1 if the slot
// is masked or its key not finite -- the head's deficit, as the schedule's full collecting pass counts it
__device__ __forceinline__ uint32_t harvest_key(const AttnArgs& a, const HarvestCtx& h, int64_t slot, float mn, int pos) {
  const uint32_t key = float_to_key(mn);
  const bool in_range = pos <= h.bound && pos >= a.hv.num_sinks;          // (metrics.py:539-544)
  if (in_range && key < h.pivot) {
    const uint32_t at = atomicAdd(&a.hv.cnt[h.g], 1u);
    if (at < (uint32_t)KREC) a.hv.lists[(int64_t)h.g * KREC + at] = ((unsigned long long)key << 32) | (uint32_t)slot;
  }
  return (!in_range || key >= KEY_INF) ? 1u : 0u;
}
// the same with the candidates staged in LDS (the single-pass kernel: one workgroup owns the head, so ONE returning
// global atomic per workgroup reserves the list entries of all its candidates instead of one round trip per wave and
// step -- measured 1 % of the kernel); entries beyond the staging area go straight to the list
__device__ __forceinline__ uint32_t harvest_key_staged(const AttnArgs& a, const HarvestCtx& h, int64_t slot, float mn, int pos,
                                                       unsigned long long* cand, uint32_t* cand_n, uint32_t cand_cap) {
  const uint32_t key = float_to_key(mn);
  const bool in_range = pos <= h.bound && pos >= a.hv.num_sinks;
  if (in_range && key < h.pivot) {
    const unsigned long long e = ((unsigned long long)key << 32) | (uint32_t)slot;
    const uint32_t i = atomicAdd(cand_n, 1u);
    if (i < cand_cap) cand[i] = e;
    else {
      const uint32_t at = atomicAdd(&a.hv.cnt[h.g], 1u);
      if (at < (uint32_t)KREC) a.hv.lists[(int64_t)h.g * KREC + at] = e;
    }
  }
  return (!in_range || key >= KEY_INF) ? 1u : 0u;
}
// a key of the context OUTSIDE the metric window (no new sum): masked whenever the window reaches the eviction bound
// (a head whose window does not is poisoned above)
__device__ __forceinline__ uint32_t harvest_outside(const AttnArgs& a, const HarvestCtx& h, int pos) {
  return (pos > h.bound || pos < a.hv.num_sinks) ? 1u : 0u;
}
// the deficit a wave counted, once per wave (every lane of the wave calls)
__device__ __forceinline__ void harvest_flush_def(const AttnArgs& a, const HarvestCtx& h, uint32_t ndef) {
  if (h.g < 0) return;                                                     // wave-uniform
  const uint32_t n = wave_reduce_sum(ndef);
  if (lane_id() == 0 && n) atomicAdd(&a.hv.def[h.g], n);
}

// fused aggregation (what CompressionMetrics.aggregate_decode does with the stored weights,
// vllm/kvcompress/metrics.py:429-439): metrics[slot] += sum_q p_q^2 (L2) or sum_q p_q, summed in
// query order with individually rounded operations, so the result is bit-identical to writing
// kv_metric_out and aggregating afterwards
__device__ __forceinline__ float metric_term(float acc, float p, int use_l2) {
  return __fadd_rn(acc, use_l2 ? __fmul_rn(p, p) : p);
}

// one token's metric row: the nq weights val(q) of query heads q0 .. q0 + nq - 1 either go to
// mo[slot, q0 + q] (16-byte non-temporal stores when rows of 4 heads are aligned: a plain store
// allocates in L2 and costs 15 % of the whole kernel) or are folded into metrics[slot]
template <typename F>
__device__ __forceinline__ uint32_t put_metric_row(const AttnArgs& a, float* mo, bool fuse, int64_t slot,
                                                   int qpk, int q0, int nq, F val, const HarvestCtx& hc, int pos) {
  if (fuse) {
    float acc = 0.0f;
    for (int q = 0; q < nq; ++q) acc = metric_term(acc, val(q), a.use_l2);
    const float mn = __fadd_rn(a.fused_metrics[slot], acc);
    a.fused_metrics[slot] = mn;
    return hc.g >= 0 ? harvest_key(a, hc, slot, mn, pos) : 0u;
  } else if (((qpk | nq) & 3) == 0) {
    for (int q = 0; q < nq; q += 4) {
      f32x4 v;
      v[0] = val(q); v[1] = val(q + 1); v[2] = val(q + 2); v[3] = val(q + 3);
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(mo + slot * qpk + q0 + q));
    }
  } else {
    for (int q = 0; q < nq; ++q) __builtin_nontemporal_store(val(q), mo + slot * qpk + q0 + q);
  }
  return 0u;
}

__device__ __forceinline__ float group_max(float v) {     // over the 4 lanes sharing lane&15
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// dynamic LDS: 2 x [ATT_WAVES][nqr][max(ATT_CHUNK, HD)] floats (P tiles | per-wave outputs)
template <typename T, int HD, int BS, int KVD, int NQM = 0>      // NQM > 0: KVC_LAYOUT_SLOT_MAJOR
// (slot-major operands cost ~30 registers more when left to the compiler: held to four waves per SIMD up to hd 128)
__global__ __launch_bounds__(256, (NQM > 0 && HD <= 128) ? 4 : 1) void paged_attention_decode_kernel(AttnArgs a) {
  constexpr bool SLOTS = NQM > 0;
  constexpr bool SLOTS_K = SLOTS && KVC_SM_EXP != 2, SLOTS_V = SLOTS && KVC_SM_EXP != 1;
  using M = Mma<T>;
  using V8 = typename M::V8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red_max[ATT_WAVES][ATT_NQ];
  __shared__ float red_sum[ATT_WAVES][ATT_NQ];
  constexpr int ROW = ATT_CHUNK > HD ? ATT_CHUNK : HD;
  constexpr int KS = HD / 32;            // QK k-steps
  constexpr int DT = HD / 16;            // output dim tiles
  const int qpk = a.num_heads / a.num_kv_heads;
  const int ngroups = (qpk + ATT_NQ - 1) / ATT_NQ;
  const int seq = blockIdx.z, hk = blockIdx.y / ngroups, qg = blockIdx.y % ngroups;
  const int part = blockIdx.x;
  // a context longer than max_context_len (a caller bug) is truncated instead of overrunning
  // the buffers that were sized from it
  const int ctx = min(a.context_lens[seq * a.num_kv_heads + hk], a.max_ctx);
  if (part * ATT_PART >= ctx) {
    // an empty head (nothing cached) attends to nothing: its output is zero, like the
    // reference's empty accumulation (.cu:332-420 with num_tokens = 0)
    if (ctx <= 0 && part == 0) {
      const int q0e = qg * ATT_NQ, nqe = min(ATT_NQ, qpk - q0e);
      for (int idx = threadIdx.x; idx < nqe * HD; idx += 256)
        reinterpret_cast<T*>(a.out)[((int64_t)seq * a.num_heads + hk * qpk + q0e + idx / HD) * HD + idx % HD] = (T)0.0f;
      if (threadIdx.x == 0 && qg == 0) harvest_empty_head(a, seq, hk);
    }
    return;
  }
  const int nparts = (ctx + ATT_PART - 1) / ATT_PART;
  const int q0 = qg * ATT_NQ;
  const int nq = min(ATT_NQ, qpk - q0);
  const int nqr = min(ATT_NQ, qpk);                       // LDS rows allocated
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  using KF = KvFrag<T, KVD>;
  constexpr int X = KF::X;
  const int32_t* bt = a.block_tables + (int64_t)(seq * a.num_kv_heads + hk) * a.max_blocks;
  const int head0 = hk * qpk + q0;                        // first query head of this group

  // ---- Q fragments (B operand): lane (query c, dim group g)
  V8 qf[KS];
  {
    const T* qp = reinterpret_cast<const T*>(a.q) + (int64_t)seq * a.q_stride + (int64_t)(head0 + c) * HD;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      au32x4 raw = {0u, 0u, 0u, 0u};
      if (c < nq) raw = *reinterpret_cast<const au32x4*>(qp + 32 * s + 8 * g);
      qf[s] = __builtin_bit_cast(V8, raw);
    }
  }
  const float slope = (a.alibi_slopes != nullptr && c < nq) ? a.alibi_slopes[head0 + c] : 0.0f;

  const int tok_w0 = part * ATT_PART + w * ATT_CHUNK;
  // metric bookkeeping of "my" tokens (one lane per token, see below): the block-table and
  // position loads are issued first so that their latency hides behind the K stream
  int64_t mslot[ATT_CHUNK / 64];
  int mpos[ATT_CHUNK / 64];
#pragma unroll
  for (int k = 0; k < ATT_CHUNK / 64; ++k) {
    const int tok = tok_w0 + k * 64 + lane;
    mslot[k] = 0;
    mpos[k] = 0x7FFFFFFF;
    if (a.record && tok < ctx) {
      mslot[k] = (int64_t)bt[tok / BS] * BS + (tok % BS);
      mpos[k] = a.kv_position[mslot[k]];
    }
  }

  // ---- QK^T: S[sb][j] = logit(token tok_w0 + 16 sb + 4 g + j, query c)
  f32x4 S[ATT_NSUB];
  float mloc = -INFINITY;
  using KSL = KSlots<T, SLOTS ? HD : 128, KVD>;
#ifndef KVC_KPF
#define KVC_KPF 2
#endif
  constexpr int KPF = KVC_KPF;                            // slot-major: tiles requested ahead
  typename KSL::Raw kraw[SLOTS_K ? KPF : 1];
  uint8_t* kst = reinterpret_cast<uint8_t*>(lds + (int64_t)2 * ATT_WAVES * nqr * ROW) + w * KSLOTS_STAGE;
  auto load_tile = [&](int sb, typename KSL::Raw& r) {
    const int t0 = tok_w0 + sb * 16;
    if (t0 < ctx)                                         // wave-uniform; (a tile lies inside one block: bs >= 16)
      KSL::load(r, a.k_cache, (int64_t)bt[t0 / BS] * a.kv_block_stride + (int64_t)(t0 % BS) * HD, lane);
  };
  if constexpr (SLOTS_K) {
#pragma unroll
    for (int d = 0; d < KPF; ++d) load_tile(d, kraw[d]);
  }
#pragma unroll
  for (int sb = 0; sb < ATT_NSUB; ++sb) {
    const int t0 = tok_w0 + sb * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (SLOTS_K) {
      if (t0 < ctx) {                                     // wave-uniform
        typename KF::Raw kk[KS];
        KSL::operands(kraw[sb % KPF], kst, kk, lane);
        if (sb + KPF < ATT_NSUB) load_tile(sb + KPF, kraw[sb % KPF]);
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = M::mma(KF::convert(kk[s], a.k_scale), qf[s], acc);
      }
    } else if (t0 < ctx) {                                // wave-uniform
      // dims 32 s + 8 g .. + 7 of token t0 + c: vector (dim / X), element (dim % X).  A 16-token
      // sub-block lies inside one cache block for BS >= 16; for BS = 8 it spans two (16 for BS = 1), so every lane
      // looks its own block up (tokens past the context are clamped onto the last one and masked below)
      const int tk = BS >= 16 ? t0 + c : min(t0 + c, ctx - 1);
      const int64_t phys = BS >= 16 ? bt[t0 / BS] : bt[tk / BS];
      const int64_t kb = phys * a.kv_block_stride + (int64_t)(tk % BS) * X;
      typename KF::Raw kk[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s)
        kk[s] = KF::load(a.k_cache, kb + (int64_t)((32 * s + 8 * g) / X) * BS * X + (32 * s + 8 * g) % X);
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = M::mma(KF::convert(kk[s], a.k_scale), qf[s], acc);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tok = t0 + 4 * g + j;
      float v = acc[j] * a.scale;
      if (slope != 0.0f) v += slope * (float)(tok - ctx + 1);       // .cu:265
      v = tok < ctx ? v : -INFINITY;
      acc[j] = v;
      mloc = fmaxf(mloc, v);
    }
    S[sb] = acc;
  }
  mloc = group_max(mloc);
  if (g == 0) red_max[w][c] = mloc;
  __syncthreads();
  const float m = fmaxf(fmaxf(red_max[0][c], red_max[1][c]), fmaxf(red_max[2][c], red_max[3][c]));

  // ---- p = exp(l - m), partition sum; P tile to LDS in [query][token] order
  float lsum = 0.0f;
  float* pw = lds + (int64_t)w * nqr * ROW;
#pragma unroll
  for (int sb = 0; sb < ATT_NSUB; ++sb) {
    f32x4 p;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float l = S[sb][j];
      p[j] = l == -INFINITY ? 0.0f : __expf(l - m);
      lsum += p[j];
    }
    S[sb] = p;
    if (c < nq) *reinterpret_cast<f32x4*>(pw + c * ROW + sb * 16 + 4 * g) = p;
  }
  lsum = group_sum(lsum);
  if (g == 0) red_sum[w][c] = lsum;
  __syncthreads();
  const float L = red_sum[0][c] + red_sum[1][c] + red_sum[2][c] + red_sum[3][c];
  const float inv = __fdividef(1.0f, L + 1e-6f);                   // .cu:298
  const bool single = nparts == 1;

  // ---- P.V: O[i][j] = out(dim 16 i + 4 g + j, query c) over this wave's tokens
  f32x4 O[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) O[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  using PV = PvSlots<T, SLOTS ? HD : 128, KVD>;
#pragma unroll
  for (int pr = 0; pr < ATT_NSUB / 2; ++pr) {
    const int t0 = tok_w0 + pr * 32;
    if (t0 >= ctx) break;                                 // wave-uniform
    // B operand: P[query c][tokens t0 + 8 g .. + 7], rounded to the cache type (.cu:332-420)
    V8 pb;
    {
      f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
      if (c < nq) {
        lo = *reinterpret_cast<const f32x4*>(pw + c * ROW + pr * 32 + 8 * g);
        hi = *reinterpret_cast<const f32x4*>(pw + c * ROW + pr * 32 + 8 * g + 4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { pb[e] = (T)lo[e]; pb[4 + e] = (T)hi[e]; }
    }
    if constexpr (SLOTS_V) {
      // A operand assembled in the lane from its 8 tokens' pieces (PvSlots)
      typename PV::Group gv;
      PV::load(gv, a, bt, t0, ctx, c, g, a.kv_block_stride, BS);
      PV::mma(gv, pb, O, a.v_scale);
      continue;
    }
    // A operand: V[dim 16 i + c][tokens t0 + 8 g .. + 7]
    const int tok = t0 + 8 * g;
    const bool live = tok < ctx;
    const bool tail = t0 + 32 > ctx;                      // wave-uniform: mask stale tokens
    typename KF::Raw vr[DT];
    if constexpr (BS >= 8) {
      const int64_t phys = live ? bt[tok / BS] : 0;
      const int64_t vb = phys * a.kv_block_stride + (int64_t)c * BS + (tok % BS);
#pragma unroll
      for (int i = 0; i < DT; ++i) vr[i] = live ? KF::load(a.v_cache, vb + (int64_t)i * 16 * BS) : KF::zero();
    } else {
      int64_t pe[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        pe[e] = tok + e < ctx ? (int64_t)bt[(tok + e) / BS] * a.kv_block_stride + (tok + e) % BS : -1;
#pragma unroll
      for (int i = 0; i < DT; ++i) vr[i] = KF::gather(a.v_cache, pe, (int64_t)(16 * i + c) * BS);
    }
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      V8 vv = KF::convert(vr[i], a.v_scale);
      if (tail) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (tok + e >= ctx) vv[e] = (T)0.0f;
      }
      O[i] = M::mma(vv, pb, O[i]);
    }
  }

  // ---- combine the four waves (same max, so a plain sum), normalise, store
  float* ow = lds + (int64_t)(ATT_WAVES + w) * nqr * ROW;   // second LDS region
  if constexpr (SLOTS_V) {
    if (c < nq) PV::store(ow + c * ROW, O, g);
  } else if (c < nq) {
#pragma unroll
    for (int i = 0; i < DT; ++i) *reinterpret_cast<f32x4*>(ow + c * ROW + 16 * i + 4 * g) = O[i];
  }
  __syncthreads();
  for (int idx = tid; idx < nq * HD; idx += 256) {
    const int qq = idx / HD, d = idx % HD;
    float o = 0.0f;
#pragma unroll
    for (int ww = 0; ww < ATT_WAVES; ++ww) o += lds[((int64_t)(ATT_WAVES + ww) * nqr + qq) * ROW + d];
    // per-query normaliser: the sums sit in red_sum (row qq)
    const float Lq = red_sum[0][qq] + red_sum[1][qq] + red_sum[2][qq] + red_sum[3][qq];
    o *= __fdividef(1.0f, Lq + 1e-6f);
    const int head = head0 + qq;
    if (single) {
      reinterpret_cast<T*>(a.out)[((int64_t)seq * a.num_heads + head) * HD + d] = (T)o;
    } else {
      reinterpret_cast<T*>(a.tmp_out)[(((int64_t)seq * a.num_heads + head) * a.max_parts + part) * HD + d] = (T)o;
    }
  }
  if (!single && tid < nq) {
    const int head = head0 + tid;
    const int64_t o = ((int64_t)seq * a.num_heads + head) * a.max_parts + part;
    a.exp_sums[o] = red_sum[0][tid] + red_sum[1][tid] + red_sum[2][tid] + red_sum[3][tid];
    a.max_logits[o] = fmaxf(fmaxf(red_max[0][tid], red_max[1][tid]), fmaxf(red_max[2][tid], red_max[3][tid]));
  }

  // ---- metric output (normalised within the partition, like the reference's tmp buffer).
  // LAST thing the wave does: vmcnt is in-order on gfx9, so a store issued before the V
  // loads (or before a barrier) puts its full write latency on the critical path of every
  // wave - measured 180 us of 870 at batch 256.  The P tiles live in their own LDS region.
  // one lane per TOKEN reads the wave's P tile back from LDS and stores all query heads of
  // the token at once (16 bytes for qpk = 4); the position test is done once per token
  if (a.record) {
    const int max_pos = a.last_position[seq] - a.kv_metric_buffer_len[seq];     // .cu:124
    float* mo = single ? a.kv_metric_out : a.tmp_kv_metric_out;
    const bool fuse = single && a.fused_metrics != nullptr;    // else: the rescale pass accumulates
    // per-query normaliser, once per workgroup (row q of red_max is free again: reuse it)
    __syncthreads();
    if (tid < nq) red_max[0][tid] = __fdividef(1.0f, red_sum[0][tid] + red_sum[1][tid] + red_sum[2][tid] + red_sum[3][tid] + 1e-6f);
    __syncthreads();
    const float* inv_q = red_max[0];
    const HarvestCtx hc = fuse ? harvest_ctx(a, seq, hk, max_pos, ctx, BS, tid == 0 && q0 == 0) : HarvestCtx{0u, -1, 0};
    uint32_t ndef = 0;
#pragma unroll
    for (int k = 0; k < ATT_CHUNK / 64; ++k) {
      const int tl = k * 64 + lane;
      if (mpos[k] > max_pos) {                             // .cu:305-312 (also: token >= ctx)
        if (hc.g >= 0 && mpos[k] != 0x7FFFFFFF) ndef += harvest_outside(a, hc, mpos[k]);
        continue;
      }
      ndef += put_metric_row(a, mo, fuse, mslot[k], qpk, q0, nq,
                             [&](int q) { return __fmul_rn(pw[q * ROW + tl], inv_q[q]); }, hc, mpos[k]);
    }
    harvest_flush_def(a, hc, ndef);
  }
}

// ------------------------------------------------------------------ single-pass variant
// One workgroup walks the WHOLE context of a (sequence, KV head) in 512-token steps with
// wave-local online softmax, keeps every unnormalised weight in LDS ([query][token] fp32)
// and writes the metric exactly once at the end: no partition buffers, no second kernels,
// no scattered tmp traffic.  Used when the weights of the longest context fit in LDS
// (two workgroups per CU: qpk * max_context * 4 B <= ~68 KiB - the continual-compression
// regime, e.g. 4k-token caps at qpk 4) and there are enough (sequence, KV head) pairs to fill the chip.
// dynamic LDS: P [nqr][prow] | O [4][nqr][HD] | mrec [niter][4][16]
template <typename T, int HD, int BS, int KVD, int NW, int NQM = 0>
__global__ __launch_bounds__(64 * NW) KVC_WHOLE_ATTR void paged_attention_decode_whole_kernel(AttnArgs a, int prow, int niter_max) {
  constexpr bool SLOTS = NQM > 0;
  constexpr bool SLOTS_K = SLOTS && KVC_SM_EXP != 2, SLOTS_V = SLOTS && KVC_SM_EXP != 1;
  using M = Mma<T>;
  using V8 = typename M::V8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red_max[NW][ATT_NQ];
  __shared__ float red_sum[NW][ATT_NQ];
  constexpr int STEP = NW * ATT_CHUNK;                    // tokens per workgroup iteration
  auto wg_max = [&](int q) { float v = red_max[0][q];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) v = fmaxf(v, red_max[ww][q]);
    return v; };
  constexpr int KS = HD / 32;
  constexpr int DT = HD / 16;
  const int qpk = a.num_heads / a.num_kv_heads;
  const int ngroups = (qpk + ATT_NQ - 1) / ATT_NQ;
  const int seq = blockIdx.y, hk = blockIdx.x / ngroups, qg = blockIdx.x % ngroups;
  // a context longer than max_context_len (a caller bug) is truncated instead of overrunning
  // the buffers that were sized from it
  const int ctx = min(a.context_lens[seq * a.num_kv_heads + hk], a.max_ctx);
  if (ctx <= 0) {                                         // empty head: zero output (see above)
    const int q0e = qg * ATT_NQ, nqe = min(ATT_NQ, qpk - q0e);
    for (int idx = threadIdx.x; idx < nqe * HD; idx += 64 * NW)
      reinterpret_cast<T*>(a.out)[((int64_t)seq * a.num_heads + hk * qpk + q0e + idx / HD) * HD + idx % HD] = (T)0.0f;
    if (threadIdx.x == 0 && qg == 0) harvest_empty_head(a, seq, hk);
    return;
  }
  const int q0 = qg * ATT_NQ;
  const int nq = min(ATT_NQ, qpk - q0);
  const int nqr = min(ATT_NQ, qpk);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  using KF = KvFrag<T, KVD>;
  constexpr int X = KF::X;
  const int32_t* bt = a.block_tables + (int64_t)(seq * a.num_kv_heads + hk) * a.max_blocks;
  const int head0 = hk * qpk + q0;
  // (what the metric epilogue needs from memory is requested here, a whole kernel ahead of its use)
  const int max_pos = a.record ? a.last_position[seq] - a.kv_metric_buffer_len[seq] : 0;
  const HarvestCtx hc = (a.record && a.fused_metrics != nullptr)
                            ? harvest_ctx(a, seq, hk, max_pos, ctx, BS, tid == 0 && q0 == 0) : HarvestCtx{0u, -1, 0};
  float* P = lds;                                          // [nqr][prow]
  float* Ol = lds + (int64_t)nqr * prow;                   // [4][nqr][HD]
  // (slot-major blocks: the waves' K tiles, KSlots, live in the O region until the loop over the context is over)
  const int o_floats = SLOTS_K ? max(4 * nqr * HD, NW * KSLOTS_STAGE / 4) : 4 * nqr * HD;
  float* mrec = Ol + o_floats;                             // [niter_max][NW][16]
  using KSL = KSlots<T, SLOTS ? HD : 128, KVD>;
  uint8_t* kst = reinterpret_cast<uint8_t*>(Ol) + w * KSLOTS_STAGE;

  V8 qf[KS];
  {
    const T* qp = reinterpret_cast<const T*>(a.q) + (int64_t)seq * a.q_stride + (int64_t)(head0 + c) * HD;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      au32x4 raw = {0u, 0u, 0u, 0u};
      if (c < nq) raw = *reinterpret_cast<const au32x4*>(qp + 32 * s + 8 * g);
      qf[s] = __builtin_bit_cast(V8, raw);
    }
  }
  const float slope = (a.alibi_slopes != nullptr && c < nq) ? a.alibi_slopes[head0 + c] : 0.0f;

  f32x4 O[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) O[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  using PV = PvSlots<T, SLOTS ? HD : 128, KVD>;
  float m_run = -INFINITY, l_run = 0.0f;
  const int niter = (ctx + STEP - 1) / STEP;
  for (int it = 0; it < niter; ++it) {
    const int tok_w0 = it * STEP + w * ATT_CHUNK;
    if (tok_w0 >= ctx) break;                              // wave-uniform
    // ---- QK^T, K fragments prefetched KVC_PF sub-blocks ahead (two waves per SIMD live here,
    // so the wave itself has to keep enough loads in flight; the scheduler is pinned with
    // sched_barrier because it otherwise sinks the loads next to their MFMAs)
    f32x4 S[ATT_NSUB];
    float mloc = -INFINITY;
    constexpr int PF = KVC_PF;
    typename KF::Raw kk[ATT_NSUB][KS];
    typename KSL::Raw kraw[SLOTS_K ? PF : 1];            // slot-major: whole tiles, operands through the wave's LDS tile (KSlots)
    auto load_k = [&](int sb) {
      const int t0 = tok_w0 + sb * 16;
      if constexpr (SLOTS_K) {
        if (t0 < ctx)
          KSL::load(kraw[sb % PF], a.k_cache, (int64_t)bt[t0 / BS] * a.kv_block_stride + (int64_t)(t0 % BS) * HD, lane);
      } else if (t0 < ctx) {
        const int tk = BS >= 16 ? t0 + c : min(t0 + c, ctx - 1);
        const int64_t phys = BS >= 16 ? bt[t0 / BS] : bt[tk / BS];
        const int64_t kb = phys * a.kv_block_stride + (int64_t)(tk % BS) * X;
#pragma unroll
        for (int s = 0; s < KS; ++s)
          kk[sb][s] = KF::load(a.k_cache, kb + (int64_t)((32 * s + 8 * g) / X) * BS * X + (32 * s + 8 * g) % X);
      } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) kk[sb][s] = KF::zero();
      }
    };
#pragma unroll
    for (int sb = 0; sb < PF && sb < ATT_NSUB; ++sb) load_k(sb);
#pragma unroll
    for (int sb = 0; sb < ATT_NSUB; ++sb) {
      const int t0 = tok_w0 + sb * 16;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if constexpr (SLOTS_K) {
        if (t0 < ctx) {                                     // wave-uniform
          typename KF::Raw kf[KS];
          KSL::operands(kraw[sb % PF], kst, kf, lane);
          if (sb + PF < ATT_NSUB) load_k(sb + PF);
#pragma unroll
          for (int s = 0; s < KS; ++s) acc = M::mma(KF::convert(kf[s], a.k_scale), qf[s], acc);
        }
      } else {
        if (sb + PF < ATT_NSUB) load_k(sb + PF);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = M::mma(KF::convert(kk[sb][s], a.k_scale), qf[s], acc);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int tok = t0 + 4 * g + j;
        float v = acc[j] * a.scale;
        if (slope != 0.0f) v += slope * (float)(tok - ctx + 1);
        v = tok < ctx ? v : -INFINITY;
        acc[j] = v;
        mloc = fmaxf(mloc, v);
      }
      S[sb] = acc;
    }
    // ---- wave-local online softmax (the chunk holds at least one live token)
    const float m_new = fmaxf(m_run, group_max(mloc));
    const float alpha = __expf(m_run - m_new);             // 0 on the first chunk
    float lsum = 0.0f;
    float* pw = P + tok_w0;                                // + c * prow per query row
#pragma unroll
    for (int sb = 0; sb < ATT_NSUB; ++sb) {
      f32x4 p;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float l = S[sb][j];
        p[j] = l == -INFINITY ? 0.0f : __expf(l - m_new);
        lsum += p[j];
      }
      if (c < nq) *reinterpret_cast<f32x4*>(pw + c * prow + sb * 16 + 4 * g) = p;
    }
    l_run = l_run * alpha + group_sum(lsum);
    m_run = m_new;
    if (g == 0) mrec[(it * NW + w) * ATT_NQ + c] = m_new;
    constexpr int NPR = ATT_NSUB / 2;
#pragma unroll
    for (int i = 0; i < DT; ++i) O[i] *= alpha;
    // ---- P.V, V fragments one 32-token pair ahead
    typename KF::Raw vv[2][DT];
    constexpr int VBUF = (SLOTS_V && HD > 128) ? 1 : 2;   // (hd 256: a group is 64 registers -- one at a time)
    typename PV::Group gv[SLOTS_V ? VBUF : 1];
    auto load_v = [&](int pr, int bufi) {
      if constexpr (SLOTS_V) {
        PV::load(gv[bufi], a, bt, tok_w0 + pr * 32, ctx, c, g, a.kv_block_stride, BS);
        return;
      }
      const int tok = tok_w0 + pr * 32 + 8 * g;
      const bool live = tok < ctx;
      if constexpr (SLOTS_V) {
      } else if constexpr (BS >= 8) {
        const int64_t phys = live ? bt[tok / BS] : 0;
        const int64_t vb = phys * a.kv_block_stride + (int64_t)c * BS + (tok % BS);
#pragma unroll
        for (int i = 0; i < DT; ++i)
          vv[bufi][i] = live ? KF::load(a.v_cache, vb + (int64_t)i * 16 * BS) : KF::zero();
      } else {
        int64_t pe[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          pe[e] = tok + e < ctx ? (int64_t)bt[(tok + e) / BS] * a.kv_block_stride + (tok + e) % BS : -1;
#pragma unroll
        for (int i = 0; i < DT; ++i) vv[bufi][i] = KF::gather(a.v_cache, pe, (int64_t)(16 * i + c) * BS);
      }
    };
    load_v(0, 0);
#pragma unroll
    for (int pr = 0; pr < NPR; ++pr) {
      const int t0 = tok_w0 + pr * 32;
      if (t0 >= ctx) break;
      if (VBUF == 2 && pr + 1 < NPR) load_v(pr + 1, (pr + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      V8 pb;
      {
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
        if (c < nq) {
          lo = *reinterpret_cast<const f32x4*>(pw + c * prow + pr * 32 + 8 * g);
          hi = *reinterpret_cast<const f32x4*>(pw + c * prow + pr * 32 + 8 * g + 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { pb[e] = (T)lo[e]; pb[4 + e] = (T)hi[e]; }
      }
      if constexpr (SLOTS_V) {
        PV::mma(gv[pr & (VBUF - 1)], pb, O, a.v_scale);
        if (VBUF == 1 && pr + 1 < NPR) load_v(pr + 1, 0);
        continue;
      }
      const int tok = t0 + 8 * g;
      const bool tail = t0 + 32 > ctx;                     // wave-uniform: mask stale tokens
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        V8 vf = KF::convert(vv[pr & 1][i], a.v_scale);
        if (tail) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (tok + e >= ctx) vf[e] = (T)0.0f;
        }
        O[i] = M::mma(vf, pb, O[i]);
      }
    }
  }

  if constexpr (SLOTS_K) __syncthreads();                  // (the O region held the waves' K tiles until here)
  // ---- combine the waves (each has its own running max), four at a time through the
  // [4][nqr][HD] output region so that the 8-wave variant needs no more LDS than the 4-wave one
  if (g == 0) { red_max[w][c] = m_run; red_sum[w][c] = l_run; }
  constexpr int OUTS = (ATT_NQ * HD + 64 * NW - 1) / (64 * NW);    // outputs per thread (upper bound)
  float oacc[OUTS];
#pragma unroll
  for (int k = 0; k < OUTS; ++k) oacc[k] = 0.0f;
#pragma unroll
  for (int hlf = 0; hlf < NW / 4; ++hlf) {
    if constexpr (SLOTS_V) {
      if (w / 4 == hlf && c < nq) PV::store(Ol + ((int64_t)(w % 4) * nqr + c) * HD, O, g);
    } else if (w / 4 == hlf && c < nq) {
#pragma unroll
      for (int i = 0; i < DT; ++i)
        *reinterpret_cast<f32x4*>(Ol + ((int64_t)(w % 4) * nqr + c) * HD + 16 * i + 4 * g) = O[i];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < OUTS; ++k) {
      const int idx = tid + k * 64 * NW;
      if (idx < nq * HD) {
        const int qq = idx / HD, d = idx % HD;
        const float Mq = wg_max(qq);
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
          const float mw = red_max[4 * hlf + ww][qq];
          const float sc = mw == -INFINITY ? 0.0f : __expf(mw - Mq);
          oacc[k] += Ol[((int64_t)ww * nqr + qq) * HD + d] * sc;
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < OUTS; ++k) {
    const int idx = tid + k * 64 * NW;
    if (idx < nq * HD) {
      const int qq = idx / HD, d = idx % HD;
      const float Mq = wg_max(qq);
      float Lq = 0.0f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww)
        Lq += red_max[ww][qq] == -INFINITY ? 0.0f : red_sum[ww][qq] * __expf(red_max[ww][qq] - Mq);
      reinterpret_cast<T*>(a.out)[((int64_t)seq * a.num_heads + head0 + qq) * HD + d] =
          (T)(oacc[k] * __fdividef(1.0f, Lq + 1e-6f));
    }
  }

  // ---- metrics: p = p~ * exp(m_used - M) / (L + 1e-6), one lane per token, written once
  if (a.record) {
    // per query: global max M and normaliser 1 / (L + 1e-6), once per workgroup, then per wave
    // and iteration the factor exp(m_used - M) / (L + 1e-6) of each query head
    __shared__ float fin_m[ATT_NQ], fin_i[ATT_NQ], wfac[NW][ATT_NQ];
    __shared__ uint32_t cand_n, cand_base;
    if (tid < nq) {
      const float Mg = wg_max(tid);
      float L = 0.0f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww)
        L += red_max[ww][tid] == -INFINITY ? 0.0f : red_sum[ww][tid] * __expf(red_max[ww][tid] - Mg);
      fin_m[tid] = Mg;
      fin_i[tid] = __fdividef(1.0f, L + 1e-6f);
    }
    if (tid == 0) cand_n = 0;
    __syncthreads();
    const bool fuse = a.fused_metrics != nullptr;
    // harvest: candidates are staged where the waves' partial outputs were (free behind the barrier above)
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(Ol);
    const uint32_t cand_cap = (uint32_t)(4 * nqr * HD / 2);
    if (fuse) {
      // metrics[slot] += sum_q p^2 is a read-modify-write: with one token after the other every store stands between
      // the loads behind it (the compiler cannot tell that slots differ) and a wave pays a full memory round trip per
      // 64 tokens -- measured 6 % of the whole kernel at 256 x 4k contexts.  The block-table entries, positions and
      // old sums of GI iterations are requested together, before anything is stored.
#ifndef KVC_ATT_GI
#define KVC_ATT_GI 4
#endif
      constexpr int GI = KVC_ATT_GI, K2 = ATT_CHUNK / 64;
      uint32_t ndef = 0;
      struct Grp { int64_t slot[GI][K2]; int kpos[GI][K2]; float mold[GI][K2]; };
      auto has_group = [&](int it0) { return it0 < niter && it0 * STEP + w * ATT_CHUNK < ctx; };      // wave-uniform
      auto load_group = [&](Grp& G, int it0) {
#pragma unroll
        for (int j = 0; j < GI; ++j)
#pragma unroll
          for (int k = 0; k < K2; ++k) {
            const int tok = (it0 + j) * STEP + w * ATT_CHUNK + k * 64 + lane;
            G.slot[j][k] = (it0 + j < niter && tok < ctx) ? (int64_t)bt[tok / BS] * BS + (tok % BS) : (int64_t)-1;
          }
#pragma unroll
        for (int j = 0; j < GI; ++j)
#pragma unroll
          for (int k = 0; k < K2; ++k) G.kpos[j][k] = G.slot[j][k] >= 0 ? a.kv_position[G.slot[j][k]] : 0x7FFFFFFF;
#pragma unroll
        for (int j = 0; j < GI; ++j)
#pragma unroll
          for (int k = 0; k < K2; ++k) G.mold[j][k] = G.kpos[j][k] <= max_pos ? a.fused_metrics[G.slot[j][k]] : 0.0f;
      };
      auto process_group = [&](const Grp& G, int it0) {
#pragma unroll
        for (int j = 0; j < GI; ++j) {
          const int it = it0 + j;
          const int tok_w0 = it * STEP + w * ATT_CHUNK;
          if (it >= niter || tok_w0 >= ctx) break;
          const float* mr = mrec + (it * NW + w) * ATT_NQ;
          if (lane < nq) wfac[w][lane] = __expf(mr[lane] - fin_m[lane]) * fin_i[lane];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          const float* fq = wfac[w];
#pragma unroll
          for (int k = 0; k < K2; ++k) {
            if (G.kpos[j][k] > max_pos) {                    // (also: token >= ctx)
              if (hc.g >= 0 && G.slot[j][k] >= 0) ndef += harvest_outside(a, hc, G.kpos[j][k]);
              continue;
            }
            const int tok = tok_w0 + k * 64 + lane;
            float acc = 0.0f;
            for (int q = 0; q < nq; ++q) acc = metric_term(acc, __fmul_rn(P[q * prow + tok], fq[q]), a.use_l2);
            const float mn = __fadd_rn(G.mold[j][k], acc);
            a.fused_metrics[G.slot[j][k]] = mn;
            if (hc.g >= 0) ndef += harvest_key_staged(a, hc, G.slot[j][k], mn, G.kpos[j][k], cand, &cand_n, cand_cap);
          }
          __builtin_amdgcn_wave_barrier();             // wfac[w] is rewritten by the next iteration
        }
      };
      // ... and the next group's operands are requested before the current group is worked off (two register sets):
      // one exposed round trip per wave instead of one per group
      Grp ga, gb;
      if (has_group(0)) load_group(ga, 0);
      for (int it0 = 0; has_group(it0); it0 += 2 * GI) {
        if (has_group(it0 + GI)) load_group(gb, it0 + GI);
        process_group(ga, it0);
        if (!has_group(it0 + GI)) break;
        if (has_group(it0 + 2 * GI)) load_group(ga, it0 + 2 * GI);
        process_group(gb, it0 + GI);
      }
      harvest_flush_def(a, hc, ndef);
      if (hc.g >= 0) {                                 // (workgroup-uniform: one head per workgroup)
        __syncthreads();
        const uint32_t n = min(cand_n, cand_cap);
        if (n) {
          if (tid == 0) cand_base = atomicAdd(&a.hv.cnt[hc.g], n);
          __syncthreads();
          for (uint32_t i = tid; i < n; i += 64 * NW) {
            const uint32_t at = cand_base + i;
            if (at < (uint32_t)KREC) a.hv.lists[(int64_t)hc.g * KREC + at] = cand[i];
          }
        }
      }
      return;
    }
    for (int it = 0; it < niter; ++it) {
      const int tok_w0 = it * STEP + w * ATT_CHUNK;
      if (tok_w0 >= ctx) break;
      const float* mr = mrec + (it * NW + w) * ATT_NQ;
      if (lane < nq) wfac[w][lane] = __expf(mr[lane] - fin_m[lane]) * fin_i[lane];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const float* fq = wfac[w];
#pragma unroll
      for (int k = 0; k < ATT_CHUNK / 64; ++k) {
        const int tok = tok_w0 + k * 64 + lane;
        if (tok >= ctx) continue;
        const int64_t slot = (int64_t)bt[tok / BS] * BS + (tok % BS);
        const int kpos = a.kv_position[slot];
        if (kpos > max_pos) continue;
        put_metric_row(a, a.kv_metric_out, false, slot, qpk, q0, nq,
                       [&](int q) { return __fmul_rn(P[q * prow + tok], fq[q]); }, hc, kpos);
      }
      __builtin_amdgcn_wave_barrier();               // wfac[w] is rewritten by the next iteration
    }
  }
}

// second pass for heads with more than one partition            .cu:532-651
template <typename T, int HD, int BS>
__global__ __launch_bounds__(256) void paged_attention_reduce_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];      // [2 * max_parts]
  __shared__ float red[8];
  const int head = blockIdx.x, seq = blockIdx.y;
  const int qpk = a.num_heads / a.num_kv_heads;
  const int hk = head / qpk, qoff = head % qpk;
  // a context longer than max_context_len (a caller bug) is truncated instead of overrunning
  // the buffers that were sized from it
  const int ctx = min(a.context_lens[seq * a.num_kv_heads + hk], a.max_ctx);
  const int nparts = (ctx + ATT_PART - 1) / ATT_PART;
  if (nparts <= 1) return;                                 // finished by the first kernel
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float* smax = lds;
  float* ssum = lds + a.max_parts;
  const int64_t base = ((int64_t)seq * a.num_heads + head) * a.max_parts;
  float mx = -INFINITY;
  for (int i = tid; i < nparts; i += 256) { const float l = a.max_logits[base + i]; smax[i] = l; mx = fmaxf(mx, l); }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float tot = 0.0f;
  for (int i = tid; i < nparts; i += 256) {
    const float r = a.exp_sums[base + i] * expf(smax[i] - mx);
    ssum[i] = r;
    tot += r;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) tot += __shfl_xor(tot, d, 64);
  if (lane == 0) red[4 + w] = tot;
  __syncthreads();
  tot = red[4] + red[5] + red[6] + red[7];
  const float inv = __fdividef(1.0f, tot + 1e-6f);
  // out[d] = sum_j tmp_out[j][d] * share_j: the partitions are split over 256 / HD thread
  // groups and each thread keeps 8 loads in flight (this kernel is pure latency: at one
  // sequence of 32k tokens it used to cost a third of the whole attention call)
  const T* tp = reinterpret_cast<const T*>(a.tmp_out) + base * HD;
  constexpr int NG = HD <= 256 ? 256 / HD : 1;              // thread groups over partitions
  __shared__ float part_acc[256];
  const int d = tid % HD, grp = tid / HD;
  float acc = 0.0f;
  if (grp < NG) {
    int j = grp;
    for (; j + 7 * NG < nparts; j += 8 * NG) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (float)tp[(int64_t)(j + u * NG) * HD + d];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u] * (ssum[j + u * NG] * inv);
    }
    for (; j < nparts; j += NG) acc += (float)tp[(int64_t)j * HD + d] * (ssum[j] * inv);
  }
  if (NG > 1) {
    part_acc[tid] = acc;
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int g2 = 1; g2 < NG; ++g2) acc += part_acc[g2 * HD + d];
    }
  }
  if (grp == 0 && d < HD) reinterpret_cast<T*>(a.out)[((int64_t)seq * a.num_heads + head) * HD + d] = (T)acc;
}

// metric side of the second pass: kv_metric_out = tmp * (partition's share of the softmax
// denominator)  (.cu:642-650), one workgroup per 1024 tokens of a (sequence, KV head), all
// query heads of the KV head at once (16-byte rows for qpk = 4) instead of one strided
// 4-byte column per query head.
constexpr int ATT_RS_TOK = 1024;
template <int BS>
__global__ __launch_bounds__(256) void paged_attention_metric_rescale_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float fac[];     // [qpk][2]
  const int chunk = blockIdx.x, hk = blockIdx.y, seq = blockIdx.z;
  // a context longer than max_context_len (a caller bug) is truncated instead of overrunning
  // the buffers that were sized from it
  const int ctx = min(a.context_lens[seq * a.num_kv_heads + hk], a.max_ctx);
  const int nparts = (ctx + ATT_PART - 1) / ATT_PART;
  if (nparts <= 1 || chunk * ATT_RS_TOK >= ctx) return;
  const int qpk = a.num_heads / a.num_kv_heads;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int p0 = chunk * (ATT_RS_TOK / ATT_PART);
  for (int q = w; q < qpk; q += 4) {                       // one wave per query head
    const int64_t base = ((int64_t)seq * a.num_heads + hk * qpk + q) * a.max_parts;
    float mx = -INFINITY;
    for (int j = lane; j < nparts; j += 64) mx = fmaxf(mx, a.max_logits[base + j]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    float tot = 0.0f;
    for (int j = lane; j < nparts; j += 64) tot += a.exp_sums[base + j] * expf(a.max_logits[base + j] - mx);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) tot += __shfl_xor(tot, d, 64);
    const float inv = __fdividef(1.0f, tot + 1e-6f);
    if (lane < 2) {
      const int j = p0 + lane;
      fac[q * 2 + lane] = j < nparts ? a.exp_sums[base + j] * expf(a.max_logits[base + j] - mx) * inv : 0.0f;
    }
  }
  __syncthreads();
  const int max_pos = a.last_position[seq] - a.kv_metric_buffer_len[seq];
  const int32_t* bt = a.block_tables + (int64_t)(seq * a.num_kv_heads + hk) * a.max_blocks;
  constexpr int K = ATT_RS_TOK / 256;
  const bool fuse = a.fused_metrics != nullptr;
  const HarvestCtx hc = fuse ? harvest_ctx(a, seq, hk, max_pos, ctx, BS, tid == 0 && chunk == 0) : HarvestCtx{0u, -1, 0};
  if ((qpk & 3) == 0 && !(fuse && qpk != 4)) {
    // all address loads, then all position + tmp loads, then the stores: four independent
    // chains per lane keep enough bytes in flight for a pass that is pure streaming; query
    // heads go in aligned groups of four (16-byte rows)
    int64_t slot[K];
    bool ok[K];
    int posv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = chunk * ATT_RS_TOK + k * 256 + tid;
      ok[k] = i < ctx;
      slot[k] = ok[k] ? (int64_t)bt[i / BS] * BS + (i % BS) : 0;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) posv[k] = ok[k] ? a.kv_position[slot[k]] : 0x7FFFFFFF;
    float mold[K];                                     // (fuse: the old sums, requested with the weights -- see the single-pass kernel)
#pragma unroll
    for (int k = 0; k < K; ++k) mold[k] = (fuse && ok[k] && posv[k] <= max_pos) ? a.fused_metrics[slot[k]] : 0.0f;
    uint32_t ndef = 0;
    if (hc.g >= 0)
#pragma unroll
      for (int k = 0; k < K; ++k)
        if (ok[k] && posv[k] > max_pos) ndef += harvest_outside(a, hc, posv[k]);
    for (int qb = 0; qb < qpk; qb += 4) {
      f32x4 t[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        t[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok[k]) t[k] = *reinterpret_cast<const f32x4*>(a.tmp_kv_metric_out + slot[k] * qpk + qb);
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (!ok[k] || posv[k] > max_pos) continue;
        const int pj = (chunk * ATT_RS_TOK + k * 256 + tid) / ATT_PART - p0;
        f32x4 v = t[k];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = __fmul_rn(v[q], fac[2 * (qb + q) + pj]);
        if (fuse) {                                  // qpk == 4 here
          float acc = 0.0f;
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = metric_term(acc, v[q], a.use_l2);
          const float mn = __fadd_rn(mold[k], acc);
          a.fused_metrics[slot[k]] = mn;
          if (hc.g >= 0) ndef += harvest_key(a, hc, slot[k], mn, posv[k]);
        } else {
          *reinterpret_cast<f32x4*>(a.kv_metric_out + slot[k] * qpk + qb) = v;
        }
      }
    }
    harvest_flush_def(a, hc, ndef);
  } else {
    uint32_t ndef = 0;
    for (int k = 0; k < K; ++k) {
      const int i = chunk * ATT_RS_TOK + k * 256 + tid;
      if (i >= ctx) break;
      const int64_t slot = (int64_t)bt[i / BS] * BS + (i % BS);
      const int kpos = a.kv_position[slot];
      if (kpos > max_pos) {
        if (hc.g >= 0) ndef += harvest_outside(a, hc, kpos);
        continue;
      }
      const int pj = i / ATT_PART - p0;
      float acc = 0.0f;
      for (int q = 0; q < qpk; ++q) {
        const float v = __fmul_rn(a.tmp_kv_metric_out[slot * qpk + q], fac[q * 2 + pj]);
        if (fuse) acc = metric_term(acc, v, a.use_l2);
        else a.kv_metric_out[slot * qpk + q] = v;
      }
      if (fuse) {
        const float mn = __fadd_rn(a.fused_metrics[slot], acc);
        a.fused_metrics[slot] = mn;
        if (hc.g >= 0) ndef += harvest_key(a, hc, slot, mn, kpos);
      }
    }
    harvest_flush_def(a, hc, ndef);
  }
}

// which schedule a call takes (shared by the launcher and kvc_paged_attention_decode_uses_partitions)
struct AttnPlan { bool whole; int nw, prow; size_t whole_lds; };
// schedule: 0 automatic, 1 always partitioned, 2 single pass whenever it fits (kvc_attention_params.schedule)
// LDS per WAVE for slot-major blocks: the K tile in flight (KSlots).  The single-pass kernel keeps it where the
// waves' partial outputs go afterwards.
inline int attention_stage_bytes(int head_size, int layout) { (void)head_size; return layout == KVC_LAYOUT_SLOT_MAJOR ? KSLOTS_STAGE : 0; }
inline AttnPlan attention_plan(int num_seqs, int num_heads, int num_kv_heads, int head_size, int max_ctx, int schedule,
                               int stage_bytes = 0) {
  const int qpk = num_heads / num_kv_heads;
  const int ngroups = (qpk + ATT_NQ - 1) / ATT_NQ;
  const int nqr = qpk < ATT_NQ ? qpk : ATT_NQ;
  const int max_parts = (max_ctx + ATT_PART - 1) / ATT_PART;
  // single pass: the fp32 weights of the longest context in LDS; 4 waves with two workgroups
  // per CU when that fits (<= 79 KiB each), else 8 waves with one workgroup per CU.
  // row = the longest context rounded to a wave chunk, + 4 to stagger the query rows over the banks
  AttnPlan p;
  p.prow = (max_ctx + ATT_CHUNK - 1) / ATT_CHUNK * ATT_CHUNK + 4;
  auto whole_bytes = [&](int nw) {
    const int niter = (max_ctx + nw * ATT_CHUNK - 1) / (nw * ATT_CHUNK);
    const size_t o_bytes = std::max((size_t)4 * nqr * head_size * sizeof(float), (size_t)nw * stage_bytes);
    return ((size_t)nqr * p.prow + (size_t)niter * nw * ATT_NQ) * sizeof(float) + o_bytes;
  };
  p.nw = whole_bytes(4) <= 79 * 1024 ? 4 : 8;
  p.whole_lds = whole_bytes(p.nw);
  const bool fits = max_parts > 1 && p.whole_lds <= (size_t)(p.nw == 4 ? 79 : 155) * 1024;   // 1 partition is one pass anyway
  const int64_t wgs = (int64_t)num_seqs * num_kv_heads * ngroups;
  p.whole = schedule == 2 ? fits
          : (schedule == 1 ? false : (fits && wgs >= (p.nw == 4 ? 512 : 256)));
  return p;
}

template <typename T, int HD, int BS, int KVD, int NQM>
int launch_attention_layout(const AttnArgs& a, int num_seqs, hipStream_t s);

template <typename T, int HD, int BS, int KVD>
int launch_attention(const AttnArgs& a, int num_seqs, hipStream_t s) {
  if (a.layout == KVC_LAYOUT_SLOT_MAJOR) {
    if constexpr (slot_major_shape<HD, BS>()) {
      return launch_attention_layout<T, HD, BS, KVD, 1>(a, num_seqs, s);
    } else {
      return fail_invalid("paged_attention_decode: slot-major blocks support head sizes 64 / 128 / 256 with block sizes 16 / 32");
    }
  }
  return launch_attention_layout<T, HD, BS, KVD, 0>(a, num_seqs, s);
}

template <typename T, int HD, int BS, int KVD, int NQM>
int launch_attention_layout(const AttnArgs& a, int num_seqs, hipStream_t s) {
  const int qpk = a.num_heads / a.num_kv_heads;
  const int ngroups = (qpk + ATT_NQ - 1) / ATT_NQ;
  const int nqr = qpk < ATT_NQ ? qpk : ATT_NQ;
  const int STAGE = attention_stage_bytes(HD, NQM > 0 ? KVC_LAYOUT_SLOT_MAJOR : KVC_LAYOUT_REFERENCE);
  const AttnPlan plan = attention_plan(num_seqs, a.num_heads, a.num_kv_heads, HD, a.max_ctx, a.schedule, STAGE);
  const int prow = plan.prow, nw = plan.nw;
  const size_t whole_lds = plan.whole_lds;
  const bool whole = plan.whole;
  if (whole) {
    const int niter = (a.max_ctx + nw * ATT_CHUNK - 1) / (nw * ATT_CHUNK);
    if (nw == 4) {
      if (whole_lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(paged_attention_decode_whole_kernel<T, HD, BS, KVD, 4, NQM>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)whole_lds);
      hipLaunchKernelGGL((paged_attention_decode_whole_kernel<T, HD, BS, KVD, 4, NQM>), dim3(a.num_kv_heads * ngroups, num_seqs),
                         dim3(256), whole_lds, s, a, prow, niter);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(paged_attention_decode_whole_kernel<T, HD, BS, KVD, 8, NQM>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)whole_lds);
      hipLaunchKernelGGL((paged_attention_decode_whole_kernel<T, HD, BS, KVD, 8, NQM>), dim3(a.num_kv_heads * ngroups, num_seqs),
                         dim3(512), whole_lds, s, a, prow, niter);
    }
    return check_launch("paged_attention_decode");
  }
  if (a.max_parts > 1 && (a.exp_sums == nullptr || a.max_logits == nullptr || a.tmp_out == nullptr ||
                          (a.record && a.tmp_kv_metric_out == nullptr)))
    return fail_invalid("paged_attention_decode: this shape needs the partition buffers");
  constexpr int ROW = ATT_CHUNK > HD ? ATT_CHUNK : HD;
  const size_t lds_bytes = (size_t)2 * ATT_WAVES * nqr * ROW * sizeof(float) + (size_t)ATT_WAVES * STAGE;
  if (lds_bytes > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(paged_attention_decode_kernel<T, HD, BS, KVD, NQM>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipLaunchKernelGGL((paged_attention_decode_kernel<T, HD, BS, KVD, NQM>),
                     dim3(a.max_parts, a.num_kv_heads * ngroups, num_seqs), dim3(256), lds_bytes, s, a);
  if (a.max_parts > 1) {
    hipLaunchKernelGGL((paged_attention_reduce_kernel<T, HD, BS>), dim3(a.num_heads, num_seqs), dim3(256),
                       (size_t)2 * a.max_parts * sizeof(float), s, a);
    if (a.record)
      hipLaunchKernelGGL((paged_attention_metric_rescale_kernel<BS>),
                         dim3((a.max_parts * ATT_PART + ATT_RS_TOK - 1) / ATT_RS_TOK, a.num_kv_heads, num_seqs),
                         dim3(256), (size_t)qpk * 2 * sizeof(float), s, a);
  }
  return check_launch("paged_attention_decode");
}

}  // namespace kvc

// every instantiated (head size, block size): "auto" caches / fp8 caches.  The kernels are
// compiled in four translation units (kvc_attention_inst_*.hip: fp16 / bf16 x auto / fp8) so that
// the build runs them side by side; kvc_attention.hip only dispatches.
#define KVC_ATT_SHAPES(X) X(128, 1) X(64, 8) X(128, 8) X(64, 16) X(64, 32) X(96, 16) X(96, 32) \
  X(128, 16) X(128, 32) X(256, 16) X(256, 32)
#define KVC_ATT_F8_SHAPES(X) X(64, 16) X(64, 32) X(128, 16) X(128, 32)
