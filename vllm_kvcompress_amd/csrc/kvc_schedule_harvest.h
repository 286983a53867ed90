// kvc_schedule_harvest.h -- A2a + A3: the decode step's aggregation that harvests the small-eviction schedule's candidates
// (one translation unit: included by kvc_schedule.hip behind kvc_schedule_small.h; see the overview there)
#pragma once
#include "kvc_common.h"
#include "kvc_schedule_common.h"
#include "kvc_schedule_small.h"
#include "../../include/kvc_mi355x.h"

namespace kvc {

// ------------------------------------------------------------------ 10. harvest-ahead
// With compression_interval = 1 every decode step sweeps the metric store twice: aggregate_decode
// (reference metrics.py:429-439: metrics += sum_q temp^2, the whole cache, 4 * qpk + 8 B per slot)
// and, right behind it, stream_collect_kernel (section 7: 4 + 1 B per candidate slot, to find the
// ~1 % of the keys below each sequence's pivot).  The sums pass through registers in the first
// sweep: aggregate_harvest_kernel is aggregate_decode_kernel with section 7's harvest behind the
// add -- the second sweep disappears.  What it cannot know is the pivot, a quantile of the sums it
// is making; it uses the one the PREVIOUS schedule call left in the harvest buffer.  Exactness
// does not depend on how good that guess is: the records made from these lists hold EVERY
// evictable key below the pivot that was used, so the selection is exact as soon as they list k'
// thresholds -- or the flag is raised, as for any record that falls short (section 7).
//
// The next pivot comes from the lists themselves (harvest_pivot_kernel, one workgroup per sequence
// behind the selection): a call's lists -- its own collecting pass's or harvested ones -- are ALL
// the sequence's evictable keys below the pivot they were made with, sorted per head, and the
// selection has just said which of them leave.  Metrics only grow (sums of squares / of softmax
// weights), so what lies below a pivot at the next step is what is left of the list, less the
// keys the step's attention lifts over it, plus the keys that leave the protected window.  With
// target = (1 + widen) x Tgt (Tgt = k * bs + sum(hang - 1): the count that guarantees k listed
// thresholds, section 7):  enough keys left -> the pivot is the target-th smallest of them (an
// exact count, no sample, no margin for a sample's error: a 1/64 sample needs 2.6 x Tgt for the
// same safety);  fewer -> the old pivot moved up by the width the missing keys would take at the
// density of the list's upper quarter, twice over (overshoot is corrected by the next step's exact
// count; metrics space, not key space: linear where the density is).  No sampling pass and no
// pivot kernel on harvested steps -- nor on the steps of a caller that never harvests: the
// schedule's own collecting pass takes the same pivots (harvest bit 2, "pivot memory": they are one
// decode step of attention old either way), and handles 1.3 x Tgt candidates instead of 2.6 x.
//
// Layout of the aggregation: the plain kernel's -- a lane per slot, rows of 64 consecutive slots,
// the temp row as qpk / 4 16-byte loads per lane -- so that the arithmetic (and its order) is the
// plain kernel's.  A wave iteration covers 64 blocks = BS rows: lane b looks after the metadata of
// block b (one coalesced load per table, as in section 7) and hands pivot / head / position bound
// to the rows' lanes by shuffles; U = 8 rows are in flight at a time (a wave's 64 blocks are
// contiguous: the TILE form of csrc/kvc_aggregate.hip).  STREAM: a store several times the
// Infinity Cache (>= 1 GiB of metrics) has its metrics loaded and stored non-temporally, as there.
// (the buffer's layout: kvc_harvest_layout.h)

// QV = qpk / 4 (qpk 4 or 8: the temp row as 16-byte loads), or 0: any qpk, the row as scalar loads summed while they
// arrive (the generic loop of aggregate_decode_kernel: the same additions in the same order).
// LAZY as in section 7: keys that do not depend on positions and sequences that do not need each other's counts of
// evictable keys look a position up only for the ~1 % of the slots below the pivot.  Otherwise (averaged metrics, a
// position bias, the reference's batch > 1 rule) the position rows are streamed next to the sums (+ 4 B per slot),
// every key is made in full (slot_key) and the masked / non-finite slots of every block are counted for their head
// (st_def), exactly as the full collecting pass does.
template <int BS, int QV, bool STREAM, bool LAZY>
__global__ __launch_bounds__(256) void aggregate_harvest_kernel(kvc_schedule_params p, SchedWs ws, float* __restrict__ temp,
                                                                const uint32_t* __restrict__ hv_pivot, int qpk, int use_l2,
                                                                int clear_temp) {
  constexpr int ROWS = BS;                           // 64 blocks x BS slots = BS rows of 64 slots
#ifndef KVC_HV_U
#define KVC_HV_U 8
#endif
  constexpr int U = KVC_HV_U;                        // rows in flight
  constexpr int BPR = 64 / BS;                       // blocks per row
  static_assert(ROWS % U == 0, "block sizes 8 / 16 / 32");
  __shared__ uint32_t qk[4][128], qs[4][128], qg[4][128];
  __shared__ int32_t ql[4][128];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int L = p.num_layers, H = p.num_kv_heads;
  float* __restrict__ metrics = const_cast<float*>(p.metrics);
  f32x4* __restrict__ temp4 = reinterpret_cast<f32x4*>(temp);
  const int64_t num_slots = p.num_blocks * BS;
  unsigned long long* lists = reinterpret_cast<unsigned long long*>(ws.rec64);
  uint32_t claimed = 0;
  int qn = 0;
  auto drain = [&](int n) {                          // pops the top n (<= 64) queue entries (section 7, LAZY)
    wave_lds_sync();
    if (lane < n) {
      const int e = qn - n + lane;
      const uint32_t g = qg[w][e];
      bool in_range = true;                          // (the full key has the mask in it)
      if constexpr (LAZY) {                          // metrics.py:539-544, for the few that matter
        const int tp = p.token_positions[qs[w][e]];
        in_range = tp <= ql[w][e] && tp >= p.num_sinks;
      }
      if (in_range) {
        const uint32_t pos = atomicAdd(&ws.st_cnt[g], 1u);
        if (pos < (uint32_t)KREC) lists[(int64_t)g * KREC + pos] = ((unsigned long long)qk[w][e] << 32) | qs[w][e];
      }
    }
    qn -= n;
    wave_lds_sync();
  };
  const int64_t wave = (int64_t)blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4;
  for (int64_t b0 = wave * 64; b0 < p.num_blocks; b0 += nwaves * 64) {
    const int64_t slot0 = b0 * BS;
    f32x4 t[U][QV > 0 ? QV : 1];
    float m[U], gsum[U];
    int tpos[U];
    auto load_rows = [&](int r0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t s = slot0 + (int64_t)(r0 + u) * 64 + lane;
        gsum[u] = 0.f;
        if (s < num_slots) {
          if constexpr (QV > 0) {
#pragma unroll
            for (int v = 0; v < QV; ++v) t[u][v] = __builtin_nontemporal_load(temp4 + s * QV + v);
          } else {
            float a = 0.0f;
            for (int q = 0; q < qpk; ++q) {
              float x = temp[s * qpk + q];
              if (use_l2) x = __fmul_rn(x, x);
              a = __fadd_rn(a, x);
            }
            gsum[u] = a;
          }
          m[u] = STREAM ? __builtin_nontemporal_load(metrics + s) : metrics[s];
          tpos[u] = LAZY ? 0 : __builtin_nontemporal_load(p.token_positions + s);
        } else {
#pragma unroll
          for (int v = 0; v < (QV > 0 ? QV : 1); ++v) t[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
          m[u] = 0.f;
          tpos[u] = 0;
        }
      }
    };
    load_rows(0);                                    // the rows do not wait for the metadata
    const int64_t mb = b0 + lane;
    const bool have = mb < p.num_blocks;
    const BlockMeta mt = load_meta(p, mb, have);
    bool ok = have && mt.s >= 0 && mt.s < p.seq_slot_len;
    int i = p.seq_slot_of_seq[ok ? mt.s : 0];
    ok = ok && i >= 0 && mt.l >= 0 && mt.l < L && mt.h >= 0 && mt.h < H;
    const int l = ok ? mt.l : 0, h = ok ? mt.h : 0;
    if (!ok) i = 0;
    // (context_lens == nullptr: a harvest for the NEXT iteration's call, whose context lengths are not known yet -- a
    // block belongs to the batch by its metadata alone; the schedule call checks the walked blocks against its own N)
    const int ctx = p.context_lens != nullptr ? p.context_lens[(l * p.num_seqs + i) * H + h] : 0;
    ok = ok && mt.lbn >= 0 && (p.context_lens == nullptr || mt.lbn < (ctx + BS - 1) / BS);
    claimed += (uint32_t)__popcll(__ballot(ok));
    const int g = ok ? (i * L + l) * H + h : -1;
    const uint32_t pex = ok ? hv_pivot[i] : 0u;      // (0: no key lies below it)
    const int seq_pos = p.seq_positions[i] + p.harvest_position_delta, prot = p.num_protected[i];
    const int bound = seq_pos - prot;
#pragma unroll
    for (int r0 = 0; r0 < ROWS; r0 += U) {
      if (r0 > 0) load_rows(r0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t s = slot0 + (int64_t)(r0 + u) * 64 + lane;
        const bool in = s < num_slots;
        float acc = 0.0f;
        if constexpr (QV > 0) {
#pragma unroll
          for (int v = 0; v < QV; ++v) {
            f32x4 x = t[u][v];
            if (use_l2) { x.x = __fmul_rn(x.x, x.x); x.y = __fmul_rn(x.y, x.y); x.z = __fmul_rn(x.z, x.z); x.w = __fmul_rn(x.w, x.w); }
            acc = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(acc, x.x), x.y), x.z), x.w);
          }
        } else {
          acc = gsum[u];
        }
        const float mn = __fadd_rn(m[u], acc);
        if (in) {
          if constexpr (STREAM) __builtin_nontemporal_store(mn, metrics + s); else metrics[s] = mn;
          if (clear_temp) {
            if constexpr (QV > 0) {
#pragma unroll
              for (int v = 0; v < QV; ++v) temp4[s * QV + v] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
              for (int q = 0; q < qpk; ++q) temp[s * qpk + q] = 0.0f;
            }
          }
        }
        const int src = (r0 + u) * BPR + lane / BS;  // the lane that looks after this slot's block
        const uint32_t pvv = (uint32_t)__shfl((int)pex, src, 64);
        uint32_t key;
        int gg = -1;
        if constexpr (LAZY) {
          key = float_to_key(mn);
        } else {
          gg = __shfl(g, src, 64);
          const int spp = __shfl(seq_pos, src, 64), prr = __shfl(prot, src, 64);
          int ll = 0, hh = 0;
          if (p.bias != nullptr) { ll = __shfl(l, src, 64); hh = __shfl(h, src, 64); }
          key = slot_key(p, mn, tpos[u], spp, prr, ll, hh);
          // masked / non-finite slots of the block (its BS lanes are adjacent) -> the head's deficit
          int ninf = (gg >= 0 && key >= KEY_INF) ? 1 : 0;
#pragma unroll
          for (int d = 1; d < BS; d <<= 1) ninf += __shfl_xor(ninf, d, 64);
          if (lane % BS == 0 && gg >= 0 && ninf > 0) atomicAdd(&ws.st_def[gg], (uint32_t)ninf);
        }
        const bool c = in && key < pvv;              // (pvv <= KEY_INF)
        const unsigned long long bal = __ballot(c);
        if (bal) {                                   // wave-uniform
          if constexpr (LAZY) gg = __shfl(g, src, 64);
          const int bb = __shfl(bound, src, 64);
          if (c) {
            const int pos = qn + __popcll(bal & ((1ull << lane) - 1ull));
            qk[w][pos] = key; qs[w][pos] = (uint32_t)s; qg[w][pos] = (uint32_t)gg; ql[w][pos] = bb;
          }
          qn += __popcll(bal);
          if (qn >= 64) drain(64);
        }
      }
    }
  }
  if (qn > 0) drain(qn);
  // blocks that are logical blocks of the batch (stream_records_kernel wants every one): as in section 7
  __shared__ uint32_t claimed_s;
  if (threadIdx.x == 0) claimed_s = 0;
  __syncthreads();
  if (lane == 0 && claimed) atomicAdd(&claimed_s, claimed);
  __syncthreads();
  if (threadIdx.x == 0 && claimed_s) atomicAdd(&ws.st_claimed[(blockIdx.x % CLAIM_SHARDS) * 32], claimed_s);
}

// A collecting pass that takes its pivots from the harvest buffer instead of a sample (kvc_schedule_params.harvest
// bit 2): st_seqrec as stream_pivot_kernel would have left it
__global__ __launch_bounds__(256) void seqrec_from_pivots_kernel(kvc_schedule_params p, SchedWs ws, const uint32_t* __restrict__ hv_pivot) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.num_seqs) return;
  SeqRec rec;
  rec.seq_pos = p.seq_positions[i]; rec.prot = p.num_protected[i]; rec.pad = 0u;
  rec.pivot_excl = p.evicted_blocks_per_seq[i] > 0 ? hv_pivot[i] : 0u;
  ws.st_seqrec[i] = rec;
}

__device__ __forceinline__ float key_to_float(uint32_t k) {      // inverse of float_to_key
  return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// The pivot for the next decode step from the R keys that are left of a sequence's lists (val(x), x < R: any order)
// once the selection has said what leaves -- see the head of this section.  Called by every thread of the workgroup.
template <typename ValF>
__device__ __forceinline__ uint32_t next_pivot_from_keys(uint32_t* hist, uint32_t* bc, uint32_t R, ValF val, int k, int bs,
                                                         uint32_t hang_sum, uint32_t used, float widen) {
  auto all = [&](int) { return true; };
  const double tgt = (double)k * bs + (double)hang_sum;
  const double want = ceil(tgt * (1.0 + (double)widen)) + 8.0;
  const uint32_t target = want < 4.0e9 ? (uint32_t)want : 0xFFFFFFFFu;
  uint32_t next;
  if (R >= target) {
    uint32_t P, r2, e2;
    block_radix_select(hist, bc, (int)R, target, val, all, P, r2, e2);
    next = P + 1u;                                   // (P < used <= KEY_INF)
  } else if (used >= KEY_INF) {
    next = KEY_INF;                                  // every evictable key was a candidate, and they are fewer than the target
  } else if (R < 2u || used == 0u) {
    next = used;                                     // (nothing to measure a density on)
  } else {
    const uint32_t m = max(R / 4u, 1u);
    uint32_t Kq, r2, e2;
    block_radix_select(hist, bc, (int)R, R - m, val, all, Kq, r2, e2);
    const float fH = key_to_float(used - 1u), fq = key_to_float(Kq);
    float df = (fH - fq) * ((float)(target - R) / (float)m) * 2.0f;
    if (!(df > 0.0f)) df = fabsf(fH) * 1e-3f + 1e-30f;           // (ties at the top of the list)
    const float fn = fH + df;
    next = (fn == fn && fn < __builtin_inff()) ? min(float_to_key(fn) + 1u, KEY_INF) : KEY_INF;
    if (next < used) next = used;
  }
  return next;
}

constexpr int HVP_LDS_KEYS = 12288;                  // keys of a sequence staged in LDS (more: read from L2 every round)

// One workgroup per sequence, behind seq_select_topk_kernel: the pivot for the harvest of the next
// decode step from what is left of this call's lists (see the head of this section).
//   used_pivot: the pivots the lists were made with (the harvest buffer's own for harvested lists
//   -- read before they are overwritten -- or nullptr: st_seqrec's, this call's collecting pass)
// (the body: harvest_pivot_kernel's, and topk_fused_kernel's last phase -- the LDS is the caller's)
struct PivotLds {
  uint32_t* hist;                                    // [RADIX], 16-byte aligned
  uint32_t* bc;                                      // [4]
  uint32_t* pre_s;                                   // [LH + 1] exclusive prefix of the heads' remaining entries
  uint16_t* start_s;                                 // [LH] first remaining entry of a head's record (= its evicted count)
  uint32_t* wsum_s;                                  // [16]
  uint32_t* hang_s;                                  // [1]
  uint32_t* keys_s;                                  // [keys_cap] a sequence's remaining keys (more: read from L2 every round)
  uint32_t keys_cap;
};
__device__ __forceinline__ void harvest_pivot_body(const kvc_schedule_params& p, const SchedWs& ws, uint32_t* hv_pivot, int from_harvest,
                                                   float widen, int i, const PivotLds& S) {
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  const int LH = p.num_layers * p.num_kv_heads, bs = p.block_size;
  if (tid == 0) *S.hang_s = 0;
  __syncthreads();
  {
    uint32_t r = 0, hs = 0;
    if (tid < LH) {                                  // LH <= 1024 = blockDim
      const int64_t g = (int64_t)i * LH + tid;
      const uint32_t C = min(ws.st_cnt[g], (uint32_t)KREC);
      const uint32_t cnt = min((uint32_t)max(p.evicted_kv_count[g], 0), C);
      r = C - cnt;
      S.start_s[tid] = (uint16_t)cnt;
      const uint32_t hang = (uint32_t)p.hanging_token_count[g];
      hs = hang >= 1u ? hang - 1u : 0u;
    }
    const uint32_t inc = wave_inclusive_scan_full(r);
    if (lane == WAVE - 1) S.wsum_s[w] = inc;
    hs = wave_reduce_sum_full(hs);
    if (lane == 0 && hs) atomicAdd(S.hang_s, hs);
    __syncthreads();
    uint32_t woff = 0;
    for (int q = 0; q < w; ++q) woff += S.wsum_s[q];
    if (tid < LH) S.pre_s[tid] = woff + inc - r;
    if (tid == LH - 1) S.pre_s[LH] = woff + inc;
  }
  __syncthreads();
  const uint32_t R = S.pre_s[LH];
  const int k = p.evicted_blocks_per_seq[i];
  const uint32_t used = from_harvest ? hv_pivot[i] : ws.st_seqrec[i].pivot_excl;
  if (k <= 0) {                                      // nothing asked of this sequence: lists made for nothing say nothing new
    if (tid == 0 && !from_harvest) hv_pivot[i] = 0u;
    return;
  }
  __syncthreads();                                   // (everybody has read the old pivot before it is written below)
  const uint32_t* pre_s = S.pre_s;
  const uint16_t* start_s = S.start_s;
  auto key_at = [&](uint32_t x) -> uint32_t {        // flat index x < R -> the key of that remaining entry
    int lo = 0, hi = LH;                             // pre_s[lo] <= x < pre_s[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (pre_s[mid] <= x) lo = mid; else hi = mid;
    }
    const int64_t e = ((int64_t)i * LH + lo) * KREC + start_s[lo] + (x - pre_s[lo]);
    // (past the L1: inside topk_fused_kernel the records were put in rank order by this very workgroup)
    return (uint32_t)(__atomic_load_n(ws.rec64 + e, __ATOMIC_RELAXED) >> 32);
  };
  uint32_t* keys_s = S.keys_s;
  const bool staged = R <= S.keys_cap;
  if (staged) {
    for (uint32_t x = tid; x < R; x += 1024u) keys_s[x] = key_at(x);
    __syncthreads();
  }
  auto val = [&](int x) -> uint32_t { return staged ? keys_s[x] : key_at((uint32_t)x); };
  const uint32_t next = next_pivot_from_keys(S.hist, S.bc, R, val, k, bs, *S.hang_s, used, widen);
  if (tid == 0) hv_pivot[i] = next;
}

__global__ __launch_bounds__(1024) void harvest_pivot_kernel(kvc_schedule_params p, SchedWs ws, uint32_t* hv_pivot,
                                                              int from_harvest, float widen) {
  __shared__ __attribute__((aligned(16))) uint32_t hist[RADIX];
  __shared__ uint32_t bc[4];
  __shared__ uint32_t pre_s[PIV_MAXLH + 1];
  __shared__ uint16_t start_s[PIV_MAXLH];
  __shared__ uint32_t wsum_s[16];
  __shared__ uint32_t hang_s;
  __shared__ __attribute__((aligned(16))) uint32_t keys_s[HVP_LDS_KEYS];
  const PivotLds S{hist, bc, pre_s, start_s, wsum_s, &hang_s, keys_s, (uint32_t)HVP_LDS_KEYS};
  harvest_pivot_body(p, ws, hv_pivot, from_harvest, widen, blockIdx.x, S);
}

}  // namespace kvc
