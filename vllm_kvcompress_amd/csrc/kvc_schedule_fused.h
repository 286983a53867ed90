// kvc_schedule_fused.h -- A3: records + selection + emission (+ the next pivots) of the small-eviction schedule in ONE launch
// (one translation unit: included by kvc_schedule.hip behind kvc_schedule_harvest.h; see the overview there)
#pragma once
#include "kvc_common.h"
#include "kvc_schedule_common.h"
#include "kvc_schedule_general.h"
#include "kvc_schedule_small.h"
#include "kvc_schedule_harvest.h"
#include "../../include/kvc_mi355x.h"

namespace kvc {

// ------------------------------------------------------------------ 7b. records + selection + emission (+ next pivots) in ONE launch
// On lists somebody else made (the aggregation pass, the attention's epilogue) or with remembered pivots the
// schedule is a handful of tiny dependent launches -- records 39 us, selection 22, emission 42, next pivots 20 at
// 65 536 heads of ~20 entries.  Those are not latency: rocprofv3 shows the bitonic sorts (42 ds_bpermute per 64-bit
// list, 21 per emitted list, every wave of a CU through the one LDS crossbar) as the cost.  Nothing here needs a SORTED
// list:
//   * a chunk threshold is the entry of RANK hang - 1 + c * bs, the evicted set is the entries of rank < cnt, the
//     emission wants them at their rank by LOGICAL index -- ranks, not orders;
//   * a list of C <= 64 entries is one 64-bit value per lane, and a lane's rank is the number of entries below its
//     own: C steps of v_readlane (a scalar broadcast: no LDS, no cross-lane network) + compare + add.
// One workgroup per sequence (16 waves, HPW heads per wave).  Phases:
//   ranks     two heads per loop (two independent chains for a wave's in-order issue), the lists requested a pair
//             ahead; the record goes back to global memory in RANK order (lane -> rec[rank]) and nothing of it stays
//             in registers; thresholds into a dense LDS list (~2 per head)
//   select    the k'-th smallest threshold by six radix rounds of the workgroup
//   counts    a lane per head (two coalesced stores per wave); the evicted entries are rec[0 .. ce): their slots,
//             then their logical block numbers, two rounds of loads for all the wave's heads
//   emit      rank by logical index among the evicted (lanes 0 .. ce - 1, all pairs, two heads per loop); the kept
//             output list's dirty map a lane per (head, map word) (eli_dirty_apply_owners)
//   pivots    (PIVOT) harvest_pivot_kernel's body: the next decode step's pivots from rec[ce .. C), staged in the
//             LDS the thresholds have left
// Lists beyond 64 entries (rare: a head that takes most of a sequence's eviction) are sorted through LDS as in
// stream_records_kernel / emit_topk_kernel.  For calls whose sequences do not need each other's counts (mode 1, one
// sequence: `coupled` 0 or 2 as in seq_select_topk_kernel) with at most 16 * HPW heads per sequence; everything else
// keeps the launch chain.  The kernel is bound by instruction issue, not by memory, LDS or the instruction cache
// (profiles/r5_topk_fused_phases.txt has the stamps of every version and the counters): what counts is instructions
// per head.
// Ranks against the entries of a lane's own ROW of 16, without a scalar broadcast: step J adds [src of the row's lane J <
// me] -- v_sub_co_u32 with a DPP row_newbcast operand leaves the borrow (= the comparison) in VCC, v_addc adds it: two
// VALU instructions per step and no scalar one (v_readlane + compare + add and the lane arithmetic around them are
// five).  Steps in groups of four while J < cnt (wave-uniform); all 64 lanes active.  (s_nop: a DPP operand written by
// the instruction before, and VCC written by a VALU instruction and read by the next, want two wait states on gfx950.)
#define KVC_DPP_STEP(J) "v_sub_co_u32_dpp %1, vcc, %2, %3 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" \
                        "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
#define KVC_DPP_STEP4(A, B, C, D)                                                                             \
  asm volatile("s_nop 1\n\t" KVC_DPP_STEP(A) KVC_DPP_STEP(B) KVC_DPP_STEP(C) KVC_DPP_STEP(D)                \
               : "+v"(r), "=&v"(tmp) : "v"(src), "v"(me) : "vcc")
__device__ __forceinline__ uint32_t rank_in_rows(uint32_t src, uint32_t me, int cnt, uint32_t r) {
  uint32_t tmp;
  if (cnt > 0) KVC_DPP_STEP4(0, 1, 2, 3);
  if (cnt > 4) KVC_DPP_STEP4(4, 5, 6, 7);
  if (cnt > 8) KVC_DPP_STEP4(8, 9, 10, 11);
  if (cnt > 12) KVC_DPP_STEP4(12, 13, 14, 15);
  return r;
}
// ... and against the 32 lanes of a lane's HALF of the wave (two lists of at most 32 entries in one register, one per
// half; the entries beyond a list's end hold 0xFFFFFFFF, below nobody): the own row, then the half's other row
__device__ __forceinline__ uint32_t rank_in_halves(uint32_t v, int n) {
  uint32_t r = rank_in_rows(v, v, n < 16 ? n : 16, 0u);
  if (n > 16) r = rank_in_rows((uint32_t)__shfl_xor((int)v, 16, 64), v, 16, r);
  return r;
}

__device__ __forceinline__ bool less64(uint32_t ahi, uint32_t alo, uint32_t bhi, uint32_t blo) {
  return ahi < bhi || (ahi == bhi && alo < blo);
}

template <int HPW, bool PIVOT>
__global__ __launch_bounds__(1024) void topk_fused_kernel(kvc_schedule_params p, SchedWs ws, int P2, int lazy, int coupled,
                                                          int lds_keys, uint32_t* hv_pivot, int from_harvest, float widen) {
  extern __shared__ __attribute__((aligned(16))) uint8_t sel_lds[];
  uint64_t* arr = reinterpret_cast<uint64_t*>(sel_lds);                 // [P2] recorded thresholds, dense
  uint32_t* cnt = reinterpret_cast<uint32_t*>(arr + P2);                // [16 * HPW] freed chunks per head
  __shared__ __attribute__((aligned(16))) uint64_t sort_s[4][KREC];     // lists beyond a wave (shared by four waves each)
  __shared__ uint32_t fsum_s, k_s, nthr_s, big_lock[4], flag_s, piv_total_s, piv_hs_s, piv_n_s;
  __shared__ unsigned long long vstar_s;
  __shared__ __attribute__((aligned(16))) uint32_t sel_hist[RADIX];
  __shared__ uint32_t sel_wtot[4];
  __shared__ uint32_t sel_digit, sel_krem;
  const int i = blockIdx.x, tid = threadIdx.x, lane = lane_id();
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = p.num_layers, H = p.num_kv_heads, B = p.num_seqs, LH = L * H;
  const int bs = p.block_size, MCH = KREC / bs;
  const int64_t gbase = (int64_t)i * LH;
  const int G = B * LH;
#ifdef KVC_TOPK_STAMPS
#define KVC_STAMP(n) do { if (i == 1 && tid == 64) reinterpret_cast<unsigned long long*>(ws.bar)[n] = wall_clock64(); } while (0)
#else
#define KVC_STAMP(n) do { } while (0)
#endif
  KVC_STAMP(0);
  if (i == 0 && w == 0) {                            // (as stream_records_kernel: blocks nobody claimed, lists of another batch)
    const uint32_t c = wave_reduce_sum_full(ws.st_claimed[lane * 32]);
    if (lane == 0 && (int64_t)c != p.total_slots / bs) atomicOr(ws.fallback, 1u);
    if (ws.hv_seen_seq != nullptr) {
      bool bad = false;
      for (int b = lane; b < B; b += WAVE)
        bad |= ws.hv_seen_seq[2 * b] != p.seq_positions[b] || ws.hv_seen_seq[2 * b + 1] != p.num_protected[b];
      if (__ballot(bad) && lane == 0) atomicOr(ws.fallback, 1u);
    }
  }
  if (tid == 0) { fsum_s = 0; nthr_s = 0; vstar_s = ~0ull; piv_total_s = 0; piv_hs_s = 0; piv_n_s = 0; }
  if (tid < 4) big_lock[tid] = 0;
  for (int lh = tid; lh < 16 * HPW; lh += 1024) cnt[lh] = 0;
  __syncthreads();
  const int kreq = p.evicted_blocks_per_seq[i];
  // ---- what lane q < HPW knows about head lh0 + q (all loads requested together)
  const int lh0 = w * HPW;
  uint32_t myC = 0, myHang = 0, myFc = 0;
  int32_t myOff = 0, myEnd = 0;                      // (offsets are int32 end to end: total slots < 2^31)
  const int sh = __ffs(bs) - 1;                      // block sizes 8 / 16 / 32: shifts, not divisions
  {
    const bool mine = lane < HPW && lh0 + lane < LH;
    const int lh = mine ? lh0 + lane : 0;
    const int64_t g = gbase + lh;
    const int l = lh / H, h = lh % H;
    const int ctx = p.context_lens[(l * B + i) * H + h];
    const uint32_t c_ = ws.st_cnt[g];
    const uint32_t hang_ = (uint32_t)p.hanging_token_count[g];
    const int32_t off_ = p.evicted_kv_offsets[g];
    const int32_t end_ = (g + 1 < G) ? p.evicted_kv_offsets[g + 1] : (int32_t)p.total_slots;
    const int seen_ = ws.hv_seen_ctx != nullptr ? ws.hv_seen_ctx[g] : ctx;
    const uint32_t def_ = lazy ? 0u : ws.st_def[g];
    if (mine) {
      myC = c_; myHang = hang_; myOff = off_; myEnd = end_;
      const uint32_t nblk = (uint32_t)((ctx + bs - 1) >> sh);
      if (seen_ != ctx && seen_ != -2) atomicOr(ws.fallback, 1u);   // another batch's list (-2: made without context lengths)
      if (!lazy) {
        myFc = nchunks_freed(nblk * (uint32_t)bs - def_, myHang, (uint32_t)bs);
        ws.head_fc[g] = myFc;
        ws.head_fc[(int64_t)G + g] = nblk;
      }
      if (myC > (uint32_t)KREC) atomicOr(ws.fallback, 1u);
    }
  }
  if (coupled == 0) {
    const uint32_t f = wave_reduce_sum_full(myFc);
    if (lane == 0 && f) atomicAdd(&fsum_s, f);
  }
  KVC_STAMP(1);
  // ---- records as ranks, two heads at a time; the lists are requested a pair ahead and nothing of them stays in
  // registers: the record goes back to global memory in rank order (what harvest_pivot_kernel behind reads, and what
  // the emission below reads its first entries from), the thresholds into LDS
  uint32_t bigmask = 0;                              // heads of this wave whose list went through LDS (64 < C <= KREC)
  auto load_list = [&](int q) -> uint64_t {
    const uint32_t C = (uint32_t)__builtin_amdgcn_readlane((int)myC, q);
    uint64_t x = ~0ull;                              // (a lane beyond the list's end: a key below nobody)
    if (C >= 1u && C <= (uint32_t)WAVE && (uint32_t)lane < C) x = ws.rec64[(gbase + lh0 + q) * KREC + lane];
    return x;
  };
  static_assert(HPW % 2 == 0, "heads are ranked two at a time");
  // two lists of at most 32 entries share ONE register: head q2 in lanes 0 .. 31, head q2 + 1 in lanes 32 .. 63
  auto load_pair = [&](int q2, uint64_t& x0, uint64_t& x1) {
    const uint32_t Ca = (uint32_t)__builtin_amdgcn_readlane((int)myC, q2), Cb = (uint32_t)__builtin_amdgcn_readlane((int)myC, q2 + 1);
    if (Ca <= 32u && Cb <= 32u) {                     // wave-uniform
      const int hqv = lane >> 5;
      x0 = ~0ull; x1 = ~0ull;
      if ((uint32_t)(lane & 31) < (hqv ? Cb : Ca)) x0 = ws.rec64[(gbase + lh0 + q2 + hqv) * KREC + (lane & 31)];
    } else {
      x0 = load_list(q2); x1 = load_list(q2 + 1);
    }
  };
  uint64_t xcur[2];
  load_pair(0, xcur[0], xcur[1]);
#pragma unroll
  for (int q2 = 0; q2 < HPW; q2 += 2) {
    uint64_t xnext[2] = {~0ull, ~0ull};
    if (q2 + 2 < HPW) load_pair(q2 + 2, xnext[0], xnext[1]);
    bool pair_done = false;
    {
      const uint32_t Ca = (uint32_t)__builtin_amdgcn_readlane((int)myC, q2), Cb = (uint32_t)__builtin_amdgcn_readlane((int)myC, q2 + 1);
      if (Ca <= 32u && Cb <= 32u) {                   // the packed pair: ranks through DPP row broadcasts
        const uint32_t hqv = (uint32_t)lane >> 5, e = (uint32_t)lane & 31u;
        const uint32_t C = hqv ? Cb : Ca;
        const uint32_t key = (uint32_t)(xcur[0] >> 32);
        const bool in = e < C;
        uint32_t r = 0;
        if ((Ca | Cb) != 0u) r = rank_in_halves(key, (int)max(Ca, Cb));
        if (wave_reduce_sum_full(in ? r : 0u) == Ca * (Ca - 1u) / 2u + Cb * (Cb - 1u) / 2u) {    // wave-uniform: no two keys tie
          pair_done = true;
          const uint32_t hang = hqv ? (uint32_t)__builtin_amdgcn_readlane((int)myHang, q2 + 1) : (uint32_t)__builtin_amdgcn_readlane((int)myHang, q2);
          const uint32_t lh = (uint32_t)(lh0 + q2) + hqv;
          if (in && kreq > 0) ws.rec64[(gbase + lh) * KREC + r] = xcur[0];
          const bool thr = in && kreq > 0 && hang >= 1u && r + 1u >= hang && ((r + 1u - hang) & (uint32_t)(bs - 1)) == 0u &&
                           ((r + 1u - hang) >> sh) < (uint32_t)MCH;
          const unsigned long long tm = __ballot(thr);
          if (tm) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&nthr_s, (uint32_t)__popcll(tm));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (thr) arr[base + __popcll(tm & ((1ull << lane) - 1ull))] = ((uint64_t)key << 32) | (lh * (uint32_t)MCH + ((r + 1u - hang) >> sh));
          }
        } else {
          // (two entries share a key: the pair goes through the general form below, a list per register)
          const uint32_t lo = (uint32_t)xcur[0], hi = (uint32_t)(xcur[0] >> 32);
          const uint32_t lo2 = (uint32_t)__shfl((int)lo, lane + 32, 64), hi2 = (uint32_t)__shfl((int)hi, lane + 32, 64);
          xcur[1] = lane < 32 ? (((uint64_t)hi2 << 32) | lo2) : ~0ull;
          if (lane >= 32) xcur[0] = ~0ull;
        }
      }
    }
    if (!pair_done) {
    const uint32_t plo[2] = {(uint32_t)xcur[0], (uint32_t)xcur[1]};
    const uint32_t phi[2] = {(uint32_t)(xcur[0] >> 32), (uint32_t)(xcur[1] >> 32)};
    // ranks by key of TWO heads in one loop: two independent chains of readlane -> compare -> add for the in-order
    // issue of a wave that shares its SIMD with three others (the loop runs to the longer list: a lane beyond a list's
    // end holds 0xFFFFFFFF, which is below nobody)
    uint32_t rpair[2] = {0u, 0u};
    {
      const uint32_t Ca = (uint32_t)__builtin_amdgcn_readlane((int)myC, q2), Cb = (uint32_t)__builtin_amdgcn_readlane((int)myC, q2 + 1);
      const int n = (int)max(Ca <= (uint32_t)WAVE ? Ca : 0u, Cb <= (uint32_t)WAVE ? Cb : 0u);
      const uint32_t ma = phi[0], mb = phi[1];
      uint32_t ra = 0, rb = 0;
      int j = 0;
      for (; j + 4 <= n; j += 4) {                     // (unrolled by hand: the compiler will not unroll around readlane)
        const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)ma, j), a1 = (uint32_t)__builtin_amdgcn_readlane((int)ma, j + 1);
        const uint32_t a2 = (uint32_t)__builtin_amdgcn_readlane((int)ma, j + 2), a3 = (uint32_t)__builtin_amdgcn_readlane((int)ma, j + 3);
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)mb, j), b1 = (uint32_t)__builtin_amdgcn_readlane((int)mb, j + 1);
        const uint32_t b2 = (uint32_t)__builtin_amdgcn_readlane((int)mb, j + 2), b3 = (uint32_t)__builtin_amdgcn_readlane((int)mb, j + 3);
        ra += (a0 < ma ? 1u : 0u) + (a1 < ma ? 1u : 0u) + (a2 < ma ? 1u : 0u) + (a3 < ma ? 1u : 0u);
        rb += (b0 < mb ? 1u : 0u) + (b1 < mb ? 1u : 0u) + (b2 < mb ? 1u : 0u) + (b3 < mb ? 1u : 0u);
      }
      for (; j < n; ++j) {
        ra += (uint32_t)__builtin_amdgcn_readlane((int)ma, j) < ma ? 1u : 0u;
        rb += (uint32_t)__builtin_amdgcn_readlane((int)mb, j) < mb ? 1u : 0u;
      }
      rpair[0] = ra; rpair[1] = rb;
    }
#pragma unroll
    for (int hq = 0; hq < 2; ++hq) {
    const int q = q2 + hq;
    const uint32_t C = (uint32_t)__builtin_amdgcn_readlane((int)myC, q);
    const uint32_t hang = (uint32_t)__builtin_amdgcn_readlane((int)myHang, q);
    const int lh = lh0 + q;
    if (C == 0u || C > (uint32_t)KREC) continue;                                // wave-uniform
    uint64_t* rec = ws.rec64 + (gbase + lh) * KREC;
    if (C <= (uint32_t)WAVE) {
      // the key (high word) decides almost always: ranked by it alone above, and again with the slot as tie-break only
      // if two entries of the list share a key -- which shows in the sum of the ranks:
      // sum_i #{j : key_j < key_i} = C (C - 1) / 2 less the tied pairs
      uint32_t r = rpair[hq];
      const bool in = (uint32_t)lane < C;
      if (wave_reduce_sum_full(in ? r : 0u) != C * (C - 1u) / 2u) {            // wave-uniform
        r = 0;
        for (int j = 0; j < (int)C; ++j) {
          const uint32_t ohi = (uint32_t)__builtin_amdgcn_readlane((int)phi[hq], j);
          const uint32_t olo = (uint32_t)__builtin_amdgcn_readlane((int)plo[hq], j);
          r += less64(ohi, olo, phi[hq], plo[hq]) ? 1u : 0u;
        }
      }
      if (in && kreq > 0) rec[r] = xcur[hq];           // (a sequence nothing is asked of emits nothing and leaves no pivot)
      // thresholds sit at ranks hang - 1 + c * bs
      const bool thr = in && kreq > 0 && hang >= 1u && r + 1u >= hang && ((r + 1u - hang) & (uint32_t)(bs - 1)) == 0u &&
                       ((r + 1u - hang) >> sh) < (uint32_t)MCH;
      const unsigned long long tm = __ballot(thr);
      if (tm) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&nthr_s, (uint32_t)__popcll(tm));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (thr) {
          const uint32_t e = (uint32_t)lh * (uint32_t)MCH + ((r + 1u - hang) >> sh);
          arr[base + __popcll(tm & ((1ull << lane) - 1ull))] = ((uint64_t)phi[hq] << 32) | e;
        }
      }
    } else {
      bigmask |= 1u << q;
      // (rare) a list beyond a wave: sorted in LDS, one of four buffers, taken with a spin lock by the wave
      const int bq = w & 3;
      if (lane == 0) while (atomicCAS(&big_lock[bq], 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(2);
      wave_lds_sync();
      uint64_t* a = sort_s[bq];
      const int SZ = C <= 128u ? 128 : 256;
      for (int j = lane; j < SZ; j += WAVE) a[j] = (uint32_t)j < C ? rec[j] : ~0ull;
      wave_lds_sync();
      if (SZ == 128) wave_bitonic_sort<uint64_t, 128>(a);
      else wave_bitonic_sort<uint64_t, 256>(a);
      for (int j0 = 0; j0 < (int)C; j0 += WAVE) {
        const int j = j0 + lane;
        const bool in = j < (int)C;
        if (in) rec[j] = a[j];
        const bool thr = in && kreq > 0 && hang >= 1u && (uint32_t)j + 1u >= hang && ((uint32_t)j + 1u - hang) % (uint32_t)bs == 0u &&
                         ((uint32_t)j + 1u - hang) / (uint32_t)bs < (uint32_t)MCH;
        const unsigned long long tm = __ballot(thr);
        if (tm) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&nthr_s, (uint32_t)__popcll(tm));
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          if (thr) {
            const uint32_t e = (uint32_t)lh * (uint32_t)MCH + ((uint32_t)j + 1u - hang) / (uint32_t)bs;
            arr[base + __popcll(tm & ((1ull << lane) - 1ull))] = (a[j] & 0xFFFFFFFF00000000ull) | e;
          }
        }
      }
      wave_lds_sync();
      if (lane == 0) atomicExch(&big_lock[bq], 0u);
    }
    }
    }
    xcur[0] = xnext[0]; xcur[1] = xnext[1];
  }
  KVC_STAMP(2);
  __syncthreads();
  KVC_STAMP(3);
  // ---- selection: k' and the k'-th smallest recorded threshold by (threshold, head, chunk)
  const uint32_t nthr = nthr_s;
  if (tid == 0) {
    uint32_t kk = kreq <= 0 ? 0u : (uint32_t)kreq;
    if (coupled == 0 && kk > fsum_s) kk = fsum_s;
    k_s = kk;
    ws.seq_k[i] = (int32_t)kk;
  }
  __syncthreads();
  const uint32_t k = k_s;
  if (k > nthr) {                                    // the records do not list k' thresholds
    if (tid == 0) atomicOr(ws.fallback, 1u);
  } else if (k > 0) {
    // MSB-first radix select over the dense list (a few hundred 64-bit (threshold, head * MCH + chunk) values: one
    // pass of the workgroup per round); bytes 2 and 3 of the low word are zero (head * MCH + chunk < 2^16): six rounds
    uint64_t prefix = 0;
    uint32_t krem = k;
    for (int round = 0; round < 6; ++round) {
      const int shift = round < 4 ? 56 - 8 * round : 8 * (5 - round);
      const int pshift = round < 4 ? shift + 8 : (round == 4 ? 16 : 8);       // bits above the digit = the prefix so far
      if (tid < RADIX) sel_hist[tid] = 0;
      __syncthreads();
      for (uint32_t e0 = 0; e0 < nthr; e0 += 1024u) {      // uniform trip count (ballots inside)
        const uint32_t e = e0 + (uint32_t)tid;
        const uint64_t x = e < nthr ? arr[e] : 0ull;
        const bool in = e < nthr && (round == 0 || (x >> pshift) == prefix);
        hist_add(sel_hist, in, (uint32_t)(x >> shift) & 0xFFu);
      }
      __syncthreads();
      uint32_t c = 0, inc = 0;
      if (tid < RADIX) {
        c = sel_hist[tid];
        inc = wave_inclusive_scan_full(c);
        if ((tid & 63) == 63) sel_wtot[tid >> 6] = inc;
      }
      __syncthreads();
      if (tid < RADIX) {
        uint32_t off = 0;
        for (int qq = 0; qq < (tid >> 6); ++qq) off += sel_wtot[qq];
        const uint32_t incl = off + inc, excl = incl - c;
        if (krem > excl && krem <= incl) { sel_digit = (uint32_t)tid; sel_krem = krem - excl; }
      }
      __syncthreads();
      prefix = round == 3 ? (((prefix << 8) | sel_digit) << 16) : ((prefix << 8) | sel_digit);   // (skips the two zero bytes)
      krem = sel_krem;
    }
    if (tid == 0) vstar_s = prefix;
    for (uint32_t e = tid; e < nthr; e += 1024u) {
      const uint64_t x = arr[e];
      if (x <= prefix) atomicAdd(&cnt[(uint32_t)x / (uint32_t)MCH], 1u);
    }
  }
  if (tid == 0) flag_s = *reinterpret_cast<volatile uint32_t*>(ws.fallback);
  __syncthreads();
  KVC_STAMP(4);
  if (tid == 0) ws.seq_prefix[i] = (k > 0 && k <= nthr) ? (uint32_t)(vstar_s >> 32) : 0u;
  // ---- counts and emission: the wave that holds a head's entries emits them
  const bool flagged = flag_s != 0u;                 // (somebody's lists fell short: the general pipeline behind rewrites everything)
  uint32_t ceV = 0;                                  // lane q < HPW: head lh0 + q's number of evicted entries
  {                                                  // counts out: a lane per head, two coalesced stores per wave
    const bool mine = lane < HPW && lh0 + lane < LH;
    uint32_t n = 0;
    if (mine && k > 0 && k <= nthr) n = cnt[lh0 + lane];
    ceV = n > 0 ? (n - 1u) * (uint32_t)bs + myHang : 0u;
    if (mine) {
      p.evicted_block_count[gbase + lh0 + lane] = (int32_t)n;
      p.evicted_kv_count[gbase + lh0 + lane] = (int32_t)ceV;
    }
  }
  // The evicted entries are the first ce of the head's record, which the ranks phase left in rank order (written by
  // this wave before the barriers above; read past the L1 all the same).  Two heads that evict at most 32 entries each
  // (nearly all) share ONE register, as the lists did: head q2 in lanes 0 .. 31 of u2[q2 / 2], head q2 + 1 in lanes
  // 32 .. 63; 0xFFFFFFFF = no entry.  Two loops: every pair's slots requested, then every pair's logical block numbers
  // -- two round trips to the L2 for the wave, not two per head.  The other pairs (more than 32 evicted entries of a
  // head, a list that went through LDS) are emitted head by head below.
  constexpr int NP = HPW / 2;
  uint32_t u2[NP];
  uint32_t packmask = 0;                             // bit pr: pair pr is packed
  const uint32_t hqv = (uint32_t)lane >> 5, el = (uint32_t)lane & 31u;
#pragma unroll
  for (int pr = 0; pr < NP; ++pr) {
    u2[pr] = 0xFFFFFFFFu;
    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)ceV, 2 * pr), c1 = (uint32_t)__builtin_amdgcn_readlane((int)ceV, 2 * pr + 1);
    if (flagged || (bigmask & (3u << (2 * pr))) || c0 > 32u || c1 > 32u) continue;        // wave-uniform
    packmask |= 1u << pr;
    if (lh0 + 2 * pr + (int)hqv < LH && el < (hqv ? c1 : c0))
      u2[pr] = __atomic_load_n(reinterpret_cast<const uint32_t*>(ws.rec64 + (gbase + lh0 + 2 * pr + hqv) * KREC + el), __ATOMIC_RELAXED);
  }
#pragma unroll
  for (int pr = 0; pr < NP; ++pr) {
    const uint32_t slot = u2[pr];
    if (slot != 0xFFFFFFFFu) u2[pr] = ((uint32_t)p.logical_block_num_by_block[slot >> sh] << sh) | (slot & (uint32_t)(bs - 1));
  }
  KVC_STAMP(5);
  if (!flagged) {
  const bool tracked = p.eli_dirty_map != nullptr && !(p.lean & 1);
  // the dirty map's words of all the wave's heads, a lane per (head, word); false: a head of more than 64 words
  // (2048 blocks) -- those take the per-head form below
  const int nheads = min(HPW, LH - lh0);
  const bool dirty_done = !tracked || nheads <= 0 ||
                          eli_dirty_apply_owners(p.eli_dirty_map, p.evicted_logical_indices, myOff, myEnd, ceV, nheads, sh, p.null_value, lane);
#pragma unroll
  for (int pr = 0; pr < NP; ++pr) {
    const int q2 = 2 * pr;
    if (lh0 + q2 >= LH) break;                        // wave-uniform
    if (packmask >> pr & 1u) {
      // both heads' evicted entries in one register: ranks by logical index through DPP row broadcasts, one store
      const uint32_t me = u2[pr];
      const unsigned long long vm = __ballot(me != 0xFFFFFFFFu);
      const int n = max(__popc((uint32_t)vm), __popc((uint32_t)(vm >> 32)));
      if (n > 0) {
        const uint32_t r2 = rank_in_halves(me, n);
        const int32_t off2 = hqv ? __builtin_amdgcn_readlane(myOff, q2 + 1) : __builtin_amdgcn_readlane(myOff, q2);
        if (me != 0xFFFFFFFFu) p.evicted_logical_indices[off2 + (int32_t)r2] = (int32_t)me;
      }
    }
#pragma unroll
    for (int hq = 0; hq < 2; ++hq) {
      const int q = q2 + hq;
      const int lh = lh0 + q;
      if (lh >= LH) break;                            // wave-uniform
      const int64_t g = gbase + lh;
      const uint32_t ce = (uint32_t)__builtin_amdgcn_readlane((int)ceV, q);
      const int32_t off = __builtin_amdgcn_readlane(myOff, q), end = __builtin_amdgcn_readlane(myEnd, q);
      if (!dirty_done)
        eli_dirty_update(p.eli_dirty_map, p.evicted_logical_indices, off >> sh, end >> sh, (int64_t)((ce + (uint32_t)bs - 1u) >> sh),
                         (int64_t)off + ce, bs, p.null_value, true, lane, WAVE);
      if (ce == 0 || (packmask >> pr & 1u)) continue;
      int32_t* out = p.evicted_logical_indices + off;
      const uint64_t* rec = ws.rec64 + g * KREC;
      if (!(bigmask & (1u << q))) {
        // (rare) a head that evicts 33 .. 64 entries: its slots, their logical block numbers, ranks by readlane
        uint32_t me = 0xFFFFFFFFu;
        if ((uint32_t)lane < ce) me = logical_of(p, __atomic_load_n(reinterpret_cast<const uint32_t*>(rec + lane), __ATOMIC_RELAXED));
        uint32_t r2 = 0;
        for (int j = 0; j < (int)min(ce, (uint32_t)WAVE); ++j) r2 += (uint32_t)__builtin_amdgcn_readlane((int)me, j) < me ? 1u : 0u;
        if ((uint32_t)lane < ce) out[r2] = (int32_t)me;
      } else {
        // (this wave wrote the sorted record to global memory above: read past the L1)
        const int bq = w & 3;
        if (lane == 0) while (atomicCAS(&big_lock[bq], 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(2);
        wave_lds_sync();
        uint32_t* a = reinterpret_cast<uint32_t*>(sort_s[bq]);
        for (int j = lane; j < KREC; j += WAVE)
          a[j] = (uint32_t)j < ce ? logical_of(p, (uint32_t)__atomic_load_n(rec + j, __ATOMIC_RELAXED)) : 0xFFFFFFFFu;
        wave_lds_sync();
        wave_bitonic_sort<uint32_t, KREC>(a);
        for (int j = lane; j < (int)ce; j += WAVE) out[j] = (int32_t)a[j];
        wave_lds_sync();
        if (lane == 0) atomicExch(&big_lock[bq], 0u);
      }
    }
  }
  }
  KVC_STAMP(6);
  // ---- the pivots for the next decode step's harvest from what is left of the lists (section 10): harvest_pivot_kernel's
  // body as this kernel's last phase -- the records are in rank order in global memory, the counts are out, and the
  // LDS of the thresholds is free for the keys (lds_keys: what the host sized the dynamic region for)
  if constexpr (PIVOT) {
    if (hv_pivot != nullptr) {
      __shared__ uint32_t piv_pre_s[16 * HPW + 1];
      __shared__ uint16_t piv_start_s[16 * HPW];
      __shared__ uint32_t piv_wsum_s[16], piv_hang_s;
      // what is left of the wave's lists, and its heads' hanging tokens (hang - 1 each), summed for the workgroup
      {
        const bool mine = lane < HPW && lh0 + lane < LH;
        const uint32_t Cc = min(myC, (uint32_t)KREC);
        const uint32_t rem = wave_reduce_sum_full(mine ? Cc - min(ceV, Cc) : 0u);
        const uint32_t hs = wave_reduce_sum_full(mine && myHang >= 1u ? myHang - 1u : 0u);
        if (lane == 0) { atomicAdd(&piv_total_s, rem); atomicAdd(&piv_hs_s, hs); }
      }
      const uint32_t used = from_harvest ? hv_pivot[i] : ws.st_seqrec[i].pivot_excl;
      __syncthreads();                               // (every wave is done with the thresholds' counts; the counts and records are out)
      const uint32_t R = piv_total_s;
      if (kreq <= 0) {                               // nothing asked of this sequence: lists made for nothing say nothing new
        if (tid == 0 && !from_harvest) hv_pivot[i] = 0u;
      } else if (R <= (uint32_t)lds_keys) {
        // every wave stages what is left of ITS heads' records (rank order: the entries from the evicted count on) in the
        // LDS the thresholds have left -- the lists of a pair of short heads through one load, as they were ranked
        uint32_t* keys_s = reinterpret_cast<uint32_t*>(sel_lds);
        auto put = [&](bool valid, uint32_t key) {
          const unsigned long long m = __ballot(valid);
          if (m) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&piv_n_s, (uint32_t)__popcll(m));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (valid) keys_s[base + __popcll(m & ((1ull << lane) - 1ull))] = key;
          }
        };
        const uint32_t hq5 = (uint32_t)lane >> 5, e5 = (uint32_t)lane & 31u;
        uint32_t kq[HPW / 2];
        uint32_t shortmask = 0;
#pragma unroll
        for (int pr = 0; pr < HPW / 2; ++pr) {         // the pairs of short lists: all their loads in flight together
          kq[pr] = 0xFFFFFFFFu;
          const uint32_t Ca = (uint32_t)__builtin_amdgcn_readlane((int)myC, 2 * pr), Cb = (uint32_t)__builtin_amdgcn_readlane((int)myC, 2 * pr + 1);
          if (Ca > 32u || Cb > 32u) continue;          // wave-uniform
          shortmask |= 1u << pr;
          const uint32_t c0 = min((uint32_t)__builtin_amdgcn_readlane((int)ceV, 2 * pr), Ca), c1 = min((uint32_t)__builtin_amdgcn_readlane((int)ceV, 2 * pr + 1), Cb);
          if (e5 >= (hq5 ? c1 : c0) && e5 < (hq5 ? Cb : Ca))
            kq[pr] = (uint32_t)(__atomic_load_n(ws.rec64 + (gbase + lh0 + 2 * pr + hq5) * KREC + e5, __ATOMIC_RELAXED) >> 32);
        }
#pragma unroll
        for (int pr = 0; pr < HPW / 2; ++pr) {
          if (shortmask >> pr & 1u) {
            // (a remaining key is never 0xFFFFFFFF: candidates lie below a pivot <= KEY_INF)
            put(kq[pr] != 0xFFFFFFFFu, kq[pr]);
          } else {
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
              const int q = 2 * pr + hq;
              const uint32_t C = min((uint32_t)__builtin_amdgcn_readlane((int)myC, q), (uint32_t)KREC);
              const uint32_t c = min((uint32_t)__builtin_amdgcn_readlane((int)ceV, q), C);
              for (uint32_t j0 = 0; j0 < C; j0 += WAVE) {      // wave-uniform
                const uint32_t j = j0 + (uint32_t)lane;
                const bool valid = j >= c && j < C;
                uint32_t key = 0;
                if (valid) key = (uint32_t)(__atomic_load_n(ws.rec64 + (gbase + lh0 + q) * KREC + j, __ATOMIC_RELAXED) >> 32);
                put(valid, key);
              }
            }
          }
        }
        __syncthreads();
        auto val = [&](int x) -> uint32_t { return keys_s[x]; };
        const uint32_t next = next_pivot_from_keys(sel_hist, sel_wtot, R, val, kreq, bs, piv_hs_s, used, widen);
        if (tid == 0) hv_pivot[i] = next;
      } else {
        // (more keys than the LDS holds: harvest_pivot_kernel's body, which reads them from the L2 every round)
        const PivotLds S{sel_hist, sel_wtot, piv_pre_s, piv_start_s, piv_wsum_s, &piv_hang_s, reinterpret_cast<uint32_t*>(sel_lds),
                         (uint32_t)lds_keys};
        harvest_pivot_body(p, ws, hv_pivot, from_harvest, widen, i, S);
      }
    }
  }
  KVC_STAMP(7);
}

}  // namespace kvc
