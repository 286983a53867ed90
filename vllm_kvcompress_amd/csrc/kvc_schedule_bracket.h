// kvc_schedule_bracket.h -- A3 schedule_evictions: the bracket schedule for bulk evictions
// (one translation unit: included by kvc_schedule.hip in this order; see the overview there)
#pragma once
#include "kvc_common.h"
#include "kvc_schedule_common.h"
#include "../../include/kvc_mi355x.h"

namespace kvc {

// ------------------------------------------------------------------ 9. bracket schedule (bulk evictions)
// The digit rounds of the general pipeline read every key four times to find T*, the k'-th smallest
// chunk threshold of a sequence, although a SAMPLE of the keys already says where T* lies to within
// a percent of the keys: with n_g = floor((R_g - hang_g) / bs) + 1 chunks freed by R_g keys, the
// keys at or below T* number k' * bs + sum(hang) - LH * (bs + 1) / 2 give or take LH * bs / 2,
// whatever the heads look like.  So (the reference's batch > 1 rule: bracket_totals_kernel below):
//   * build_keys leaves a sample behind: the sequence's slots in <= 32 Ki cells of 2^k slots, one
//     hashed slot per cell (sample_keys: four instructions and a hash in a pass that waits for HBM);
//   * bracket_kernel (a workgroup per sequence): the sample in registers, two order statistics of it
//     -> [lo, hi] around T*: the rank above -+ (4.5 sigma of the sample + 8), a block and a half per
//     head further down so that every head's last freed threshold is listed too;
//   * count_collect_kernel: ONE pass over the keys (logical order, as the histograms take them):
//     per head the keys below lo are counted, the keys inside the bracket go to the head's list
//     (LDS queue, one returning atomic per head and 64 entries, nobody waiting for it);
//   * bracket_records_kernel (a workgroup per head): the list, sorted (buckets over the bracket's
//     range: five barriers); the thresholds inside the bracket are every bs-th entry from the first
//     rank >= `below` that is a threshold rank, copied side by side for the next kernel;
//   * bracket_select_kernel (a workgroup per sequence): thresholds below the bracket are freed for
//     sure; the (k' - sure)-th smallest listed threshold is T* (digit rounds in LDS over the
//     bracket's range); per-head counts, ties in (head, chunk) order as finalize_body hands them out;
//   * select_emit with M = the cnt-th smallest key read off the sorted list: no digit rounds.
// keys 8 + 4 B, one counting pass 4 B, emit 4 + 4 B per slot instead of 40; 7 launches instead of
// 10, none of them a memset.  Exact whenever T* lies inside the bracket -- checked: sure < k' <=
// sure + listed, lists within their capacity (a head whose M lies below the bracket selects it from
// its keys) -- else the flag is raised and the digit rounds run (the single gated launch of section
// 8, over the keys that exist already).  Measured (MI355X, S1 of one call): config 2 (256 heads x
// 32 Ki) 189 -> 126 us, config 5 (256 x 64 Ki, bs 32) 299 -> 192, 8 x config 2 938 -> 687,
// 1 x 256 heads x 1 Ki 111 -> 57, config 4's shape 4 x 640 heads x 16 Ki 700 -> 476.
// What the kernels that are ONE workgroup per sequence cost was found with phase stamps
// (-DKVC_BR_STAMPS, tools/bracket_stamps.py), and three of the findings are general:
//   * LDS adds to one address serialise at about a lane per 8 cycles: histograms of metric keys
//     (top byte = sign and seven exponent bits) must not be taken on the raw digits -- the rounds
//     run on (key - min) << clz(max - min) (bracket_kernel 48 -> 33 us, bracket_select 29 -> 24);
//   * one CU moves ~100 GB/s: 13.8 k thresholds at a 64-byte stride were 9 us of the selection
//     kernel; the per-head kernel now leaves them side by side (2.8 us);
//   * a 55-step bitonic network over 1024 LDS keys is 12 us even with wave-local steps ordered by
//     wave barriers; a bucket sort over the bracket's range is 3.3 us.
struct BrRec { int32_t seq_pos, prot; uint32_t lo, hi; };    // (same slot as SeqRec: st_seqrec)

// -DKVC_BR_STAMPS (experiment builds, tools/bracket_stamps.py): workgroup 0 of the per-sequence
// kernels leaves the 100 MHz wall clock of its phases in head_fc (unused by this schedule)
#ifdef KVC_BR_STAMPS
#define BR_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) ws.head_fc[k] = (uint32_t)wall_clock64(); } while (0)
#else
#define BR_STAMP(k) do { } while (0)
#endif

// Digit histogram of a 1024-thread workgroup's values in wave-private LDS tables (plain LDS adds:
// the digits of a bracket are spread; the leader election of hist_add costs more than the
// conflicts it saves here), summed into hist[256].  PRIV_STRIDE = 257 words: the same digit of
// different waves lies in different banks, and so do neighbouring digits of one wave in the sum.
constexpr int PRIV_STRIDE = RADIX + 1;
constexpr int PRIV_WORDS = 16 * PRIV_STRIDE;
__device__ __forceinline__ void priv_clear(uint32_t* priv, int sets) {
  for (int j = threadIdx.x; j < sets * PRIV_WORDS; j += 1024) priv[j] = 0u;
}
__device__ __forceinline__ void priv_sum(const uint32_t* priv, uint32_t* hist, int sets) {
  const int tid = threadIdx.x;
  if (tid < sets * RADIX) {
    const uint32_t* src = priv + (tid >> 8) * PRIV_WORDS + (tid & 255);
    uint32_t t = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += src[q * PRIV_STRIDE];
    hist[tid] = t;
  }
}
// digit of the rank-th (1-based) entry of hist[256], the count below that digit and the digit's own,
// by one wave -> bc[0], bc[1], bc[2]
__device__ __forceinline__ void wave_pick_digit(const uint32_t* hist, uint32_t rank, uint32_t* bc) {
  const int l = lane_id();
  uint4 q = reinterpret_cast<const uint4*>(hist)[l];
  q.y += q.x; q.z += q.y; q.w += q.z;
  const uint32_t inc = wave_inclusive_scan(q.w);
  const uint32_t ex = inc - q.w;
  const uint32_t c[4] = {q.x + ex, q.y + ex, q.z + ex, q.w + ex};
  uint32_t prev = ex;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (prev < rank && rank <= c[t]) { bc[0] = (uint32_t)l * 4u + (uint32_t)t; bc[1] = prev; bc[2] = c[t] - prev; }
    prev = c[t];
  }
}

// wave-wide minimum / maximum in every lane's reach (lane 63 holds it, read back as a scalar): row
// rotations and the two row broadcasts of GFX9's DPP instead of six LDS-routed shuffles
template <bool MAX>
__device__ __forceinline__ uint32_t wave_reduce_minmax(uint32_t v) {
  auto op = [](uint32_t x, uint32_t y) { return MAX ? max(x, y) : min(x, y); };
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xF, 0xF, false));   // row_ror:4
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xF, 0xF, false));   // row_ror:8
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// the rank_a-th and rank_b-th smallest (1-based, rank_a <= rank_b <= their number) evictable keys
// among the R x 1024 register-resident keys of a 1024-thread workgroup: digit rounds on the
// registers, both ranks at once (they share the histogram as long as they share the prefix) -- or
// rather a value at most the one, at least the other and at most four sample keys off: the rounds
// stop when the buckets are that small.
// The digits of a metric key are badly spread (a sign, an exponent: most keys share the top byte, and
// LDS adds to one address serialise): the rounds run on (key - min) << clz(max - min) instead, as
// many of them as max - min has bytes.  finmask: which of the thread's keys are evictable; kmin, kmax:
// the thread's own extremes of those.  The keys are overwritten.
template <int R>
__device__ __forceinline__ void reg_rank_select2(uint32_t (&key)[R], uint32_t finmask, uint32_t kmin, uint32_t kmax,
                                                 uint32_t rank_a, uint32_t rank_b,
                                                 uint32_t* priv /*[2][PRIV_WORDS]*/, uint32_t* hist /*[2][RADIX]*/,
                                                 uint32_t* bc /*[6]*/, uint32_t& out_a, uint32_t& out_b, SchedWs& ws) {
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  kmin = wave_reduce_minmax<false>(kmin);
  kmax = wave_reduce_minmax<true>(kmax);
  __syncthreads();
  if (lane == 0) { hist[w] = kmin; hist[16 + w] = kmax; }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) { kmin = min(kmin, hist[q]); kmax = max(kmax, hist[16 + q]); }
  __syncthreads();
  if (kmin >= kmax) { out_a = kmin; out_b = kmin; return; }             // (uniform)
  BR_STAMP(24);
  const int sh = __builtin_clz(kmax - kmin);
  const int rounds = (32 - sh + 7) / 8;
#pragma unroll
  for (int r = 0; r < R; ++r) key[r] = (key[r] - kmin) << sh;
  uint32_t pa = 0, pb = 0;
  int done = 0;
  for (int round = 0; round < rounds; ++round) {
    const int shift = 24 - 8 * round;
    const bool split = pa != pb;                     // (uniform)
    priv_clear(priv, split ? 2 : 1);
    __syncthreads();
    uint32_t* ha = priv + w * PRIV_STRIDE;
    uint32_t* hb = priv + PRIV_WORDS + w * PRIV_STRIDE;
    if (!split) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (((finmask >> r) & 1u) && (round == 0 || (key[r] >> (shift + 8)) == pa)) atomicAdd(&ha[(key[r] >> shift) & 0xFFu], 1u);
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t top = key[r] >> (shift + 8);
        if (((finmask >> r) & 1u) && (top == pa || top == pb))
          atomicAdd(&(top == pa ? ha : hb)[(key[r] >> shift) & 0xFFu], 1u);
      }
    }
    BR_STAMP(25 + 4 * round);
    __syncthreads();
    BR_STAMP(26 + 4 * round);
    priv_sum(priv, hist, split ? 2 : 1);
    __syncthreads();
    if (w == 0) wave_pick_digit(hist, rank_a, bc);
    if (w == 1) wave_pick_digit(hist + (split ? RADIX : 0), rank_b, bc + 3);
    __syncthreads();                                 // (bc is next written three barriers on)
    pa = (pa << 8) | bc[0]; rank_a -= bc[1];
    pb = (pb << 8) | bc[3]; rank_b -= bc[4];
    const bool fine = bc[2] <= 4u && bc[5] <= 4u;    // (uniform) both buckets hold a few sample keys: near enough
    ++done;
    BR_STAMP(27 + 4 * round);
    if (fine) break;
  }
  // the bucket's lower end for a, its upper end for b (after all the rounds the bits below are zero)
  const int tail = 32 - 8 * done;
  out_a = kmin + ((pa << tail) >> sh);
  out_b = kmin + (((pb << tail) | (tail ? (1u << tail) - 1u : 0u)) >> sh);
  if (out_b > kmax) out_b = kmax;
}

// The reference's batch > 1 rule (mode 0, B > 1) couples the sequences: k' of one needs the
// finite-threshold and all chunks of every one (seq_prepare_body) -- before the bracket, which is
// placed by k'.  build_keys counted the keys of every head that are not evictable; a workgroup per
// sequence sums the chunk counts that follow (seq_tmp: F, Cn), seq_prepare_kernel makes k' of them.
// (A chunk no physical block claims keeps its 0xFFFFFFFF keys, which nobody counted: count_collect
// raises the flag when it meets one, and the digit rounds redo the call.)
__global__ __launch_bounds__(256) void bracket_totals_kernel(kvc_schedule_params p, SchedWs ws) {
  if (voided(ws)) return;
  __shared__ uint32_t red_s[2];
  const int i = blockIdx.x, tid = threadIdx.x, lane = lane_id();
  const int B = p.num_seqs, H = p.num_kv_heads, LH = p.num_layers * H, bs = p.block_size;
  const int G = B * LH;
  if (tid < 2) red_s[tid] = 0;
  __syncthreads();
  uint32_t f = 0, cn = 0;
  for (int lh = tid; lh < LH; lh += blockDim.x) {
    const int g = i * LH + lh;
    const int64_t b = p.evicted_kv_offsets[g];
    const int64_t e = (g + 1 < G) ? (int64_t)p.evicted_kv_offsets[g + 1] : true_n(p, ws);
    const uint32_t slots = (uint32_t)(e - b), nonfin = ws.bnonfin[g];
    const int ctx = p.context_lens[((lh / H) * B + i) * H + (lh % H)];
    f += nchunks_freed(slots > nonfin ? slots - nonfin : 0u, (uint32_t)p.hanging_token_count[g], (uint32_t)bs);
    cn += (uint32_t)((ctx + bs - 1) / bs);
  }
  f = wave_reduce_sum(f); cn = wave_reduce_sum(cn);
  if (lane == 0) { atomicAdd(&red_s[0], f); atomicAdd(&red_s[1], cn); }
  __syncthreads();
  if (tid == 0) { ws.seq_tmp[i] = (int32_t)red_s[0]; ws.seq_tmp[B + i] = (int32_t)red_s[1]; }
}

// One workgroup per sequence: the sample build_keys left behind (one key per cell, R x 1024 cells),
// the number of keys a k-chunk eviction takes (k bs + sum(hang) less half a block per head: the last
// threshold of a head lies anywhere inside its next block) in sample units, and the sample's keys
// at the ranks a few sigma around it: [lo, hi] holds T* unless the sample misleads (then the lists
// run over or T* is not among the listed thresholds: fallback).  Below T* the bracket reaches a
// block and a half per head further: every head's last freed threshold M, at most bs keys below
// T* in the head's own order, should be listed as well.  Also clears the counters of the passes
// behind it (the heads' three, the flag and the barrier words).
constexpr int BR_R = BR_CELLS / 1024;                // sample keys per thread
__global__ __launch_bounds__(1024) void bracket_kernel(kvc_schedule_params p, SchedWs ws) {
  if (voided(ws)) return;
  __shared__ __attribute__((aligned(16))) uint32_t priv[2 * PRIV_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t hist[2 * RADIX];
  __shared__ uint32_t bc[6];
  __shared__ uint32_t red_s[3];
  const int i = blockIdx.x, tid = threadIdx.x, lane = lane_id();
  const int B = p.num_seqs, H = p.num_kv_heads, LH = p.num_layers * H, bs = p.block_size;
  const int64_t base = p.evicted_kv_offsets[i * LH];
  const int64_t end = i + 1 < B ? (int64_t)p.evicted_kv_offsets[(i + 1) * LH] : true_n(p, ws);
  const uint32_t n = (uint32_t)(end - base);
  BR_STAMP(0);
  if (tid < 3) red_s[tid] = 0;
  if (i == 0 && tid < 128) ws.fallback[tid] = 0u;    // the flag, the stamps and the phase counters of the fallback
  __syncthreads();
  if (i == 0 && ws.bclaim != nullptr && tid < WAVE) {
    // no 0xFF fill in front of the key pass: it counted the logical blocks it found instead.  Every slot of keys,
    // chunk table and sample was written iff the count is N / bs -- else (a hole in the metadata, or counters some
    // other schedule left dirty) the flag goes up and the fallback builds everything anew.  The counters are left
    // zero for the next call.
    static_assert(CLAIM_SHARDS == WAVE, "one shard per lane");
    const uint32_t c = wave_reduce_sum(ws.bclaim[tid * 32]);
    ws.bclaim[tid * 32] = 0u;
    if (tid == 0 && (int64_t)c != true_n(p, ws) / bs) atomicOr(ws.fallback, 1u | FB_HOLES_BIT);
  }
  // (the sample's 32 keys per thread are requested first: the walk over the heads below does not depend on them)
  const int lg = bracket_stride_log2(n);
  const uint32_t stride = 1u << lg;
  const uint32_t* samp = ws.bsample + (int64_t)i * BR_CELLS;
  uint32_t key[BR_R];
  uint32_t fin = 0, finmask = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
  for (int r = 0; r < BR_R; ++r) {
    const uint32_t x = (uint32_t)r * 1024u + (uint32_t)tid;
    // (a cell whose sampled slot lies beyond the sequence holds nothing, or something stale)
    const uint64_t c0 = (uint64_t)x << lg;
    const bool have = c0 + stride <= n || (c0 < n && bracket_cell_slot(x, (uint32_t)i, lg) < n);
    key[r] = have ? samp[x] : 0xFFFFFFFFu;
  }
  {
    uint32_t hs = 0, la = 0;                         // sum of hang, heads that hold anything
    for (int lh = tid; lh < LH; lh += blockDim.x) {
      const int g = i * LH + lh;
      ws.st_cnt[g] = 0u; ws.st_def[g] = 0u;
      const int ctx = p.context_lens[((lh / H) * B + i) * H + (lh % H)];
      if (ctx > 0) { hs += (uint32_t)p.hanging_token_count[g]; la += 1u; }
    }
    hs = wave_reduce_sum(hs); la = wave_reduce_sum(la);
    if (lane == 0) { atomicAdd(&red_s[0], hs); atomicAdd(&red_s[1], la); }
  }
  BR_STAMP(1);
#pragma unroll
  for (int r = 0; r < BR_R; ++r)
    if (key[r] < KEY_INF) { finmask |= 1u << r; kmin = min(kmin, key[r]); kmax = max(kmax, key[r]); }
  fin = wave_reduce_sum((uint32_t)__popc(finmask));
  static_assert(BR_R <= 32, "finmask");
  if (lane == 0 && fin) atomicAdd(&red_s[2], fin);
  __syncthreads();
  fin = red_s[2];
  BR_STAMP(2);
  const double hs = red_s[0], la = red_s[1];
  const int k = ws.bk[i];
  BrRec rec;
  rec.seq_pos = p.seq_positions[i]; rec.prot = p.num_protected[i];
  rec.lo = 1u; rec.hi = 0u;                          // empty bracket: nothing is listed
  if (k > 0 && fin > 0u) {                           // (uniform)
    const double rstar = ((double)k * bs + hs - la * (bs + 1) * 0.5) / (double)stride;
    const double rho = rstar < 1.0 ? 1.0 : (rstar > (double)fin ? (double)fin : rstar);   // (over-ask: the top of the sample)
    const double sig = 4.5 * sqrt(rho * (1.0 - rho / ((double)fin + 1.0)) + 1.0) + 8.0;
    const double rlo = rho - sig - (la * bs * 1.5) / (double)stride;
    const double rhi = rho + sig + (la * bs * 0.5) / (double)stride;
    const bool open_lo = rlo < 1.0, open_hi = rhi >= (double)fin;
    uint32_t ka = 0, kb = 0;
    if (!(open_lo && open_hi)) {
      const uint32_t ra = open_lo ? 1u : (uint32_t)rlo;
      uint32_t rb = open_hi ? fin : (uint32_t)ceil(rhi);
      if (rb < ra) rb = ra;
      reg_rank_select2<BR_R>(key, finmask, kmin, kmax, ra, rb, priv, hist, bc, ka, kb, ws);
    }
    rec.lo = open_lo ? 0u : ka;
    rec.hi = open_hi ? KEY_INF - 1u : kb;
  }
  BR_STAMP(3);
  if (tid == 0) reinterpret_cast<BrRec*>(ws.st_seqrec)[i] = rec;
}

// ONE pass over the keys, tiles of HTILE keys on a persistent grid like hist_round, four consecutive
// keys per lane: per head the keys below lo (-> st_def) are counted in a register per lane and summed
// when the head changes (through LDS: one global add per workgroup and head); keys inside [lo, hi]
// are queued in LDS (their places from one wave scan per 256 keys) and appended to their heads'
// lists 64 at a time, one atomic per head and batch (st_cnt counts on beyond the capacity: overflow).
// (a ballot-compacted key per lane and step was 0.8 instructions per key: 11 us of VALU time at 8 M keys)
constexpr int CC_RUN = 256;                          // keys per wave step
constexpr int CC_QUEUE = 64 + CC_RUN;
__global__ __launch_bounds__(256) void count_collect_kernel(kvc_schedule_params p, SchedWs ws) {
  if (voided(ws)) return;
  __shared__ uint32_t qk[4][CC_QUEUE], qg[4][CC_QUEUE];
  __shared__ uint32_t wg_below[8];                   // the workgroup's first eight heads: one global add each
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  const int LH = p.num_layers * p.num_kv_heads;
  const int64_t N = true_n(p, ws);
  const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const BrRec* recs = reinterpret_cast<const BrRec*>(ws.st_seqrec);
  const int64_t ntiles = (N + HTILE - 1) / HTILE;
  const int64_t tb = ntiles * blockIdx.x / gridDim.x, te = ntiles * (blockIdx.x + 1) / gridDim.x;
  if (tb >= te) return;                              // (the whole workgroup)
  if (threadIdx.x < 8) wg_below[threadIdx.x] = 0u;
  __syncthreads();
  int qn = 0;
  // the queue's first n entries leave (they are in head order: runs of one head): one returning add
  // per run reserves their places.  Nobody waits for it here: the entries stay in registers and are
  // stored when the next batch leaves (or at the end) -- the round trip of the add, and of the
  // head's slot range the store needs, is then long over.
  bool pend = false;
  uint32_t p_key = 0, p_g = 0, p_pos0 = 0;
  int p_s0 = 0;
  int64_t p_b = 0, p_en = 0;
  auto complete = [&]() {
    if (!pend) return;                               // (uniform)
    const uint32_t pos0 = __shfl(p_pos0, p_s0, 64);
    if (p_g != 0xFFFFFFFFu) {
      const uint32_t pos = pos0 + (uint32_t)(lane - p_s0);
      if (pos < bracket_cap((uint32_t)(p_en - p_b))) ws.blist[bracket_list_at(p_b, (int)p_g) + pos] = p_key;
    }
    pend = false;
  };
  auto drain = [&](int n) {
    complete();
    wave_lds_sync();
    const bool have = lane < n;
    const uint32_t g = have ? qg[w][lane] : 0xFFFFFFFFu;
    const uint32_t gp = (have && lane > 0) ? qg[w][lane - 1] : 0xFFFFFFFEu;
    const unsigned long long starts = __ballot(have && g != gp);
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);      // lanes up to mine
    const int s0 = 63 - __builtin_clzll((starts & le) | 1ull);                        // my run's first lane
    p_pos0 = 0;
    if (have && lane == s0) {
      const unsigned long long nxt = starts & ~le;                                    // the next run's start
      const int e1 = nxt ? __ffsll((long long)nxt) - 1 : n;
      p_pos0 = atomicAdd(&ws.st_cnt[g], (uint32_t)(e1 - lane));
    }
    p_key = have ? qk[w][lane] : 0u;
    p_g = g; p_s0 = s0;
    if (have) {
      p_b = p.evicted_kv_offsets[g];
      p_en = ((int)g + 1 < G) ? (int64_t)p.evicted_kv_offsets[g + 1] : N;
    }
    pend = true;
    // what stays moves to the front (rest <= CC_RUN: up to CC_RUN / 64 entries per lane)
    const int rest = qn - n;
    uint32_t mk[CC_RUN / 64], mg[CC_RUN / 64];
#pragma unroll
    for (int q = 0; q < CC_RUN / 64; ++q)
      if (q * 64 + lane < rest) { mk[q] = qk[w][n + q * 64 + lane]; mg[q] = qg[w][n + q * 64 + lane]; }
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < CC_RUN / 64; ++q)
      if (q * 64 + lane < rest) { qk[w][q * 64 + lane] = mk[q]; qg[w][q * 64 + lane] = mg[q]; }
    qn = rest;
    wave_lds_sync();
  };
  constexpr int U = HTILE / (4 * CC_RUN);
  static_assert(U >= 1 && HTILE % (4 * CC_RUN) == 0, "a tile is U steps of four waves");
  // heads change rarely: the head of the last step, its slot range and its sequence's bracket stay in
  // (scalar) registers
  // (found below, behind the request for the first tile: what the workgroup takes is a chain of dependent loads)
  int g = 0, g0 = 0;                                 // g0: the same in every wave
  int64_t g_beg = 0, g_end = 0;
  BrRec rc{};
  int acc_g = -1;
  uint32_t acc_b = 0;                                // (per lane)
  auto flush = [&]() {
    const uint32_t tot = wave_reduce_sum(acc_b);
    if (acc_g >= 0 && lane == 0 && tot) {
      if ((unsigned)(acc_g - g0) < 8u) atomicAdd(&wg_below[acc_g - g0], tot);
      else atomicAdd(&ws.st_def[acc_g], tot);
    }
    acc_b = 0;
  };
  // the lane's four keys idx0 .. idx0 + 3, as far as they lie in [sb, se), belong to head gs (bracket lo .. hi)
  auto segment = [&](int gs, const uint4& k4, int64_t idx0, int64_t sb, int64_t se, uint32_t lo, uint32_t hi) {
    if (gs != acc_g) { flush(); acc_g = gs; }
    const uint32_t kx[4] = {k4.x, k4.y, k4.z, k4.w};
    bool in[4];
    uint32_t nin = 0;
    bool hole = false;                               // a key nobody wrote (see bracket_totals_kernel)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool mine = idx0 + c >= sb && idx0 + c < se;
      hole = hole || (mine && kx[c] == 0xFFFFFFFFu);
      acc_b += (mine && kx[c] < lo) ? 1u : 0u;
      in[c] = mine && kx[c] >= lo && kx[c] <= hi;
      nin += in[c] ? 1u : 0u;
    }
    if (ws.bnonfin != nullptr && __ballot(hole) && lane == 0) atomicOr(ws.fallback, 1u);
    if (__ballot(nin != 0u)) {
      const uint32_t inc = wave_inclusive_scan(nin);
      int pos = qn + (int)(inc - nin);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (in[c]) { qk[w][pos] = kx[c]; qg[w][pos] = (uint32_t)gs; ++pos; }
      qn += (int)__shfl(inc, 63, 64);
      while (qn >= 64) drain(64);
    }
  };
  uint4 kv[U], kn[U];
  auto load_tile = [&](uint4 (&dst)[U], int64_t t) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t idx0 = t * HTILE + (int64_t)(u * 4 + w) * CC_RUN + 4 * lane;
      if (idx0 + 3 < N) {
        dst[u] = *reinterpret_cast<const uint4*>(ws.keys + idx0);
      } else {
        dst[u].x = idx0 < N ? ws.keys[idx0] : 0xFFFFFFFFu;
        dst[u].y = idx0 + 1 < N ? ws.keys[idx0 + 1] : 0xFFFFFFFFu;
        dst[u].z = idx0 + 2 < N ? ws.keys[idx0 + 2] : 0xFFFFFFFFu;
        dst[u].w = 0xFFFFFFFFu;
      }
    }
  };
  BR_STAMP(21);
  load_tile(kn, tb);
  g = wave_upper_bound_minus1(p.evicted_kv_offsets, G, tb * HTILE);
  g_beg = p.evicted_kv_offsets[g];
  g_end = (g + 1 < G) ? (int64_t)p.evicted_kv_offsets[g + 1] : N;
  rc = recs[g / LH];
  g0 = g;
  BR_STAMP(22);
  for (int64_t t = tb; t < te; ++t) {
    const int64_t t0 = t * HTILE;
#pragma unroll
    for (int u = 0; u < U; ++u) kv[u] = kn[u];
    if (t + 1 < te) load_tile(kn, t + 1);            // the next tile's keys are on their way meanwhile (two ahead: no gain)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r0 = t0 + (int64_t)(u * 4 + w) * CC_RUN;             // the step's first key
      if (r0 >= N) break;                                                // (wave-uniform)
      const int64_t r1 = min(N, r0 + CC_RUN);
      const int64_t idx0 = r0 + 4 * lane;
      if (r0 >= g_beg && r1 <= g_end) {                                  // inside the head of the last step
        segment(g, kv[u], idx0, r0, r1, rc.lo, rc.hi);
        continue;
      }
      // head of the step's first key (scalar walk from the last one), then one segment per head inside the step
      while (g + 1 < G && (int64_t)p.evicted_kv_offsets[g + 1] <= r0) ++g;
      while (g > 0 && (int64_t)p.evicted_kv_offsets[g] > r0) --g;
      int64_t sb = r0;
      for (;;) {
        g_beg = p.evicted_kv_offsets[g];
        g_end = (g + 1 < G) ? (int64_t)p.evicted_kv_offsets[g + 1] : N;
        rc = recs[g / LH];
        const int64_t se = min(r1, g_end);
        if (se > sb) {
          segment(g, kv[u], idx0, sb, se, rc.lo, rc.hi);
          sb = se;
        }
        if (sb >= r1) break;
        ++g;
      }
    }
  }
  BR_STAMP(23);
  flush();
  if (qn > 0) drain(qn);
  BR_STAMP(14);
  complete();
  __syncthreads();
  BR_STAMP(15);
  if (threadIdx.x < 8 && wg_below[threadIdx.x]) atomicAdd(&ws.st_def[g0 + threadIdx.x], wg_below[threadIdx.x]);
}

// Ascending sort of the m keys a[0..m) (LDS; m <= SZ <= BR_SORT_MAX, SZ a power of two >= 2) by a
// 512-thread workgroup, all of them inside [lo, hi]: SZ buckets by the top bits of
// (key - lo) << clz(hi - lo) -- about one key per bucket when the bracket is a narrow quantile
// range -- an exclusive scan of the bucket counts, a scatter, and the order inside a bucket by
// counting (equal keys in the order they arrived: any order of equal keys is the sorted list).
// Five barriers instead of the 55 steps of a bitonic network (12 us at 1024 keys).  The result is in
// a[0..m); tmp[SZ] and cnt[SZ + 1] are scratch.
__device__ __forceinline__ void block_bucket_sort(uint32_t* a, uint32_t* tmp, uint32_t* cnt, uint32_t* wtot /*[8]*/,
                                                  int m, int SZ, uint32_t lo, uint32_t hi) {
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  if (hi <= lo) return;                              // (uniform) one value
  const int sh = __builtin_clz(hi - lo);
  const int down = 32 - (31 - __builtin_clz((uint32_t)SZ));             // 32 - log2(SZ)
  auto bucket = [&](uint32_t key) { return ((key - lo) << sh) >> down; };
  for (int j = tid; j <= SZ; j += 512) cnt[j] = 0u;
  __syncthreads();
  constexpr int E = BR_SORT_MAX / 512;
  uint32_t slot[E];
#pragma unroll
  for (int u = 0; u < E; ++u) {
    const int e = tid + u * 512;
    slot[u] = e < m ? atomicAdd(&cnt[bucket(a[e])], 1u) : 0u;
  }
  __syncthreads();
  {                                                  // exclusive scan of the SZ counts, in place; cnt[SZ] = m
    const int per = (SZ + 511) / 512;
    const int b0 = tid * per;
    uint32_t sum = 0;
    for (int q = 0; q < per; ++q) if (b0 + q < SZ) sum += cnt[b0 + q];
    const uint32_t inc = wave_inclusive_scan(sum);
    if (lane == WAVE - 1) wtot[w] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (int q = 0; q < w; ++q) run += wtot[q];
    for (int q = 0; q < per; ++q)
      if (b0 + q < SZ) { const uint32_t c = cnt[b0 + q]; cnt[b0 + q] = run; run += c; }
    if (tid == 0) cnt[SZ] = (uint32_t)m;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < E; ++u) {
    const int e = tid + u * 512;
    if (e < m) { const uint32_t key = a[e]; tmp[cnt[bucket(key)] + slot[u]] = key; }
  }
  __syncthreads();
  for (int q = tid; q < m; q += 512) {
    const uint32_t key = tmp[q];
    const uint32_t b = bucket(key);
    const uint32_t s = cnt[b], e = cnt[b + 1];
    uint32_t r = 0;
    for (uint32_t j = s; j < e; ++j) { const uint32_t v = tmp[j]; r += (v < key) || (v == key && j < (uint32_t)q); }
    a[s + r] = key;
  }
  __syncthreads();
}

// listed thresholds of a head: list entries first, first + bs, ... (< m)
__device__ __forceinline__ void bracket_thresholds(uint32_t below, uint32_t m, uint32_t hang, uint32_t bs,
                                                   uint32_t& first, uint32_t& tcnt) {
  // smallest c with c * bs + hang - 1 >= below
  const uint32_t c0 = below + 1u > hang ? (below + 1u - hang + bs - 1u) / bs : 0u;
  first = c0 * bs + hang - 1u - below;
  tcnt = first < m ? (m - first + bs - 1u) / bs : 0u;
}

// one workgroup per head: its list sorted in place, and its thresholds (every bs-th entry from the
// first threshold rank on) side by side in bthr from the head's first chunk on -- the selection
// kernel is one workgroup per sequence and would fetch a 64-byte line per threshold otherwise
// (9 us at config 2's 13.8 k thresholds)
__global__ __launch_bounds__(512) void bracket_records_kernel(kvc_schedule_params p, SchedWs ws) {
  if (voided(ws)) return;
  __shared__ uint32_t a[BR_SORT_MAX], tmp[BR_SORT_MAX], cnt[BR_SORT_MAX + 1];
  __shared__ uint32_t wtot[8];
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  const int g = blockIdx.x;
  BR_STAMP(16);
  const int64_t base = p.evicted_kv_offsets[g];
  const int64_t end = (g + 1 < G) ? (int64_t)p.evicted_kv_offsets[g + 1] : true_n(p, ws);
  const uint32_t m = ws.st_cnt[g];
  const uint32_t below = ws.st_def[g];
  const uint32_t hang = (uint32_t)p.hanging_token_count[g];
  const BrRec rc = reinterpret_cast<const BrRec*>(ws.st_seqrec)[g / (p.num_layers * p.num_kv_heads)];
  if (m > bracket_cap((uint32_t)(end - base))) {     // the list overflowed: the digit rounds take over
    if (threadIdx.x == 0) atomicOr(ws.fallback, 1u);
    return;
  }
  if (m == 0u) return;
  uint32_t* list = ws.blist + bracket_list_at(base, g);
  int SZ = 2;
  while ((uint32_t)SZ < m) SZ <<= 1;
  for (int j = threadIdx.x; j < (int)m; j += blockDim.x) a[j] = list[j];
  __syncthreads();
  BR_STAMP(17);
  if (m > 1u) block_bucket_sort(a, tmp, cnt, wtot, (int)m, SZ, rc.lo, rc.hi);
  BR_STAMP(18);
  if (m > 1u)
    for (int j = threadIdx.x; j < (int)m; j += blockDim.x) list[j] = a[j];
  uint32_t first, tcnt;
  bracket_thresholds(below, m, hang, (uint32_t)p.block_size, first, tcnt);
  uint32_t* thr = ws.bthr + base / p.block_size;
  for (uint32_t j = threadIdx.x; j < tcnt; j += blockDim.x) thr[j] = a[first + j * (uint32_t)p.block_size];
  BR_STAMP(19);
}

// One workgroup per sequence: k', the chunks below the bracket, T* = the (k' - those)-th smallest
// of the listed thresholds (four digit rounds over them in LDS), and the per-head counts: chunks
// with a threshold below T*, then the ones equal to it in (head, chunk) order until the total is
// k' -- finalize_body's rule.                                        metrics.py:671-729, 773-792
// dynamic LDS: arr[P] thresholds, head-major; tpre[LH + 1]
__global__ __launch_bounds__(1024) void bracket_select_kernel(kvc_schedule_params p, SchedWs ws, int P) {
  if (voided(ws)) return;
  extern __shared__ __attribute__((aligned(16))) uint8_t sel_lds[];
  uint32_t* arr = reinterpret_cast<uint32_t*>(sel_lds);
  uint32_t* tpre = arr + P;                                             // [LH + 1] exclusive prefix of the heads' listed thresholds
  uint32_t* hsrc = tpre + (p.num_layers * p.num_kv_heads + 1);          // [LH] where the head's listed thresholds are (in bthr)
  __shared__ __attribute__((aligned(16))) uint32_t priv[PRIV_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t sel_hist[RADIX];
  __shared__ uint32_t bc[3];
  __shared__ uint32_t red_s[1];
  __shared__ uint32_t wsum_s[16], wsum2_s[16];
  const int i = blockIdx.x, tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  const int LH = p.num_layers * p.num_kv_heads;
  const uint32_t bs = (uint32_t)p.block_size;
  BR_STAMP(8);
  if (tid == 0) red_s[0] = 0;
  __syncthreads();
  // per head (LH <= 1024 = blockDim: one thread each): list geometry, the chunks below the bracket
  uint32_t hang = 1, first = 0, tc = 0, sure = 0;
  int64_t hchunk = 0;                                // the head's first chunk: where its thresholds are in bthr
  if (tid < LH) {
    const int g = i * LH + tid;
    const int64_t b = p.evicted_kv_offsets[g];
    const int64_t e = (g + 1 < G) ? (int64_t)p.evicted_kv_offsets[g + 1] : true_n(p, ws);
    const uint32_t below = ws.st_def[g];
    const uint32_t m = min(ws.st_cnt[g], bracket_cap((uint32_t)(e - b)));
    hang = (uint32_t)p.hanging_token_count[g];
    hchunk = b / bs;
    if (e > b) {
      bracket_thresholds(below, m, hang, bs, first, tc);
      sure = nchunks_freed(below, hang, bs);                             // thresholds of rank < below
    }
  }
  uint32_t my_pre;
  {
    const uint32_t inc = wave_inclusive_scan(tc);
    if (lane == WAVE - 1) wsum_s[w] = inc;
    const uint32_t s1 = wave_reduce_sum(sure);
    if (lane == 0 && s1) atomicAdd(&red_s[0], s1);
    __syncthreads();
    uint32_t woff = 0;
    for (int q = 0; q < w; ++q) woff += wsum_s[q];
    my_pre = woff + inc - tc;
    if (tid < LH) { tpre[tid] = my_pre; hsrc[tid] = (uint32_t)hchunk; }
    if (tid == LH - 1) tpre[LH] = woff + inc;
  }
  __syncthreads();
  const uint32_t T = tpre[LH];
  BR_STAMP(9);
  const int kk = ws.bk[i];
  const uint32_t sure_all = red_s[0];
  const BrRec rc = reinterpret_cast<const BrRec*>(ws.st_seqrec)[i];
  // k' = min(k, finite-threshold chunks): a bracket that is open above lists every threshold from
  // lo on, so the finite-threshold chunks are the sure ones and the listed ones
  uint32_t need = 0;
  bool active = kk > 0;
  if (active) {                                      // (uniform)
    bool ok = (uint32_t)kk > sure_all && T <= (uint32_t)P;
    if (ok) {
      need = (uint32_t)kk - sure_all;
      if (need > T) { if (rc.hi >= KEY_INF - 1u) need = T; else ok = false; }
    } else if ((uint32_t)kk == sure_all && T == 0u && rc.hi >= KEY_INF - 1u) {
      ok = true;                                     // exactly the chunks below an open bracket
    }
    if (!ok) {
      if (tid == 0) atomicOr(ws.fallback, 1u);       // T* is not among the listed thresholds
      return;
    }
  }
  uint32_t lt = 0, eq = 0;                           // my head's listed thresholds below T*, equal to it
  uint32_t need_eq = 0;
  if (active && need > 0u) {
    // the listed thresholds into LDS, head-major: a group of threads per head
    int tph = 1;
    while (tph * 2 * LH <= 1024) tph <<= 1;          // threads per head
    {
      const int lh = tid / tph, sub = tid % tph;
      if (lh < LH) {
        const uint32_t n_h = tpre[lh + 1] - tpre[lh];
        const uint32_t* src = ws.bthr + hsrc[lh];
        uint32_t* dst = arr + tpre[lh];
        for (uint32_t j = (uint32_t)sub; j < n_h; j += (uint32_t)tph) dst[j] = src[j];
      }
    }
    __syncthreads();
    BR_STAMP(10);
    // every listed threshold lies in [lo, hi]: the rounds run on (v - lo) << clz(hi - lo), whose
    // digits are spread (the bytes of the keys themselves are nearly constant over a bracket, and
    // LDS adds to one address serialise)
    const int sh = rc.hi > rc.lo ? __builtin_clz(rc.hi - rc.lo) : 32;
    const int rounds = (32 - sh + 7) / 8;
    uint32_t prefix = 0, krem = need;
    for (int round = 0; round < rounds; ++round) {
      const int shift = 24 - 8 * round;
      priv_clear(priv, 1);
      __syncthreads();
      uint32_t* hw = priv + w * PRIV_STRIDE;
      for (int e = tid; e < (int)T; e += blockDim.x) {
        const uint32_t v = (arr[e] - rc.lo) << sh;
        if (round == 0 || (v >> (shift + 8)) == prefix) atomicAdd(&hw[(v >> shift) & 0xFFu], 1u);
      }
      __syncthreads();
      priv_sum(priv, sel_hist, 1);
      __syncthreads();
      if (w == 0) wave_pick_digit(sel_hist, krem, bc);
      __syncthreads();                               // (bc is next written three barriers on)
      prefix = (prefix << 8) | bc[0];
      krem -= bc[1];
    }
    const uint32_t Tstar = rounds > 0 ? rc.lo + ((prefix << (32 - 8 * rounds)) >> sh) : rc.lo;
    BR_STAMP(11);
    need_eq = krem;                                  // thresholds equal to T* still to hand out
    if (tid < LH && tc > 0u) {                       // my head's thresholds ascend: two bisections
      const uint32_t* mine = arr + my_pre;
      uint32_t lo = 0, hi = tc;                      // first entry >= T*
      while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (mine[mid] < Tstar) lo = mid + 1u; else hi = mid; }
      lt = lo;
      hi = tc;                                       // first entry > T*
      while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (mine[mid] <= Tstar) lo = mid + 1u; else hi = mid; }
      eq = lo - lt;
    }
  }
  BR_STAMP(12);
  // ties in (head, chunk) order: exclusive scan of eq over the heads
  {
    const uint32_t inc = wave_inclusive_scan(eq);
    __syncthreads();
    if (lane == WAVE - 1) wsum2_s[w] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int q = 0; q < w; ++q) woff += wsum2_s[q];
    const uint32_t excl = woff + inc - eq;
    if (tid < LH) {
      const int g = i * LH + tid;
      uint32_t nfree = 0;
      if (active) {
        const uint32_t room = need_eq > excl ? need_eq - excl : 0u;
        nfree = sure + lt + (eq < room ? eq : room);
      }
      p.evicted_block_count[g] = (int32_t)nfree;
      p.evicted_kv_count[g] = nfree > 0 ? (int32_t)((nfree - 1u) * bs + hang) : 0;
    }
  }
  BR_STAMP(13);
}


}  // namespace kvc
