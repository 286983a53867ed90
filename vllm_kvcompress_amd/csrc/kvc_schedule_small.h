// kvc_schedule_small.h -- A3 schedule_evictions: the small-eviction schedule of the continual steady state
// (one translation unit: included by kvc_schedule.hip in this order; see the overview there)
#pragma once
#include "kvc_common.h"
#include "kvc_schedule_common.h"
#include "../../include/kvc_mi355x.h"

namespace kvc {

// ------------------------------------------------------------------ 7. small-eviction schedule
// Continual compression frees about one block per head and step, from thousands of short heads
// (config 3: 65 536 heads of ~4 k slots).  The general pipeline above writes a key per slot and
// then reads every key five times (four sequence-level digit rounds + the per-head select) to
// evict 0.4 % of them.  Here the metric store is read ONCE, in PHYSICAL block order -- a plain
// coalesced stream -- and no key array, no chunk table exists:
//   * stream_sample_kernel: a sample of the physical blocks (those whose index hashes to 0 modulo
//     the stride; only their rows and metadata are touched), keys written to a dense per-head slot;
//   * stream_pivot_kernel (one workgroup per sequence): a pivot P_i such that the sequence holds,
//     with a wide margin, at least Tgt_i = k_i * bs + sum_g (hang_g - 1) evictable keys <= P_i --
//     with that many the chunk thresholds <= P_i number at least k_i, whatever their spread over
//     the heads (n_g = floor((R_g - hang_g) / bs) + 1 >= (R_g - hang_g + 1) / bs);
//   * stream_collect_kernel: the one pass over metrics / (positions) / metadata: per block the keys
//     are made on the fly, those <= P_i are queued in LDS and appended to their head's candidate
//     list (one returning atomic per candidate, issued 64 at a time); blocks with masked slots
//     add their number to the head's deficit (finite keys of a head = slots - deficit) -- or, when
//     keys do not depend on positions and sequences not on each other (LAZY), the position rows
//     are not streamed at all and only the candidates' positions are looked up;
//   * stream_records_kernel (one wave per head): the list sorted by (key, physical slot) -- the
//     canonical tie order -- is the head's record;
//   * chunk thresholds are every bs-th entry of a record, so the sequence-level selection (one
//     workgroup per sequence: the k'-th smallest of the recorded thresholds of its heads by
//     (threshold, head, chunk)) and the emission (the first cnt record entries, re-sorted by
//     logical index) never touch the metrics again.
// HBM: 1 B (metadata) + 8 B (metrics, positions; 4 B when LAZY) + 4 B (null padding of the output)
// per candidate slot = the 12.75 B lower bound of SURVEY 8(d) (LAZY: below it) + the sample.
// Exactness never depends on the sample: a record holds EVERY evictable key <= P_i of its head,
// every threshold it does not list is > P_i, so the selection is exact as soon as the records of a
// sequence list k' thresholds.  If they do not (pivot too low), a head has more candidates than a
// record holds (KREC; e.g. all metrics tied), or the per-block metadata does not cover every
// logical block of the batch, `fallback` is raised and the general pipeline -- enqueued behind,
// gated on that flag -- recomputes everything.  Chosen by the host from
// kvc_schedule_params.max_evicted_blocks_hint (average <= 256 / bs / 8 blocks per head).

// ascending bitonic sort of SZ (power of two >= 128) LDS elements by one wave
template <typename T, int SZ>
__device__ void wave_bitonic_sort(T* a) {
  const int lane = lane_id();
  for (int k = 2; k <= SZ; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < SZ / 2; t += WAVE) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i + j;
        const bool up = (i & k) == 0;
        const T x = a[i], y = a[l];
        if ((x > y) == up) { a[i] = y; a[l] = x; }
      }
      wave_lds_sync();
    }
}

struct SeqRec { int32_t seq_pos, prot; uint32_t pivot_excl, pad; };   // candidates: key < pivot_excl

__device__ __forceinline__ uint32_t strat_hash(uint32_t g, uint32_t j) {
  uint32_t x = (g * 0x9E3779B1u) ^ ((j + 0x7F4A7C15u) * 0x85EBCA77u);
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
  return x;
}

// per-block metadata of 64 consecutive blocks, one per lane, all four loads requested together
struct BlockMeta { int s, l, h, lbn; };
__device__ __forceinline__ BlockMeta load_meta(const kvc_schedule_params& p, int64_t blk, bool in) {
  BlockMeta m{-1, 0, 0, 0};
  if (in) {
    m.s = p.seq_index_by_block[blk]; m.l = p.layer_index_by_block[blk];
    m.h = p.head_index_by_block[blk]; m.lbn = p.logical_block_num_by_block[blk];
  }
  return m;
}
// The sample: every physical block whose index hashes to 0 mod 2^sshift -- no pass over the
// metadata, and no pattern of the allocator or of the logical order can alias with it.  A wave
// walks 64 block indices per step (arithmetic only), queues the chosen ones in LDS and works them
// off 64 / (BS / 4) at a time, BS / 4 lanes per block: metadata -> owner -> a slot in the head's
// sample (one returning atomic per block: keys[off_g + slot * bs ...], at most one slot per block
// of the head) -> metric / position row -> keys.
__device__ __forceinline__ bool block_sampled(uint32_t blk, uint32_t smask) {
  return (strat_hash(blk, 0x51ED270Bu) & smask) == 0u;
}

template <int BS>
__global__ __launch_bounds__(256) void stream_sample_kernel(kvc_schedule_params p, SchedWs ws, int sshift) {
  constexpr int LPB = BS / 4, BPD = 64 / LPB;        // lanes per block, blocks per drain
  constexpr int QCAP = 64 + BPD;
  __shared__ uint32_t q_blk[4][QCAP];
  __shared__ uint32_t q1_blk[4][128];                // first stage: hashed-in blocks, membership not looked at yet
  const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int64_t wave = (int64_t)blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4;
  const int L = p.num_layers, H = p.num_kv_heads;
  const uint32_t smask = (1u << sshift) - 1u;
  int qn = 0;
  auto drain = [&](int n) {                          // pops the top n (<= BPD) queued blocks
    wave_lds_sync();
    const int e = qn - n + lane / LPB;
    bool ok = lane / LPB < n;
    const int64_t blk = ok ? (int64_t)q_blk[w][e] : 0;
    const BlockMeta mt = load_meta(p, blk, ok);
    ok = ok && mt.s >= 0 && mt.s < p.seq_slot_len;
    int i = p.seq_slot_of_seq[ok ? mt.s : 0];
    ok = ok && i >= 0 && mt.l >= 0 && mt.l < L && mt.h >= 0 && mt.h < H && mt.lbn >= 0;
    const int l = ok ? mt.l : 0, h = ok ? mt.h : 0;
    if (!ok) i = 0;
    const int g = (i * L + l) * H + h;
    const int ctx = p.context_lens[(l * p.num_seqs + i) * H + h];
    const int64_t off = p.evicted_kv_offsets[g];
    const int seq_pos = p.seq_positions[i], prot = p.num_protected[i];
    const float4 m = reinterpret_cast<const float4*>(p.metrics + blk * BS)[lane % LPB];
    const int4 q = reinterpret_cast<const int4*>(p.token_positions + blk * BS)[lane % LPB];
    ok = ok && mt.lbn < (ctx + BS - 1) / BS;         // (else: not a logical block of its head)
    uint32_t slot = 0;
    if (ok && lane % LPB == 0) slot = atomicAdd(&ws.st_samp[g], 1u);
    slot = (uint32_t)__shfl((int)slot, lane & ~(LPB - 1), 64);
    // more physical blocks naming a head than the head has logical blocks (duplicate or stale
    // metadata; consistent state cannot get here): the head's sample region holds nblk blocks --
    // the surplus is dropped and the call handed to the general pipeline
    if (ok && slot >= (uint32_t)((ctx + BS - 1) / BS)) {
      if (lane % LPB == 0) atomicOr(ws.fallback, 1u);
      ok = false;
    }
    if (ok) {
      uint4 k;
      k.x = slot_key(p, m.x, q.x, seq_pos, prot, l, h);
      k.y = slot_key(p, m.y, q.y, seq_pos, prot, l, h);
      k.z = slot_key(p, m.z, q.z, seq_pos, prot, l, h);
      k.w = slot_key(p, m.w, q.w, seq_pos, prot, l, h);
      reinterpret_cast<uint4*>(ws.keys + off + (int64_t)slot * BS)[lane % LPB] = k;
    }
    qn -= n;
    wave_lds_sync();
  };
  // First stage: 64 hashed-in blocks at a time, one per lane -- is the block's sequence in the
  // batch at all?  (In an engine-sized cache most sampled blocks belong to other sequences or to
  // nobody: the 4-lane drain with its four metadata gathers per block is for the batch's only.)
  int q1n = 0;
  auto filter = [&](int n) {                         // pops the top n (<= 64) first-stage entries
    wave_lds_sync();
    bool in = lane < n;
    const uint32_t blk = in ? q1_blk[w][q1n - n + lane] : 0u;
    int sq = -1;
    if (in) sq = p.seq_index_by_block[blk];
    in = in && sq >= 0 && sq < p.seq_slot_len;
    int i = -1;
    if (in) i = p.seq_slot_of_seq[sq];
    in = in && i >= 0;
    q1n -= n;
    const unsigned long long bal = __ballot(in);
    if (bal) {                                       // wave-uniform
      if (in) q_blk[w][qn + __popcll(bal & ((1ull << lane) - 1ull))] = blk;
      qn += __popcll(bal);
      while (qn >= BPD) drain(BPD);
    }
    wave_lds_sync();
  };
  for (int64_t b0 = wave * 64; b0 < p.num_blocks; b0 += nwaves * 64) {
    const int64_t blk = b0 + lane;
    const bool take = blk < p.num_blocks && block_sampled((uint32_t)blk, smask);
    const unsigned long long bal = __ballot(take);
    if (bal) {                                       // wave-uniform
      if (take) q1_blk[w][q1n + __popcll(bal & ((1ull << lane) - 1ull))] = (uint32_t)blk;
      q1n += __popcll(bal);
      if (q1n >= 64) filter(64);
    }
  }
  if (q1n > 0) filter(q1n);
  while (qn > 0) drain(min(qn, BPD));
}

// one workgroup per sequence: the rho-th smallest evictable key of its sample, rho = the sample's
// share of Tgt + 12 sigma + 8 (sigma^2 = that share: a binomial count, taken twice over for keys
// that cluster by block); a sample that is everything (stride 1) gives the Tgt-th key itself.
// The heads' samples (st_samp[g] blocks at keys[off_g ...]) form one flat key space through a
// prefix sum in LDS; a thread finds the head of its flat index by bisection.  A sample of up to
// PIV_R x 1024 keys is read ONCE into registers and the four digit rounds of the select run on
// the registers; a longer one (a sequence far longer than the batch average) is re-read from L2
// every round.
#ifndef KVC_PIV_SIGMAS
#define KVC_PIV_SIGMAS 12.0                          // (experiment builds: tools/, profiles/DESIGN_history_r1_r4.md section 6)
#endif
constexpr int PIV_R = 48;
constexpr int PIV_MAXLH = 1024;                      // heads per sequence (the host checked)
__global__ __launch_bounds__(1024) void stream_pivot_kernel(kvc_schedule_params p, SchedWs ws, int sshift) {
  __shared__ __attribute__((aligned(16))) uint32_t hist[RADIX];
  __shared__ uint32_t bc[4];
  __shared__ uint32_t tot_s[3];                      // blocks, sampled blocks, sum(hang - 1)
  __shared__ uint32_t fin_s;
  __shared__ uint32_t pre_s[PIV_MAXLH + 1];          // exclusive prefix of the heads' sample lengths (keys)
  __shared__ uint32_t wsum_s[16];
  const int i = blockIdx.x, tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  const int B = p.num_seqs, H = p.num_kv_heads, LH = p.num_layers * H, bs = p.block_size;
  if (tid < 3) tot_s[tid] = 0;
  if (tid == 0) fin_s = 0;
  __syncthreads();
  {
    uint32_t nb = 0, ns = 0, hs = 0;
    if (tid < LH) {                                  // LH <= 1024 = blockDim
      const int ctx = p.context_lens[((tid / H) * B + i) * H + (tid % H)];
      const uint32_t nblk = (uint32_t)((ctx + bs - 1) / bs);
      if (nblk) {
        ns = min(ws.st_samp[i * LH + tid], nblk);    // (the counter counts on past what the sampling pass stored)
        nb = nblk; hs = (uint32_t)p.hanging_token_count[i * LH + tid] - 1u;
      }
    }
    // block-wide exclusive scan of ns * bs -> pre_s
    const uint32_t len = ns * (uint32_t)bs;
    const uint32_t inc = wave_inclusive_scan(len);
    if (lane == WAVE - 1) wsum_s[w] = inc;
    nb = wave_reduce_sum(nb); const uint32_t nss = wave_reduce_sum(ns); hs = wave_reduce_sum(hs);
    if (lane == 0) { atomicAdd(&tot_s[0], nb); atomicAdd(&tot_s[1], nss); atomicAdd(&tot_s[2], hs); }
    __syncthreads();
    uint32_t woff = 0;
    for (int q = 0; q < w; ++q) woff += wsum_s[q];
    if (tid < LH) pre_s[tid] = woff + inc - len;
    if (tid == LH - 1) pre_s[LH] = woff + inc;
  }
  __syncthreads();
  const uint32_t nb = tot_s[0], ns = tot_s[1], hs = tot_s[2], n_keys = pre_s[LH];
  const int k = p.evicted_blocks_per_seq[i];
  SeqRec rec;
  rec.seq_pos = p.seq_positions[i]; rec.prot = p.num_protected[i]; rec.pivot_excl = 0u; rec.pad = 0u;
  // flat index x < n_keys -> address in the key scratch
  auto locate = [&](uint32_t x) {
    int lo = 0, hi = LH;                             // pre_s[lo] <= x < pre_s[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (pre_s[mid] <= x) lo = mid; else hi = mid;
    }
    return (int64_t)p.evicted_kv_offsets[i * LH + lo] + (x - pre_s[lo]);
  };
  if (k > 0 && nb > 0) {
    const double tgt = (double)k * bs + (double)hs;
    double rho = tgt;
    if (sshift > 0) {
      const double x = tgt * (double)ns / (double)nb;
      rho = ceil(x + KVC_PIV_SIGMAS * sqrt(x) + 8.0);
    }
    if (n_keys == 0u) {
      rec.pivot_excl = KEY_INF;                      // an empty sample: every evictable key is a candidate
    } else if (n_keys <= (uint32_t)PIV_R * 1024u) {
      // ---- the sample in registers: a thread takes units of 8 consecutive keys (32 B; sample
      // lengths are multiples of bs >= 8), one bisection per unit
      uint32_t key[PIV_R];
#pragma unroll
      for (int r = 0; r < PIV_R; r += 8) {
        const uint32_t x = ((uint32_t)(r / 8) * 1024u + (uint32_t)tid) * 8u;
        uint4 k0 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu), k1 = k0;
        if ((uint32_t)(r / 8) * 8192u < n_keys) {    // (uniform)
          if (x < n_keys) {
            const uint4* src = reinterpret_cast<const uint4*>(ws.keys + locate(x));
            k0 = src[0]; k1 = src[1];
          }
        }
        key[r] = k0.x; key[r + 1] = k0.y; key[r + 2] = k0.z; key[r + 3] = k0.w;
        key[r + 4] = k1.x; key[r + 5] = k1.y; key[r + 6] = k1.z; key[r + 7] = k1.w;
      }
      uint32_t prefix = 0, rank = 0;
      bool all = false;
      for (int round = 0; round < 4; ++round) {
        const int shift = 24 - 8 * round;
        if (tid < RADIX) hist[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PIV_R; ++r) {
          if ((uint32_t)(r / 8) * 8192u >= n_keys) break;  // (uniform)
          const bool valid = key[r] < KEY_INF && (round == 0 || (key[r] >> (shift + 8)) == prefix);
          hist_add(hist, valid, (key[r] >> shift) & 0xFFu);
        }
        __syncthreads();
        if (tid < WAVE) {                            // 256-bin inclusive scan, 4 bins per lane
          uint4 q = reinterpret_cast<uint4*>(hist)[tid];
          q.y += q.x; q.z += q.y; q.w += q.z;
          const uint32_t inc = wave_inclusive_scan(q.w);
          const uint32_t ex = inc - q.w;
          uint32_t rk = rank;
          if (round == 0) {                          // all evictable keys of the sample = the last bin's count
            const uint32_t fin = (uint32_t)__shfl((int)inc, WAVE - 1, 64);
            rk = (fin == 0u || rho >= (double)fin) ? 0u : (uint32_t)rho;
            if (tid == 0) bc[2] = rk;
          }
          const uint32_t c[4] = {q.x + ex, q.y + ex, q.z + ex, q.w + ex};
          uint32_t prev = ex;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (prev < rk && rk <= c[t]) { bc[0] = (uint32_t)tid * 4u + (uint32_t)t; bc[1] = prev; }
            prev = c[t];
          }
        }
        __syncthreads();
        if (round == 0) {
          rank = bc[2];
          if (rank == 0u) { all = true; break; }     // (uniform) every evictable key is a candidate
        }
        prefix = (prefix << 8) | bc[0];
        rank -= bc[1];
        __syncthreads();
      }
      rec.pivot_excl = all ? KEY_INF : prefix + 1u;  // prefix < KEY_INF
    } else {
      // ---- a sample too long for the registers: every round re-reads it
      auto pred = [&](int x) { return ws.keys[locate((uint32_t)x)] < KEY_INF; };
      auto val = [&](int x) { return ws.keys[locate((uint32_t)x)]; };
      const int n = (int)n_keys;
      uint32_t fin = 0;
      for (int x = tid; x < n; x += blockDim.x) fin += pred(x) ? 1u : 0u;
      fin = wave_reduce_sum(fin);
      if (lane == 0 && fin) atomicAdd(&fin_s, fin);
      __syncthreads();
      fin = fin_s;
      if (fin == 0 || rho >= (double)fin) {
        rec.pivot_excl = KEY_INF;                    // every evictable key is a candidate
      } else {
        uint32_t P, r2, e2;
        block_radix_select(hist, bc, n, (uint32_t)rho, val, pred, P, r2, e2);
        rec.pivot_excl = P + 1u;                     // P < KEY_INF
      }
    }
  }
  if (tid == 0) ws.st_seqrec[i] = rec;
}

// THE pass: metrics / positions / per-block metadata in physical order.  BS/4 lanes own a block's
// row (16 B of each store per lane); the metadata of the 64 blocks of a wave iteration is loaded
// once, coalesced, and handed to the row lanes by shuffles.  DENSE: the rows are requested before
// the metadata is looked at (most blocks belong to the batch); otherwise only the rows of the
// batch's blocks are touched (an engine-sized cache holding a small batch).
// LAZY: the position rows are not streamed at all.  A key needs its position only for the mask
// (no averaging, no position bias), and only the ~1 % of the slots whose METRIC lies below the
// pivot can become candidates: their positions are fetched when the queue is drained (one 4 B
// gather per entry, masked ones dropped there).  What is lost is the count of evictable keys per
// head, which only says whether a sequence can free the k chunks it was asked for -- and that the
// records answer themselves: k listed thresholds exist, or the flag is raised.  (The reference's
// batch > 1 rule counts the inf thresholds of every sequence and keeps the full pass.)
// 8 B + 1 B of the 12.75 B per candidate slot are then 4 B + 1 B.
template <int BS, bool DENSE, bool LAZY>
__global__ __launch_bounds__(256) void stream_collect_kernel(kvc_schedule_params p, SchedWs ws) {
  constexpr int LPB = BS / 4;                        // lanes per block
  constexpr int BPL = 64 / LPB;                      // blocks per wave load
  constexpr int U = LPB >= 4 ? 4 : 64 / BPL;         // wave loads per iteration: 64 blocks (bs 8: 2 x 32)
  constexpr int BPW = BPL * U;
  static_assert(BPW <= 64, "one metadata load covers the iteration's blocks");
  __shared__ uint32_t qk[4][128], qs[4][128], qg[4][128];
  __shared__ int32_t ql[LAZY ? 4 : 1][128];          // LAZY: highest evictable position of the entry's sequence
  __shared__ uint32_t list_s[DENSE ? 1 : SPARSE_CHUNK];   // !DENSE: (batch position << 12) | block - chunk base
  __shared__ uint32_t n_s;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int L = p.num_layers, H = p.num_kv_heads;
  unsigned long long* lists = reinterpret_cast<unsigned long long*>(ws.rec64);
  uint32_t claimed = 0;
  int qn = 0;
  auto drain = [&](int n) {                          // pops the top n (<= 64) queue entries
    wave_lds_sync();
    if (lane < n) {
      const int e = qn - n + lane;
      const uint32_t g = qg[w][e];
      bool in_range = true;
      if constexpr (LAZY) {                            // metrics.py:539-544, for the few that matter
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
  // One wave iteration: lane j < BPW looks after block mb (have: there is one); i_known >= 0: its
  // batch position is known already (sparse sweep), else the sequence index is looked up here.
  auto iteration = [&](int64_t mb, bool have, int i_known) {
    f32x4 m[U];
    i32x4 q[U];
    auto load_rows = [&](unsigned long long want) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int src = u * BPL + lane / LPB;
        const int64_t blk = DENSE ? mb - lane + src : (int64_t)(uint32_t)__shfl((int)(uint32_t)mb, src, 64);
        if ((want >> src) & 1ull) {
          m[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.metrics + blk * BS) + (lane % LPB));
          if constexpr (!LAZY)
            q[u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(p.token_positions + blk * BS) + (lane % LPB));
          else
            q[u] = i32x4{0, 0, 0, 0};
        } else {
          m[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          q[u] = i32x4{0, 0, 0, 0};
        }
      }
    };
    const unsigned long long havem = __ballot(have);
    if constexpr (DENSE) load_rows(havem);           // the rows do not wait for the metadata
    BlockMeta mt{-1, 0, 0, 0};
    if constexpr (DENSE) {
      mt = load_meta(p, mb, have);
    } else if (have) {                               // (the sweep has looked at the sequence index already)
      mt.l = p.layer_index_by_block[mb]; mt.h = p.head_index_by_block[mb];
      mt.lbn = p.logical_block_num_by_block[mb];
    }
    bool ok = have && (i_known >= 0 || (mt.s >= 0 && mt.s < p.seq_slot_len));
    int i = i_known >= 0 ? i_known : p.seq_slot_of_seq[ok ? mt.s : 0];
    ok = ok && i >= 0 && mt.l >= 0 && mt.l < L && mt.h >= 0 && mt.h < H;
    const int l = ok ? mt.l : 0, h = ok ? mt.h : 0;
    if (!ok) i = 0;
    const int ctx = p.context_lens[(l * p.num_seqs + i) * H + h];
    const SeqRec r = ws.st_seqrec[i];
    ok = ok && mt.lbn >= 0 && mt.lbn < (ctx + BS - 1) / BS;
    const unsigned long long okm = __ballot(ok);
    if (okm == 0ull) return;                         // wave-uniform
    claimed += (uint32_t)__popcll(okm);
    if constexpr (!DENSE) load_rows(okm);
    const int g = ok ? (i * L + l) * H + h : -1;
    const int seq_pos = r.seq_pos, prot = r.prot;
    const uint32_t pex = r.pivot_excl;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int src = u * BPL + lane / LPB;
      const int gg = __shfl(g, src, 64);
      const int spp = __shfl(seq_pos, src, 64), prr = __shfl(prot, src, 64);
      const uint32_t pvv = (uint32_t)__shfl((int)pex, src, 64);
      const uint32_t blk32 = DENSE ? (uint32_t)(mb - lane + src) : (uint32_t)__shfl((int)(uint32_t)mb, src, 64);
      int ll = 0, hh = 0;
      if (p.bias != nullptr) { ll = __shfl(l, src, 64); hh = __shfl(h, src, 64); }
      const float mm[4] = {m[u].x, m[u].y, m[u].z, m[u].w};
      const int qq[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
      const uint32_t slot0 = blk32 * (uint32_t)BS + (uint32_t)(lane % LPB) * 4u;
      int ninf = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t key = LAZY ? float_to_key(mm[k]) : slot_key(p, mm[k], qq[k], spp, prr, ll, hh);
        ninf += (gg >= 0 && key >= KEY_INF) ? 1 : 0;
        const bool c = gg >= 0 && key < pvv;         // (pvv <= KEY_INF)
        const unsigned long long bal = __ballot(c);
        if (bal) {                                   // wave-uniform
          if (c) {
            const int pos = qn + __popcll(bal & ((1ull << lane) - 1ull));
            qk[w][pos] = key; qs[w][pos] = slot0 + (uint32_t)k; qg[w][pos] = (uint32_t)gg;
            if constexpr (LAZY) ql[w][pos] = spp - prr;
          }
          qn += __popcll(bal);
          if (qn >= 64) drain(64);
        }
      }
      if constexpr (!LAZY) {
        // masked / non-finite slots of the block (its LPB lanes are adjacent)
#pragma unroll
        for (int d = 1; d < LPB; d <<= 1) ninf += __shfl_xor(ninf, d, 64);
        if (lane % LPB == 0 && ninf > 0) atomicAdd(&ws.st_def[gg], (uint32_t)ninf);
      }
    }
  };
  if constexpr (DENSE) {
    const int64_t wave = (int64_t)blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4;
    for (int64_t b0 = wave * BPW; b0 < p.num_blocks; b0 += nwaves * BPW) {
      const int64_t mb = b0 + lane;
      iteration(mb, lane < BPW && mb < p.num_blocks, -1);
    }
  } else {
    // An engine sizes its cache to HBM: most blocks do not belong to the batch.  A workgroup sweeps
    // SPARSE_CHUNK consecutive blocks -- every thread requests its share of the sequence indices at
    // once (one round trip), the batch's blocks are compacted into an LDS list -- and the list is
    // then worked off densely, 64 blocks per wave iteration like above (the per-block chain of
    // lookups run for every block of a 30 M-block cache cost 0.3 ms for a batch of 1 M blocks).
    const int tid = threadIdx.x;
    int sidx[SPARSE_SCAN], snext[SPARSE_SCAN];
    auto request = [&](int64_t base, int* dst) {       // the chunk's sequence indices, one round trip
#pragma unroll
      for (int u = 0; u < SPARSE_SCAN; ++u) {
        const int64_t blk = base + u * 256 + tid;
        dst[u] = blk < p.num_blocks ? p.seq_index_by_block[blk] : -1;
      }
    };
    const int64_t stride = (int64_t)gridDim.x * SPARSE_CHUNK;
    int64_t base = (int64_t)blockIdx.x * SPARSE_CHUNK;
    if (base < p.num_blocks) request(base, snext);
    for (; base < p.num_blocks; base += stride) {
      __syncthreads();
      if (tid == 0) n_s = 0;
      __syncthreads();
#pragma unroll
      for (int u = 0; u < SPARSE_SCAN; ++u) sidx[u] = snext[u];
      // the next chunk's indices are requested now and arrive while this chunk's list is worked off
      if (base + stride < p.num_blocks) request(base + stride, snext);
#pragma unroll
      for (int u = 0; u < SPARSE_SCAN; ++u) {
        const int sq = sidx[u];
        int i = -1;
        if (sq >= 0 && sq < p.seq_slot_len) i = p.seq_slot_of_seq[sq];
        const unsigned long long mask = __ballot(i >= 0);
        if (mask == 0ull) continue;                     // wave-uniform
        uint32_t wbase = 0;
        if (lane == 0) wbase = atomicAdd(&n_s, (uint32_t)__popcll(mask));
        wbase = (uint32_t)__shfl((int)wbase, 0, 64);
        if (i >= 0) list_s[wbase + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = ((uint32_t)i << 12) | (uint32_t)(u * 256 + tid);
      }
      __syncthreads();
      const int n = (int)n_s;
      for (int e0 = w * BPW; e0 < n; e0 += 4 * BPW) {
        const bool have = lane < BPW && e0 + lane < n;
        const uint32_t ent = have ? list_s[e0 + lane] : 0u;
        iteration(base + (int64_t)(ent & 4095u), have, have ? (int)(ent >> 12) : -1);
      }
    }
  }
  if (qn > 0) drain(qn);
  // blocks that are logical blocks of the batch (every one must be there, else fallback): one
  // atomic per workgroup, on one of CLAIM_SHARDS counters a cache line apart (a single word takes
  // ~12 ns per atomic: 16 k waves on it would outlast the whole pass)
  __shared__ uint32_t claimed_s;
  if (threadIdx.x == 0) claimed_s = 0;
  __syncthreads();
  if (lane == 0 && claimed) atomicAdd(&claimed_s, claimed);
  __syncthreads();
  if (threadIdx.x == 0 && claimed_s) atomicAdd(&ws.st_claimed[(blockIdx.x % CLAIM_SHARDS) * 32], claimed_s);
}

// ascending sort of one 64-bit value per lane across the wave (bitonic, shuffles only)
__device__ __forceinline__ uint64_t wave_sort64(uint64_t v) {
  const int lane = lane_id();
#pragma unroll
  for (int k = 2; k <= WAVE; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)v, j, 64);
      const uint32_t ohi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), j, 64);
      const uint64_t o = ((uint64_t)ohi << 32) | olo;
      const bool up = (lane & k) == 0, lower = (lane & j) == 0;
      v = (lower == up) ? (v < o ? v : o) : (v > o ? v : o);
    }
  return v;
}

// The candidate list sorted by (key, physical slot) is the head's record.  A wave takes HPW
// consecutive heads at once (their counts, lists and sorts are independent: one round trip and
// interleaved shuffles instead of HPW of each); lists beyond 64 entries are sorted in LDS.
template <int WAVES, int HPW>
__global__ __launch_bounds__(64 * WAVES) void stream_records_kernel(kvc_schedule_params p, SchedWs ws, int lazy) {
  __shared__ __attribute__((aligned(16))) uint64_t sort_s[WAVES][KREC];
  const int lane = lane_id();
  const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int L = p.num_layers, H = p.num_kv_heads, B = p.num_seqs;
  const int G = B * L * H;
  const int g0 = (blockIdx.x * WAVES + w) * HPW;
  if (g0 >= G) return;
  const int bs = p.block_size;
  if (g0 == 0) {                                     // a logical block of the batch has no physical block?
    const uint32_t c = wave_reduce_sum(ws.st_claimed[lane * 32]);
    static_assert(CLAIM_SHARDS == WAVE, "one shard per lane");
    if (lane == 0 && (int64_t)c != p.total_slots / bs) atomicOr(ws.fallback, 1u);
    if (ws.hv_seen_seq != nullptr) {                 // lists that carry what they were made with (harvest bit 3)
      bool bad = false;
      for (int i = lane; i < B; i += WAVE)
        bad |= ws.hv_seen_seq[2 * i] != p.seq_positions[i] || ws.hv_seen_seq[2 * i + 1] != p.num_protected[i];
      if (__ballot(bad) && lane == 0) atomicOr(ws.fallback, 1u);
    }
  }
  // lane q < HPW looks after head g0 + q: finite keys -> finite-threshold chunks of the head
  uint32_t myC = 0;
  if (lane < HPW && g0 + lane < G) {
    const int g = g0 + lane;
    const int i_seq = g / (L * H), l = (g / H) % L, h = g % H;
    const int ctx = p.context_lens[(l * B + i_seq) * H + h];
    const uint32_t nblk = (uint32_t)((ctx + bs - 1) / bs);
    myC = ws.st_cnt[g];
    if (ws.hv_seen_ctx != nullptr && ws.hv_seen_ctx[g] != ctx && ws.hv_seen_ctx[g] != -2) atomicOr(ws.fallback, 1u);   // another batch's list (-2: not recorded)
    if (!lazy) {                                     // (lazy: nobody counted the masked slots, nobody needs them)
      const uint32_t F = nblk * (uint32_t)bs - ws.st_def[g];
      ws.head_fc[g] = nchunks_freed(F, (uint32_t)p.hanging_token_count[g], (uint32_t)bs);   // finite-threshold chunks
      ws.head_fc[G + g] = nblk;                                                              // all chunks
    }
    if (myC > (uint32_t)KREC) atomicOr(ws.fallback, 1u);
  }
  uint32_t C[HPW];
  uint64_t v[HPW];
#pragma unroll
  for (int q = 0; q < HPW; ++q) {
    C[q] = (uint32_t)__shfl((int)myC, q, 64);
    v[q] = ~0ull;
    if (C[q] > 1u && C[q] <= (uint32_t)WAVE && (uint32_t)lane < C[q]) v[q] = ws.rec64[(int64_t)(g0 + q) * KREC + lane];
  }
#pragma unroll
  for (int q = 0; q < HPW; ++q)
    if (C[q] > 1u && C[q] <= (uint32_t)WAVE) v[q] = wave_sort64(v[q]);          // wave-uniform condition
#pragma unroll
  for (int q = 0; q < HPW; ++q)
    if (C[q] > 1u && C[q] <= (uint32_t)WAVE && (uint32_t)lane < C[q]) ws.rec64[(int64_t)(g0 + q) * KREC + lane] = v[q];
#pragma unroll
  for (int q = 0; q < HPW; ++q) {
    if (C[q] <= (uint32_t)WAVE || C[q] > (uint32_t)KREC) continue;              // wave-uniform
    uint64_t* rec = ws.rec64 + (int64_t)(g0 + q) * KREC;
    uint64_t* a = sort_s[w];
    const int SZ = C[q] <= 128u ? 128 : 256;
    wave_lds_sync();
    for (int j = lane; j < SZ; j += WAVE) a[j] = (uint32_t)j < C[q] ? rec[j] : ~0ull;
    wave_lds_sync();
    if (SZ == 128) wave_bitonic_sort<uint64_t, 128>(a);
    else wave_bitonic_sort<uint64_t, 256>(a);
    for (int j = lane; j < (int)C[q]; j += WAVE) rec[j] = a[j];
  }
}

// per sequence: finite-threshold chunks and all chunks, from the per-head counts stream_records left
// -> seq_tmp, where seq_prepare_kernel expects them (only the reference's batch > 1 rule needs this
// and the launch behind it: otherwise seq_select_topk_kernel finds its k' itself)
__global__ __launch_bounds__(256) void seq_sums_topk_kernel(kvc_schedule_params p, SchedWs ws) {
  __shared__ uint32_t red[2][4];
  const int B = p.num_seqs, LH = p.num_layers * p.num_kv_heads, G = B * LH;
  const int i = blockIdx.x;
  uint32_t f = 0, cn = 0;
  for (int lh = threadIdx.x; lh < LH; lh += blockDim.x) {
    f += ws.head_fc[(int64_t)i * LH + lh];
    cn += ws.head_fc[(int64_t)G + (int64_t)i * LH + lh];
  }
  f = wave_reduce_sum(f);
  cn = wave_reduce_sum(cn);
  if (lane_id() == 0) { red[0][threadIdx.x / WAVE] = f; red[1][threadIdx.x / WAVE] = cn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ws.seq_tmp[i] = (int32_t)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    ws.seq_tmp[B + i] = (int32_t)(red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// one workgroup per sequence: sort the recorded thresholds of its heads by (threshold, head,
// chunk); the first k' are the freed chunks (metrics.py:704-729 + 773-792)
__global__ __launch_bounds__(1024) void seq_select_topk_kernel(kvc_schedule_params p, SchedWs ws, int P2, int coupled) {
  extern __shared__ __attribute__((aligned(16))) uint8_t sel_lds[];
  uint64_t* arr = reinterpret_cast<uint64_t*>(sel_lds);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(arr + P2);
  const int i = blockIdx.x;
  const int LH = p.num_layers * p.num_kv_heads;
  const uint32_t bs = (uint32_t)p.block_size;
  const int MCH = KREC / p.block_size;               // thresholds a record holds
  const int tid = threadIdx.x;
  __shared__ uint32_t fsum_s;
  if (coupled == 2) {
    // lazy pass: the evictable keys were not counted.  k' = min(k, finite-threshold chunks) is k
    // whenever the records list k thresholds (all of them finite); if they do not, the flag is
    // raised below like for any record that falls short
    if (tid == 0) ws.seq_k[i] = max(p.evicted_blocks_per_seq[i], 0);
    __syncthreads();
  } else if (!coupled) {
    // k' = min(k, finite-threshold chunks of the sequence): what seq_prepare_body gives for
    // mode 1 or a single sequence (the reference's batch > 1 rule ran seq_prepare_kernel instead)
    if (tid == 0) fsum_s = 0;
    __syncthreads();
    uint32_t f = 0;
    for (int lh = tid; lh < LH; lh += blockDim.x) f += ws.head_fc[(int64_t)i * LH + lh];
    f = wave_reduce_sum(f);
    if (lane_id() == 0 && f) atomicAdd(&fsum_s, f);
    __syncthreads();
    if (tid == 0) {
      const int kk = p.evicted_blocks_per_seq[i];
      ws.seq_k[i] = kk <= 0 ? 0 : (int32_t)((uint32_t)kk < fsum_s ? (uint32_t)kk : fsum_s);
    }
    __syncthreads();
  }
  const uint32_t k = (uint32_t)ws.seq_k[i];
  for (int e = tid; e < P2; e += blockDim.x) {
    const int lh = e / MCH, c = e % MCH;
    uint64_t v = ~0ull;
    if (lh < LH && k > 0) {
      const int64_t g = (int64_t)i * LH + lh;
      const uint32_t hang = (uint32_t)p.hanging_token_count[g];
      const uint32_t have = min(ws.st_cnt[g], (uint32_t)KREC);
      const uint32_t r = hang - 1u + (uint32_t)c * bs;          // rank - 1 of threshold c
      if (hang >= 1u && r < have) v = (ws.rec64[g * KREC + r] & 0xFFFFFFFF00000000ull) | (uint32_t)e;
    }
    arr[e] = v;
  }
  for (int lh = tid; lh < LH; lh += blockDim.x) cnt[lh] = 0;
  __syncthreads();
  // the k'-th smallest entry by an MSB-first radix select over the 64-bit (threshold, head,
  // chunk) values in LDS -- eight byte rounds of one histogram each (a full bitonic sort of the
  // 4096 entries of 256 heads took 40 of this kernel's 54 us, for k' = 16)
  __shared__ uint32_t sel_hist[RADIX];
  __shared__ uint32_t sel_wtot[4];
  __shared__ uint32_t sel_digit, sel_krem;
  uint64_t vstar = ~0ull;
  if (k > 0 && k <= (uint32_t)P2) {
    uint64_t prefix = 0;
    uint32_t krem = k;
    for (int round = 0; round < 8; ++round) {
      const int shift = 56 - 8 * round;
      if (tid < RADIX) sel_hist[tid] = 0;
      __syncthreads();
      for (int e0 = 0; e0 < P2; e0 += blockDim.x) {          // uniform trip count (ballots inside)
        const int e = e0 + tid;
        const uint64_t v = e < P2 ? arr[e] : 0ull;
        const bool in = e < P2 && (round == 0 || (v >> (shift + 8)) == prefix);
        hist_add(sel_hist, in, (uint32_t)(v >> shift) & 0xFFu);
      }
      __syncthreads();
      uint32_t c = 0, inc = 0;
      if (tid < RADIX) {
        c = sel_hist[tid];
        inc = wave_inclusive_scan(c);
        if ((tid & 63) == 63) sel_wtot[tid >> 6] = inc;
      }
      __syncthreads();
      if (tid < RADIX) {
        uint32_t off = 0;
        for (int q = 0; q < (tid >> 6); ++q) off += sel_wtot[q];
        const uint32_t incl = off + inc, excl = incl - c;
        if (krem > excl && krem <= incl) { sel_digit = (uint32_t)tid; sel_krem = krem - excl; }
      }
      __syncthreads();
      prefix = (prefix << 8) | sel_digit;
      krem = sel_krem;
    }
    vstar = prefix;
    // (k' > number of recorded thresholds: the select ends on the ~0 padding)
    if (vstar == ~0ull) { if (tid == 0) atomicOr(ws.fallback, 1u); }
    else
      for (int e = tid; e < P2; e += blockDim.x) {
        const uint64_t v = arr[e];
        if (v <= vstar) atomicAdd(&cnt[(uint32_t)v / (uint32_t)MCH], 1u);
      }
  } else if (k > (uint32_t)P2) {
    if (tid == 0) atomicOr(ws.fallback, 1u);             // the records do not hold k' thresholds
  }
  __syncthreads();
  const uint32_t Tstar = vstar != ~0ull ? (uint32_t)(vstar >> 32) : 0u;
  if (tid == 0) ws.seq_prefix[i] = Tstar;
  for (int lh = tid; lh < LH; lh += blockDim.x) {
    const int64_t g = (int64_t)i * LH + lh;
    const uint32_t hang = (uint32_t)p.hanging_token_count[g];
    const uint32_t n = k > 0 ? cnt[lh] : 0u;
    // (every threshold a record does not list is a key above the sequence's pivot, hence above
    // every listed one: nothing to check here; a list that overflowed raised the flag already)
    p.evicted_block_count[g] = (int32_t)n;
    p.evicted_kv_count[g] = n > 0 ? (int32_t)((n - 1) * bs + hang) : 0;
  }
}

// logical slot index of a physical slot (the block's own metadata row)
__device__ __forceinline__ uint32_t logical_of(const kvc_schedule_params& p, uint32_t phys_slot) {
  const uint32_t bs = (uint32_t)p.block_size;
  return (uint32_t)p.logical_block_num_by_block[phys_slot / bs] * bs + phys_slot % bs;
}

// one wave per head: the first cnt record entries, ascending by logical index  (metrics.py:822-834)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void emit_topk_kernel(kvc_schedule_params p, SchedWs ws) {
  __shared__ uint32_t sort_s[WAVES][KREC];
  if (*ws.fallback != 0u) return;                    // the general pipeline (gated behind) writes everything
  const int lane = lane_id();
  const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  const int g = blockIdx.x * WAVES + w;
  if (g >= G) return;
  const uint32_t cnt = (uint32_t)p.evicted_kv_count[g];
  if (p.eli_dirty_map != nullptr && !(p.lean & 1)) {
    // a tracked output buffer: no null fill of the whole list -- what earlier calls left behind in
    // this head's segment beyond the cnt entries written below is cleared here, and marked
    const int64_t off = p.evicted_kv_offsets[g];
    const int64_t end = (g + 1 < G) ? (int64_t)p.evicted_kv_offsets[g + 1] : p.total_slots;
    const int bsz = p.block_size;
    eli_dirty_update(p.eli_dirty_map, p.evicted_logical_indices, off / bsz, end / bsz, ((int64_t)cnt + bsz - 1) / bsz,
                     off + cnt, bsz, p.null_value, true, lane, WAVE);
  }
  if (cnt == 0) return;
  int32_t* out = p.evicted_logical_indices + p.evicted_kv_offsets[g];
  if (cnt <= (uint32_t)WAVE) {
    // the usual case (a block or two per head): one index per lane, bitonic sort across the lanes
    uint32_t v = 0xFFFFFFFFu;
    if ((uint32_t)lane < cnt) v = logical_of(p, (uint32_t)ws.rec64[(int64_t)g * KREC + lane]);
#pragma unroll
    for (int k = 2; k <= WAVE; k <<= 1)
#pragma unroll
      for (int j = k >> 1; j > 0; j >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, j, 64);
        const bool up = (lane & k) == 0, lower = (lane & j) == 0;
        v = (lower == up) ? (v < o ? v : o) : (v > o ? v : o);
      }
    if ((uint32_t)lane < cnt) out[lane] = (int32_t)v;
    return;
  }
  uint32_t* a = sort_s[w];
  for (int j = lane; j < KREC; j += WAVE)
    a[j] = (uint32_t)j < cnt ? logical_of(p, (uint32_t)ws.rec64[(int64_t)g * KREC + j]) : 0xFFFFFFFFu;
  wave_lds_sync();
  wave_bitonic_sort<uint32_t, KREC>(a);
  for (int j = lane; j < (int)cnt; j += WAVE) out[j] = (int32_t)a[j];
}

// general pipeline behind the small-eviction schedule (gated): the chunk table is cleared by a gated
// kernel instead of a memset (nothing runs unless the flag was raised), and the keys of chunks
// nobody claimed, which no memset cleared on that path, are set afterwards
__device__ __forceinline__ void clear_chunk_table_body(const kvc_schedule_params& p, SchedWs& ws, unsigned bid, unsigned nb) {
  const int64_t nchunks = p.total_slots / p.block_size;
  for (int64_t c = (int64_t)bid * blockDim.x + threadIdx.x; c < nchunks; c += (int64_t)nb * blockDim.x)
    ws.chunk_phys[c] = -1;
}
__device__ __forceinline__ void fix_unclaimed_body(const kvc_schedule_params& p, SchedWs& ws, unsigned bid, unsigned nb) {
  const int64_t nchunks = p.total_slots / p.block_size;
  for (int64_t c = (int64_t)bid * blockDim.x + threadIdx.x; c < nchunks; c += (int64_t)nb * blockDim.x)
    if (ws.chunk_phys[c] < 0)
      for (int o = 0; o < p.block_size; ++o) ws.keys[c * p.block_size + o] = 0xFFFFFFFFu;
}
__global__ __launch_bounds__(256) void clear_chunk_table_kernel(kvc_schedule_params p, SchedWs ws) {
  if (gated_off(ws)) return;
  clear_chunk_table_body(p, ws, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void fix_unclaimed_kernel(kvc_schedule_params p, SchedWs ws) {
  if (gated_off(ws)) return;
  fix_unclaimed_body(p, ws, blockIdx.x, gridDim.x);
}


}  // namespace kvc
