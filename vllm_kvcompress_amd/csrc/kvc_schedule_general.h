// kvc_schedule_general.h -- A3 schedule_evictions: the general pipeline (keys, digit rounds, scan + pick, select + emit)
// (one translation unit: included by kvc_schedule.hip in this order; see the overview there)
#pragma once
#include "kvc_common.h"
#include "kvc_schedule_common.h"
#include "../../include/kvc_mi355x.h"

namespace kvc {

// ------------------------------------------------------------------ 0. keys
// effective metric -> order-preserving key                   metrics.py:495-544
__device__ __forceinline__ uint32_t slot_key(const kvc_schedule_params& p, float m, int pos, int seq_pos,
                                             int prot, int l, int h) {
  if (p.use_average) m = __fdiv_rn(m, (float)(seq_pos - pos));          // :495-501
  if (p.bias != nullptr) {                                               // :503-506, :54-81
    int cnt = 0;
    for (int k = 0; k < p.num_bins; ++k) cnt += pos >= p.position_bins[k];
    int bi = cnt - 1;
    if (bi < 0) bi += p.num_bins;
    float b = p.bias[((int64_t)l * p.num_kv_heads + h) * p.num_bins + bi];
    if (pos < 0) b = 0.0f;
    m = __fadd_rn(m, __fmul_rn(b, p.bias_weight));
  }
  const bool in_range = pos <= seq_pos - prot && pos >= p.num_sinks;   // :539-544
  return in_range ? float_to_key(m) : KEY_INF;
}

// one thread per VEC consecutive slots of a physical block (VEC = 4: 16 B loads and stores)
// the counters of the later passes <- 0 (workgroup bid of nb)
__device__ __forceinline__ void zero_body(uint4* zero16, int64_t zero_vecs, unsigned bid, unsigned nb) {
  for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < zero_vecs; i += (int64_t)nb * 256)
    zero16[i] = make_uint4(0u, 0u, 0u, 0u);
}

// bracket schedule: the key of a cell's sampled slot, if it is one of the four (the one) just built at dst
__device__ __forceinline__ void sample_keys(const kvc_schedule_params& p, SchedWs& ws, int i, int64_t dst, const uint4& k) {
  const int LH = p.num_layers * p.num_kv_heads;
  const int64_t sb = p.evicted_kv_offsets[i * LH];
  const int64_t se = i + 1 < p.num_seqs ? (int64_t)p.evicted_kv_offsets[(i + 1) * LH] : true_n(p, ws);
  const int lg = bracket_stride_log2((uint32_t)(se - sb));
  const uint32_t at0 = (uint32_t)(dst - sb);
  const uint32_t cell = at0 >> lg;
  const uint32_t t = bracket_cell_slot(cell, (uint32_t)i, lg) - at0;
  if (t < 4u) ws.bsample[(int64_t)i * BR_CELLS + cell] = t == 0u ? k.x : (t == 1u ? k.y : (t == 2u ? k.z : k.w));
}
__device__ __forceinline__ void sample_key(const kvc_schedule_params& p, SchedWs& ws, int i, int64_t dst, uint32_t k) {
  const int LH = p.num_layers * p.num_kv_heads;
  const int64_t sb = p.evicted_kv_offsets[i * LH];
  const int64_t se = i + 1 < p.num_seqs ? (int64_t)p.evicted_kv_offsets[(i + 1) * LH] : true_n(p, ws);
  const int lg = bracket_stride_log2((uint32_t)(se - sb));
  const uint32_t at = (uint32_t)(dst - sb);
  const uint32_t cell = at >> lg;
  if (bracket_cell_slot(cell, (uint32_t)i, lg) == at) ws.bsample[(int64_t)i * BR_CELLS + cell] = k;
}

// (bodies take the workgroup's index and the number of workgroups as arguments: the kernels below
// pass blockIdx / gridDim, the single-launch fallback of the small-eviction schedule its own)
// one add per workgroup on one of CLAIM_SHARDS counters a cache line apart (what the workgroup's threads counted)
__device__ __forceinline__ void claim_flush(uint32_t* shards, uint32_t mine, unsigned bid) {
  __shared__ uint32_t claim_s;
  if (threadIdx.x == 0) claim_s = 0u;
  __syncthreads();
  const uint32_t w = wave_reduce_sum(mine);
  if (lane_id() == 0 && w) atomicAdd(&claim_s, w);
  __syncthreads();
  if (threadIdx.x == 0 && claim_s) atomicAdd(&shards[(bid % (unsigned)CLAIM_SHARDS) * 32], claim_s);
}

// COUNT: the logical blocks that found their physical block are counted into ws.bclaim (the bracket schedule's
// stand-in for the 0xFF fill: every slot was written if the count is N / bs)
template <int VEC, bool COUNT = false>
__device__ __forceinline__ void build_keys_body(const kvc_schedule_params& p, SchedWs& ws, unsigned bid, unsigned data_blocks) {
  const int bs = p.block_size;
  const int per_blk = bs / VEC;
  uint32_t claimed = 0;
  const BlockDiv pd(per_blk);                         // (thread -> block, offset without a run-time division per thread)
  const bool narrow = p.num_blocks * per_blk < ((int64_t)1 << 31);
  // (grid-stride: behind the small-eviction schedule this kernel is launched gated, with a small grid)
  for (int64_t tid = (int64_t)bid * blockDim.x + threadIdx.x; tid < p.num_blocks * per_blk;
       tid += (int64_t)data_blocks * blockDim.x) {
  int64_t blk;
  int off;
  if (narrow) { int q, r; pd.divmod((int)tid, q, r); blk = q; off = r * VEC; }
  else { blk = tid / per_blk; off = (int)(tid % per_blk) * VEC; }
  // free blocks (an engine's cache is sized to HBM: most blocks do not belong to the batch) cost
  // their 4 B of sequence index and nothing else; for the others the wide loads do not depend on
  // the rest of the metadata chain below and are issued first
  const int s = p.seq_index_by_block[blk];
  // (this kernel runs when most blocks belong to the batch -- a sparse cache takes build_keys_sparse_body --: the other
  // three metadata rows are requested with the first, not behind the two tests below: one round trip less per thread)
  const int l = p.layer_index_by_block[blk], h = p.head_index_by_block[blk];
  const int lbn = p.logical_block_num_by_block[blk];
  if (s < 0 || s >= p.seq_slot_len) continue;
  float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f);
  int4 q4 = make_int4(0, 0, 0, 0);
  if constexpr (VEC == 4) {
    m4 = *reinterpret_cast<const float4*>(p.metrics + blk * bs + off);
    q4 = *reinterpret_cast<const int4*>(p.token_positions + blk * bs + off);
  }
  const int i = p.seq_slot_of_seq[s];
  if (i < 0) continue;
  const int L = p.num_layers, H = p.num_kv_heads, B = p.num_seqs;
  // (a block whose rows do not name a head of the batch is inconsistent metadata: skipped like a free one)
  if (l < 0 || l >= L || h < 0 || h >= H) continue;
  const int g = (i * L + l) * H + h;
  // (requested together, in front of the test that needs only the first)
  const int ctx = p.context_lens[(l * B + i) * H + h];
  const int seq_pos = p.seq_positions[i], prot = p.num_protected[i];
  const int64_t base = p.evicted_kv_offsets[g];
  const int nblk = (ctx + bs - 1) / bs;
  if (lbn < 0 || lbn >= nblk) continue;        // not part of the head's slot range
  const int64_t src = blk * bs + off, dst = base + (int64_t)lbn * bs + off;
  if constexpr (VEC == 4) {
    const float4 m = m4;
    const int4 q = q4;
    uint4 k;
    k.x = slot_key(p, m.x, q.x, seq_pos, prot, l, h);
    k.y = slot_key(p, m.y, q.y, seq_pos, prot, l, h);
    k.z = slot_key(p, m.z, q.z, seq_pos, prot, l, h);
    k.w = slot_key(p, m.w, q.w, seq_pos, prot, l, h);
    *reinterpret_cast<uint4*>(ws.keys + dst) = k;
    if (ws.bsample != nullptr) sample_keys(p, ws, i, dst, k);
    if (ws.bnonfin != nullptr) {
      const uint32_t c = (k.x >= KEY_INF) + (k.y >= KEY_INF) + (k.z >= KEY_INF) + (k.w >= KEY_INF);
      if (c) atomicAdd(&ws.bnonfin[g], c);
    }
  } else {
    const uint32_t k1 = slot_key(p, p.metrics[src], p.token_positions[src], seq_pos, prot, l, h);
    ws.keys[dst] = k1;
    if (ws.bsample != nullptr) sample_key(p, ws, i, dst, k1);
    if (ws.bnonfin != nullptr && k1 >= KEY_INF) atomicAdd(&ws.bnonfin[g], 1u);
  }
  if (off == 0) { ws.chunk_phys[base / bs + lbn] = (int32_t)blk; claimed += 1u; }
  }
  if constexpr (COUNT) claim_flush(ws.bclaim, claimed, bid);
}

template <int VEC>
__global__ __launch_bounds__(256) void build_keys_kernel(kvc_schedule_params p, SchedWs ws,
                                                         unsigned data_blocks, uint4* zero16, int64_t zero_vecs) {
  if (gated_off(ws)) return;
  if (blockIdx.x >= data_blocks) {      // tail workgroups clear the counters of the later passes
    zero_body(zero16, zero_vecs, blockIdx.x - data_blocks, gridDim.x - data_blocks);
    return;
  }
  if (ws.bclaim != nullptr) build_keys_body<VEC, true>(p, ws, blockIdx.x, data_blocks);
  else build_keys_body<VEC>(p, ws, blockIdx.x, data_blocks);
}

// The same pass for bs in {4, 8, 16, 32, 64} (16 B per thread), organised so that blocks OUTSIDE the
// batch cost one coalesced 4 B read and nothing else.  An engine sizes its cache to HBM: most
// blocks do not belong to the sequences being compressed.  A workgroup sweeps SPARSE_CHUNK
// consecutive blocks: every thread requests its share of the sequence indices at once (one round
// trip), the blocks of the batch are compacted into an LDS list, and the list is then worked off
// densely, one thread per 4 slots like build_keys_kernel.  (History: one thread per 4 slots of
// EVERY block 0.42 ms for a 32 M-block cache holding one 32k sequence, bound by the latency of the
// per-thread index load; one wave per 64 blocks 0.18 ms, bound by the dependent loads of the few
// batch blocks a wave finds; this form 0.07 ms.)
constexpr int SPARSE_SCAN = 16;                       // index loads in flight per thread
constexpr int SPARSE_CHUNK = 256 * SPARSE_SCAN;       // blocks per workgroup sweep
template <bool COUNT = false>
__device__ __forceinline__ void build_keys_sparse_body(const kvc_schedule_params& p, SchedWs& ws, unsigned bid, unsigned data_blocks) {
  uint32_t claimed = 0;
  __shared__ uint32_t list_s[SPARSE_CHUNK];           // (batch position of the sequence << 12) | block - chunk base
  static_assert(SPARSE_CHUNK <= 4096, "12 bits of block offset");
  __shared__ uint32_t n_s;
  const int bs = p.block_size;
  const int per_blk = bs / 4;
  const int tid = threadIdx.x, lane = lane_id();
  const int L = p.num_layers, H = p.num_kv_heads, B = p.num_seqs;
  for (int64_t base = (int64_t)bid * SPARSE_CHUNK; base < p.num_blocks; base += (int64_t)data_blocks * SPARSE_CHUNK) {
    if (tid == 0) n_s = 0;
    __syncthreads();
    int sidx[SPARSE_SCAN];
#pragma unroll
    for (int u = 0; u < SPARSE_SCAN; ++u) {
      const int64_t blk = base + u * 256 + tid;
      sidx[u] = blk < p.num_blocks ? p.seq_index_by_block[blk] : -1;
    }
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
      if (i >= 0) {
        const uint32_t pos = wbase + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        list_s[pos] = ((uint32_t)i << 12) | (uint32_t)(u * 256 + tid);
      }
    }
    __syncthreads();
    const int items = (int)n_s * per_blk;
    for (int it = tid; it < items; it += 256) {
      const int e = it / per_blk;
      const int off = (it % per_blk) * 4;
      const uint32_t ent = list_s[e];
      const int64_t blk = base + (ent & 4095u);
      const int i = (int)(ent >> 12);
      const float4 m = *reinterpret_cast<const float4*>(p.metrics + blk * bs + off);
      const int4 q = *reinterpret_cast<const int4*>(p.token_positions + blk * bs + off);
      const int l = p.layer_index_by_block[blk], h = p.head_index_by_block[blk];
      const int lbn = p.logical_block_num_by_block[blk];
      const int g = (i * L + l) * H + h;
      const int ctx = p.context_lens[(l * B + i) * H + h];
      if (lbn < 0 || lbn >= (ctx + bs - 1) / bs) continue;       // not part of the head's slot range
      const int seq_pos = p.seq_positions[i], prot = p.num_protected[i];
      const int64_t base_g = p.evicted_kv_offsets[g];
      uint4 kq;
      kq.x = slot_key(p, m.x, q.x, seq_pos, prot, l, h);
      kq.y = slot_key(p, m.y, q.y, seq_pos, prot, l, h);
      kq.z = slot_key(p, m.z, q.z, seq_pos, prot, l, h);
      kq.w = slot_key(p, m.w, q.w, seq_pos, prot, l, h);
      *reinterpret_cast<uint4*>(ws.keys + base_g + (int64_t)lbn * bs + off) = kq;
      if (ws.bsample != nullptr) sample_keys(p, ws, i, base_g + (int64_t)lbn * bs + off, kq);
      if (ws.bnonfin != nullptr) {
        const uint32_t c = (kq.x >= KEY_INF) + (kq.y >= KEY_INF) + (kq.z >= KEY_INF) + (kq.w >= KEY_INF);
        if (c) atomicAdd(&ws.bnonfin[g], c);
      }
      if (off == 0) { ws.chunk_phys[base_g / bs + lbn] = (int32_t)blk; claimed += 1u; }
    }
    __syncthreads();
  }
  if constexpr (COUNT) claim_flush(ws.bclaim, claimed, bid);
}

__global__ __launch_bounds__(256) void build_keys_sparse_kernel(kvc_schedule_params p, SchedWs ws,
                                                                unsigned data_blocks, uint4* zero16, int64_t zero_vecs) {
  if (gated_off(ws)) return;
  if (blockIdx.x >= data_blocks) {      // tail workgroups clear the counters of the later passes
    zero_body(zero16, zero_vecs, blockIdx.x - data_blocks, gridDim.x - data_blocks);
    return;
  }
  if (ws.bclaim != nullptr) build_keys_sparse_body<true>(p, ws, blockIdx.x, data_blocks);
  else build_keys_sparse_body(p, ws, blockIdx.x, data_blocks);
}

// The same keys in LOGICAL order through the caller's block tables (kvc_schedule_params.block_tables,
// optional): a thread takes four consecutive slots of a head, looks its physical block up and reads
// the two rows.  For a batch that is sparse in its cache -- an engine sizes the cache to HBM -- this
// replaces the sweep over every block's sequence index and the five scattered accesses per batch
// block that follow it (layer, head, logical number; key and chunk-table stores) by two row reads
// and one 4-byte check (the block must still name the sequence as its owner: a detached block is an
// unclaimed chunk, as in the sweep); keys, chunk table and sample are written side by side and
// completely, so nothing has to be cleared first.  (108 -> 36 us for one 32k sequence in a 222 GiB cache.)
__global__ __launch_bounds__(256) void build_keys_tables_kernel(kvc_schedule_params p, SchedWs ws, unsigned data_blocks,
                                                                uint4* zero16, int64_t zero_vecs) {
  if (gated_off(ws)) return;
  if (blockIdx.x >= data_blocks) {      // tail workgroups clear the counters of the later passes
    zero_body(zero16, zero_vecs, blockIdx.x - data_blocks, gridDim.x - data_blocks);
    return;
  }
  const int H = p.num_kv_heads, LH = p.num_layers * H, G = p.num_seqs * LH, bs = p.block_size;
  const int64_t N = true_n(p, ws);
  for (int64_t t0 = (int64_t)blockIdx.x * 1024; t0 < N; t0 += (int64_t)data_blocks * 1024) {
    const int64_t idx0 = t0 + 4 * threadIdx.x;
    if (idx0 >= N) continue;
    int g = upper_bound_minus1(p.evicted_kv_offsets, G, t0);       // (the same walk in every thread of the workgroup)
    while (g + 1 < G && (int64_t)p.evicted_kv_offsets[g + 1] <= idx0) ++g;
    const int64_t base = p.evicted_kv_offsets[g];
    const int lbn = (int)((idx0 - base) / bs), off = (int)((idx0 - base) % bs);
    const int i = g / LH, l = (g % LH) / H, h = g % H;
    const int sq = p.seq_index_of_slot[i];
    int64_t blk = -1;
    if (lbn < p.block_tables_width && sq >= 0 && sq < p.max_num_seqs)
      blk = p.block_tables[(((int64_t)l * p.max_num_seqs + sq) * H + h) * p.block_tables_width + lbn];
    const bool ok = blk >= 0 && blk < p.num_blocks && p.seq_index_by_block[blk] == sq;
    uint4 k = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    if (ok) {
      const float4 m = *reinterpret_cast<const float4*>(p.metrics + blk * bs + off);
      const int4 q = *reinterpret_cast<const int4*>(p.token_positions + blk * bs + off);
      const int seq_pos = p.seq_positions[i], prot = p.num_protected[i];
      k.x = slot_key(p, m.x, q.x, seq_pos, prot, l, h);
      k.y = slot_key(p, m.y, q.y, seq_pos, prot, l, h);
      k.z = slot_key(p, m.z, q.z, seq_pos, prot, l, h);
      k.w = slot_key(p, m.w, q.w, seq_pos, prot, l, h);
      if (ws.bnonfin != nullptr) {
        const uint32_t c = (k.x >= KEY_INF) + (k.y >= KEY_INF) + (k.z >= KEY_INF) + (k.w >= KEY_INF);
        if (c) atomicAdd(&ws.bnonfin[g], c);
      }
    }
    *reinterpret_cast<uint4*>(ws.keys + idx0) = k;
    if (ws.bsample != nullptr) sample_keys(p, ws, i, idx0, k);
    if (off == 0) ws.chunk_phys[base / bs + lbn] = ok ? (int32_t)blk : -1;
  }
}

// ------------------------------------------------------------------ 1. per-head histograms
// flat tiles of TILE keys; a tile inside one head (the common case) accumulates in LDS.
constexpr int HTILE = 2048;
constexpr int HSEG_MAX = 8;      // head segments of a tile handled by LDS passes; more -> global atomics
// Persistent: every workgroup walks a contiguous range of HTILE-key tiles.  The head of the
// first tile is found by one binary search, later tiles advance it incrementally; counts of
// consecutive tiles of one head stay in LDS and are flushed once per head.
__device__ __forceinline__ void hist_round_body(const kvc_schedule_params& p, SchedWs& ws, int round, unsigned bid, unsigned nb) {
  __shared__ uint32_t sh[RADIX];
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  const int LH = p.num_layers * p.num_kv_heads;
  const int64_t N = true_n(p, ws);
  const int shift = 24 - 8 * round;
  const int64_t ntiles = (N + HTILE - 1) / HTILE;
  const int64_t tb = ntiles * bid / nb, te = ntiles * (bid + 1) / nb;
  if (tb >= te) return;
  int g = wave_upper_bound_minus1(p.evicted_kv_offsets, G, tb * HTILE);
  int64_t g_end = (g + 1 < G) ? (int64_t)p.evicted_kv_offsets[g + 1] : N;
  int cur_g = -1;                                   // head whose counts sit in sh
  auto flush = [&]() {                              // uniform call sites only
    __syncthreads();
    if (cur_g >= 0)
      for (int k = threadIdx.x; k < RADIX; k += blockDim.x) {
        const uint32_t v = sh[k];
        if (v) atomicAdd(&ws.hist[(int64_t)cur_g * RADIX + k], v);
      }
    __syncthreads();
    for (int k = threadIdx.x; k < RADIX; k += blockDim.x) sh[k] = 0;
    __syncthreads();
  };
  flush();
  constexpr int U = HTILE / 256;
  for (int64_t t = tb; t < te; ++t) {
    const int64_t t0 = t * HTILE, t1 = min(N, t0 + HTILE);
    uint32_t kv[U];                                 // all loads of the tile first (independent)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t idx = t0 + threadIdx.x + (int64_t)u * 256;
      kv[u] = idx < t1 ? ws.keys[idx] : 0xFFFFFFFFu;
    }
    while (t0 >= g_end && g + 1 < G) { ++g; g_end = (g + 1 < G) ? (int64_t)p.evicted_kv_offsets[g + 1] : N; }
    if (t1 <= g_end) {                              // the whole tile belongs to head g
      if (g != cur_g) { flush(); cur_g = g; }
      const int i = g / LH;
      if (round > 0 && ws.seq_k[i] == 0) continue;  // inactive sequence
      const uint32_t prefix = ws.seq_prefix[i];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t key = kv[u];
        const bool valid = key < KEY_INF && (round == 0 || (key >> (shift + 8)) == prefix);
        hist_add(sh, valid, (key >> shift) & 0xFFu);
      }
      continue;
    }
    // head boundaries inside the tile (scalar walk, capped)
    int nseg = 1;
    for (int gk = g + 1; gk < G && nseg <= HSEG_MAX && (int64_t)p.evicted_kv_offsets[gk] < t1; ++gk) ++nseg;
    if (nseg <= HSEG_MAX) {
      // one LDS pass per head segment of the tile (keys stay in registers).
      // With heads of a few thousand slots (continual-compression steady state) every
      // second or third tile has a boundary; per-key global atomics there cost 3x the
      // whole pass because the top digits are degenerate.
      int64_t seg_b = t0;
      for (int sgi = 0; sgi < nseg; ++sgi) {
        while (seg_b >= g_end && g + 1 < G) { ++g; g_end = (g + 1 < G) ? (int64_t)p.evicted_kv_offsets[g + 1] : N; }
        const int64_t seg_e = min(t1, g_end);
        if (g != cur_g) { flush(); cur_g = g; }
        const int i = g / LH;
        if (round == 0 || ws.seq_k[i] != 0) {       // else: inactive sequence
          const uint32_t prefix = ws.seq_prefix[i];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int64_t idx = t0 + threadIdx.x + (int64_t)u * 256;
            const uint32_t key = kv[u];
            const bool valid = idx >= seg_b && idx < seg_e && key < KEY_INF &&
                (round == 0 || (key >> (shift + 8)) == prefix);
            hist_add(sh, valid, (key >> shift) & 0xFFu);
          }
        }
        seg_b = seg_e;
      }
    } else {                                        // many tiny heads in this tile
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t idx = t0 + threadIdx.x + (int64_t)u * 256;
        int gk = g;
        int64_t ek = g_end;
        while (idx >= ek && gk + 1 < G) { ++gk; ek = (gk + 1 < G) ? (int64_t)p.evicted_kv_offsets[gk + 1] : N; }
        const int i = gk / LH;
        const uint32_t key = kv[u];
        const bool valid = idx < t1 && key < KEY_INF &&
            (round == 0 || (ws.seq_k[i] != 0 && (key >> (shift + 8)) == ws.seq_prefix[i]));
        hist_add(ws.hist, valid, (uint32_t)gk * RADIX + ((key >> shift) & 0xFFu));
      }
    }
  }
  flush();
}

__global__ __launch_bounds__(256) void hist_round_kernel(kvc_schedule_params p, SchedWs ws, int round) {
  if (gated_off(ws)) return;
  hist_round_body(p, ws, round, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------ 2. chunks per sequence
// (the per-head scan, the per-sequence totals and the pick of the digit live in scan_pick_body, 5a)
// ... and from them the number of chunks k'_i each sequence really frees   metrics.py:704-729
// (f_s = finite-threshold chunks, cn_s = all chunks of every sequence, already in LDS)
__device__ __forceinline__ void seq_prepare_body(const kvc_schedule_params& p, SchedWs& ws, int64_t* un_s,
                                 int32_t* f_s, int32_t* cn_s, int32_t* off_s, int32_t* pinf_s) {
  const int B = p.num_seqs;
  __syncthreads();
  if (threadIdx.x == 0) {                            // exclusive prefixes: all chunks, inf-threshold chunks
    int64_t o = 0, q = 0;
    for (int i = 0; i < B; ++i) {
      off_s[i] = (int32_t)o; pinf_s[i] = (int32_t)q;
      o += cn_s[i]; q += cn_s[i] - f_s[i];
    }
  }
  __syncthreads();
  // #inf thresholds among the first x entries of the (seq, threshold)-ordered chunk list:
  // everything of the sequences in front of the one that holds entry x, plus its share
  // (the sum over all sequences of clamp(x - off_j - f_j, 0, I_j), by bisection instead of a
  // loop: the loop made this kernel 66 us at 256 sequences)
  auto inf_prefix = [&](int64_t x) {
    int lo = 0, hi = B - 1;                          // largest j with off_j <= x  (off_0 = 0 <= x)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if ((int64_t)off_s[mid] <= x) lo = mid; else hi = mid - 1;
    }
    int64_t v = x - off_s[lo] - f_s[lo];
    const int64_t Ij = cn_s[lo] - f_s[lo];
    v = v < 0 ? 0 : (v > Ij ? Ij : v);
    return (int64_t)pinf_s[lo] + v;
  };
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const int64_t x = (int64_t)off_s[i] + p.evicted_blocks_per_seq[i];
    int64_t ninf = inf_prefix(x);
    if (p.mode == 1) ninf -= inf_prefix(off_s[i]);
    un_s[i] = x - ninf;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    int64_t e = un_s[i];
    if (p.mode == 0)
      for (int j = i + 1; j < B; ++j) e = un_s[j] < e ? un_s[j] : e;   // later seqs un-evict
    int64_t k = e - off_s[i];
    k = k < 0 ? 0 : k;
    k = k > f_s[i] ? f_s[i] : k;    // thresholds beyond the finite ones are never freed
    ws.seq_k[i] = (int32_t)k;
    ws.seq_prefix[i] = 0;
    ws.seq_tmp[2 * B + i] = off_s[i];
  }
}

// everything lives in LDS: the loops are O(B^2) over three small tables, and walking them in
// global memory cost 117 us at 256 sequences.  Any number of sequences: tables of B entries in
// dynamic LDS (24 B per sequence: up to 6500).
__device__ __forceinline__ void seq_prepare_tables(const kvc_schedule_params& p, SchedWs& ws, uint8_t* lds) {
  const int B = p.num_seqs;
  int64_t* un_s = reinterpret_cast<int64_t*>(lds);
  int32_t* f_s = reinterpret_cast<int32_t*>(un_s + B);
  int32_t* cn_s = f_s + B;
  int32_t* off_s = cn_s + B;
  int32_t* pinf_s = off_s + B;
  for (int i = threadIdx.x; i < B; i += blockDim.x) { f_s[i] = ws.seq_tmp[i]; cn_s[i] = ws.seq_tmp[B + i]; }
  seq_prepare_body(p, ws, un_s, f_s, cn_s, off_s, pinf_s);
}
__global__ __launch_bounds__(1024) void seq_prepare_kernel(kvc_schedule_params p, SchedWs ws) {
  if (gated_off(ws)) return;
  extern __shared__ __attribute__((aligned(16))) uint8_t prep_lds[];
  seq_prepare_tables(p, ws, prep_lds);
}

// ------------------------------------------------------------------ 5. per-head counts
// chunks with threshold < T* are freed; chunks with threshold == T* are handed out in
// (head, chunk) order until the sequence total is k'.           metrics.py:773-792
// (one workgroup of NW waves per sequence; wave_tot: NW words, carry_s / lt_total_s: one each)
template <int NW>
__device__ __forceinline__ void finalize_body(const kvc_schedule_params& p, SchedWs& ws, int i, uint32_t* wave_tot,
                              uint32_t* carry_s, uint32_t* lt_total_s) {
  const int LH = p.num_layers * p.num_kv_heads;
  const uint32_t bs = (uint32_t)p.block_size;
  const uint32_t k = (uint32_t)ws.seq_k[i];
  const int tid = threadIdx.x, lane = lane_id(), w = tid / WAVE;
  // pass 1: total of sure chunks
  uint32_t part = 0;
  for (int lh = tid; lh < LH; lh += NW * WAVE) {
    const int g = i * LH + lh;
    if (k) part += nchunks_freed(ws.less[g], (uint32_t)p.hanging_token_count[g], bs);
  }
  part = wave_reduce_sum(part);
  __syncthreads();
  if (lane == 0) wave_tot[w] = part;
  if (tid == 0) *carry_s = 0;
  __syncthreads();
  if (tid == 0) {
    uint32_t t = 0;
    for (int q = 0; q < NW; ++q) t += wave_tot[q];
    *lt_total_s = t;
  }
  __syncthreads();
  const uint32_t need = k - (k ? *lt_total_s : 0u);     // tie chunks still to hand out
  for (int base = 0; base < LH; base += NW * WAVE) {
    const int lh = base + tid;
    const int g = i * LH + lh;
    uint32_t n_lt = 0, e = 0, hang = 1;
    if (lh < LH) {
      hang = (uint32_t)p.hanging_token_count[g];
      if (k) {
        n_lt = nchunks_freed(ws.less[g], hang, bs);
        e = nchunks_freed(ws.less[g] + ws.eq[g], hang, bs) - n_lt;
      }
    }
    const uint32_t inc = wave_inclusive_scan(e);
    __syncthreads();
    if (lane == 63) wave_tot[w] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int q = 0; q < w; ++q) woff += wave_tot[q];
    const uint32_t excl = *carry_s + woff + inc - e;
    if (lh < LH) {
      const uint32_t room = need > excl ? need - excl : 0u;
      const uint32_t n = n_lt + (e < room ? e : room);
      p.evicted_block_count[g] = (int32_t)n;
      p.evicted_kv_count[g] = n > 0 ? (int32_t)((n - 1) * bs + hang) : 0;
    }
    __syncthreads();
    if (tid == NW * WAVE - 1) *carry_s = excl + e;
    __syncthreads();
  }
}

// ------------------------------------------------------------------ 5a. scan + pick (+ totals, + counts) in one launch
// One workgroup per sequence does what used to be a launch each -- the per-head scan, (the
// sequence's totals, k',) the pick of the digit and, in the last round, the per-head counts: its 16 waves scan the digit histograms of
// the sequence's heads, the chunk counts per digit are summed in LDS (no [G,256] array), the digit
// is picked and the heads' `less` / `eq` updated.  Round 0 also needs k': per sequence it is
// min(k, finite-threshold chunks) -- what seq_prepare_body gives for mode 1 or a single sequence;
// the reference's batch > 1 rule (mode 0) couples the sequences (parts 1 and 2 below).
// (NW waves per workgroup: 16 in the kernel of its own, 4 inside the single-launch fallback)
// part: 0 = everything in one go; the reference's batch > 1 rule (round 0, mode 0) needs every
// sequence's totals before any k' exists, so its round 0 runs as part 1 (scan + the sequence's
// chunk totals -> seq_tmp), seq_prepare, part 2 (the per-digit chunk counts once more from the
// stored cumulative counts, pick, update)
template <int NW, int SU>
__device__ __forceinline__ void scan_pick_body(const kvc_schedule_params& p, SchedWs& ws, int round, int i, int part = 0) {
  __shared__ __attribute__((aligned(16))) uint32_t csum[NW][RADIX];
  __shared__ uint32_t wave_tot[NW];
  __shared__ uint32_t carry_s, lt_total_s;
  __shared__ int dstar_s;
  __shared__ uint32_t k_s;
  const int tid = threadIdx.x, lane = lane_id(), w = tid / WAVE;
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  const int LH = p.num_layers * p.num_kv_heads;
  const uint32_t bs = (uint32_t)p.block_size;
  const int bs_shift = (bs & (bs - 1u)) == 0u ? 31 - __builtin_clz(bs) : -1;
  const bool active = round == 0 || ws.seq_k[i] != 0;
  if (active) {
    reinterpret_cast<uint4*>(csum[w])[lane] = make_uint4(0u, 0u, 0u, 0u);
    // the scan: a wave takes every NW-th head, eight at a time (their loads, scans and stores
    // are independent: with a single sequence this workgroup is alone on the chip and a round
    // trip to the histograms -- last touched by atomics -- is what it waits for)
    for (int lh0 = w; lh0 < LH; lh0 += NW * SU) {
      uint4 v[SU];
      uint32_t less[SU], hang[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int lh = lh0 + NW * u;
        v[u] = make_uint4(0u, 0u, 0u, 0u); less[u] = 0; hang[u] = 1;
        if (lh < LH) {                                 // wave-uniform
          const int g = i * LH + lh;
          v[u] = part == 2 ? reinterpret_cast<const uint4*>(ws.cum + ((int64_t)round * G + g) * RADIX)[lane]
                           : reinterpret_cast<uint4*>(ws.hist + (int64_t)g * RADIX)[lane];   // 4 bins per lane
          less[u] = ws.less[g]; hang[u] = (uint32_t)p.hanging_token_count[g];
        }
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int lh = lh0 + NW * u;
        if (lh >= LH) break;                           // wave-uniform
        const int g = i * LH + lh;
        if (part != 2) {
          reinterpret_cast<uint4*>(ws.hist + (int64_t)g * RADIX)[lane] = make_uint4(0u, 0u, 0u, 0u);   // ready for the next round
          v[u].y += v[u].x; v[u].z += v[u].y; v[u].w += v[u].z;
          const uint32_t inc = wave_inclusive_scan(v[u].w);
          const uint32_t ex = inc - v[u].w;
          v[u].x += ex; v[u].y += ex; v[u].z += ex; v[u].w += ex;
          reinterpret_cast<uint4*>(ws.cum + ((int64_t)round * G + g) * RADIX)[lane] = v[u];
        }
        uint4 c = reinterpret_cast<uint4*>(csum[w])[lane];
        c.x += nchunks_freed_s(less[u] + v[u].x, hang[u], bs, bs_shift); c.y += nchunks_freed_s(less[u] + v[u].y, hang[u], bs, bs_shift);
        c.z += nchunks_freed_s(less[u] + v[u].z, hang[u], bs, bs_shift); c.w += nchunks_freed_s(less[u] + v[u].w, hang[u], bs, bs_shift);
        reinterpret_cast<uint4*>(csum[w])[lane] = c;
      }
    }
    __syncthreads();
    if (tid < RADIX) {                                 // chunks freed if the digit were d, over all heads
      uint32_t t = 0;
#pragma unroll
      for (int q = 0; q < NW; ++q) t += csum[q][tid];
      csum[0][tid] = t;
    }
    if (tid == 0) dstar_s = 255;
    __syncthreads();
    if (part == 1) {                                   // the sequence's totals, for seq_prepare
      uint32_t cn = 0;
      const int B = p.num_seqs, H = p.num_kv_heads;
      for (int lh = tid; lh < LH; lh += blockDim.x) {
        const int ctx = p.context_lens[((lh / H) * B + i) * H + (lh % H)];
        cn += (uint32_t)((ctx + (int)bs - 1) / (int)bs);
      }
      cn = wave_reduce_sum(cn);
      if (tid == 0) lt_total_s = 0;
      __syncthreads();
      if (lane == 0 && cn) atomicAdd(&lt_total_s, cn);
      __syncthreads();
      if (tid == 0) { ws.seq_tmp[i] = (int32_t)csum[0][255]; ws.seq_tmp[B + i] = (int32_t)lt_total_s; }
      return;
    }
    if (round == 0 && part == 0 && tid == 0) {         // seq_totals + seq_prepare, per sequence
      const int kk = p.evicted_blocks_per_seq[i];
      const uint32_t f = csum[0][255];                 // finite-threshold chunks
      const uint32_t k = kk <= 0 ? 0u : ((uint32_t)kk < f ? (uint32_t)kk : f);
      ws.seq_k[i] = (int32_t)k;
      ws.seq_prefix[i] = 0;
      k_s = k;
    }
    if ((round != 0 || part == 2) && tid == 0) k_s = (uint32_t)ws.seq_k[i];
    __syncthreads();
    const uint32_t k = k_s;
    if (k != 0) {                                      // the pick
      if (tid < RADIX) {
        const uint32_t sd = csum[0][tid];
        if (sd >= k && (tid == 0 || csum[0][tid - 1] < k)) dstar_s = tid;     // non-decreasing in d
      }
      __syncthreads();
      const int ds = dstar_s;
      if (tid == 0) ws.seq_prefix[i] = ((round == 0 ? 0u : ws.seq_prefix[i]) << 8) | (uint32_t)ds;
      for (int h2 = tid; h2 < LH; h2 += blockDim.x) {
        const int g = i * LH + h2;
        const uint32_t* cum = ws.cum + ((int64_t)round * G + g) * RADIX;
        const uint32_t below = ds > 0 ? cum[ds - 1] : 0u;
        ws.less[g] += below;
        if (round == 3) ws.eq[g] = cum[ds] - below;
      }
    }
  }
  if (round == 3) {
    __syncthreads();                                   // (the heads' less / eq just written by this workgroup)
    finalize_body<NW>(p, ws, i, wave_tot, &carry_s, &lt_total_s);
  }
}

__global__ __launch_bounds__(1024) void scan_pick_kernel(kvc_schedule_params p, SchedWs ws, int round, int part) {
  if (gated_off(ws)) return;
  scan_pick_body<16, 8>(p, ws, round, blockIdx.x, part);
}

// ------------------------------------------------------------------ 5b. uniform_evict: per-head counts
// The reference's other selection rule (metrics.py:639-666; its scheduler never passes it): every
// head of sequence i frees evicted_blocks_per_seq[i] / (L*H) chunks, its own lowest ones.  The
// reference asserts that those thresholds are finite (:660); here a head frees at most its
// finite-threshold chunks.  One wave per head over the round-0 histogram (the head's evictable keys).
__global__ __launch_bounds__(256) void uniform_counts_kernel(kvc_schedule_params p, SchedWs ws) {
  const int LH = p.num_layers * p.num_kv_heads, G = p.num_seqs * LH;
  const int g = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (g >= G) return;
  const int lane = lane_id();
  const uint4 v = reinterpret_cast<const uint4*>(ws.hist + (int64_t)g * RADIX)[lane];
  const uint32_t finite = wave_reduce_sum(v.x + v.y + v.z + v.w);
  if (lane == 0) {
    const int k = p.evicted_blocks_per_seq[g / LH];
    const uint32_t per_head = k > 0 ? (uint32_t)k / (uint32_t)LH : 0u;
    const uint32_t hang = (uint32_t)p.hanging_token_count[g], bs = (uint32_t)p.block_size;
    const uint32_t f = nchunks_freed(finite, hang, bs);
    const uint32_t n = per_head < f ? per_head : f;
    p.evicted_block_count[g] = (int32_t)n;
    p.evicted_kv_count[g] = n > 0 ? (int32_t)((n - 1) * bs + hang) : 0;
  }
}

// ------------------------------------------------------------------ 6. select + emit
// one workgroup per head: cnt-th smallest (key, physical slot) by radix select, then the
// ascending logical indices of everything at or below it.       metrics.py:822-834

// radix-select the rank-th (1-based) smallest value of f(idx) over idx in [0,n) where
// pred(idx); returns the value, and the 1-based rank among equals / number of equals.
// (first_round, prefix0): the top first_round digits are already known to be prefix0 and
// `rank` counts within that bucket; first_round == 4 returns prefix0 with out_eq untouched.
template <typename ValF, typename PredF>
__device__ __forceinline__ void block_radix_select(uint32_t* hist, uint32_t* bc, int n, uint32_t rank, ValF val,
                                   PredF pred, uint32_t& out_val, uint32_t& out_rank_in_eq,
                                   uint32_t& out_eq, int first_round = 0, uint32_t prefix0 = 0) {
  uint32_t prefix = prefix0;
  for (int round = first_round; round < 4; ++round) {
    const int shift = 24 - 8 * round;
    for (int k = threadIdx.x; k < RADIX; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    constexpr int U = 8;
    const int step = blockDim.x * U;
    for (int base = 0; base < n; base += step) {
      uint32_t vv[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {                 // independent loads first
        const int idx = base + u * blockDim.x + threadIdx.x;
        ok[u] = idx < n && pred(idx);
        vv[u] = ok[u] ? val(idx) : 0u;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool valid = ok[u] && (round == 0 || (vv[u] >> (shift + 8)) == prefix);
        hist_add(hist, valid, (vv[u] >> shift) & 0xFFu);
      }
    }
    __syncthreads();
    // 256-bin inclusive scan by the first 4 waves' worth of threads (one wave does it)
    if (threadIdx.x < WAVE) {
      uint4 q = reinterpret_cast<uint4*>(hist)[threadIdx.x];
      q.y += q.x; q.z += q.y; q.w += q.z;
      const uint32_t inc = wave_inclusive_scan(q.w);
      const uint32_t ex = inc - q.w;
      q.x += ex; q.y += ex; q.z += ex; q.w += ex;
      const uint32_t c[4] = {q.x, q.y, q.z, q.w};
      uint32_t prev = ex;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (prev < rank && rank <= c[t]) { bc[0] = threadIdx.x * 4 + t; bc[1] = prev; bc[2] = c[t] - prev; }
        prev = c[t];
      }
    }
    __syncthreads();
    prefix = (prefix << 8) | bc[0];
    rank -= bc[1];
    out_eq = bc[2];
    __syncthreads();
  }
  out_val = prefix;
  out_rank_in_eq = rank;
}

#ifdef KVC_BR_STAMPS
#define SE_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && bracket == 1) ws.head_fc[k] = (uint32_t)wall_clock64(); } while (0)
#else
#define SE_STAMP(k) do { } while (0)
#endif
// lds_cap = number of keys the dynamic LDS buffer can stage (0 = read keys from global/L2)
// bracket = 1 (section 9): M comes from the head's sorted bracket list instead of the digit rounds
template <int SEL_THREADS>
__device__ __forceinline__ void select_emit_head(const kvc_schedule_params& p, SchedWs& ws, int lds_cap, int g, uint32_t* lds_keys,
                                                 int bracket = 0) {
  __shared__ __attribute__((aligned(16))) uint32_t hist[RADIX];
  __shared__ uint32_t bc[4];
  __shared__ uint32_t scan_buf[8 * (SEL_THREADS / WAVE) + 1];
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  const int bs = p.block_size;
  const int64_t base = p.evicted_kv_offsets[g];
  const int64_t end = (g + 1 < G) ? (int64_t)p.evicted_kv_offsets[g + 1] : true_n(p, ws);
  const int n = (int)(end - base);
  const uint32_t cnt = (uint32_t)p.evicted_kv_count[g];
  const uint32_t* gkeys = ws.keys + base;
  int32_t* out = p.evicted_logical_indices + base;
  const int tid = threadIdx.x, lane = lane_id(), w = tid / WAVE;
  // (a call on a tracked output buffer that ended up here -- the small-eviction schedule fell back --
  // writes the whole segment like any other; the map only has to say what it holds afterwards)
  if (p.eli_dirty_map != nullptr && !(p.lean & 1))
    eli_dirty_update(p.eli_dirty_map, p.evicted_logical_indices, base / bs, end / bs, ((int64_t)cnt + bs - 1) / bs, 0, bs,
                     p.null_value, false, tid, SEL_THREADS);
  if (cnt == 0) {
    if (!(p.lean & 1))
      for (int idx = tid; idx < n; idx += blockDim.x) out[idx] = p.null_value;
    return;
  }
  SE_STAMP(4);
  // (bracket schedule: the head's count below the bracket, its list length and the list entry that is M are requested
  // here, in front of the staging pass and its barrier -- two dependent round trips that the staging hides)
  uint32_t br_below = 0, br_m = 0, br_M = 0;
  bool br_listed = false;
  if (bracket == 1) {
    br_below = ws.st_def[g];
    br_m = min(ws.st_cnt[g], bracket_cap((uint32_t)n));
    br_listed = cnt > br_below && cnt - 1u - br_below < br_m;
    if (br_listed) br_M = (ws.blist + bracket_list_at(base, g))[cnt - 1u - br_below];
  }
  // stage the head's keys in LDS once; every later pass (4 select rounds + emit) reads LDS
  const bool staged = n <= lds_cap;
  if (staged) {
    for (int idx = tid * 4; idx < n; idx += SEL_THREADS * 4) {
      if (idx + 3 < n && ((base & 3) == 0)) {
        *reinterpret_cast<uint4*>(lds_keys + idx) = *reinterpret_cast<const uint4*>(gkeys + idx);
      } else {
        for (int q = idx; q < min(n, idx + 4); ++q) lds_keys[q] = gkeys[q];
      }
    }
    __syncthreads();
  }
  SE_STAMP(5);
  auto key_at = [&](int idx) { return staged ? lds_keys[idx] : gkeys[idx]; };
  // Warm start from the sequence-level rounds: cum[r][g][d] counts this head's keys that share
  // T*'s top r digits and have digit r <= d.  The cnt-th smallest key M is at most T* and at
  // most a block's worth of keys below it, so it normally shares two or three digits with
  // T*: find the first round r* whose below-T* count L_r reaches cnt, read M's digit r* off
  // the stored histogram, and only run the remaining rounds r*+1..3 over the keys.
  uint32_t M, take, eqn = 0;
  bool from_list = false;
  if (bracket == 2) {
    // uniform_evict: no sequence-level T* exists -- the head's own cnt-th smallest key, by a full select
    block_radix_select(hist, bc, n, cnt, key_at, [&](int) { return true; }, M, take, eqn);
    from_list = true;
  } else if (bracket) {
    // the cnt-th smallest key of the head lies in its bracket list (keys in [lo, hi], sorted; `below`
    // keys of the head are smaller than lo) unless the head frees only chunks below the bracket
    const uint32_t below = br_below;
    const uint32_t m = br_m;
    const uint32_t* list = ws.blist + bracket_list_at(base, g);
    if (br_listed) {
      from_list = true;
      M = br_M;
      uint32_t lt = 0, eq = 0;                       // entries below M / equal to M: one parallel pass over the list
      for (uint32_t j = tid; j < m; j += SEL_THREADS) { const uint32_t v = list[j]; lt += v < M; eq += v == M; }
      lt = wave_reduce_sum(lt); eq = wave_reduce_sum(eq);
      if (tid == 0) { bc[0] = 0; bc[1] = 0; }
      __syncthreads();
      if (lane == 0) { atomicAdd(&bc[0], lt); atomicAdd(&bc[1], eq); }
      __syncthreads();
      eqn = bc[1];
      take = cnt - below - bc[0];
      __syncthreads();
    } else {
      block_radix_select(hist, bc, n, cnt, key_at, [&](int) { return true; }, M, take, eqn);
      from_list = true;
    }
  }
  if (!from_list) {
    const int i_seq = g / (p.num_layers * p.num_kv_heads);
    const uint32_t Tstar = ws.seq_prefix[i_seq];
    if (tid < 4) {                                   // the four lookups in parallel (latency)
      const uint32_t ds = (Tstar >> (24 - 8 * tid)) & 0xFFu;
      bc[tid] = ds ? ws.cum[((int64_t)tid * G + g) * RADIX + ds - 1] : 0u;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t L = 0;
      int rstar = 4;
      uint32_t base_rank = 0;
      for (int r = 0; r < 4; ++r) {
        const uint32_t below = bc[r];
        if (cnt <= L + below) { rstar = r; base_rank = L; break; }
        L += below;
      }
      if (rstar == 4) base_rank = L;                 // M == T*: rank among the equal keys
      bc[0] = (uint32_t)rstar;
      bc[1] = base_rank;
    }
    __syncthreads();
    const int rstar = (int)bc[0];
    const uint32_t base_rank = bc[1];
    __syncthreads();
    if (rstar == 4) {
      if (cnt - base_rank <= ws.eq[g]) {
        M = Tstar; take = cnt - base_rank; eqn = ws.eq[g];
      } else {                                       // not expected (finalize caps cnt): full select
        block_radix_select(hist, bc, n, cnt, key_at, [&](int) { return true; }, M, take, eqn);
      }
    } else {
      // digit r* of M: first d with cum[r*][d] >= cnt - base_rank (d < T*'s digit by construction)
      const uint32_t* cr = ws.cum + ((int64_t)rstar * G + g) * RADIX;
      const uint32_t tgt = cnt - base_rank;
      for (int d = tid; d < RADIX; d += blockDim.x) {
        const uint32_t c = cr[d], c0 = d ? cr[d - 1] : 0u;
        if (c0 < tgt && tgt <= c) { bc[0] = (uint32_t)d; bc[1] = c0; bc[2] = c - c0; }
      }
      __syncthreads();
      const uint32_t dig = bc[0], c0 = bc[1], cw = bc[2];
      __syncthreads();
      const uint32_t hi = rstar ? (Tstar >> (32 - 8 * rstar)) : 0u;       // shared top digits
      const uint32_t prefix = (hi << 8) | dig;
      eqn = cw;                                      // only final when r* == 3
      block_radix_select(hist, bc, n, tgt - c0, key_at, [&](int) { return true; }, M, take, eqn,
                         rstar + 1, prefix);
    }
  }
  SE_STAMP(6);
  // ties on the metric: the `take` entries with the smallest (physical block, offset)
  uint32_t Fstar = 0xFFFFFFFFu;
  const int32_t* cphys = ws.chunk_phys + base / bs;
  auto fkey = [&](int idx) { return (uint32_t)cphys[idx / bs] * (uint32_t)bs + (uint32_t)(idx % bs); };
  if (take < eqn) {
    uint32_t r2, e2;
    block_radix_select(hist, bc, n, take, fkey, [&](int idx) { return key_at(idx) == M; }, Fstar, r2, e2);
  }
  // emit: flags for U rows of SEL_THREADS consecutive indices at a time, one block-wide
  // exclusive scan of the U x (waves) ballot counts (two barriers per U*SEL_THREADS keys),
  // compact; then pad with null
  constexpr int U = 8;
  constexpr int NWAVES = SEL_THREADS / WAVE;
  uint32_t carry = 0;
  const bool tie_cut = Fstar != 0xFFFFFFFFu;
  for (int base0 = 0; base0 < n; base0 += SEL_THREADS * U) {
    uint32_t kk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {                   // independent loads first
      const int idx = base0 + u * SEL_THREADS + tid;
      kk[u] = idx < n ? key_at(idx) : 0xFFFFFFFFu;
    }
    uint32_t lane_ex[U];
    bool sel[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base0 + u * SEL_THREADS + tid;
      sel[u] = idx < n && (kk[u] < M || (kk[u] == M && (!tie_cut || fkey(idx) <= Fstar)));
      const unsigned long long bal = __ballot(sel[u]);
      lane_ex[u] = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) scan_buf[u * NWAVES + w] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    if (w == 0) {                                   // exclusive scan of U*NWAVES counts (row-major)
      constexpr int PER = (U * NWAVES + WAVE - 1) / WAVE;
      uint32_t v[PER], run = 0;
#pragma unroll
      for (int q = 0; q < PER; ++q) { const int e = lane * PER + q; v[q] = e < U * NWAVES ? scan_buf[e] : 0u; run += v[q]; }
      const uint32_t inc = wave_inclusive_scan(run);
      uint32_t ex = inc - run;
#pragma unroll
      for (int q = 0; q < PER; ++q) { const int e = lane * PER + q; if (e < U * NWAVES) scan_buf[e] = ex; ex += v[q]; }
      if (lane == WAVE - 1) scan_buf[U * NWAVES] = inc;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base0 + u * SEL_THREADS + tid;
      if (sel[u]) out[carry + scan_buf[u * NWAVES + w] + lane_ex[u]] = idx;   // logical index == position in head
    }
    carry += scan_buf[U * NWAVES];
    __syncthreads();                                // scan_buf is rewritten by the next batch
  }
  SE_STAMP(7);
  if (!(p.lean & 1))
    for (int idx = (int)cnt + tid; idx < n; idx += blockDim.x) out[idx] = p.null_value;
  SE_STAMP(20);
}

// one workgroup per head; behind the small-eviction schedule (gated: a launch that normally finds
// the flag down) the grid is capped and a workgroup walks several heads -- 65 536 workgroups that
// only read the flag took 15 us, a capped grid takes what every gated launch takes
template <int SEL_THREADS>
__global__ __launch_bounds__(SEL_THREADS) void select_emit_kernel(kvc_schedule_params p, SchedWs ws, int lds_cap, int bracket) {
  if (gated_off(ws)) return;
  if (bracket == 1 && *ws.fallback != 0u) return;    // the bracket missed: the gated pipeline behind writes everything
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_keys[];
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  for (int g = blockIdx.x; g < G; g += gridDim.x) {
    select_emit_head<SEL_THREADS>(p, ws, lds_cap, g, lds_keys, bracket);
    __syncthreads();
  }
}


}  // namespace kvc
