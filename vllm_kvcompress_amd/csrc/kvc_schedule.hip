// A3 CompressionMetrics.schedule_evictions for gfx950 -- without a single sort.
//
// The reference (vllm/kvcompress/metrics.py:441-847) masks the candidate metrics, sorts
// them globally (twice), gathers one threshold per block-sized chunk, sorts those
// (twice), walks the sequences on the host, counts leading evicted chunks per head and
// sorts the surviving logical indices (twice) -- six device sorts over N slots and ~8N
// of temporaries.  What it computes is an order-statistics problem:
//
//   * per head g the chunk thresholds are every bs-th order statistic of the head's
//     masked metrics, thr[g,c] = (c*bs + hang_g)-th smallest;
//   * per sequence the k smallest thresholds are selected; because thr[g,.] increases
//     with c, head g frees n_g = #{c : thr[g,c] <= T*} chunks where T* is the k-th
//     smallest threshold of the sequence, i.e. n_g = floor((R_g(T*) - hang_g)/bs) + 1
//     with R_g(T) = #{finite keys of head g that are <= T};
//   * the evicted slots of head g are its cnt_g = (n_g-1)*bs + hang_g smallest keys,
//     listed by ascending logical index.
//
// So: one pass builds order-preserving 32-bit keys in head-contiguous *logical* order;
// T* per sequence is found by an MSB-first radix select (4 rounds of per-head 256-bin
// histograms, a per-head scan that turns cumulative counts into chunk counts, and a
// per-sequence pick of the digit); a per-head radix select finds the cnt_g-th smallest
// key, and a flag + prefix-sum pass emits the logical indices already in ascending order.
// Everything is a streaming pass over N x 4 B; ties are resolved exactly in the canonical
// order of DESIGN.md ((metric, physical block, offset) within a head, (threshold, head,
// chunk) within a sequence).  The reference's batch>1 quirk (metrics.py:718-721) only
// changes how many chunks each sequence may free, which is computed on device in
// seq_prepare_kernel for mode 0.
#include "kvc_common.h"
#include "../../include/kvc_mi355x.h"
#include <atomic>

namespace kvc {

constexpr int RADIX = 256;

struct SeqRec;
struct SchedWs {
  uint32_t* keys;        // [N]      order-preserving metric keys, index off_g + lambda
  int32_t* chunk_phys;   // [N/bs]   physical block of logical chunk
  uint32_t* hist;        // [G,256]  per-head digit histogram (re-zeroed by the scan)
  uint32_t* cum;         // [4,G,256] inclusive cumulative counts of every round (select_emit reuses them)
  uint32_t* less;        // [G]      keys strictly below the current prefix
  uint32_t* eq;          // [G]      keys equal to T* (after the last round)
  uint32_t* seq_prefix;  // [B]
  int32_t* seq_k;        // [B]      chunks this sequence frees (k'), 0 = inactive
  int32_t* seq_tmp;      // [3B]     F (finite chunks), Cn (all chunks), offset
  // small-eviction schedule (section 7 below)
  uint64_t* rec64;       // [G,KREC] per head: (key << 32 | physical slot) of every evictable key below the
                         //          sequence's pivot; sorted ascending (canonical tie order) = the head's record
  uint32_t* st_cnt;      // [G]      entries of that list (counts on beyond KREC: overflow)
  uint32_t* st_def;      // [G]      masked / non-finite slots of the head's blocks
  uint32_t* st_samp;     // [G]      sampled blocks of the head
  uint32_t* st_claimed;  // [64 x 32] physical blocks that are logical blocks of the batch, sharded over 64 cache lines
  struct SeqRec* st_seqrec;  // [B]  per sequence: position, protected window, pivot
  uint32_t* head_fc;     // [2G]     per head: finite-threshold chunks, all chunks (stream_records)
  uint32_t* bsample;     // [B, BR_CELLS] bracket schedule: the sample build_keys leaves behind (nullptr: none wanted)
  uint32_t* bnonfin;     // [G] bracket schedule under the batch > 1 rule: build_keys counts the head's keys that are
                         //     not evictable here (= st_samp; nullptr: not wanted)
  const int32_t* bk;     // [B] bracket schedule: chunks a sequence frees -- the caller's k (k' = min(k, finite) is found
                         //     on the way) or, under the batch > 1 rule, seq_k = k' itself
  uint32_t* bthr;        // [N / bs] bracket schedule: per head (from its first chunk on) its listed thresholds, ascending
  uint32_t* blist;       // [N / 8 + 32 G]  bracket schedule: per head the keys inside the sequence's bracket (then sorted)
  uint32_t* fallback;    // [1]      != 0: the small-eviction schedule could not finish exactly
  uint32_t* bar;         // [32+64]  single-launch fallback: phase stamps, then claim / done counters of its phases
  const uint32_t* gate;  // general-path kernels run only if gate == nullptr or *gate != 0
};

constexpr int KREC = 256;   // record length of the small-eviction schedule (keys per head)

__device__ __forceinline__ bool gated_off(const SchedWs& ws) { return ws.gate != nullptr && *ws.gate == 0u; }

// bracket schedule (section 9): the bracket list of head g (its slots start at off_g) lives at
// blist + off_g / BR_DIV + g * BR_PAD and holds an eighth of the head's slots plus BR_PAD entries,
// BR_SORT_MAX at most (what one workgroup sorts in LDS); the lists of neighbours do not overlap
constexpr int BR_DIV = 8;
constexpr int BR_PAD = 32;
constexpr uint32_t BR_SORT_MAX = 4096;
__device__ __forceinline__ uint32_t bracket_cap(uint32_t head_slots) {
  const uint32_t c = head_slots / BR_DIV + BR_PAD;
  return c < BR_SORT_MAX ? c : BR_SORT_MAX;
}
__device__ __forceinline__ int64_t bracket_list_at(int64_t head_base, int g) {
  return head_base / BR_DIV + (int64_t)g * BR_PAD;
}
// the bracket's sample: a sequence's slots in at most BR_CELLS cells of 2^k >= 4 slots (the four
// slots a build_keys thread writes lie in one cell), one sampled slot per cell at a hashed place
// inside it (no pattern of the layout aliases with the sample); build_keys leaves its key in
// bsample[seq * BR_CELLS + cell]
constexpr uint32_t BR_CELLS = 32768;
// log2 of the cell size: the power of two (>= 4) that covers the sequence with at most BR_CELLS cells
__device__ __forceinline__ int bracket_stride_log2(uint32_t seq_slots) {
  const uint32_t per = (seq_slots + BR_CELLS - 1u) / BR_CELLS;
  const int lg = per <= 1u ? 0 : 32 - __builtin_clz(per - 1u);
  return lg < 2 ? 2 : lg;
}
__device__ __forceinline__ uint32_t bracket_cell_slot(uint32_t cell, uint32_t seq, int stride_log2) {
  uint32_t x = (cell * 0x9E3779B1u) ^ ((seq + 0x7F4A7C15u) * 0x85EBCA77u);
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
  return (cell << stride_log2) + (x & ((1u << stride_log2) - 1u));
}

__device__ __forceinline__ uint32_t nchunks_freed(uint32_t r, uint32_t hang, uint32_t bs) {
  return r >= hang ? (r - hang) / bs + 1u : 0u;
}
// the same with the division as a shift when bs is a power of two (bs_shift >= 0): the scans do
// four of these per lane and head, and a 32-bit division is ~40 instructions
__device__ __forceinline__ uint32_t nchunks_freed_s(uint32_t r, uint32_t hang, uint32_t bs, int bs_shift) {
  if (r < hang) return 0u;
  return (bs_shift >= 0 ? (r - hang) >> bs_shift : (r - hang) / bs) + 1u;
}

// evicted_logical_indices as a buffer the caller keeps between calls (kvc_schedule_params.
// eli_dirty_map): null everywhere except the leading entries of the head segments the last call wrote.
// One bit per chunk of bs entries (head segments start at multiples of bs) says where; the owner of
// the chunks [c0, c1) -- a head -- marks its first new_chunks chunks, clears the rest and, with
// fill_old, writes null over what older calls left behind from entry keep_from on.  Words that
// straddle the owner's ends are shared with the neighbouring heads, who do the same to their bits
// at the same time: atomics there, plain accesses inside.
__device__ __forceinline__ void eli_dirty_update(uint32_t* map, int32_t* eli, int64_t c0, int64_t c1, int64_t new_chunks,
                                                 int64_t keep_from, int bs, int32_t null_value, bool fill_old,
                                                 int tid, int nthreads) {
  if (c0 >= c1) return;
  const int64_t w0 = c0 >> 5, w1 = (c1 - 1) >> 5;
  const int64_t cn = c0 + new_chunks;
  for (int64_t w = w0 + tid; w <= w1; w += nthreads) {
    const int64_t lo = max(c0, w << 5), hi = min(c1, (w + 1) << 5);
    const uint32_t mask = (hi - lo >= 32) ? 0xFFFFFFFFu : (((1u << (hi - lo)) - 1u) << (lo & 31));
    const int64_t nh = min(hi, cn);
    const uint32_t fresh = nh > lo ? ((nh - lo >= 32) ? 0xFFFFFFFFu : (((1u << (nh - lo)) - 1u) << (lo & 31))) : 0u;
    uint32_t old;
    if (mask == 0xFFFFFFFFu) {
      old = map[w];
      if (old != fresh) map[w] = fresh;
    } else {
      old = atomicAnd(&map[w], ~mask) & mask;
      if (fresh) atomicOr(&map[w], fresh);
    }
    if (!fill_old) continue;
    while (old) {
      const int bit = __ffs((int)old) - 1;
      old &= old - 1u;
      const int64_t eb = max(((w << 5) + bit) * (int64_t)bs, keep_from), ee = (((w << 5) + bit) + 1) * (int64_t)bs;
      for (int64_t e = eb; e < ee; ++e) eli[e] = null_value;
    }
  }
}

// wave-aggregated shared-memory histogram add: metric keys are often degenerate in their
// top digits (all lanes hit one bin), which would serialise 64 LDS atomics; up to two
// leader-elected groups are folded into one atomic each, the rest go one by one.
// (also used on global memory with digit = head * 256 + digit)
__device__ __forceinline__ void hist_add(uint32_t* hist, bool valid, uint32_t digit) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const unsigned long long act = __ballot(valid);
    if (!act) return;
    const int leader = __ffsll((long long)act) - 1;
    const uint32_t d0 = __shfl(digit, leader, 64);
    const bool same = valid && digit == d0;
    const unsigned long long grp = __ballot(same);
    if (lane_id() == leader) atomicAdd(&hist[d0], (uint32_t)__popcll(grp));
    valid = valid && !same;
  }
  if (valid) atomicAdd(&hist[digit], 1u);
}

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
  const int64_t se = i + 1 < p.num_seqs ? (int64_t)p.evicted_kv_offsets[(i + 1) * LH] : p.total_slots;
  const int lg = bracket_stride_log2((uint32_t)(se - sb));
  const uint32_t at0 = (uint32_t)(dst - sb);
  const uint32_t cell = at0 >> lg;
  const uint32_t t = bracket_cell_slot(cell, (uint32_t)i, lg) - at0;
  if (t < 4u) ws.bsample[(int64_t)i * BR_CELLS + cell] = t == 0u ? k.x : (t == 1u ? k.y : (t == 2u ? k.z : k.w));
}
__device__ __forceinline__ void sample_key(const kvc_schedule_params& p, SchedWs& ws, int i, int64_t dst, uint32_t k) {
  const int LH = p.num_layers * p.num_kv_heads;
  const int64_t sb = p.evicted_kv_offsets[i * LH];
  const int64_t se = i + 1 < p.num_seqs ? (int64_t)p.evicted_kv_offsets[(i + 1) * LH] : p.total_slots;
  const int lg = bracket_stride_log2((uint32_t)(se - sb));
  const uint32_t at = (uint32_t)(dst - sb);
  const uint32_t cell = at >> lg;
  if (bracket_cell_slot(cell, (uint32_t)i, lg) == at) ws.bsample[(int64_t)i * BR_CELLS + cell] = k;
}

// (bodies take the workgroup's index and the number of workgroups as arguments: the kernels below
// pass blockIdx / gridDim, the single-launch fallback of the small-eviction schedule its own)
template <int VEC>
__device__ __forceinline__ void build_keys_body(const kvc_schedule_params& p, SchedWs& ws, unsigned bid, unsigned data_blocks) {
  const int bs = p.block_size;
  const int per_blk = bs / VEC;
  // (grid-stride: behind the small-eviction schedule this kernel is launched gated, with a small grid)
  for (int64_t tid = (int64_t)bid * blockDim.x + threadIdx.x; tid < p.num_blocks * per_blk;
       tid += (int64_t)data_blocks * blockDim.x) {
  const int64_t blk = tid / per_blk;
  const int off = (int)(tid % per_blk) * VEC;
  // free blocks (an engine's cache is sized to HBM: most blocks do not belong to the batch) cost
  // their 4 B of sequence index and nothing else; for the others the wide loads do not depend on
  // the rest of the metadata chain below and are issued first
  const int s = p.seq_index_by_block[blk];
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
  const int l = p.layer_index_by_block[blk], h = p.head_index_by_block[blk];
  const int lbn = p.logical_block_num_by_block[blk];
  const int g = (i * L + l) * H + h;
  const int ctx = p.context_lens[(l * B + i) * H + h];
  const int nblk = (ctx + bs - 1) / bs;
  if (lbn < 0 || lbn >= nblk) continue;        // not part of the head's slot range
  const int seq_pos = p.seq_positions[i], prot = p.num_protected[i];
  const int64_t base = p.evicted_kv_offsets[g];
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
  if (off == 0) ws.chunk_phys[base / bs + lbn] = (int32_t)blk;
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void build_keys_kernel(kvc_schedule_params p, SchedWs ws,
                                                         unsigned data_blocks, uint4* zero16, int64_t zero_vecs) {
  if (gated_off(ws)) return;
  if (blockIdx.x >= data_blocks) {      // tail workgroups clear the counters of the later passes
    zero_body(zero16, zero_vecs, blockIdx.x - data_blocks, gridDim.x - data_blocks);
    return;
  }
  build_keys_body<VEC>(p, ws, blockIdx.x, data_blocks);
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
__device__ __forceinline__ void build_keys_sparse_body(const kvc_schedule_params& p, SchedWs& ws, unsigned bid, unsigned data_blocks) {
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
      if (off == 0) ws.chunk_phys[base_g / bs + lbn] = (int32_t)blk;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void build_keys_sparse_kernel(kvc_schedule_params p, SchedWs ws,
                                                                unsigned data_blocks, uint4* zero16, int64_t zero_vecs) {
  if (gated_off(ws)) return;
  if (blockIdx.x >= data_blocks) {      // tail workgroups clear the counters of the later passes
    zero_body(zero16, zero_vecs, blockIdx.x - data_blocks, gridDim.x - data_blocks);
    return;
  }
  build_keys_sparse_body(p, ws, blockIdx.x, data_blocks);
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
  const int64_t N = p.total_slots;
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
  const int64_t N = p.total_slots;
  const int shift = 24 - 8 * round;
  const int64_t ntiles = (N + HTILE - 1) / HTILE;
  const int64_t tb = ntiles * bid / nb, te = ntiles * (bid + 1) / nb;
  if (tb >= te) return;
  int g = upper_bound_minus1(p.evicted_kv_offsets, G, tb * HTILE);
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
  const int64_t end = (g + 1 < G) ? (int64_t)p.evicted_kv_offsets[g + 1] : p.total_slots;
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
  auto key_at = [&](int idx) { return staged ? lds_keys[idx] : gkeys[idx]; };
  // Warm start from the sequence-level rounds: cum[r][g][d] counts this head's keys that share
  // T*'s top r digits and have digit r <= d.  The cnt-th smallest key M is at most T* and at
  // most a block's worth of keys below it, so it normally shares two or three digits with
  // T*: find the first round r* whose below-T* count L_r reaches cnt, read M's digit r* off
  // the stored histogram, and only run the remaining rounds r*+1..3 over the keys.
  uint32_t M, take, eqn = 0;
  bool from_list = false;
  if (bracket) {
    // the cnt-th smallest key of the head lies in its bracket list (keys in [lo, hi], sorted; `below`
    // keys of the head are smaller than lo) unless the head frees only chunks below the bracket
    const uint32_t below = ws.st_def[g];
    const uint32_t m = min(ws.st_cnt[g], bracket_cap((uint32_t)n));
    const uint32_t* list = ws.blist + bracket_list_at(base, g);
    if (cnt > below && cnt - 1u - below < m) {
      from_list = true;
      M = list[cnt - 1u - below];
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
  if (!(p.lean & 1))
    for (int idx = (int)cnt + tid; idx < n; idx += blockDim.x) out[idx] = p.null_value;
}

// one workgroup per head; behind the small-eviction schedule (gated: a launch that normally finds
// the flag down) the grid is capped and a workgroup walks several heads -- 65 536 workgroups that
// only read the flag took 15 us, a capped grid takes what every gated launch takes
template <int SEL_THREADS>
__global__ __launch_bounds__(SEL_THREADS) void select_emit_kernel(kvc_schedule_params p, SchedWs ws, int lds_cap, int bracket) {
  if (gated_off(ws)) return;
  if (bracket && *ws.fallback != 0u) return;         // the bracket missed: the gated pipeline behind writes everything
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_keys[];
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  for (int g = blockIdx.x; g < G; g += gridDim.x) {
    select_emit_head<SEL_THREADS>(p, ws, lds_cap, g, lds_keys, bracket);
    __syncthreads();
  }
}

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
constexpr int CLAIM_SHARDS = 64;     // counters of claimed blocks, 128 B apart

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
#define KVC_PIV_SIGMAS 12.0                          // (experiment builds: tools/)
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
  }
  // lane q < HPW looks after head g0 + q: finite keys -> finite-threshold chunks of the head
  uint32_t myC = 0;
  if (lane < HPW && g0 + lane < G) {
    const int g = g0 + lane;
    const int i_seq = g / (L * H), l = (g / H) % L, h = g % H;
    const int ctx = p.context_lens[(l * B + i_seq) * H + h];
    const uint32_t nblk = (uint32_t)((ctx + bs - 1) / bs);
    myC = ws.st_cnt[g];
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
    const int64_t e = (g + 1 < G) ? (int64_t)p.evicted_kv_offsets[g + 1] : p.total_slots;
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
  __shared__ __attribute__((aligned(16))) uint32_t priv[2 * PRIV_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t hist[2 * RADIX];
  __shared__ uint32_t bc[6];
  __shared__ uint32_t red_s[3];
  const int i = blockIdx.x, tid = threadIdx.x, lane = lane_id();
  const int B = p.num_seqs, H = p.num_kv_heads, LH = p.num_layers * H, bs = p.block_size;
  const int64_t base = p.evicted_kv_offsets[i * LH];
  const int64_t end = i + 1 < B ? (int64_t)p.evicted_kv_offsets[(i + 1) * LH] : p.total_slots;
  const uint32_t n = (uint32_t)(end - base);
  BR_STAMP(0);
  if (tid < 3) red_s[tid] = 0;
  if (i == 0 && tid < 128) ws.fallback[tid] = 0u;    // the flag, the stamps and the phase counters of the fallback
  __syncthreads();
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
  __shared__ uint32_t qk[4][CC_QUEUE], qg[4][CC_QUEUE];
  __shared__ uint32_t wg_below[8];                   // the workgroup's first eight heads: one global add each
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  const int LH = p.num_layers * p.num_kv_heads;
  const int64_t N = p.total_slots;
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
  int g = upper_bound_minus1(p.evicted_kv_offsets, G, tb * HTILE);
  int64_t g_beg = p.evicted_kv_offsets[g];
  int64_t g_end = (g + 1 < G) ? (int64_t)p.evicted_kv_offsets[g + 1] : N;
  BrRec rc = recs[g / LH];
  const int g0 = g;                                  // (the same in every wave)
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
  load_tile(kn, tb);
  for (int64_t t = tb; t < te; ++t) {
    const int64_t t0 = t * HTILE;
#pragma unroll
    for (int u = 0; u < U; ++u) kv[u] = kn[u];
    if (t + 1 < te) load_tile(kn, t + 1);            // the next tile's keys are on their way meanwhile
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
  flush();
  if (qn > 0) drain(qn);
  complete();
  __syncthreads();
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
  __shared__ uint32_t a[BR_SORT_MAX], tmp[BR_SORT_MAX], cnt[BR_SORT_MAX + 1];
  __shared__ uint32_t wtot[8];
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  const int g = blockIdx.x;
  BR_STAMP(16);
  const int64_t base = p.evicted_kv_offsets[g];
  const int64_t end = (g + 1 < G) ? (int64_t)p.evicted_kv_offsets[g + 1] : p.total_slots;
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
    const int64_t e = (g + 1 < G) ? (int64_t)p.evicted_kv_offsets[g + 1] : p.total_slots;
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

// ------------------------------------------------------------------ 8. the fallback in ONE launch
// HIP has no conditional enqueue: behind the small-eviction schedule the general pipeline used to
// be 13 launches that read the flag and return, ~4.6 us each -- 60 us of a 150 us schedule at 16
// resident sequences.  This kernel is the whole general pipeline (for sequences that do not couple:
// mode 1 or a single one; the batch > 1 rule up to FB_MAX_COUPLED sequences) in ONE launch; with
// the flag down it is one launch that returns.
//
// Its phases depend on each other across workgroups, and nothing guarantees that a grid is resident
// at once (another stream, another process or a CU mask may hold compute units whatever the
// occupancy query says): a barrier that waits for every WORKGROUP to arrive can wait for one that
// has not started.  So the phases do not wait for workgroups, they wait for WORK: a phase is cut
// into V virtual workgroups (the bodies take their index and count as arguments), the real
// workgroups claim them from a counter until none is left and then wait until V of them are done.
// Whatever is resident does all of the work; a workgroup that starts late finds the counters of
// the finished phases full and falls through them.  Every claimed piece is being executed by a
// workgroup that runs, so every wait ends: correctness does not depend on co-residency, only speed
// does (the host still sizes the grid to what the occupancy query says is resident at once).
// Publishing a piece is the release / acquire recipe of cdna_hip_programming.md Guideline 16: every
// wave's stores are complete at the workgroup barrier, lane 0 writes the XCD's L2 back (release,
// agent scope) and adds to the phase's done counter; a waiter polls it with relaxed loads and a
// sleep, invalidates the CU's L1 (acquire), and the workgroup barrier hands that to the other waves.
// A wait that does not end within ten seconds of the 100 MHz wall clock (a device that lost a
// workgroup: must not happen) raises bit 1 of the flag word, which is sticky: every workgroup that
// sees it stops and overwrites the outputs with the schedule that evicts NOTHING (zero counts, a
// null list) -- never a partial one -- and the host raises when it reads the bit (metrics.py).
constexpr int FB_PHASES = 32;                        // claim / done counters (14 phases at most)
constexpr uint32_t FB_TIMEOUT_BIT = 2u;
struct FbSync {
  uint32_t* claim;       // [FB_PHASES] virtual workgroups handed out
  uint32_t* done;        // [FB_PHASES] ... finished
  uint32_t* flag;        // the schedule's flag word (bit 1: a wait timed out, results void)
};

// runs body(v, V) for the virtual workgroups v this workgroup can claim, then waits for all V;
// false = the wait was given up (or somebody else gave up): stop
template <typename F>
__device__ __forceinline__ bool fb_phase(const FbSync& fs, uint32_t phase, uint32_t V, uint32_t* word_s, F&& body) {
  uint32_t* claim = fs.claim + phase;
  uint32_t* done = fs.done + phase;
  for (;;) {
    __syncthreads();                                 // (word_s and the body's LDS are free again)
    if (threadIdx.x == 0) *word_s = __hip_atomic_fetch_add(claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t v = *word_s;
    if (v >= V) break;
    body(v, V);
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t ok = 1u;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < V) {
      __builtin_amdgcn_s_sleep(16);
      if (__hip_atomic_load(fs.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & FB_TIMEOUT_BIT) { ok = 0u; break; }
      if (wall_clock64() - t0 > 1000000000ull) { atomicOr(fs.flag, FB_TIMEOUT_BIT); ok = 0u; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *word_s = ok;
  }
  __syncthreads();
  return *word_s != 0u;
}

// the schedule that evicts nothing (what a call leaves behind when a wait was given up)
__device__ __forceinline__ void fb_void_outputs(const kvc_schedule_params& p, unsigned bid, unsigned nb) {
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  for (int64_t g = (int64_t)bid * 256 + threadIdx.x; g < G; g += (int64_t)nb * 256) {
    p.evicted_kv_count[g] = 0;
    p.evicted_block_count[g] = 0;
  }
  for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < p.total_slots; i += (int64_t)nb * 256)
    p.evicted_logical_indices[i] = p.null_value;
}

constexpr int FB_MAX_COUPLED = 256;                  // sequences whose batch > 1 rule fits the static tables below
// have_keys: the key pass ran already (the bracket schedule's) -- straight to the digit rounds
// vgrid: virtual workgroups of the streaming phases (the grid the host would like to be resident)
__global__ __launch_bounds__(256, 2) void fallback_general_kernel(kvc_schedule_params p, SchedWs ws, int sparse,
                                                               uint4* zero16, int64_t zero_vecs, int have_keys,
                                                               unsigned vgrid) {
  const uint32_t flag0 = __hip_atomic_load(ws.fallback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (flag0 == 0u) return;                           // flag down: this launch is all the fallback costs
  __shared__ __attribute__((aligned(16))) uint8_t prep_s[FB_MAX_COUPLED * 24];
  __shared__ uint32_t word_s;
  const bool coupled = p.mode == 0 && p.num_seqs > 1;
  const unsigned bid = blockIdx.x, nb = gridDim.x;
  if (flag0 & FB_TIMEOUT_BIT) { fb_void_outputs(p, bid, nb); return; }
  FbSync fs;
  fs.claim = ws.bar + 32;
  fs.done = ws.bar + 32 + FB_PHASES;
  fs.flag = ws.fallback;
  uint32_t phase = 0;
  bool alive = true;
  // (workgroup 0 leaves the 100 MHz wall clock of every phase end behind the counter: tools/fallback_cost.py)
  auto run = [&](uint32_t V, auto&& body) {
    if (!alive) return;
    alive = fb_phase(fs, phase, V, &word_s, body);
    ++phase;
    if (bid == 0 && threadIdx.x == 0 && phase < 15) ws.bar[1 + phase] = (uint32_t)wall_clock64();
  };
  if (bid == 0 && threadIdx.x == 0) ws.bar[1] = (uint32_t)wall_clock64();
  const int B = p.num_seqs, G = B * p.num_layers * p.num_kv_heads;
  const uint32_t VS = (uint32_t)B < 4u * vgrid ? (uint32_t)B : 4u * vgrid;     // per-sequence phases
  const uint32_t VH = (uint32_t)G < 8u * vgrid ? (uint32_t)G : 8u * vgrid;     // the per-head phase
  if (have_keys) {
    run(vgrid, [&](unsigned v, unsigned V) { zero_body(zero16, zero_vecs, v, V); });
  } else {
    // every logical block of the batch has a physical block (the collecting pass counted them:
    // the same for all workgroups) -> nothing to clear, nothing to fix: two phases less
    uint32_t claimed = 0;
    for (int q = 0; q < CLAIM_SHARDS; ++q) claimed += ws.st_claimed[q * 32];
    const bool holes = (int64_t)claimed != p.total_slots / p.block_size && !(p.lean & 2);
    auto keys = [&](unsigned v, unsigned V) {
      if (sparse) build_keys_sparse_body(p, ws, v, V);
      else build_keys_body<4>(p, ws, v, V);
    };
    if (holes) {
      run(vgrid, [&](unsigned v, unsigned V) { zero_body(zero16, zero_vecs, v, V); clear_chunk_table_body(p, ws, v, V); });
      run(vgrid, keys);
      run(vgrid, [&](unsigned v, unsigned V) { fix_unclaimed_body(p, ws, v, V); });
    } else {
      run(vgrid, [&](unsigned v, unsigned V) { zero_body(zero16, zero_vecs, v, V); keys(v, V); });
    }
  }
  for (int round = 0; round < 4; ++round) {
    run(vgrid, [&](unsigned v, unsigned V) { hist_round_body(p, ws, round, v, V); });
    if (round == 0 && coupled) {                       // the reference's batch > 1 rule: totals, k', then the pick
      run(VS, [&](unsigned v, unsigned V) {
        for (int i = (int)v; i < B; i += (int)V) { scan_pick_body<4, 4>(p, ws, 0, i, 1); __syncthreads(); }
      });
      run(1u, [&](unsigned, unsigned) { seq_prepare_tables(p, ws, prep_s); });
      run(VS, [&](unsigned v, unsigned V) {
        for (int i = (int)v; i < B; i += (int)V) { scan_pick_body<4, 4>(p, ws, 0, i, 2); __syncthreads(); }
      });
    } else {
      run(VS, [&](unsigned v, unsigned V) {
        for (int i = (int)v; i < B; i += (int)V) { scan_pick_body<4, 4>(p, ws, round, i); __syncthreads(); }
      });
    }
  }
  if (!alive) { fb_void_outputs(p, bid, nb); return; }
  // the last phase: nobody waits for it (the kernel's end does)
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) word_s = __hip_atomic_fetch_add(fs.claim + phase, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t v = word_s;
    if (v >= VH) break;
    for (int g = (int)v; g < G; g += (int)VH) {
      select_emit_head<256>(p, ws, 0, g, nullptr);
      __syncthreads();
    }
  }
  if (bid == 0 && threadIdx.x == 0) ws.bar[17] = (uint32_t)wall_clock64();     // (workgroup 0's own end)
}

}  // namespace kvc

// --------------------------------------------------------------------------- host side
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// The null padding of the output list is 4 B per candidate slot of pure HBM writes with nobody
// waiting for it until the emission: on the small-eviction schedule it runs on a side stream,
// forked off the caller's stream behind the sampling kernel and joined in front of emit_topk, so
// that it runs next to the pivot kernel (registers and LDS: the HBM is idle) and into the start
// of the collecting pass.  Measured at 256 x 1 M slots (step = S1 + S2 + S3): inline 1.92 ms;
// forked at the very start 1.84 (the sampling pass lives on random accesses and is slowed down
// as much as the fill is hidden); forked behind the sampling kernel 1.76; split with a part next
// to the record / selection kernels 1.78-1.81 (S1 itself is 20 us shorter, but the list is then
// still dirty in the caches when schedule_cache_moves starts, which pays 50 us for it).
// One non-blocking stream and two events per (host thread, device), made on first use and kept for
// the life of the thread (they are not destroyed at thread exit: by then the HIP runtime may be
// unloading); never created under stream capture (a call that is being captured before any other
// gets the fill inline).  Any failure of the fork / join calls is answered by doing without the
// side stream (the fill inline, or waited for on the host) -- never by an unordered fill.
struct SideStream { hipStream_t s2 = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool failed = false; };
static SideStream* side_stream(hipStream_t main) {
  thread_local SideStream tab[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  SideStream& t = tab[dev & 63];
  if (t.failed) return nullptr;
  if (t.s2 == nullptr) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      return nullptr;
    }
    if (hipStreamCreateWithFlags(&t.s2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&t.join, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      t.failed = true;
      return nullptr;
    }
  }
  return &t;
}

// > 64 KiB of dynamic LDS needs an opt-in per function AND per device (gfx950 has 160 KiB per CU):
// done once per (function, device), whatever thread or device the caller is on
static void allow_dynamic_lds(const void* fn, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done.fetch_or(bit, std::memory_order_release);
}

struct WsLayout {
  size_t keys, zero_begin, chunk_phys, bsample, hist, less, eq, seq_prefix, seq_k, zero_end, cum,
      seq_tmp, tz_begin, fallback, st_claimed, st_cnt, st_def, st_samp, tz_end, st_seqrec, head_fc, rec64, blist, bthr, total;
};

static WsLayout ws_layout(int64_t N, int32_t G, int32_t B, int32_t bs) {
  WsLayout l;
  size_t o = 0;
  l.keys = o;        o = align_up(o + (size_t)N * 4, 256);      // keys + chunk_phys (+ bsample on the bracket schedule): ONE 0xFF memset
  l.chunk_phys = o;  o = align_up(o + (size_t)(N / bs + 1) * 4, 256);
  l.bsample = o;     o = align_up(o + (size_t)B * kvc::BR_CELLS * 4, 256);   // (bracket schedule only; the memset's tail)
  l.zero_begin = o;  // everything up to zero_end is cleared by build_keys' tail workgroups
  l.hist = o;        o = align_up(o + (size_t)G * kvc::RADIX * 4, 256);
  l.less = o;        o = align_up(o + (size_t)G * 4, 256);
  l.eq = o;          o = align_up(o + (size_t)G * 4, 256);
  l.seq_prefix = o;  o = align_up(o + (size_t)B * 4, 256);
  l.seq_k = o;       o = align_up(o + (size_t)B * 4, 256);
  l.zero_end = o;
  l.cum = o;         o = align_up(o + (size_t)4 * G * kvc::RADIX * 4, 256);
  l.seq_tmp = o;     o = align_up(o + (size_t)B * 12, 256);
  l.tz_begin = o;    // one memset at the head of the small-eviction schedule
  l.fallback = o;    o = align_up(o + 512, 256);   // flag word | +128: stamps | +256: claim / done counters of the fallback's phases
  l.st_claimed = o;  o = align_up(o + (size_t)kvc::CLAIM_SHARDS * 128, 256);
  l.st_cnt = o;      o = align_up(o + (size_t)G * 4, 256);
  l.st_def = o;      o = align_up(o + (size_t)G * 4, 256);
  l.st_samp = o;     o = align_up(o + (size_t)G * 4, 256);
  l.tz_end = o;
  l.st_seqrec = o;   o = align_up(o + (size_t)B * 16, 256);
  l.head_fc = o;     o = align_up(o + (size_t)G * 8, 256);
  l.rec64 = o;       o = align_up(o + (size_t)G * kvc::KREC * 8, 256);
  l.blist = o;       o = align_up(o + (size_t)(N / kvc::BR_DIV + (int64_t)G * kvc::BR_PAD + 4) * 4, 256);
  l.bthr = o;        o = align_up(o + (size_t)(N / bs + 1) * 4, 256);
  l.total = o;
  return l;
}

// grid of the single-launch fallback (section 8): what is resident at once on an idle device, less
// one workgroup per CU (the occupancy query can be one too high where SGPRs decide,
// MI355X_MICROARCH.md), at most 3 per CU (4 measured slower: the cost of a phase end grows with the
// number of waiters); asked once per device.  Only speed depends on it: the kernel's phases wait for
// work, not for workgroups (kvc_schedule_params.fallback_grid launches any other number: tests)
static int fallback_grid() {
  static std::atomic<int> grid[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (grid[dev].load(std::memory_order_relaxed) == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kvc::fallback_general_kernel, 256, 0) != hipSuccess)
      per_cu = 1;
    per_cu = per_cu - 1 < 1 ? 1 : (per_cu - 1 > 3 ? 3 : per_cu - 1);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    grid[dev].store(per_cu * cus, std::memory_order_relaxed);
  }
  return grid[dev].load(std::memory_order_relaxed);
}

// small-eviction schedule (section 7) or not: the host knows how many blocks a sequence frees at
// most (the reference passes a Python list); eligible when that is on average <= 1/8 of what a
// head's record covers and a sequence's thresholds fit one workgroup's LDS.
// p2 = padded threshold count per sequence (0: not eligible), sshift = log2 of the sample stride;
// returns why not (KVC_WHY_*, include/kvc_mi355x.h; KVC_WHY_TAKEN = eligible)
static int topk_plan(const kvc_schedule_params& p, int& p2_out, int& sshift) {
  p2_out = 0; sshift = 0;
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  if (G < 1 || p.total_slots <= 0) return KVC_WHY_EMPTY;
  const int LH = p.num_layers * p.num_kv_heads;
  const int bsz = p.block_size;
  if (p.schedule_path == 1 || p.schedule_path == 4) return KVC_WHY_FORCED_PATH;
  if (!(bsz == 8 || bsz == 16 || bsz == 32)) return KVC_WHY_BLOCK_SIZE;
  // (per-head tables of the pivot kernel in LDS; a record entry packs the physical slot into 32 bits)
  if (LH > kvc::PIV_MAXLH) return KVC_WHY_HEADS_PER_SEQ;
  if (p.num_seqs > 65535 || p.num_blocks * (int64_t)bsz >= (int64_t)1 << 32) return KVC_WHY_INDEX_RANGE;
  const int mch = kvc::KREC / bsz;
  int p2 = 128;
  while (p2 < LH * mch && p2 <= 16384) p2 <<= 1;
  if (p2 > 16384) return KVC_WHY_THRESHOLDS_LDS;
  if (p.schedule_path != 2 && p.schedule_path != 3) {
    if (p.max_evicted_blocks_hint < 0) return KVC_WHY_HINT_UNKNOWN;
    if ((int64_t)p.max_evicted_blocks_hint * 8 > (int64_t)mch * LH) return KVC_WHY_BULK;
  }
  p2_out = p2;
  // sample stride: about 16 Ki sampled keys per sequence (a sequence's pivot is a low quantile of
  // its sample, held in the registers of one workgroup; a sampled block costs ~9 random accesses,
  // a candidate ~3: at 256 x 1 M slots stride 64 is where the two meet); small sequences are
  // sampled whole
  int64_t stride = p.total_slots / p.num_seqs / 16384;
  if (p.sample_stride > 0) stride = p.sample_stride;
  while (sshift < 8 && (2ll << sshift) <= stride) ++sshift;
  return KVC_WHY_TAKEN;
}

// bracket schedule (section 9) or the digit rounds, for calls the small-eviction schedule does not
// take: a head per thread of one workgroup, list indices in 32 bits.
// Chosen by itself from 64 Ki slots per sequence and 64 blocks per head on: below that the digit
// rounds are as fast, and a head's list (an eighth of its slots) gets too short for the bracket.
// Returns why not (KVC_WHY_TAKEN = the bracket schedule).
static int bracket_why(const kvc_schedule_params& p) {
  const int LH = p.num_layers * p.num_kv_heads;
  const int64_t G = (int64_t)p.num_seqs * LH;
  if (G < 1 || p.total_slots <= 0 || p.block_size < 1) return KVC_WHY_EMPTY;
  if (p.schedule_path != 0 && p.schedule_path != 4) return KVC_WHY_FORCED_PATH;
  // (the reference's batch > 1 rule: as many sequences as the single-launch fallback has tables for)
  if (p.mode == 0 && p.num_seqs > kvc::FB_MAX_COUPLED) return KVC_WHY_COUPLED_BATCH;
  if (LH > kvc::PIV_MAXLH) return KVC_WHY_HEADS_PER_SEQ;
  if (p.total_slots >= (int64_t)1 << 32) return KVC_WHY_INDEX_RANGE;
  if (p.schedule_path == 4) return KVC_WHY_TAKEN;
  if (!(p.total_slots / p.num_seqs >= 65536 && p.total_slots / G >= 64 * (int64_t)p.block_size)) return KVC_WHY_SMALL_BATCH;
  return KVC_WHY_TAKEN;
}
static bool bracket_plan(const kvc_schedule_params& p) { return bracket_why(p) == KVC_WHY_TAKEN; }

// introspection for tests and bench.py: which schedule a call with these parameters enqueues
// (0 = the digit rounds, 1 = small-eviction, 2 = bracket)
extern "C" int32_t kvc_schedule_evictions_plan(const kvc_schedule_params* p) {
  if (p == nullptr) return 0;
  int p2 = 0, sshift = 0;
  topk_plan(*p, p2, sshift);
  if (p2 > 0) return 1;
  return bracket_plan(*p) ? 2 : 0;
}

// ... and why: bits 0-7 why not the small-eviction schedule, bits 8-15 why not the bracket schedule
// (looked at only when the small-eviction one is not taken), bits 16-23 the form of the fallback
// behind a taken schedule (0 = the single launch of section 8, KVC_WHY_COUPLED_BATCH = the gated
// launch chain: the reference's batch > 1 rule over more sequences than its tables hold)
extern "C" int32_t kvc_schedule_evictions_plan_reason(const kvc_schedule_params* p) {
  if (p == nullptr) return KVC_WHY_EMPTY | (KVC_WHY_EMPTY << 8);
  int p2 = 0, sshift = 0;
  const int w1 = topk_plan(*p, p2, sshift);
  if (w1 == KVC_WHY_TAKEN)
    return (p->mode == 0 && p->num_seqs > kvc::FB_MAX_COUPLED) ? (KVC_WHY_COUPLED_BATCH << 16) : 0;
  return w1 | (bracket_why(*p) << 8);
}

// the key pass through the caller's block tables (build_keys_tables_kernel) or from the per-block
// metadata: tables given, a batch that takes less than half of the cache, 16-byte rows
static bool tables_plan(const kvc_schedule_params& p) {
  return p.block_tables != nullptr && p.seq_index_of_slot != nullptr && p.block_tables_width > 0 && p.max_num_seqs > 0 &&
         p.block_size >= 4 && p.block_size % 4 == 0 && p.total_slots < (int64_t)p.num_blocks * p.block_size / 2;
}
extern "C" int32_t kvc_schedule_evictions_uses_block_tables(const kvc_schedule_params* p) {
  if (p == nullptr || !tables_plan(*p)) return 0;
  int p2 = 0, sshift = 0;
  topk_plan(*p, p2, sshift);
  return p2 > 0 ? 0 : 1;                             // (the small-eviction schedule streams the store: no tables)
}

// introspection for tests and bench.py: 1 if a call with these parameters enqueues the
// small-eviction schedule; byte offset of its `fallback` word inside the workspace (non-zero
// after the call = the general pipeline behind it recomputed the result)
extern "C" int32_t kvc_schedule_evictions_uses_small_eviction_schedule(const kvc_schedule_params* p) {
  int p2 = 0, sshift = 0;
  if (p != nullptr) topk_plan(*p, p2, sshift);
  return p2 > 0 ? 1 : 0;
}
static WsLayout ws_layout(int64_t N, int32_t G, int32_t B, int32_t bs);
extern "C" size_t kvc_schedule_evictions_fallback_offset(int64_t total_slots, int32_t total_heads,
                                                         int32_t num_seqs, int32_t block_size) {
  if (block_size < 1) return 0;
  return ws_layout(total_slots, total_heads, num_seqs, block_size).fallback;
}

#ifdef KVC_BR_STAMPS
extern "C" size_t kvc_br_stamps_offset(int64_t total_slots, int32_t total_heads, int32_t num_seqs, int32_t block_size) {
  return ws_layout(total_slots, total_heads, num_seqs, block_size).head_fc;
}
#endif

extern "C" size_t kvc_schedule_evictions_workspace_bytes(int64_t total_slots, int32_t total_heads,
                                                         int32_t num_seqs, int32_t block_size) {
  if (block_size < 1) return 0;
  return ws_layout(total_slots, total_heads, num_seqs, block_size).total;
}

extern "C" int kvc_schedule_evictions(const kvc_schedule_params* pp, void* workspace,
                                      size_t workspace_bytes, kvc_stream_t stream) {
  using namespace kvc;
  const kvc_schedule_params p = *pp;
  if (p.block_size < 1) return fail_invalid("Unsupported block size: " + std::to_string(p.block_size));
  // (the per-sequence tables of seq_prepare live in LDS: 20 B per sequence of the 160 KiB)
  if (p.num_seqs < 1 || p.num_seqs > 6500)
    return fail_invalid("schedule_evictions: num_seqs must be in [1,6500]");
  if (p.mode != 0 && p.mode != 1) return fail_invalid("schedule_evictions: mode must be 0 or 1");
  if (p.total_slots < 0 || p.total_slots >= (int64_t)2147483647)
    return fail_invalid("schedule_evictions: total slots must stay below 2^31 (int32 offsets)");
  if (p.total_slots % p.block_size != 0)
    return fail_invalid("schedule_evictions: total_slots must be a multiple of block_size");
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  const int B = p.num_seqs;
  const WsLayout l = ws_layout(p.total_slots, G, B, p.block_size);
  if (workspace_bytes < l.total) return fail_invalid("schedule_evictions: workspace too small");
  if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0)
    return fail_invalid("schedule_evictions: workspace must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  uint8_t* wb = reinterpret_cast<uint8_t*>(workspace);
  SchedWs ws;
  ws.keys = reinterpret_cast<uint32_t*>(wb + l.keys);
  ws.chunk_phys = reinterpret_cast<int32_t*>(wb + l.chunk_phys);
  ws.bsample = nullptr;
  ws.bnonfin = nullptr;
  ws.bk = p.evicted_blocks_per_seq;
  ws.hist = reinterpret_cast<uint32_t*>(wb + l.hist);
  ws.cum = reinterpret_cast<uint32_t*>(wb + l.cum);
  ws.less = reinterpret_cast<uint32_t*>(wb + l.less);
  ws.eq = reinterpret_cast<uint32_t*>(wb + l.eq);
  ws.seq_prefix = reinterpret_cast<uint32_t*>(wb + l.seq_prefix);
  ws.seq_k = reinterpret_cast<int32_t*>(wb + l.seq_k);
  ws.seq_tmp = reinterpret_cast<int32_t*>(wb + l.seq_tmp);
  ws.rec64 = reinterpret_cast<uint64_t*>(wb + l.rec64);
  ws.st_cnt = reinterpret_cast<uint32_t*>(wb + l.st_cnt);
  ws.st_def = reinterpret_cast<uint32_t*>(wb + l.st_def);
  ws.st_samp = reinterpret_cast<uint32_t*>(wb + l.st_samp);
  ws.st_claimed = reinterpret_cast<uint32_t*>(wb + l.st_claimed);
  ws.st_seqrec = reinterpret_cast<SeqRec*>(wb + l.st_seqrec);
  ws.fallback = reinterpret_cast<uint32_t*>(wb + l.fallback);
  ws.bar = reinterpret_cast<uint32_t*>(wb + l.fallback + 128);     // (stamps; +128 / +256 the phase counters: same zeroed region)
  ws.head_fc = reinterpret_cast<uint32_t*>(wb + l.head_fc);
  ws.blist = reinterpret_cast<uint32_t*>(wb + l.blist);
  ws.bthr = reinterpret_cast<uint32_t*>(wb + l.bthr);
  ws.gate = nullptr;
  if (p.total_slots == 0) {
    hipMemsetAsync(p.evicted_kv_count, 0, (size_t)G * 4, s);
    hipMemsetAsync(p.evicted_block_count, 0, (size_t)G * 4, s);
    return check_launch("schedule_evictions(empty)");
  }
  // keys default to "not evictable" (0xFFFFFFFF > KEY_INF) and the chunk table to -1 for slots
  // no physical block claims (inconsistent metadata); histograms and counters are zeroed by
  // the tail workgroups of build_keys
  const int LH = p.num_layers * p.num_kv_heads;
  int topk_p2 = 0, sshift = 0;
  topk_plan(p, topk_p2, sshift);
  const bool topk = topk_p2 > 0;
  const size_t prep_lds = (size_t)B * 24;
  if (prep_lds > 64 * 1024) {
    static std::atomic<uint64_t> prep_done{0};
    allow_dynamic_lds(reinterpret_cast<const void*>(seq_prepare_kernel), 160 * 1024 - 1024, prep_done);
  }
  if (topk) {
    // ---- small-eviction schedule (section 7): one memset of its counters, the output fill (side
    // stream), 6 launches (+ 2 for the reference's batch > 1 rule); the general pipeline is enqueued
    // behind it -- one gated launch (section 8) -- and runs only if the flag was raised
    hipMemsetAsync(wb + l.tz_begin, 0, l.tz_end - l.tz_begin, s);
    SideStream* side = nullptr;
    if (!(p.lean & 1) && p.eli_dirty_map == nullptr) {
      if (p.total_slots >= (1 << 22)) side = side_stream(s);
      if (side == nullptr)
        hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p.evicted_logical_indices), p.null_value,
                          (size_t)p.total_slots, s);
    }
    static std::atomic<uint64_t> sel_done{0};
    allow_dynamic_lds(reinterpret_cast<const void*>(seq_select_topk_kernel), 156 * 1024, sel_done);   // + its static tables
    // positions only for the slots whose metric lies below the pivot (stream_collect_kernel, LAZY):
    // keys that do not depend on the position, sequences that do not need each other's inf counts
    const bool lazy = !p.use_average && p.bias == nullptr && !(p.mode == 0 && B > 1) && p.schedule_path != 3;
    {
      int64_t sb = (p.num_blocks + 2047) / 2048;     // >= 8 steps of 64 block indices per wave
      sb = sb < 1 ? 1 : (sb > 4096 ? 4096 : sb);
      const dim3 grid((unsigned)sb), blk(256);
      if (p.block_size == 8) hipLaunchKernelGGL(stream_sample_kernel<8>, grid, blk, 0, s, p, ws, sshift);
      else if (p.block_size == 16) hipLaunchKernelGGL(stream_sample_kernel<16>, grid, blk, 0, s, p, ws, sshift);
      else hipLaunchKernelGGL(stream_sample_kernel<32>, grid, blk, 0, s, p, ws, sshift);
    }
    if (side != nullptr) {
      if (hipEventRecord(side->fork, s) != hipSuccess || hipStreamWaitEvent(side->s2, side->fork, 0) != hipSuccess) {
        (void)hipGetLastError();                     // (no side stream after all: inline)
        side = nullptr;
      }
      hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p.evicted_logical_indices), p.null_value,
                        (size_t)p.total_slots, side != nullptr ? side->s2 : s);
      if (side != nullptr && hipEventRecord(side->join, side->s2) != hipSuccess) {
        // no event to wait for: the fill is waited for here and now (a later wait on `join` would
        // refer to an earlier call's record), and this thread fills inline from now on
        (void)hipGetLastError();
        (void)hipStreamSynchronize(side->s2);
        side->failed = true;
        side = nullptr;
      }
    }
    hipLaunchKernelGGL(stream_pivot_kernel, dim3(B), dim3(1024), 0, s, p, ws, sshift);
    {
      // blocks of the batch / blocks of the cache: a dense cache requests the rows before it has
      // looked at the metadata, a sparse one (engine-sized cache, small batch) only the batch's rows
      const bool dense = p.total_slots >= (int64_t)p.num_blocks * p.block_size / 2;
      int64_t cb = dense ? (p.num_blocks + 255) / 256 : (p.num_blocks + kvc::SPARSE_CHUNK - 1) / kvc::SPARSE_CHUNK;
      cb = cb < 1 ? 1 : (cb > 4096 ? 4096 : cb);
      const dim3 grid((unsigned)cb), blk(256);
#define KVC_COLLECT(BSV)                                                                                  \
      if (lazy) {                                                                                         \
        if (dense) hipLaunchKernelGGL((stream_collect_kernel<BSV, true, true>), grid, blk, 0, s, p, ws);  \
        else hipLaunchKernelGGL((stream_collect_kernel<BSV, false, true>), grid, blk, 0, s, p, ws);       \
      } else {                                                                                            \
        if (dense) hipLaunchKernelGGL((stream_collect_kernel<BSV, true, false>), grid, blk, 0, s, p, ws); \
        else hipLaunchKernelGGL((stream_collect_kernel<BSV, false, false>), grid, blk, 0, s, p, ws);      \
      }
      if (p.block_size == 8) { KVC_COLLECT(8); }
      else if (p.block_size == 16) { KVC_COLLECT(16); }
      else { KVC_COLLECT(32); }
#undef KVC_COLLECT
    }
    hipLaunchKernelGGL((stream_records_kernel<4, 4>), dim3((G + 15) / 16), dim3(256), 0, s, p, ws, lazy ? 1 : 0);
    const int coupled_tk = (p.mode == 0 && B > 1) ? 1 : (lazy ? 2 : 0);
    if (coupled_tk == 1) {
      hipLaunchKernelGGL(seq_sums_topk_kernel, dim3(B), dim3(256), 0, s, p, ws);
      hipLaunchKernelGGL(seq_prepare_kernel, dim3(1), dim3(1024), prep_lds, s, p, ws);
    }
    hipLaunchKernelGGL(seq_select_topk_kernel, dim3(B), dim3(1024), (size_t)topk_p2 * 8 + (size_t)LH * 4, s, p, ws, topk_p2, coupled_tk);
    if (side != nullptr && hipStreamWaitEvent(s, side->join, 0) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipStreamSynchronize(side->s2);          // (the emission below must not race the fill)
      side->failed = true;
    }
    hipLaunchKernelGGL(emit_topk_kernel<4>, dim3((G + 3) / 4), dim3(256), 0, s, p, ws);
    ws.gate = ws.fallback;
  }
  if (topk && !(p.mode == 0 && B > kvc::FB_MAX_COUPLED)) {
    // ---- the general pipeline as ONE gated launch (section 8)
    uint4* z16 = reinterpret_cast<uint4*>(wb + l.zero_begin);
    const int64_t zv = (int64_t)((l.zero_end - l.zero_begin) / 16);
    const int sparse = p.total_slots < (int64_t)p.num_blocks * p.block_size / 2 ? 1 : 0;
    const unsigned vgrid = (unsigned)fallback_grid();
    hipLaunchKernelGGL(fallback_general_kernel, dim3(p.fallback_grid > 0 ? (unsigned)p.fallback_grid : vgrid), dim3(256), 0, s,
                       p, ws, sparse, z16, zv, 0, vgrid);
    return check_launch("schedule_evictions");
  }
  // ---- general pipeline
  // keys default to "not evictable" (0xFFFFFFFF > KEY_INF) and the chunk table to -1 for slots
  // no physical block claims (inconsistent metadata); histograms and counters are zeroed by
  // the tail workgroups of build_keys.  Behind the small-eviction schedule (gated) the clear is a
  // gated kernel instead of a memset.
  // bulk evictions: T* from a bracket around a sample's quantile
  // instead of four digit rounds (section 9, bracket_plan)
  const bool bracket = !topk && bracket_plan(p);
  // a batch that is sparse in its cache, with the caller's block tables at hand: the keys in logical
  // order through the tables (build_keys_tables_kernel; every slot is written: no clearing)
  const bool by_tables = !topk && tables_plan(p);
  // (the bracket's sample -- 128 KiB per sequence -- lies behind the keys and the chunk table: cleared with them
  // only when that schedule runs)
  if (!topk && !(p.lean & 2) && !by_tables) hipMemsetAsync(ws.keys, 0xFF, (bracket ? l.zero_begin : l.bsample) - l.keys, s);
  const bool bracket_coupled = bracket && p.mode == 0 && B > 1;
  if (bracket) ws.bsample = reinterpret_cast<uint32_t*>(wb + l.bsample);   // build_keys leaves the sample behind
  if (bracket_coupled) {                             // ... and counts the keys that are not evictable
    hipMemsetAsync(wb + l.tz_begin, 0, l.tz_end - l.tz_begin, s);
    ws.bnonfin = ws.st_samp;
    ws.bk = ws.seq_k;
  }
  if (topk && !(p.lean & 2)) hipLaunchKernelGGL(clear_chunk_table_kernel, dim3(1024), dim3(256), 0, s, p, ws);
  {
    uint4* z16 = reinterpret_cast<uint4*>(wb + l.zero_begin);
    const int64_t zv = (int64_t)((l.zero_end - l.zero_begin) / 16);
    const int64_t zb64 = (zv + 1023) / 1024;
    const unsigned zb = (unsigned)(zb64 < 1 ? 1 : (zb64 > 2048 ? 2048 : zb64));
    const int64_t threads = p.block_size % 4 == 0 ? p.num_blocks * (p.block_size / 4) : p.num_blocks * p.block_size;
    // grid-stride: at most 16 Ki workgroups (an engine-sized cache has tens of millions of blocks,
    // most of them outside the batch: one workgroup per 64 of them cost 0.5 ms in dispatch alone);
    // half of that behind the small-eviction schedule, where the launch is a no-op unless the flag was raised
    int64_t db64 = (threads + 255) / 256;
    const int64_t cap = topk ? 8192 : 16384;
    if (db64 > cap) db64 = cap;
    const unsigned db = (unsigned)db64;
    const int bsz = p.block_size;
    // blocks of the batch / blocks of the cache: a dense cache is faster with one independent thread
    // per 4 slots (build_keys_kernel), a sparse one with the compacting sweep
    const bool sparse = p.total_slots < (int64_t)p.num_blocks * bsz / 2;
    if (by_tables) {
      int64_t tb64 = (p.total_slots + 1023) / 1024;
      if (tb64 > cap) tb64 = cap;
      const unsigned tbk = (unsigned)(tb64 < 1 ? 1 : tb64);
      hipLaunchKernelGGL(build_keys_tables_kernel, dim3(tbk + zb), dim3(256), 0, s, p, ws, tbk, z16, zv);
    } else if (sparse && (bsz == 4 || bsz == 8 || bsz == 16 || bsz == 32 || bsz == 64)) {
      // one workgroup per SPARSE_CHUNK blocks (grid-stride when capped)
      int64_t wb64 = (p.num_blocks + kvc::SPARSE_CHUNK - 1) / kvc::SPARSE_CHUNK;
      if (wb64 > cap) wb64 = cap;
      const unsigned wbk = (unsigned)(wb64 < 1 ? 1 : wb64);
      hipLaunchKernelGGL(build_keys_sparse_kernel, dim3(wbk + zb), dim3(256), 0, s, p, ws, wbk, z16, zv);
    } else if (p.block_size % 4 == 0)
      hipLaunchKernelGGL(build_keys_kernel<4>, dim3(db + zb), dim3(256), 0, s, p, ws, db, z16, zv);
    else
      hipLaunchKernelGGL(build_keys_kernel<1>, dim3(db + zb), dim3(256), 0, s, p, ws, db, z16, zv);
  }
  if (topk && !(p.lean & 2)) hipLaunchKernelGGL(fix_unclaimed_kernel, dim3(1024), dim3(256), 0, s, p, ws);
  const int64_t htiles_all = (p.total_slots + HTILE - 1) / HTILE;
  // persistent grid: 4 workgroups per CU up to 4M keys per round-pass, growing to 16 per CU for
  // very large batches (measured: 1024 is best at 8M keys, 4096 is 18 % faster at 270M)
  int64_t hgrid = htiles_all / 32;
  hgrid = hgrid < 1024 ? 1024 : (hgrid > 4096 ? 4096 : hgrid);
  const unsigned htiles = (unsigned)(htiles_all < hgrid ? htiles_all : hgrid);
  if (bracket) {
    // ---- bracket schedule (section 9) over the keys just built; the digit rounds behind it as the
    // single gated launch of section 8
    static std::atomic<uint64_t> bsel_done{0};
    allow_dynamic_lds(reinterpret_cast<const void*>(bracket_select_kernel), 140 * 1024, bsel_done);   // (+ 17.7 KiB static)
    int p2 = 1024;
    while (p2 < LH * 128 && p2 < 32768) p2 <<= 1;      // room for ~128 listed thresholds per head
    const size_t sel_lds = (size_t)p2 * 4 + (size_t)(2 * LH + 1) * 4;
    if (bracket_coupled) {                           // k' of every sequence first (the batch > 1 rule)
      hipLaunchKernelGGL(bracket_totals_kernel, dim3(B), dim3(256), 0, s, p, ws);
      hipLaunchKernelGGL(seq_prepare_kernel, dim3(1), dim3(1024), prep_lds, s, p, ws);
    }
    hipLaunchKernelGGL(bracket_kernel, dim3(B), dim3(1024), 0, s, p, ws);
    hipLaunchKernelGGL(count_collect_kernel, dim3(htiles), dim3(256), 0, s, p, ws);
    hipLaunchKernelGGL(bracket_records_kernel, dim3(G), dim3(512), 0, s, p, ws);
    hipLaunchKernelGGL(bracket_select_kernel, dim3(B), dim3(1024), sel_lds, s, p, ws, p2);
    {
      const int64_t avg = p.total_slots / G;
      int64_t want = (avg + avg / 4 + 2047) / 2048 * 2048;
      const int lds_cap = (int)(want < 2048 ? 2048 : (want > 32768 ? 32768 : want));
      if (avg <= 8192) {
        hipLaunchKernelGGL(select_emit_kernel<256>, dim3(G), dim3(256), (size_t)lds_cap * 4, s, p, ws, lds_cap, 1);
      } else {
        static std::atomic<uint64_t> long_done_b{0};
        allow_dynamic_lds(reinterpret_cast<const void*>(select_emit_kernel<1024>), 32768 * 4, long_done_b);
        hipLaunchKernelGGL(select_emit_kernel<1024>, dim3(G), dim3(1024), (size_t)lds_cap * 4, s, p, ws, lds_cap, 1);
      }
    }
    ws.gate = ws.fallback;
    uint4* z16 = reinterpret_cast<uint4*>(wb + l.zero_begin);
    const int64_t zv = (int64_t)((l.zero_end - l.zero_begin) / 16);
    const unsigned vgrid = (unsigned)fallback_grid();
    hipLaunchKernelGGL(fallback_general_kernel, dim3(p.fallback_grid > 0 ? (unsigned)p.fallback_grid : vgrid), dim3(256), 0, s,
                       p, ws, 0, z16, zv, 1, vgrid);
    return check_launch("schedule_evictions");
  }
  // per round: the histograms, then ONE launch for scan + pick (round 0: + the chunk totals and k',
  // round 3: + the per-head counts).  Only the reference's batch > 1 rule (mode 0, B > 1) couples
  // the sequences in round 0 and takes three launches there (totals | k' | pick).  10 (12) launches
  // + the memset.
  const bool coupled = p.mode == 0 && B > 1;
  for (int round = 0; round < 4; ++round) {
    hipLaunchKernelGGL(hist_round_kernel, dim3(htiles), dim3(256), 0, s, p, ws, round);
    if (round == 0 && coupled) {
      hipLaunchKernelGGL(scan_pick_kernel, dim3(B), dim3(1024), 0, s, p, ws, 0, 1);
      hipLaunchKernelGGL(seq_prepare_kernel, dim3(1), dim3(1024), prep_lds, s, p, ws);
      hipLaunchKernelGGL(scan_pick_kernel, dim3(B), dim3(1024), 0, s, p, ws, 0, 2);
    } else {
      hipLaunchKernelGGL(scan_pick_kernel, dim3(B), dim3(1024), 0, s, p, ws, round, 0);
    }
  }
  {
    // stage a head's keys in LDS when the average head fits with 25 % slack (ragged heads
    // that do not fit read from L2).  Small heads (the continual-compression steady state:
    // thousands of heads of a few thousand slots) take 256-thread workgroups and a small
    // buffer so that 6+ heads share a CU; the per-head passes are barrier-latency bound.
    const int64_t avg = p.total_slots / G;
    int64_t want = (avg + avg / 4 + 2047) / 2048 * 2048;
    // (heads beyond 32k slots read their keys from L2: measured faster than one 144 KiB
    // staging workgroup per CU)
    const int lds_cap = (int)(want < 2048 ? 2048 : (want > 32768 ? 32768 : want));
    const unsigned sel_grid = (unsigned)((topk && G > 2048) ? 2048 : G);   // (gated: see select_emit_kernel)
    // > 64 KiB of dynamic LDS needs the opt-in (gfx950 has 160 KiB per CU); per device, cheap
    if (avg <= 8192) {
      hipLaunchKernelGGL(select_emit_kernel<256>, dim3(sel_grid), dim3(256), (size_t)lds_cap * 4, s, p, ws, lds_cap, 0);
    } else {
      static std::atomic<uint64_t> long_done{0};
      allow_dynamic_lds(reinterpret_cast<const void*>(select_emit_kernel<1024>), 32768 * 4, long_done);
      hipLaunchKernelGGL(select_emit_kernel<1024>, dim3(sel_grid), dim3(1024), (size_t)lds_cap * 4, s, p, ws, lds_cap, 0);
    }
  }
  return check_launch("schedule_evictions");
}
