// kvc_schedule_common.h -- A3 schedule_evictions: the scratch layout every schedule shares, small device helpers
// (one translation unit: included by kvc_schedule.hip in this order; see the overview there)
#pragma once
#include "kvc_common.h"
#include "kvc_harvest_layout.h"
#include "../../include/kvc_mi355x.h"

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
  uint32_t* bclaim;      // [64 x 32] bracket schedule without the 0xFF fill: the key pass counts the logical blocks it
                         //     finds a physical block for here (= st_claimed; nullptr: not wanted); bracket_kernel checks
                         //     the sum against N / bs and leaves the counters zero for the next call
  const int32_t* bk;     // [B] bracket schedule: chunks a sequence frees -- the caller's k (k' = min(k, finite) is found
                         //     on the way) or, under the batch > 1 rule, seq_k = k' itself
  uint32_t* bthr;        // [N / bs] bracket schedule: per head (from its first chunk on) its listed thresholds, ascending
  uint32_t* blist;       // [N / 8 + 32 G]  bracket schedule: per head the keys inside the sequence's bracket (then sorted)
  uint32_t* fallback;    // [1]      != 0: the small-eviction schedule could not finish exactly
  uint32_t* bar;         // [32+64]  single-launch fallback: phase stamps, then claim / done counters of its phases
  const uint32_t* gate;  // general-path kernels run only if gate == nullptr or *gate != 0
  const int64_t* n_dev;  // kvc_schedule_params.total_slots_dev (ABI 8): [0] the true N, [1] != 0: the call is void; nullptr: total_slots is N
  const int32_t* hv_seen_ctx;   // [G] lists made by the attention's epilogue: the context length every head's list was made
  const int32_t* hv_seen_seq;   // [2B] with, every sequence's (position, protected window) -- or nullptr: nothing to verify
};


// flag word bit: the bracket schedule's key pass did not find every logical block (or its counters were not clean):
// the fallback makes keys, chunk table and holes anew instead of starting from the keys that exist
constexpr uint32_t FB_HOLES_BIT = 8u;

// N as the kernels see it: a host that launches before it knows N passes an upper bound in total_slots (the layout,
// the grids) and the number itself through device memory (kvc_schedule_params.total_slots_dev, ABI version 8)
__device__ __forceinline__ int64_t true_n(const kvc_schedule_params& p, const SchedWs& ws) {
  return ws.n_dev != nullptr ? ws.n_dev[0] : p.total_slots;
}
// ... and a bound that turned out too small voids the call: nothing is read or written
__device__ __forceinline__ bool voided(const SchedWs& ws) { return ws.n_dev != nullptr && ws.n_dev[1] != 0; }

__device__ __forceinline__ bool gated_off(const SchedWs& ws) { return voided(ws) || (ws.gate != nullptr && *ws.gate == 0u); }

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
      // (only this owner changes these bits: a plain look at them is safe next to the neighbours'
      // atomics on theirs, and in the steady state nothing has to change)
      old = map[w] & mask;
      if (old & ~fresh) atomicAnd(&map[w], ~(old & ~fresh));
      if (fresh & ~old) atomicOr(&map[w], fresh & ~old);
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

// eli_dirty_update for an owner of at most 64 map words whose words were loaded beforehand (lane t: word
// (c0 >> 5) + t, `old_word`; topk_fused_kernel requests the words of all its heads together): the same result
__device__ __forceinline__ void eli_dirty_apply(uint32_t* map, int32_t* eli, int32_t c0, int32_t c1, int32_t new_chunks,
                                                int32_t keep_from, int bs_shift, int32_t null_value, uint32_t old_word, int lane) {
  // (called by all 64 lanes of the wave: the words are per lane, the entries that go back to null are the wave's)
  const int32_t w0 = c0 >> 5, w1 = (c1 - 1) >> 5;
  const int32_t w = w0 + lane;
  uint32_t old = 0;
  if (w <= w1) {
    const int32_t cn = c0 + new_chunks;
    const int32_t lo = max(c0, w << 5), hi = min(c1, (w + 1) << 5);
    const uint32_t mask = (hi - lo >= 32) ? 0xFFFFFFFFu : (((1u << (hi - lo)) - 1u) << (lo & 31));
    const int32_t nh = min(hi, cn);
    const uint32_t fresh = nh > lo ? ((nh - lo >= 32) ? 0xFFFFFFFFu : (((1u << (nh - lo)) - 1u) << (lo & 31))) : 0u;
    if (mask == 0xFFFFFFFFu) {
      old = old_word;
      if (old != fresh) map[w] = fresh;
    } else {
      old = old_word & mask;
      if (old & ~fresh) atomicAnd(&map[w], ~(old & ~fresh));
      if (fresh & ~old) atomicOr(&map[w], fresh & ~old);
    }
  }
  // everything from keep_from to the end of the owner's last dirty chunk goes back to null: a chunk that is not dirty
  // in between holds nulls already (that is what its bit says), so one strided fill by the wave does what a walk over
  // the dirty bits would -- one lane, one entry at a time
  const unsigned long long any = __ballot(old != 0u);
  if (any) {                                         // wave-uniform
    const int top = 63 - __clzll((long long)any);
    const uint32_t tw = (uint32_t)__builtin_amdgcn_readlane((int)old, top);
    const int32_t last = ((w0 + top) << 5) + (31 - __clz((int)tw));
    const int32_t ee = (last + 1) << bs_shift;
    for (int32_t e = keep_from + lane; e < ee; e += WAVE) eli[e] = null_value;
  }
}

// The same for up to 64 owners at once, by ONE wave with all its lanes active: lane q < n_owners of off_v / end_v / ce_v
// holds owner q's first entry, end and number of fresh entries.  A lane per (owner, map word) -- 64 / Wp owners per
// pass, Wp = the owners' largest word count rounded up to a power of two -- instead of a pass per owner with a
// handful of lanes at work (topk_fused_kernel: 16 heads of 8 words each are two passes, not sixteen).  The entries that
// go back to null: from the owner's fresh entries' end to the end of its last dirty chunk, filled by the owner's Wp
// lanes (chunks in between that are not dirty hold nulls already).  false (nothing done): an owner of more than 64 words.
__device__ __forceinline__ bool eli_dirty_apply_owners(uint32_t* map, int32_t* eli, int32_t off_v, int32_t end_v, uint32_t ce_v,
                                                       int n_owners, int bs_shift, int32_t null_value, int lane) {
  uint32_t nw = 0;
  if (lane < n_owners) {
    const int32_t c0 = off_v >> bs_shift, c1 = end_v >> bs_shift;
    nw = c0 < c1 ? (uint32_t)(((c1 - 1) >> 5) - (c0 >> 5) + 1) : 0u;
  }
  uint32_t W = 0;
  for (int d = 32; d > 0; d >>= 1) nw = max(nw, (uint32_t)__shfl_xor((int)nw, d, 64));
  W = (uint32_t)__builtin_amdgcn_readfirstlane((int)nw);
  if (W == 0u) return true;
  if (W > (uint32_t)WAVE) return false;
  const int lg = W <= 1u ? 0 : 32 - __clz((int)(W - 1u));
  const int per_pass = WAVE >> lg;
  const int t = lane & ((1 << lg) - 1);
  for (int q0 = 0; q0 < n_owners; q0 += per_pass) {    // wave-uniform
    const int q = q0 + (lane >> lg);
    const int qs = min(q, n_owners - 1);
    const int32_t off = __shfl(off_v, qs, 64), end = __shfl(end_v, qs, 64);
    const int32_t fresh_n = (int32_t)__shfl((int)ce_v, qs, 64);
    const int32_t c0 = off >> bs_shift, c1 = end >> bs_shift;
    const int32_t w = (c0 >> 5) + t;
    uint32_t old = 0;
    if (q < n_owners && c0 < c1 && w <= ((c1 - 1) >> 5)) {
      const int32_t cn = c0 + ((fresh_n + (1 << bs_shift) - 1) >> bs_shift);
      const int32_t lo = max(c0, w << 5), hi = min(c1, (w + 1) << 5);
      const uint32_t mask = (hi - lo >= 32) ? 0xFFFFFFFFu : (((1u << (hi - lo)) - 1u) << (lo & 31));
      const int32_t nh = min(hi, cn);
      const uint32_t fresh = nh > lo ? ((nh - lo >= 32) ? 0xFFFFFFFFu : (((1u << (nh - lo)) - 1u) << (lo & 31))) : 0u;
      const uint32_t old_word = map[w];
      if (mask == 0xFFFFFFFFu) {
        old = old_word;
        if (old != fresh) map[w] = fresh;
      } else {
        old = old_word & mask;
        if (old & ~fresh) atomicAnd(&map[w], ~(old & ~fresh));
        if (fresh & ~old) atomicOr(&map[w], fresh & ~old);
      }
    }
    int32_t last = old ? (w << 5) + 31 - __clz((int)old) : -1;      // the owner's highest dirty chunk: max over its lanes
    for (int d = 1; d < (1 << lg); d <<= 1) last = max(last, __shfl_xor(last, d, 64));
    if (q < n_owners && last >= 0) {
      const int32_t ee = (last + 1) << bs_shift;
      for (int32_t e = off + fresh_n + t; e < ee; e += 1 << lg) eli[e] = null_value;
    }
  }
  return true;
}

// Fills (`bytes` a multiple of 4, `dst` 4-byte aligned) as KERNEL launches, not hipMemsetAsync: on
// ROCm 7.2 a memset node recorded into a HIP graph fills its range on the first replay only (later
// replays leave it partly or wholly untouched: measured, tests/test_gpu_configs.py replays the
// captured step several times) -- and every schedule starts by clearing its counters.
__global__ __launch_bounds__(256) void fill32_kernel(uint32_t* __restrict__ dst, uint32_t value, int64_t words) {
  // leading words up to 16-byte alignment, 16-byte stores, trailing words
  const int64_t head = min(words, (int64_t)(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u) >> 2));
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nthreads = (int64_t)gridDim.x * 256;
  if (tid < head) dst[tid] = value;
  uint4* d16 = reinterpret_cast<uint4*>(dst + head);
  const int64_t vecs = (words - head) >> 2;
  const uint4 v = make_uint4(value, value, value, value);
  for (int64_t i = tid; i < vecs; i += nthreads) d16[i] = v;
  const int64_t tail0 = head + (vecs << 2);
  if (tail0 + tid < words) dst[tail0 + tid] = value;          // (< 4 words)
}
inline void fill32_async(void* dst, uint32_t value, size_t bytes, hipStream_t s) {
  const int64_t words = (int64_t)(bytes >> 2);
  if (words <= 0) return;
  int64_t blocks = (words / 4 + 255) / 256;                    // one 16-byte store per thread up to 16 Ki workgroups
  blocks = blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks);
  hipLaunchKernelGGL(fill32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<uint32_t*>(dst), value, words);
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


}  // namespace kvc
