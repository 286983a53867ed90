// A4 count_block_evictions and A5 schedule_t1_cache_moves for gfx950.
//
// Both reference kernels run ONE THREAD PER HEAD with a serial loop
// (csrc/kvcompress_eviction_kernels.cu:190-221, 223-289).  Here every loop iteration is
// an independent work item:
//
//  * count: one 64-lane wave per head; lanes test 64 chunk heads per step, the first
//    null chunk is found with a ballot.
//  * moves: the serial two-pointer walk is put in closed form (derivation in DESIGN.md):
//    with E the head's ascending evicted indices, cnt = |E|, R_k = E[cnt-1-k],
//    the k-th "skip" iteration is t_k = max(k, ctx-1-R_k) (strictly increasing), the
//    j-th emitted move comes from iteration i = j + k*, k* = min{k : t_k - k > j}
//    (binary search), and it exists iff i < cnt and E[j] < ctx-1-i.  One thread per ROW
//    of the [rows,2] move table, so the kernel also produces the zero fill of the
//    reference wrapper (vllm/_custom_ops.py:1168) in the same pass, writing each row once.
#include "kvc_common.h"
#include "../../include/kvc_mi355x.h"

namespace kvc {

// ------------------------------------------------------------------------------- A4
__global__ __launch_bounds__(256) void count_block_evictions_kernel(
    int32_t* __restrict__ evicted_block_count, int32_t* __restrict__ idx,
    const int32_t* __restrict__ offs, const int32_t* __restrict__ hang, int total_heads,
    int64_t total_kvs, int bs, int null_value) {
  const int g = blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x / WAVE);
  if (g >= total_heads) return;
  const int lane = lane_id();
  const int64_t start = offs[g];
  const int64_t end = (g + 1 >= total_heads) ? total_kvs : (int64_t)offs[g + 1];
  const int64_t nchunks = (end - start + bs - 1) / bs;   // loop "i < end; i += bs"
  int64_t run = nchunks;                                 // leading evicted chunks
  for (int64_t c0 = 0; c0 < nchunks; c0 += WAVE) {
    const int64_t c = c0 + lane;
    bool is_null = false;
    if (c < nchunks) is_null = idx[start + c * bs] == null_value;
    const unsigned long long m = __ballot(is_null);
    if (m) { run = c0 + __ffsll((long long)m) - 1; break; }
  }
  if (lane == 0) evicted_block_count[g] = (int)run;
  if (run > 0) {
    const int64_t last_end = start + run * bs;
    for (int64_t i = last_end - bs + hang[g] + lane; i < last_end; i += WAVE) idx[i] = null_value;
  }
}

// ------------------------------------------------------------------------------- A5
struct MoveHead {
  const int32_t* E;   // evicted indices of this head (ascending), cnt entries
  int cnt;
  int ctx;
};

// iteration index of the j-th emitted move, or -1
__device__ __forceinline__ int move_iteration(const MoveHead& h, int j) {
  // smallest k in [0,cnt] with (k == cnt) or max(0, ctx-1-E[cnt-1-k]-k) > j
  int lo = 0, hi = h.cnt;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    int d = h.ctx - 1 - h.E[h.cnt - 1 - mid] - mid;
    d = d > 0 ? d : 0;
    if (d > j) hi = mid; else lo = mid + 1;
  }
  const int i = j + lo;
  if (i >= h.cnt) return -1;
  if (h.E[j] >= h.ctx - 1 - i) return -1;   // "dst >= src": the walk has stopped
  return i;
}

// One workgroup per head (plus a few that clear the rows behind the last head).
//
// Regular heads (every evicted index is a real slot, E[cnt-1] < ctx -- always true when
// protected_window >= 1): the walk degenerates to "the j-th hole below new_len = ctx-cnt
// receives the j-th surviving slot of the tail [new_len, ctx), counted from the top".  The
// tail's evicted slots are marked in an LDS bitmap, the tail is scanned top-down with
// ballot prefix sums, and M = #holes below new_len moves are written, coalesced.
// Irregular heads and tails beyond the bitmap use the closed form above.
constexpr int MOVE_BITMAP_WORDS = 16384;

// rows [b, e) of the move workspace <- (0, 0): 16 B stores (the wrapper's fill_(0) is
// most of this op's bytes: 8 B per candidate slot, 2.2 GB at 256 resident sequences)
template <int THREADS>
__device__ __forceinline__ void zero_rows(int2* mv2, int64_t b, int64_t e, int tid, int nthreads) {
  typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
  if (b >= e) return;
  if ((reinterpret_cast<uintptr_t>(mv2 + b) & 15u) != 0) {      // odd leading row
    if (tid == 0) mv2[b] = make_int2(0, 0);
    ++b;
  }
  const int64_t pairs = (e - b) >> 1;
  i32x4* m4 = reinterpret_cast<i32x4*>(mv2 + b);
  const i32x4 z = {0, 0, 0, 0};
  // (plain stores: streaming ones made this op 0.13 ms slower at 256 resident sequences and the
  // compaction behind it 0.06 ms faster -- a loss)
  for (int64_t i = tid; i < pairs; i += nthreads) m4[i] = z;
  if (((e - b) & 1) && tid == 0) mv2[e - 1] = make_int2(0, 0);
}      // 512 Ki tail slots per head in LDS (64 KiB)

template <int THREADS, int BITMAP_WORDS>
__global__ __launch_bounds__(THREADS) void schedule_moves_heads_kernel(
    int32_t* __restrict__ moves, int64_t rows, int32_t* __restrict__ count,
    const int32_t* __restrict__ evicted, const int32_t* __restrict__ ekc,
    const int32_t* __restrict__ offs, const int32_t* __restrict__ block_tables,
    const int32_t* __restrict__ context_lens, int B, int L, int H, int M, int bs, int zero_fill) {
  __shared__ uint32_t bitmap[BITMAP_WORDS];       // 64 KiB for long heads, 4 KiB for short ones (occupancy)
  __shared__ uint32_t wave_tot[2][THREADS / WAVE];
  const int G = B * L * H;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int2* mv2 = reinterpret_cast<int2*>(moves);
  // slots of the last head -> total number of rows that belong to heads
  const int lastg = G - 1;
  const int lb = lastg / (L * H), llh = lastg % (L * H);
  const int last_ctx = context_lens[((llh / H) * B + lb) * H + (llh % H)];
  const int64_t n_total = (int64_t)offs[lastg] + (int64_t)((last_ctx + bs - 1) / bs) * bs;
  if ((int)blockIdx.x >= G) {                       // rows behind the last head: zeros
    if (!zero_fill) return;
    const int64_t nt = gridDim.x - G, part = (rows - n_total + nt - 1) / nt;   // one contiguous piece per workgroup
    const int64_t b = n_total + (int64_t)(blockIdx.x - G) * part;
    zero_rows<THREADS>(mv2, b, min(rows, b + part), tid, THREADS);
    return;
  }
  const int g = blockIdx.x;
  const int b = g / (L * H), lh = g % (L * H), l = lh / H, hh = lh % H;
  const int lbh = (l * B + b) * H + hh;
  const int cnt = ekc[g];
  const int ctx = context_lens[lbh];
  const int64_t off = offs[g];
  const int64_t seg_end = min((int64_t)rows, (g + 1 < G) ? (int64_t)offs[g + 1] : n_total);
  const int32_t* E = evicted + off;
  const int32_t* bt = block_tables + (int64_t)lbh * M;
  // the zeros behind the head's cnt entries depend on nothing below: they go out first, so that
  // the stores fly while the head's dependent loads (E, the block table) are still coming in --
  // with thousands of short heads (one move each, 33 KB of zeros) the kernel was waiting for those
  // loads with an idle store queue
  if (zero_fill) zero_rows<THREADS>(mv2, off + cnt, seg_end, tid, THREADS);
  int nmoves = 0;
  const bool regular = cnt > 0 && E[cnt - 1] < ctx && (cnt + 31) / 32 <= BITMAP_WORDS;
  if (cnt > 0 && regular) {
    const int new_len = ctx - cnt;
    // holes below new_len = lower_bound(E, new_len)
    int lo = 0, hi = cnt;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (E[mid] < new_len) lo = mid + 1; else hi = mid; }
    nmoves = lo;
    const int words = (cnt + 31) / 32;
    for (int i = tid; i < words; i += THREADS) bitmap[i] = 0;
    __syncthreads();
    for (int k = nmoves + tid; k < cnt; k += THREADS) {       // evicted slots inside the tail
      const int t = E[k] - new_len;
      atomicOr(&bitmap[t >> 5], 1u << (t & 31));
    }
    __syncthreads();
    uint32_t carry = 0;
    int buf = 0;
    for (int base = 0; base < cnt; base += THREADS) {         // i-th slot from the top
      const int i = base + tid;
      bool surv = false;
      int slot = 0;
      if (i < cnt) {
        slot = ctx - 1 - i;
        const int t = slot - new_len;
        surv = !((bitmap[t >> 5] >> (t & 31)) & 1u);
      }
      const unsigned long long bal = __ballot(surv);
      const uint32_t lane_ex = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wave_tot[buf][w] = (uint32_t)__popcll(bal);
      __syncthreads();
      uint32_t woff = 0, tot = 0;
#pragma unroll
      for (int q = 0; q < THREADS / WAVE; ++q) { const uint32_t c = wave_tot[buf][q]; if (q < w) woff += c; tot += c; }
      if (surv) {
        const int j = (int)(carry + woff + lane_ex);
        const int dst = E[j];
        if (off + j < rows)
          mv2[off + j] = make_int2(bt[dst / bs] * bs + dst % bs, bt[slot / bs] * bs + slot % bs);
      }
      carry += tot;
      buf ^= 1;
    }
  } else if (cnt > 0) {
    MoveHead h{E, cnt, ctx};
    for (int j = tid; j < cnt; j += THREADS) {
      const int i = move_iteration(h, j);
      if (i >= 0) {
        const int src = ctx - 1 - i, dst = E[j];
        if (off + j < rows)
          mv2[off + j] = make_int2(bt[dst / bs] * bs + dst % bs, bt[src / bs] * bs + src % bs);
      }
    }
    int lo = 0, hi = cnt;                                    // first j that is not a move
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (move_iteration(h, mid) >= 0) lo = mid + 1; else hi = mid; }
    nmoves = lo;
  }
  if (tid == 0) count[g] = nmoves;
  if (zero_fill) zero_rows<THREADS>(mv2, off + nmoves, min(seg_end, off + cnt), tid, THREADS);
}

}  // namespace kvc

extern "C" int kvc_count_block_evictions(int32_t* evicted_block_count,
                                         int32_t* evicted_logical_indices,
                                         const int32_t* evicted_kv_offsets,
                                         const int32_t* hanging_token_count,
                                         int32_t total_heads, int64_t total_kvs,
                                         int32_t block_size, int32_t null_value,
                                         kvc_stream_t stream) {
  if (block_size < 1) return kvc::fail_invalid("Unsupported block size: " + std::to_string(block_size));
  if (total_heads <= 0) return KVC_OK;
  const int waves_per_block = 4;
  dim3 grid((total_heads + waves_per_block - 1) / waves_per_block), block(waves_per_block * kvc::WAVE);
  hipLaunchKernelGGL(kvc::count_block_evictions_kernel, grid, block, 0, (hipStream_t)stream,
                     evicted_block_count, evicted_logical_indices, evicted_kv_offsets,
                     hanging_token_count, total_heads, total_kvs, block_size, null_value);
  return kvc::check_launch("count_block_evictions");
}

extern "C" int kvc_schedule_t1_cache_moves(
    int32_t* cache_moves_idx, int64_t cache_moves_rows, int32_t* cache_moves_count,
    const int32_t* evicted_logical_indices, const int32_t* evicted_kv_count,
    const int32_t* evicted_kv_offsets, const int32_t* block_tables,
    const int32_t* context_lens, int32_t num_seqs, int32_t num_layers, int32_t num_kv_heads,
    int32_t max_num_blocks_per_seq, int32_t block_size, int32_t zero_fill,
    kvc_stream_t stream) {
  if (block_size < 1) return kvc::fail_invalid("Unsupported block size: " + std::to_string(block_size));
  const int G = num_seqs * num_layers * num_kv_heads;
  if (G <= 0) return KVC_OK;
  hipStream_t s = (hipStream_t)stream;
  // extra workgroups clear the rows behind the last head's segment (wrapper zero fill)
  const int tail_wgs = zero_fill ? 64 : 0;
  const int64_t rows_per_head = cache_moves_rows / G;
#define KVC_LAUNCH_HEADS(T, W)                                                                 \
  hipLaunchKernelGGL((kvc::schedule_moves_heads_kernel<T, W>), dim3(G + tail_wgs), dim3(T), 0, s,     \
                     cache_moves_idx, cache_moves_rows, cache_moves_count,                      \
                     evicted_logical_indices, evicted_kv_count, evicted_kv_offsets,             \
                     block_tables, context_lens, num_seqs, num_layers, num_kv_heads,            \
                     max_num_blocks_per_seq, block_size, zero_fill)
  if (rows_per_head >= 8192) KVC_LAUNCH_HEADS(1024, kvc::MOVE_BITMAP_WORDS); else KVC_LAUNCH_HEADS(256, 1024);
#undef KVC_LAUNCH_HEADS
  return kvc::check_launch("schedule_t1_cache_moves");
}
