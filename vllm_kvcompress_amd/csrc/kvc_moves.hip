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

template <int BS>
__global__ __launch_bounds__(256) void schedule_moves_rows_kernel(
    int32_t* __restrict__ moves, int64_t rows, const int32_t* __restrict__ evicted,
    const int32_t* __restrict__ ekc, const int32_t* __restrict__ offs,
    const int32_t* __restrict__ block_tables, const int32_t* __restrict__ context_lens,
    int B, int L, int H, int M, int bs_rt, int zero_fill) {
  const int bs = BS > 0 ? BS : bs_rt;
  const int G = B * L * H;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  int g = upper_bound_minus1(offs, G, r);
  const int j = (int)(r - offs[g]);
  int2 out = make_int2(0, 0);
  bool have = false;
  const int cnt = ekc[g];
  if (j < cnt) {
    const int b = g / (L * H), lh = g % (L * H), l = lh / H, hh = lh % H;
    const int lbh = (l * B + b) * H + hh;
    MoveHead h{evicted + offs[g], cnt, context_lens[lbh]};
    const int i = move_iteration(h, j);
    if (i >= 0) {
      const int src = h.ctx - 1 - i;
      const int dst = h.E[j];
      const int32_t* bt = block_tables + (int64_t)lbh * M;
      out.x = bt[dst / bs] * bs + dst % bs;
      out.y = bt[src / bs] * bs + src % bs;
      have = true;
    }
  }
  if (have || zero_fill) reinterpret_cast<int2*>(moves)[r] = out;
}

// one thread per head: number of emitted moves (validity is monotone in j -> bisection)
__global__ __launch_bounds__(256) void schedule_moves_count_kernel(
    int32_t* __restrict__ count, const int32_t* __restrict__ evicted,
    const int32_t* __restrict__ ekc, const int32_t* __restrict__ offs,
    const int32_t* __restrict__ context_lens, int B, int L, int H) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * L * H) return;
  const int b = g / (L * H), lh = g % (L * H), l = lh / H, hh = lh % H;
  const int lbh = (l * B + b) * H + hh;
  MoveHead h{evicted + offs[g], ekc[g], context_lens[lbh]};
  int lo = 0, hi = h.cnt;                  // first j that is NOT a move
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (move_iteration(h, mid) >= 0) lo = mid + 1; else hi = mid;
  }
  count[g] = lo;
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
  if (cache_moves_rows > 0) {
    dim3 block(256), grid((unsigned)((cache_moves_rows + 255) / 256));
#define KVC_LAUNCH_ROWS(BS)                                                                     \
  hipLaunchKernelGGL(kvc::schedule_moves_rows_kernel<BS>, grid, block, 0, s, cache_moves_idx,   \
                     cache_moves_rows, evicted_logical_indices, evicted_kv_count,               \
                     evicted_kv_offsets, block_tables, context_lens, num_seqs, num_layers,      \
                     num_kv_heads, max_num_blocks_per_seq, block_size, zero_fill)
    switch (block_size) {
      case 16: KVC_LAUNCH_ROWS(16); break;
      case 32: KVC_LAUNCH_ROWS(32); break;
      default: KVC_LAUNCH_ROWS(0); break;
    }
#undef KVC_LAUNCH_ROWS
    int rc = kvc::check_launch("schedule_t1_cache_moves(rows)");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(kvc::schedule_moves_count_kernel, dim3((G + 255) / 256), dim3(256), 0, s,
                     cache_moves_count, evicted_logical_indices, evicted_kv_count,
                     evicted_kv_offsets, context_lens, num_seqs, num_layers, num_kv_heads);
  return kvc::check_launch("schedule_t1_cache_moves(count)");
}
