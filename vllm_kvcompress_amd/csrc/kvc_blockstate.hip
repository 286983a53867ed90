// F2: the block-state side of a compression step, on device.
//
// After the moves are scheduled the reference frees, per head, the last
// `evicted_block_count` allocated blocks and shrinks the head's context length:
//   BlockSpaceManagerKVC.free_compressed_blocks     vllm/kvcompress/block_manager.py:466-530
//   BlockStateView.last_n_allocated_block_mask      vllm/kvcompress/block.py:367-379
//   BlockState.remove_trailing_blocks               vllm/kvcompress/block.py:184-210
//   ParallelBlockAllocator.free                     vllm/kvcompress/block_manager.py:112-118
//   CompressionMetrics.remove_metadata              vllm/kvcompress/metrics.py:366-370
// all of which are boolean-mask gathers over [L,B,H,M] in torch (each with a host sync).
// Here: one scan over the (layer, seq, head) counts, one pass that lists the freed blocks in
// the reference's order (layer, batch position, head, logical block ascending), marks them
// free, detaches their metadata and updates context_lens.
#include "kvc_common.h"
#include "../../include/kvc_mi355x.h"

namespace kvc {

// counts in (l, b, h) order -> workspace prefix (exclusive); nfree clipped to the head's blocks
__global__ __launch_bounds__(1024) void blockstate_scan_kernel(
    int32_t* __restrict__ prefix, const int32_t* __restrict__ freed_count_blh,
    const int32_t* __restrict__ context_lens, const int32_t* __restrict__ seq_slots, int L, int B,
    int S, int H, int bs) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  const int n = L * B * H;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    uint32_t v = 0;
    if (i < n) {
      const int l = i / (B * H), b = (i / H) % B, h = i % H;
      const int ctx = context_lens[((int64_t)l * S + seq_slots[b]) * H + h];
      const int nblk = (ctx + bs - 1) / bs;
      int f = freed_count_blh[((int64_t)b * L + l) * H + h];
      f = f < 0 ? 0 : (f > nblk ? nblk : f);
      v = (uint32_t)f;
    }
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) wave_tot[w] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < w; ++k) woff += wave_tot[k];
    const uint32_t carry = carry_s;
    if (i < n) prefix[i] = (int32_t)(carry + woff + inc - v);
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (tid == 0) prefix[n] = (int32_t)carry_s;
}

// one wave per (l, b, h)
__global__ __launch_bounds__(256) void blockstate_free_kernel(
    int32_t* __restrict__ context_lens, int32_t* __restrict__ seq_index_by_block,
    uint8_t* __restrict__ free_mask, int32_t* __restrict__ freed_blocks, int32_t freed_capacity,
    int32_t* __restrict__ freed_total, const int32_t* __restrict__ prefix,
    const int32_t* __restrict__ block_tables, const int32_t* __restrict__ freed_count_blh,
    const int32_t* __restrict__ seq_slots, int L, int B, int S, int H, int M, int bs) {
  const int n = L * B * H;
  const int i = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  if (i >= n) return;
  const int lane = lane_id();
  const int l = i / (B * H), b = (i / H) % B, h = i % H;
  const int64_t lsh = ((int64_t)l * S + seq_slots[b]) * H + h;
  const int ctx = context_lens[lsh];
  const int nblk = (ctx + bs - 1) / bs;
  const int nfree = prefix[i + 1] - prefix[i];
  if (i == 0 && lane == 0) *freed_total = prefix[n];
  if (nfree == 0) return;
  const int32_t* bt = block_tables + lsh * M;
  for (int j = lane; j < nfree; j += WAVE) {
    const int blk = bt[nblk - nfree + j];
    const int o = prefix[i] + j;
    if (o < freed_capacity) freed_blocks[o] = blk;
    seq_index_by_block[blk] = -1;                      // remove_metadata
    if (free_mask != nullptr) free_mask[blk] = 1;      // allocator.free
  }
  if (lane == 0) {
    // remove_trailing_blocks: ctx -= clamp(n*bs - (bs - hanging), 0)
    const int rem = ctx % bs;
    const int hang = rem == 0 ? bs : rem;
    int removed = nfree * bs - (bs - hang);
    removed = removed < 0 ? 0 : removed;
    context_lens[lsh] = ctx - removed;
  }
}

}  // namespace kvc

extern "C" size_t kvc_free_compressed_blocks_workspace_bytes(int32_t num_layers, int32_t batch,
                                                             int32_t num_kv_heads) {
  return ((size_t)num_layers * batch * num_kv_heads + 1) * sizeof(int32_t);
}

extern "C" int kvc_free_compressed_blocks(int32_t* context_lens, int32_t* seq_index_by_block,
                                          uint8_t* free_mask, int32_t* freed_blocks,
                                          int32_t freed_capacity, int32_t* freed_total,
                                          const int32_t* block_tables,
                                          const int32_t* freed_block_count,
                                          const int32_t* seq_slots, int32_t num_layers,
                                          int32_t batch, int32_t max_num_seqs,
                                          int32_t num_kv_heads, int32_t max_num_blocks_per_seq,
                                          int32_t block_size, void* workspace,
                                          size_t workspace_bytes, kvc_stream_t stream) {
  using namespace kvc;
  if (block_size < 1) return fail_invalid("Unsupported block size: " + std::to_string(block_size));
  const int n = num_layers * batch * num_kv_heads;
  if (n <= 0) return KVC_OK;
  if (workspace_bytes < kvc_free_compressed_blocks_workspace_bytes(num_layers, batch, num_kv_heads))
    return fail_invalid("free_compressed_blocks: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int32_t* prefix = reinterpret_cast<int32_t*>(workspace);
  hipLaunchKernelGGL(blockstate_scan_kernel, dim3(1), dim3(1024), 0, s, prefix, freed_block_count,
                     context_lens, seq_slots, num_layers, batch, max_num_seqs, num_kv_heads, block_size);
  hipLaunchKernelGGL(blockstate_free_kernel, dim3((n + 3) / 4), dim3(256), 0, s, context_lens,
                     seq_index_by_block, free_mask, freed_blocks, freed_capacity, freed_total, prefix,
                     block_tables, freed_block_count, seq_slots, num_layers, batch, max_num_seqs,
                     num_kv_heads, max_num_blocks_per_seq, block_size);
  return check_launch("free_compressed_blocks");
}
