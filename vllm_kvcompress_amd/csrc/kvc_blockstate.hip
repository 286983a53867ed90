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

// ------------------------------------------------------------------------- append side
// One decode step appends one KV per head (token_count = 1):
//   BlockSpaceManagerKVC._append_to_sequence_batch  vllm/kvcompress/block_manager.py:269-294
//   ParallelBlockAllocator.allocate                 vllm/kvcompress/block_manager.py:103-110
//   BlockStateView.get_batch_new_block_metadata     vllm/kvcompress/block.py:513-620
//   CompressionMetrics.insert_metadata              vllm/kvcompress/metrics.py:344-361
// In torch: two boolean masks over [L,B,H,M], a masked gather of the free list, a masked scatter
// into the block tables, a metadata gather and five indexed stores, with a host sync for the
// count.  Here: a scan over the (layer, batch position, head) heads that sit on a block boundary,
// a tiled rank of the free list (the n lowest-numbered free blocks, like block_numbers[free_mask][:n]),
// and one pass that wires block r to the r-th such head, writes its metadata and position row
// and bumps every context length.
struct AppendWs {
  int32_t* prefix;      // [n + 1]   exclusive scan of "needs a new block" in (l, b, h) order
  int32_t* tile_off;    // [tiles + 1] exclusive scan of free blocks per tile
  int32_t* alloc;       // [n]       the blocks handed out, ascending
  int32_t* ok;          // [1]       1 = enough free blocks and every new logical block fits the table
};
constexpr int FREE_TILE = 4096;

__global__ __launch_bounds__(1024) void append_scan_kernel(AppendWs ws, const int32_t* __restrict__ context_lens,
                                                           const int32_t* __restrict__ seq_slots, int L, int B,
                                                           int S, int H, int M, int bs) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  __shared__ int fits_s;
  const int n = L * B * H;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) { carry_s = 0; fits_s = 1; }
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    uint32_t v = 0;
    if (i < n) {
      const int l = i / (B * H), b = (i / H) % B, h = i % H;
      const int ctx = context_lens[((int64_t)l * S + seq_slots[b]) * H + h];
      v = ctx % bs == 0 ? 1u : 0u;
      if (v && ctx / bs >= M) fits_s = 0;            // the block table has no room for it
    }
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) wave_tot[w] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < w; ++k) woff += wave_tot[k];
    const uint32_t carry = carry_s;
    if (i < n) ws.prefix[i] = (int32_t)(carry + woff + inc - v);
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (tid == 0) { ws.prefix[n] = (int32_t)carry_s; *ws.ok = fits_s; }
}

__global__ __launch_bounds__(256) void free_tile_count_kernel(AppendWs ws, const uint8_t* __restrict__ free_mask,
                                                              int64_t num_blocks) {
  __shared__ uint32_t red[4];
  const int64_t t0 = (int64_t)blockIdx.x * FREE_TILE;
  uint32_t c = 0;
  for (int k = threadIdx.x; k < FREE_TILE; k += 256) {
    const int64_t blk = t0 + k;
    c += (blk < num_blocks && free_mask[blk]) ? 1u : 0u;
  }
  c = wave_reduce_sum(c);
  if (lane_id() == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) ws.tile_off[blockIdx.x] = (int32_t)(red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(1024) void free_tile_scan_kernel(AppendWs ws, int tiles, int n_heads,
                                                              int32_t* __restrict__ status) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < tiles; base += 1024) {
    const int i = base + tid;
    const uint32_t v = i < tiles ? (uint32_t)ws.tile_off[i] : 0u;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) wave_tot[w] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < w; ++k) woff += wave_tot[k];
    const uint32_t carry = carry_s;
    if (i < tiles) ws.tile_off[i] = (int32_t)(carry + woff + inc - v);
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (tid == 0) {
    const int need = ws.prefix[n_heads];
    ws.tile_off[tiles] = (int32_t)carry_s;
    status[0] = need;
    status[1] = *ws.ok ? (int32_t)carry_s : -1;      // -1: a head's block table is full
    if ((int)carry_s < need) *ws.ok = 0;             // "Out of memory!" (block_manager.py:104-106)
  }
}

// the `need` lowest-numbered free blocks, ascending; they leave the free list
__global__ __launch_bounds__(256) void free_pick_kernel(AppendWs ws, uint8_t* __restrict__ free_mask,
                                                        int64_t num_blocks, int n_heads) {
  __shared__ uint32_t wave_cnt[4];
  if (*ws.ok == 0) return;
  const int need = ws.prefix[n_heads];
  uint32_t rank0 = (uint32_t)ws.tile_off[blockIdx.x];
  if ((int)rank0 >= need) return;
  const int64_t t0 = (int64_t)blockIdx.x * FREE_TILE;
  const int lane = lane_id(), w = threadIdx.x >> 6;
  for (int k0 = 0; k0 < FREE_TILE; k0 += 256) {
    const int64_t blk = t0 + k0 + threadIdx.x;
    const bool f = blk < num_blocks && free_mask[blk];
    const unsigned long long bal = __ballot(f);
    if (lane == 0) wave_cnt[w] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t off = rank0;
    for (int q = 0; q < w; ++q) off += wave_cnt[q];
    const uint32_t r = off + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    if (f && (int)r < need) { ws.alloc[r] = (int32_t)blk; free_mask[blk] = 0; }
    rank0 += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
}

// one thread per (l, b, h)
__global__ __launch_bounds__(256) void append_apply_kernel(
    AppendWs ws, int32_t* __restrict__ context_lens, int32_t* __restrict__ block_tables,
    int32_t* __restrict__ seq_index_by_block, int32_t* __restrict__ layer_index_by_block,
    int32_t* __restrict__ head_index_by_block, int32_t* __restrict__ logical_block_num_by_block,
    int32_t* __restrict__ token_positions, const int32_t* __restrict__ seq_slots,
    const int32_t* __restrict__ last_token_position, int L, int B, int S, int H, int M, int bs,
    int write_token_position) {
  if (*ws.ok == 0) return;                           // nothing is modified when the step cannot be done
  const int n = L * B * H;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int l = i / (B * H), b = (i / H) % B, h = i % H;
  const int slot = seq_slots[b];
  const int64_t lsh = ((int64_t)l * S + slot) * H + h;
  const int ctx = context_lens[lsh];
  const int m = ctx / bs;
  const int last = last_token_position[b];
  int blk;
  if (ctx % bs == 0) {                               // first token of a new block
    blk = ws.alloc[ws.prefix[i]];
    block_tables[lsh * M + m] = blk;
    seq_index_by_block[blk] = slot;                  // insert_metadata
    layer_index_by_block[blk] = l;
    head_index_by_block[blk] = h;
    logical_block_num_by_block[blk] = m;
    for (int o = 0; o < bs; ++o) token_positions[(int64_t)blk * bs + o] = last + o;
  } else {
    blk = block_tables[lsh * M + m];
  }
  if (write_token_position) token_positions[(int64_t)blk * bs + ctx % bs] = last;
  context_lens[lsh] = ctx + 1;
}

// ------------------------------------------------------------------------- prefill side
// A new sequence's first allocation:
//   BlockSpaceManagerKVC._add_sequence                vllm/kvcompress/block_manager.py:196-222
//   ParallelBlockAllocator.allocate                   vllm/kvcompress/block_manager.py:103-110
//   BlockStateView.get_allocated_block_metadata       vllm/kvcompress/block.py:414-446
//   CompressionMetrics.insert_metadata                vllm/kvcompress/metrics.py:344-361
//   BlockStateView.get_prefill_slot_mapping           vllm/kvcompress/block.py:275-303
// ceil(T / bs) blocks for every (layer, head): the L * H * cnt lowest-numbered free blocks (the tiled rank of the
// append side), block r to (l, h, j) = (r / (H cnt), r / cnt % H, r % cnt); context lengths, table rows, metadata
// rows, position rows (j * bs + o: positions are logical indices before any compression) and the prefill slot
// mapping [L, T, H] in one pass.
// the scan's first element holds the request (n_heads = 0 for free_tile_scan / free_pick): need, and whether the table fits
__global__ void prefill_request_kernel(AppendWs ws, int need, int fits) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { ws.prefix[0] = need; *ws.ok = fits; }
}

// one thread per allocated block
__global__ __launch_bounds__(256) void prefill_apply_kernel(
    AppendWs ws, int32_t* __restrict__ context_lens, int32_t* __restrict__ block_tables,
    int32_t* __restrict__ seq_index_by_block, int32_t* __restrict__ layer_index_by_block,
    int32_t* __restrict__ head_index_by_block, int32_t* __restrict__ logical_block_num_by_block,
    int32_t* __restrict__ token_positions, int64_t* __restrict__ slot_mapping, int L, int S, int H, int M, int bs,
    int slot, int seq_len, int cnt) {
  if (*ws.ok == 0) return;                           // nothing is modified when the sequence does not fit
  const int need = L * H * cnt;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= need) return;
  const int l = r / (H * cnt), h = (r / cnt) % H, j = r % cnt;
  const int blk = ws.alloc[r];
  const int64_t lsh = ((int64_t)l * S + slot) * H + h;
  block_tables[lsh * M + j] = blk;
  if (j == 0) context_lens[lsh] = seq_len;
  seq_index_by_block[blk] = slot;                    // insert_metadata
  layer_index_by_block[blk] = l;
  head_index_by_block[blk] = h;
  logical_block_num_by_block[blk] = j;
  for (int o = 0; o < bs; ++o) {
    const int t = j * bs + o;
    token_positions[(int64_t)blk * bs + o] = t;
    if (slot_mapping != nullptr && t < seq_len) slot_mapping[((int64_t)l * seq_len + t) * H + h] = (int64_t)blk * bs + o;
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

static size_t append_tiles(int64_t num_blocks) { return (size_t)((num_blocks + kvc::FREE_TILE - 1) / kvc::FREE_TILE); }

extern "C" size_t kvc_append_slots_workspace_bytes(int32_t num_layers, int32_t batch, int32_t num_kv_heads,
                                                   int64_t num_blocks) {
  const size_t n = (size_t)num_layers * batch * num_kv_heads;
  return ((n + 1) + (append_tiles(num_blocks) + 1) + n + 4) * sizeof(int32_t);
}

extern "C" int kvc_append_slots(int32_t* context_lens, int32_t* block_tables, uint8_t* free_mask,
                                int32_t* seq_index_by_block, int32_t* layer_index_by_block,
                                int32_t* head_index_by_block, int32_t* logical_block_num_by_block,
                                int32_t* token_positions, const int32_t* seq_slots,
                                const int32_t* last_token_position, int32_t* status, int32_t num_layers,
                                int32_t batch, int32_t max_num_seqs, int32_t num_kv_heads,
                                int32_t max_num_blocks_per_seq, int64_t num_blocks, int32_t block_size,
                                int32_t write_token_position, void* workspace, size_t workspace_bytes,
                                kvc_stream_t stream) {
  using namespace kvc;
  if (block_size < 1) return fail_invalid("Unsupported block size: " + std::to_string(block_size));
  const int n = num_layers * batch * num_kv_heads;
  if (n <= 0) return KVC_OK;
  if (num_blocks < 1) return fail_invalid("append_slots: no blocks");
  if (workspace_bytes < kvc_append_slots_workspace_bytes(num_layers, batch, num_kv_heads, num_blocks))
    return fail_invalid("append_slots: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int tiles = (int)append_tiles(num_blocks);
  AppendWs ws;
  ws.prefix = reinterpret_cast<int32_t*>(workspace);
  ws.tile_off = ws.prefix + n + 1;
  ws.alloc = ws.tile_off + tiles + 1;
  ws.ok = ws.alloc + n;
  hipLaunchKernelGGL(append_scan_kernel, dim3(1), dim3(1024), 0, s, ws, context_lens, seq_slots, num_layers,
                     batch, max_num_seqs, num_kv_heads, max_num_blocks_per_seq, block_size);
  hipLaunchKernelGGL(free_tile_count_kernel, dim3(tiles), dim3(256), 0, s, ws, free_mask, num_blocks);
  hipLaunchKernelGGL(free_tile_scan_kernel, dim3(1), dim3(1024), 0, s, ws, tiles, n, status);
  hipLaunchKernelGGL(free_pick_kernel, dim3(tiles), dim3(256), 0, s, ws, free_mask, num_blocks, n);
  hipLaunchKernelGGL(append_apply_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ws, context_lens, block_tables,
                     seq_index_by_block, layer_index_by_block, head_index_by_block, logical_block_num_by_block,
                     token_positions, seq_slots, last_token_position, num_layers, batch, max_num_seqs,
                     num_kv_heads, max_num_blocks_per_seq, block_size, write_token_position);
  return check_launch("append_slots");
}

extern "C" size_t kvc_add_sequence_workspace_bytes(int32_t num_layers, int32_t num_kv_heads, int32_t seq_len,
                                                   int32_t block_size, int64_t num_blocks) {
  if (block_size < 1 || seq_len < 0) return 0;
  const size_t need = (size_t)num_layers * num_kv_heads * (size_t)((seq_len + block_size - 1) / block_size);
  return (1 + (append_tiles(num_blocks) + 1) + need + 4) * sizeof(int32_t);
}

extern "C" int kvc_add_sequence(int32_t* context_lens, int32_t* block_tables, uint8_t* free_mask,
                                int32_t* seq_index_by_block, int32_t* layer_index_by_block,
                                int32_t* head_index_by_block, int32_t* logical_block_num_by_block,
                                int32_t* token_positions, int64_t* slot_mapping, int32_t* status,
                                int32_t num_layers, int32_t max_num_seqs, int32_t num_kv_heads,
                                int32_t max_num_blocks_per_seq, int64_t num_blocks, int32_t block_size,
                                int32_t seq_slot, int32_t seq_len, void* workspace, size_t workspace_bytes,
                                kvc_stream_t stream) {
  using namespace kvc;
  if (block_size < 1) return fail_invalid("Unsupported block size: " + std::to_string(block_size));
  if (num_layers < 1 || num_kv_heads < 1 || seq_len < 0 || seq_slot < 0 || seq_slot >= max_num_seqs)
    return fail_invalid("add_sequence: bad arguments");
  if (num_blocks < 1) return fail_invalid("add_sequence: no blocks");
  if (workspace_bytes < kvc_add_sequence_workspace_bytes(num_layers, num_kv_heads, seq_len, block_size, num_blocks))
    return fail_invalid("add_sequence: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int cnt = (seq_len + block_size - 1) / block_size;
  const int64_t need64 = (int64_t)num_layers * num_kv_heads * cnt;
  if (need64 > 0x7FFFFFFF) return fail_invalid("add_sequence: too many blocks");
  const int need = (int)need64;
  const int tiles = (int)append_tiles(num_blocks);
  AppendWs ws;
  ws.prefix = reinterpret_cast<int32_t*>(workspace);
  ws.tile_off = ws.prefix + 1;
  ws.alloc = ws.tile_off + tiles + 1;
  ws.ok = ws.alloc + need;
  hipLaunchKernelGGL(prefill_request_kernel, dim3(1), dim3(64), 0, s, ws, need, cnt <= max_num_blocks_per_seq ? 1 : 0);
  hipLaunchKernelGGL(free_tile_count_kernel, dim3(tiles), dim3(256), 0, s, ws, free_mask, num_blocks);
  hipLaunchKernelGGL(free_tile_scan_kernel, dim3(1), dim3(1024), 0, s, ws, tiles, 0, status);
  hipLaunchKernelGGL(free_pick_kernel, dim3(tiles), dim3(256), 0, s, ws, free_mask, num_blocks, 0);
  // (a sequence of no tokens allocates nothing but still takes its batch slot: context lengths 0)
  if (need > 0)
    hipLaunchKernelGGL(prefill_apply_kernel, dim3((need + 255) / 256), dim3(256), 0, s, ws, context_lens, block_tables,
                       seq_index_by_block, layer_index_by_block, head_index_by_block, logical_block_num_by_block,
                       token_positions, slot_mapping, num_layers, max_num_seqs, num_kv_heads, max_num_blocks_per_seq,
                       block_size, seq_slot, seq_len, cnt);
  return check_launch("add_sequence");
}
