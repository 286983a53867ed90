// A6 execute_cache_moves: paged K/V compaction for gfx950 (MI355X).
//
// Reference kernel (csrc/kvcompress_eviction_kernels.cu:359-435): 128 threads per
// (seq,layer), each thread walks whole KVs serially and copies 2-byte elements one at a
// time at a 32 B stride.  This file restates the job around the memory system instead:
//
//   K block  [hd*e/16 rows][bs slots][16 B]   -> a slot is hd*e/16 chunks of 16 B
//   V block  [hd rows][bs slots * e B]        -> a slot is ONE element in each of hd rows
//
// V is the hostile half: a token's 128 values sit in 128 different 32 B sectors, so any
// move touches every sector of both its source and destination block.  The fast path
// therefore works on whole V rows: the workgroup has one thread per V row; a thread
// loads its row of the destination block once (wide loads), patches in the elements of
// every move that targets that block from the matching row of the source block(s)
// (again wide loads; consecutive moves share source blocks), and writes the row back
// once.  The slot indices are wave-uniform, so the element extract/insert runs on
// scalar-selected registers, no LDS, no barriers.  K is copied as independent 16 B
// chunks with consecutive lanes on consecutive moves.
//
// Work distribution: a tile is KVC_TM consecutive moves of one head.  Two tiny planning
// kernels (per-head monotonicity check + exclusive scan of tile counts) run first so
// that a fixed persistent grid can walk the tiles without any host synchronisation
// (move counts live in device memory).  The row-wise V path needs all moves into one
// destination block to be handled by one workgroup: that holds when a head's dst slots
// are strictly ascending (always true for schedules produced by A5); a head that fails
// the check falls back to element-wise copies, which are correct for any independent
// move list.
#include "kvc_common.h"
#include "../../include/kvc_mi355x.h"

namespace kvc {

constexpr int KVC_TM = 32;            // moves per tile

// ------------------------------------------------------------------------- planning
// one wave per head: flag[g] = dst strictly ascending, tiles[g] = ceil(cnt/TM)
__global__ __launch_bounds__(256) void compact_plan_heads_kernel(
    int32_t* __restrict__ flags, int32_t* __restrict__ tiles,
    const int32_t* __restrict__ moves, const int32_t* __restrict__ count,
    const int32_t* __restrict__ offs, int G) {
  const int g = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  if (g >= G) return;
  const int lane = lane_id();
  const int cnt = count[g];
  const int2* mv = reinterpret_cast<const int2*>(moves) + offs[g];
  bool ok = true;
  for (int j = lane; j + 1 < cnt; j += WAVE) ok &= mv[j].x < mv[j + 1].x;
  const bool all_ok = __all(ok);
  if (lane == 0) {
    flags[g] = all_ok ? 1 : 0;
    tiles[g] = (cnt + KVC_TM - 1) / KVC_TM;
  }
}

// single workgroup: in-place exclusive scan of tiles[0..G) -> prefix[0..G]
__global__ __launch_bounds__(1024) void compact_plan_scan_kernel(int32_t* __restrict__ tiles, int G) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = lane_id(), w = tid / WAVE;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < G; base += 1024) {
    const int i = base + tid;
    const uint32_t v = i < G ? (uint32_t)tiles[i] : 0u;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) wave_tot[w] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < w; ++k) woff += wave_tot[k];
    const uint32_t carry = carry_s;
    if (i < G) tiles[i] = (int32_t)(carry + woff + inc - v);
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (tid == 0) tiles[G] = (int32_t)carry_s;
}

// ------------------------------------------------------------------------- fast path
template <int NW>
struct Row { uint32_t w[NW]; };

template <int NW>
__device__ __forceinline__ Row<NW> load_row(const uint8_t* p) {
  Row<NW> r;
  if constexpr (NW % 4 == 0) {
#pragma unroll
    for (int i = 0; i < NW / 4; ++i) {
      const uint4 q = reinterpret_cast<const uint4*>(p)[i];
      r.w[4 * i] = q.x; r.w[4 * i + 1] = q.y; r.w[4 * i + 2] = q.z; r.w[4 * i + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = reinterpret_cast<const uint32_t*>(p)[i];
  }
  return r;
}

template <int NW>
__device__ __forceinline__ void store_row(uint8_t* p, const Row<NW>& r) {
  if constexpr (NW % 4 == 0) {
#pragma unroll
    for (int i = 0; i < NW / 4; ++i)
      reinterpret_cast<uint4*>(p)[i] = make_uint4(r.w[4 * i], r.w[4 * i + 1], r.w[4 * i + 2], r.w[4 * i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < NW; ++i) reinterpret_cast<uint32_t*>(p)[i] = r.w[i];
  }
}

// element `slot` (wave-uniform) of a row held in registers; E = element bytes
template <int NW, int E>
__device__ __forceinline__ uint32_t row_extract(const Row<NW>& r, int slot) {
  constexpr int PER = 4 / E;                   // elements per dword
  const int wi = slot / PER;
  uint32_t w = r.w[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) w = (wi == i) ? r.w[i] : w;   // scalar-conditioned selects
  if constexpr (E == 4) return w;
  const int sh = (slot % PER) * (8 * E);
  return (w >> sh) & ((1u << (8 * E)) - 1u);
}

template <int NW, int E>
__device__ __forceinline__ void row_insert(Row<NW>& r, int slot, uint32_t val) {
  constexpr int PER = 4 / E;
  const int wi = slot / PER;
  if constexpr (E == 4) {
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = (wi == i) ? val : r.w[i];
  } else {
    const int sh = (slot % PER) * (8 * E);
    const uint32_t mask = ((1u << (8 * E)) - 1u) << sh;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = (wi == i) ? ((r.w[i] & ~mask) | (val << sh)) : r.w[i];
  }
}

// HD = head size = V rows per block = threads per workgroup; BS = block size; E = elem bytes
template <int HD, int BS, int E>
__global__ __launch_bounds__(HD) void compact_rows_kernel(
    uint8_t* __restrict__ k_cache, uint8_t* __restrict__ v_cache, float* __restrict__ metrics,
    int32_t* __restrict__ positions, const int32_t* __restrict__ moves,
    const int32_t* __restrict__ count, const int32_t* __restrict__ offs,
    const int32_t* __restrict__ flags, const int32_t* __restrict__ tile_prefix, int G) {
  constexpr int RB = BS * E;                   // bytes per V row
  constexpr int NW = RB / 4;                   // dwords per V row
  constexpr int KR = HD * E / 16;              // 16 B chunk rows per K block
  constexpr int64_t BLOCK_BYTES = (int64_t)HD * BS * E;
  const int tid = threadIdx.x;
  const int total_tiles = tile_prefix[G];
  for (int t = blockIdx.x; t < total_tiles; t += gridDim.x) {
    const int g = upper_bound_minus1(tile_prefix, G, t);
    const int cnt = count[g];
    const int j0 = (t - tile_prefix[g]) * KVC_TM;
    const int j1 = min(cnt, j0 + KVC_TM);
    const int2* __restrict__ mv = reinterpret_cast<const int2*>(moves) + offs[g];
    const bool rows_ok = flags[g] != 0;

    // ---- metrics + positions: one lane per move ---------------------------------
    for (int j = j0 + tid; j < j1; j += HD) {
      const int2 m = mv[j];
      metrics[m.x] = metrics[m.y];
      positions[m.x] = positions[m.y];
    }
    // ---- K: independent 16 B chunks, consecutive lanes = consecutive moves ------
    for (int idx = tid; idx < KVC_TM * KR; idx += HD) {
      const int m = idx % KVC_TM, r = idx / KVC_TM;
      if (j0 + m < j1) {
        const int2 mm = mv[j0 + m];
        const int64_t so = ((int64_t)(mm.y / BS) * KR + r) * (BS * 16) + (mm.y % BS) * 16;
        const int64_t dof = ((int64_t)(mm.x / BS) * KR + r) * (BS * 16) + (mm.x % BS) * 16;
        *reinterpret_cast<uint4*>(k_cache + dof) = *reinterpret_cast<const uint4*>(k_cache + so);
      }
    }
    // ---- V ------------------------------------------------------------------------
    if (rows_ok) {
      // a run = the moves into one destination block; it belongs to the tile that holds
      // its first move.  All indices below are wave-uniform.
      for (int j = j0; j < j1; ++j) {
        const int dslot = mv[j].x;
        const int dblk = dslot / BS;
        if (j > 0 && mv[j - 1].x / BS == dblk) continue;          // not a run start
        uint8_t* drow_p = v_cache + (int64_t)dblk * BLOCK_BYTES + (int64_t)tid * RB;
        Row<NW> drow = load_row<NW>(drow_p);
        int cur_sblk = -1;
        Row<NW> srow;
        for (int jj = j; jj < cnt; ++jj) {
          const int2 m = mv[jj];
          if (m.x / BS != dblk) break;
          const int sblk = m.y / BS;
          if (sblk != cur_sblk) {
            srow = load_row<NW>(v_cache + (int64_t)sblk * BLOCK_BYTES + (int64_t)tid * RB);
            cur_sblk = sblk;
          }
          row_insert<NW, E>(drow, m.x % BS, row_extract<NW, E>(srow, m.y % BS));
        }
        store_row<NW>(drow_p, drow);
      }
    } else {
      // element-wise fallback: correct for any independent move list
      for (int j = j0; j < j1; ++j) {
        const int2 m = mv[j];
        const int64_t so = (int64_t)(m.y / BS) * BLOCK_BYTES + (int64_t)tid * RB + (m.y % BS) * E;
        const int64_t dof = (int64_t)(m.x / BS) * BLOCK_BYTES + (int64_t)tid * RB + (m.x % BS) * E;
        if constexpr (E == 1) v_cache[dof] = v_cache[so];
        else if constexpr (E == 2) *reinterpret_cast<uint16_t*>(v_cache + dof) = *reinterpret_cast<const uint16_t*>(v_cache + so);
        else *reinterpret_cast<uint32_t*>(v_cache + dof) = *reinterpret_cast<const uint32_t*>(v_cache + so);
      }
    }
  }
}

// ------------------------------------------------------------------------- generic path
// any block_size / head_size / element size / K vector width x: byte-granular copies,
// one thread per byte.
__global__ __launch_bounds__(256) void compact_generic_kernel(
    uint8_t* __restrict__ k_cache, uint8_t* __restrict__ v_cache, float* __restrict__ metrics,
    int32_t* __restrict__ positions, const int32_t* __restrict__ moves,
    const int32_t* __restrict__ count, const int32_t* __restrict__ offs,
    const int32_t* __restrict__ tile_prefix, int G, int bs, int hd, int e, int x) {
  const int tid = threadIdx.x;
  const int total_tiles = tile_prefix[G];
  const int kgroups = hd / x;                 // K vectors per slot
  const int kvec_bytes = x * e;
  const int64_t block_bytes = (int64_t)hd * bs * e;
  for (int t = blockIdx.x; t < total_tiles; t += gridDim.x) {
    const int g = upper_bound_minus1(tile_prefix, G, t);
    const int cnt = count[g];
    const int j0 = (t - tile_prefix[g]) * KVC_TM;
    const int j1 = min(cnt, j0 + KVC_TM);
    const int2* __restrict__ mv = reinterpret_cast<const int2*>(moves) + offs[g];
    for (int j = j0 + tid; j < j1; j += blockDim.x) {
      const int2 m = mv[j];
      metrics[m.x] = metrics[m.y];
      positions[m.x] = positions[m.y];
    }
    const int per_move = kgroups * kvec_bytes + hd * e;      // bytes per move (K then V)
    const int n = (j1 - j0) * per_move;
    for (int idx = tid; idx < n; idx += blockDim.x) {
      const int2 m = mv[j0 + idx / per_move];
      const int bsel = idx % per_move;
      const int sb = m.y / bs, so = m.y % bs, db = m.x / bs, dof = m.x % bs;
      if (bsel < kgroups * kvec_bytes) {
        const int r = bsel / kvec_bytes, byte = bsel % kvec_bytes;
        k_cache[(int64_t)db * block_bytes + ((int64_t)r * bs + dof) * kvec_bytes + byte] =
            k_cache[(int64_t)sb * block_bytes + ((int64_t)r * bs + so) * kvec_bytes + byte];
      } else {
        const int vb = bsel - kgroups * kvec_bytes;
        const int d = vb / e, byte = vb % e;
        v_cache[(int64_t)db * block_bytes + ((int64_t)d * bs + dof) * e + byte] =
            v_cache[(int64_t)sb * block_bytes + ((int64_t)d * bs + so) * e + byte];
      }
    }
  }
}

}  // namespace kvc

extern "C" size_t kvc_execute_cache_moves_workspace_bytes(int32_t total_heads) {
  return (size_t)(2 * (int64_t)total_heads + 2) * sizeof(int32_t);
}

extern "C" int kvc_execute_cache_moves(void* k_cache, void* v_cache, float* kv_metrics,
                                       int32_t* kv_position, const int32_t* cache_moves_idx,
                                       const int32_t* cache_moves_count,
                                       const int32_t* evicted_kv_offsets, int32_t total_heads,
                                       int64_t num_blocks, int32_t block_size,
                                       int32_t head_size, int32_t elem_bytes, int32_t vec_size,
                                       void* workspace, size_t workspace_bytes,
                                       kvc_stream_t stream) {
  using namespace kvc;
  (void)num_blocks;
  if (block_size < 1) return fail_invalid("Unsupported block size: " + std::to_string(block_size));
  if (head_size < 1) return fail_invalid("Unsupported head size: " + std::to_string(head_size));
  if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4)
    return fail_invalid("Unsupported cache element size: " + std::to_string(elem_bytes));
  const int x = vec_size;
  if (x < 1 || head_size % x != 0)
    return fail_invalid("Unsupported vec size: " + std::to_string(vec_size));
  if (total_heads <= 0) return KVC_OK;
  if (workspace_bytes < kvc_execute_cache_moves_workspace_bytes(total_heads))
    return fail_invalid("execute_cache_moves: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int32_t* flags = reinterpret_cast<int32_t*>(workspace);
  int32_t* prefix = flags + total_heads;
  const int G = total_heads;
  hipLaunchKernelGGL(compact_plan_heads_kernel, dim3((G + 3) / 4), dim3(256), 0, s, flags, prefix,
                     cache_moves_idx, cache_moves_count, evicted_kv_offsets, G);
  hipLaunchKernelGGL(compact_plan_scan_kernel, dim3(1), dim3(1024), 0, s, prefix, G);
  uint8_t* k = reinterpret_cast<uint8_t*>(k_cache);
  uint8_t* v = reinterpret_cast<uint8_t*>(v_cache);
  const int grid = 256 * 8;     // 256 CUs x 8 resident workgroups, persistent over tiles
#define KVC_ROWS(HD, BS, E)                                                                      \
  hipLaunchKernelGGL((compact_rows_kernel<HD, BS, E>), dim3(grid), dim3(HD), 0, s, k, v,         \
                     kv_metrics, kv_position, cache_moves_idx, cache_moves_count,                \
                     evicted_kv_offsets, flags, prefix, G)
  bool fast = true;
  if (x * elem_bytes != 16) fast = false;   // fast path copies K as 16 B chunks
  else if (head_size == 128 && block_size == 16 && elem_bytes == 2) KVC_ROWS(128, 16, 2);
  else if (head_size == 128 && block_size == 32 && elem_bytes == 1) KVC_ROWS(128, 32, 1);
  else if (head_size == 128 && block_size == 32 && elem_bytes == 2) KVC_ROWS(128, 32, 2);
  else if (head_size == 128 && block_size == 16 && elem_bytes == 1) KVC_ROWS(128, 16, 1);
  else if (head_size == 128 && block_size == 16 && elem_bytes == 4) KVC_ROWS(128, 16, 4);
  else if (head_size == 64 && block_size == 16 && elem_bytes == 2) KVC_ROWS(64, 16, 2);
  else if (head_size == 256 && block_size == 16 && elem_bytes == 2) KVC_ROWS(256, 16, 2);
  else fast = false;
#undef KVC_ROWS
  if (!fast) {
    hipLaunchKernelGGL(compact_generic_kernel, dim3(grid), dim3(256), 0, s, k, v, kv_metrics,
                       kv_position, cache_moves_idx, cache_moves_count, evicted_kv_offsets, prefix,
                       G, block_size, head_size, elem_bytes, x);
  }
  return check_launch("execute_cache_moves");
}
