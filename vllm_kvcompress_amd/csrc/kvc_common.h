// Shared helpers for the gfx950 (CDNA4 / MI355X) KV-Compress kernels.
// Wave = 64 lanes everywhere in this code base; nothing here is portable to 32-wide
// hardware and nothing tries to be.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#define KVC_OK 0
#define KVC_ERR_INVALID 1   // bad argument / unsupported shape  (-> RuntimeError in Python)
#define KVC_ERR_HIP 2       // HIP runtime failure

namespace kvc {

void set_error(const std::string& msg);   // defined in kvc_api.hip (thread local)

inline int fail_invalid(const std::string& msg) {
  set_error(msg);
  return KVC_ERR_INVALID;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return KVC_ERR_HIP;
  }
  return KVC_OK;
}

// kvc_aggregate.hip: max_pool1d(7) over keys + accumulate, out_kh[K,Hq] += pool(colsum[Hq,K])
int launch_epilogue_pool(float* out_kh, const float* colsum, int num_q_heads, int num_keys,
                         int use_maxpool, hipStream_t s);

constexpr int WAVE = 64;

// Order-preserving map float32 -> uint32 (ascending float order == ascending key order).
// -0.0 is canonicalised to +0.0 (torch.sort treats them as equal) and NaNs to the positive
// quiet NaN so they sort after +inf like torch.sort does.
__device__ __forceinline__ uint32_t float_to_key(float v) {
  if (v != v) return 0xFFC00000u;          // NaN -> above +inf
  v = v + 0.0f;                            // -0 -> +0
  uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
constexpr uint32_t KEY_INF = 0xFF800000u;  // float_to_key(+inf)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// wave-wide inclusive scan (64 lanes) with shuffles
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t n = __shfl_up(v, d, 64);
    if (lane_id() >= d) v += n;
  }
  return v;
}

__device__ __forceinline__ uint32_t wave_reduce_sum(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// The same two for a wave whose 64 lanes are ALL active: DPP row operations (the adds themselves permute: no trip
// through the LDS crossbar, no address arithmetic) and, for the sum, four scalar reads of the rows' totals.  A lane
// that is switched off would neither be added nor hold a row's total: only where control flow is wave-uniform.
//   quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror 0x141, row_mirror 0x140, row_shr:n 0x110 + n,
//   row_bcast:15 0x142 (rows 1, 3: row_mask 0xA), row_bcast:31 0x143 (rows 2, 3: row_mask 0xC)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_reduce_sum_full(uint32_t v) {
  v += dpp_or_zero<0xB1>(v);
  v += dpp_or_zero<0x4E>(v);
  v += dpp_or_zero<0x141>(v);
  v += dpp_or_zero<0x140>(v);
  return (uint32_t)(__builtin_amdgcn_readlane((int)v, 0) + __builtin_amdgcn_readlane((int)v, 16) +
                    __builtin_amdgcn_readlane((int)v, 32) + __builtin_amdgcn_readlane((int)v, 48));
}
__device__ __forceinline__ uint32_t wave_inclusive_scan_full(uint32_t v) {
  v += dpp_or_zero<0x111>(v);
  v += dpp_or_zero<0x112>(v);
  v += dpp_or_zero<0x114>(v);
  v += dpp_or_zero<0x118>(v);
  v += dpp_or_zero<0x142, 0xA>(v);
  v += dpp_or_zero<0x143, 0xC>(v);
  return v;
}

// last index g in [0,n) with a[g] <= x  (a ascending, a[0] <= x assumed)
__device__ __forceinline__ int upper_bound_minus1(const int32_t* __restrict__ a, int n, int64_t x) {
  int lo = 0, hi = n;                      // invariant: a[lo] <= x, (hi==n or a[hi] > x)
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if ((int64_t)a[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}

// The same by a whole wave (x wave-uniform, every lane active): 64 probes per step instead of one -- two dependent
// loads for 4096 entries where the bisection takes twelve (persistent kernels find their first head this way: the
// chain of dependent loads was a third of what a workgroup of count_collect_kernel took).
__device__ __forceinline__ int wave_upper_bound_minus1(const int32_t* __restrict__ a, int n, int64_t x) {
  const int lane = (int)(threadIdx.x & 63u);
  int lo = 0, len = n;                     // the answer lies in [lo, lo + len); a[lo] <= x
  while (len > 64) {
    const int step = (len + 63) / 64;
    const int npieces = (len + step - 1) / step;
    const bool le = lane < npieces && (int64_t)a[lo + lane * step] <= x;     // pieces that start at or below x: a prefix
    const int piece = __popcll(__ballot(le)) - 1;
    const int end = lo + len;
    lo += (piece < 0 ? 0 : piece) * step;
    len = min(step, end - lo);
  }
  const bool le = lane < len && (int64_t)a[lo + lane] <= x;
  const int c = __popcll(__ballot(le));
  return lo + (c > 0 ? c - 1 : 0);
}

// x -> (x / d, x % d) for a divisor that is a kernel argument (block sizes, slots per block): a 32-bit division by a
// run-time value is ~35 VALU instructions.  One division per thread instead: m = floor((2^32 - 1) / bs); for 0 <= x < 2^31 the
// quotient mulhi(x, m) is x / bs or one less (x (1 / bs - m / 2^32) < 1), which the remainder tells.  Any block size.
struct BlockDiv {
  uint32_t bs, m;
  __device__ __forceinline__ explicit BlockDiv(int b) : bs((uint32_t)b), m(0xFFFFFFFFu / (uint32_t)b) {}
  __device__ __forceinline__ void divmod(int x, int& q, int& r) const {
    uint32_t qq = __umulhi((uint32_t)x, m);
    uint32_t rr = (uint32_t)x - qq * bs;
    if (rr >= bs) { ++qq; rr -= bs; }
    q = (int)qq; r = (int)rr;
  }
  __device__ __forceinline__ int blk(int x) const { int q, r; divmod(x, q, r); return q; }
  __device__ __forceinline__ int off(int x) const { int q, r; divmod(x, q, r); return r; }
  // physical slot of logical slot x through the head's block table
  __device__ __forceinline__ int phys(const int32_t* bt, int x) const { int q, r; divmod(x, q, r); return bt[q] * (int)bs + r; }
};

// A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global load and store the wave
// has in flight (s_waitcnt vmcnt(0) in front of s_barrier): in a loop whose steps exchange a few words through LDS and
// then walk a chain of dependent global loads, every step then pays the whole chain's latency before the next one may
// even request its loads (schedule_moves_heads_kernel: 16 steps x ~2 us).  Only for barriers that publish LDS data.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// order LDS traffic between the lanes of one wave (no other wave shares the buffer)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A move list's PLAN: how many move tiles the heads of every workgroup of schedule_t1_cache_moves
// hold -- all execute_cache_moves needs to spread the tiles over its waves (a wave finds its first
// tile with three wave-wide scans: the workgroups' sums, one workgroup's heads, done).  At most
// MOVES_PLAN_WGS workgroups; two tile sizes side by side (the compaction's block path takes
// 64 - bs moves per tile, its byte-wise path 32) because the move scheduler does not know the cache's shape.
constexpr int MOVES_PLAN_WGS = 4096;
inline __host__ __device__ void moves_plan_shape(int total_heads, int& nwg, int& heads_per_wg) {
  heads_per_wg = (total_heads + MOVES_PLAN_WGS - 1) / MOVES_PLAN_WGS;
  if (heads_per_wg < 1) heads_per_wg = 1;
  nwg = (total_heads + heads_per_wg - 1) / heads_per_wg;
}

}  // namespace kvc
