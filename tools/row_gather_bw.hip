// How fast can randomly placed 64-byte rows (a bs-16 block's metric / position row) be gathered,
// and does the cache policy of the load change what the L2 fetches for them?  (profiling aid)
// Pattern of round 2's head_topk_kernel (replaced in round 3 by a physical-order stream): a lane loads 4 B, 16 lanes cover one row, 16 rows requested before
// the first is used.  Variants: plain, nt, sc1, sc0 sc1, sc0 sc1 nt (inline asm), and 128-byte
// rows (both halves of the line used) as the reference point.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <algorithm>

template <int MODE>
__device__ __forceinline__ float ld(const float* p) {
  float v;
  if constexpr (MODE == 0) asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (MODE == 1) asm volatile("global_load_dword %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (MODE == 2) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (MODE == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dword %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// ROWF = floats per row (16 = 64 B, 32 = 128 B); rows[]: row index of every gathered row
template <int MODE, int ROWF, int UB>
__global__ __launch_bounds__(256) void k_(const float* __restrict__ data, const int* __restrict__ rows, int nrows, float* out) {
  constexpr int LPR = ROWF;                   // lanes per row (4 B each)
  constexpr int RPW = 64 / LPR;               // rows per wave load
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  float acc = 0.f;
  for (int64_t r0 = wave * RPW * UB; r0 < nrows; r0 += nwaves * RPW * UB) {
    float v[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int64_t r = r0 + u * RPW + lane / LPR;
      const int row = r < nrows ? rows[r] : 0;
      v[u] = ld<MODE>(data + (int64_t)row * ROWF + lane % LPR);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < UB; ++u) acc += v[u];
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int MODE, int ROWF, int UB = 16>
static double run(const float* d, const int* rows, int nrows, float* out, int grid = 4096) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k_<MODE, ROWF, UB>), dim3(grid), dim3(256), 0, 0, d, rows, nrows, out);
  (void)hipEventRecord(e0);
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k_<MODE, ROWF, UB>), dim3(grid), dim3(256), 0, 0, d, rows, nrows, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return (double)nrows * ROWF * 4 / (ms / 5) / 1e6;        // useful GB/s
}

int main() {
  const int64_t total_rows64 = 1ll << 25;                   // 2 GiB of 64-byte rows
  const int nrows = 1 << 24;                                // gather half of them (distinct, random)
  float* d; (void)hipMalloc(&d, total_rows64 * 64); (void)hipMemset(d, 0, total_rows64 * 64);
  float* out; (void)hipMalloc(&out, 4);
  std::vector<int> ids(total_rows64);
  for (int64_t i = 0; i < total_rows64; ++i) ids[i] = (int)i;
  std::mt19937 rng(3);
  std::shuffle(ids.begin(), ids.end(), rng);
  int* rows; (void)hipMalloc(&rows, sizeof(int) * nrows);
  (void)hipMemcpy(rows, ids.data(), sizeof(int) * nrows, hipMemcpyHostToDevice);
  // 128-byte rows: indices below 2^24
  std::vector<int> ids2(1 << 24);
  for (int i = 0; i < (1 << 24); ++i) ids2[i] = i;
  std::shuffle(ids2.begin(), ids2.end(), rng);
  int* rows2; (void)hipMalloc(&rows2, sizeof(int) * (1 << 23));
  (void)hipMemcpy(rows2, ids2.data(), sizeof(int) * (1 << 23), hipMemcpyHostToDevice);
  printf("{\"rows_64B_useful_GBps\": {\"plain\": %.0f, \"nt\": %.0f, \"sc1\": %.0f, \"sc0_sc1\": %.0f, \"sc0_sc1_nt\": %.0f},\n",
         run<0, 16>(d, rows, nrows, out), run<1, 16>(d, rows, nrows, out), run<2, 16>(d, rows, nrows, out),
         run<3, 16>(d, rows, nrows, out), run<4, 16>(d, rows, nrows, out));
  printf(" \"rows_128B_useful_GBps\": {\"plain\": %.0f, \"nt\": %.0f, \"sc0_sc1\": %.0f}}\n",
         run<0, 32>(d, rows2, 1 << 23, out), run<1, 32>(d, rows2, 1 << 23, out), run<3, 32>(d, rows2, 1 << 23, out));
  printf("{\"rows_64B_nt_by_loads_in_flight\": {\"4\": %.0f, \"8\": %.0f, \"16\": %.0f, \"32\": %.0f},\n",
         run<1, 16, 4>(d, rows, nrows, out), run<1, 16, 8>(d, rows, nrows, out), run<1, 16, 16>(d, rows, nrows, out), run<1, 16, 32>(d, rows, nrows, out));
  printf(" \"rows_128B_nt_by_loads_in_flight\": {\"4\": %.0f, \"8\": %.0f, \"16\": %.0f, \"32\": %.0f},\n",
         run<1, 32, 4>(d, rows2, 1 << 23, out), run<1, 32, 8>(d, rows2, 1 << 23, out), run<1, 32, 16>(d, rows2, 1 << 23, out), run<1, 32, 32>(d, rows2, 1 << 23, out));
  printf(" \"rows_64B_nt_16_by_grid\": {\"1024\": %.0f, \"2048\": %.0f, \"8192\": %.0f, \"16384\": %.0f}}\n",
         run<1, 16, 16>(d, rows, nrows, out, 1024), run<1, 16, 16>(d, rows, nrows, out, 2048), run<1, 16, 16>(d, rows, nrows, out, 8192), run<1, 16, 16>(d, rows, nrows, out, 16384));
  return 0;
}
