// Plain device copy ceiling on MI355X for several kernel shapes (profiling aid):
// 16 B/lane, with/without non-temporal hints, 1/4/8 independent loads per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

template <int U, bool NT>
float run(const u32x4* s, u32x4* d, size_t n, int grid) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((copy_k<U, NT>), dim3(grid), dim3(256), 0, 0, s, d, n);
  hipEventRecord(a);
  for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((copy_k<U, NT>), dim3(grid), dim3(256), 0, 0, s, d, n);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 10;
}

int main() {
  const size_t bytes = 2ull << 30, n = bytes / 16;
  u32x4 *s, *d;
  (void)hipMalloc(&s, bytes); (void)hipMalloc(&d, bytes);
  (void)hipMemset(s, 1, bytes); (void)hipMemset(d, 2, bytes);
  for (int grid : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
    printf("grid %5d: U1 %.0f  U4 %.0f  U8 %.0f | NT U1 %.0f  U4 %.0f  U8 %.0f  GB/s (read+write)\n", grid,
           2.0 * bytes / run<1, false>(s, d, n, grid) / 1e6, 2.0 * bytes / run<4, false>(s, d, n, grid) / 1e6,
           2.0 * bytes / run<8, false>(s, d, n, grid) / 1e6, 2.0 * bytes / run<1, true>(s, d, n, grid) / 1e6,
           2.0 * bytes / run<4, true>(s, d, n, grid) / 1e6, 2.0 * bytes / run<8, true>(s, d, n, grid) / 1e6);
  }
  float ms;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  (void)hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0);
  hipEventRecord(a);
  for (int it = 0; it < 10; ++it) (void)hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0);
  hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
  printf("hipMemcpyAsync D2D: %.0f GB/s (read+write)\n", 2.0 * bytes / (ms / 10) / 1e6);
  return 0;
}
