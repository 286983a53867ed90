// Microbenchmark: bandwidth of copying randomly placed chunks (gather / scatter / both)
// on MI355X as a function of chunk size.  Profiling aid for DESIGN.md, not product code.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bw.hip -o tools/bin/gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x, uint32_t mask, int bits) {   // bijection on [0,2^bits)
  x = (x * 0x9E3779B1u) & mask; x ^= x >> (bits / 2 + 1); x = (x * 0x85EBCA6Bu) & mask;
  x ^= x >> (bits / 2); x = (x * 0xC2B2AE35u) & mask; x ^= x >> (bits / 2 + 2);
  return x & mask;
}

// mode bit0: random source, bit1: random destination
__global__ __launch_bounds__(256) void copy_chunks(const u32x4* __restrict__ src, u32x4* __restrict__ dst,
                                                   uint32_t nchunks, int lanes_per_chunk, int bits, int mode) {
  const uint32_t mask = (1u << bits) - 1u;
  const uint64_t total = (uint64_t)nchunks * lanes_per_chunk;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t c = (uint32_t)(i / lanes_per_chunk), l = (uint32_t)(i % lanes_per_chunk);
    const uint32_t sc = (mode & 1) ? mix(c, mask, bits) : c;
    const uint32_t dc = (mode & 2) ? mix(c ^ 0x5bd1e995u & mask, mask, bits) : c;
    dst[(uint64_t)dc * lanes_per_chunk + l] = src[(uint64_t)sc * lanes_per_chunk + l];
  }
}

int main() {
  const size_t total = 2ull << 30;   // bytes per buffer
  u32x4 *src, *dst;
  hipMalloc(&src, total); hipMalloc(&dst, total);
  hipMemset(src, 1, total); hipMemset(dst, 2, total);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  printf("{\n");
  const int sizes[] = {16, 32, 64, 128, 256, 1024, 4096, 16384};
  for (int si = 0; si < 8; ++si) {
    const int chunk = sizes[si];
    const uint32_t n = (uint32_t)(total / chunk);
    int bits = 0; while ((1u << bits) < n) ++bits;
    printf(" \"%d\": {", chunk);
    for (int mode = 0; mode < 4; ++mode) {
      for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(copy_chunks, dim3(256 * 16), dim3(256), 0, 0, src, dst, n, chunk / 16, bits, mode);
      hipEventRecord(a);
      const int iters = 5;
      for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(copy_chunks, dim3(256 * 16), dim3(256), 0, 0, src, dst, n, chunk / 16, bits, mode);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); ms /= iters;
      const char* names[] = {"seq", "gather", "scatter", "both"};
      printf("\"%s_GBps_rw\": %.0f%s", names[mode], 2.0 * total / ms / 1e6, mode < 3 ? ", " : "");
    }
    printf("}%s\n", si < 7 ? "," : "");
  }
  printf("}\n");
  return 0;
}
