// Does the distance between the K plane and the V plane of the unified cache [2, NB, block] matter?
// (profiling aid)  One allocation of 2 * NB images, v = k + NB * 4096 like the engine's cache.
// Patterns: "both" = per run read destination K+V, read source K+V, write destination K+V (what
// compact_runs_kernel does); "one plane" = the same traffic per image but a launch touches only
// the K plane (then a second launch the V plane).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool BOTH>
__global__ __launch_bounds__(256) void k_(uint8_t* __restrict__ k, uint8_t* __restrict__ v, const int2* __restrict__ runs, int nruns) {
  __shared__ __attribute__((aligned(16))) uint8_t lds_s[4][8192];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint8_t* lds = lds_s[wib];
  const int nw = gridDim.x * 4, wid = blockIdx.x * 4 + wib;
  const int r0 = (int)((int64_t)nruns * wid / nw), r1 = (int)((int64_t)nruns * (wid + 1) / nw);
  u32x4 kd[4], vd[4];
  for (int i = 0; i < 4; ++i) { kd[i] = u32x4{0, 0, 0, 0}; vd[i] = u32x4{0, 0, 0, 0}; }
  for (int r = r0; r < r1; ++r) {
    const int2 run = runs[r];
    uint8_t* kdp = k + (int64_t)run.x * 4096; uint8_t* vdp = v + (int64_t)run.x * 4096;
    const uint8_t* ksp = k + (int64_t)run.y * 4096; const uint8_t* vsp = v + (int64_t)run.y * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      kd[i] = __builtin_nontemporal_load((const u32x4*)(kdp + (i * 64 + lane) * 16));
      if (BOTH) vd[i] = __builtin_nontemporal_load((const u32x4*)(vdp + (i * 64 + lane) * 16));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksp + (i * 64 + lane) * 16),
                                       (__attribute__((address_space(3))) void*)(lds + i * 1024), 16, 0, 0);
      if (BOTH)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsp + (i * 64 + lane) * 16),
                                         (__attribute__((address_space(3))) void*)(lds + 4096 + i * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4 ks = *(const u32x4*)(lds + i * 1024 + (lane ^ 1) * 16);
      kd[i].x = ks.x; kd[i].z = ks.z;
      if (BOTH) { const u32x4 vs = *(const u32x4*)(lds + 4096 + i * 1024 + (lane ^ 1) * 16); vd[i].y = vs.y; vd[i].w = vs.w; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_nontemporal_store(kd[i], (u32x4*)(kdp + (i * 64 + lane) * 16));
      if (BOTH) __builtin_nontemporal_store(vd[i], (u32x4*)(vdp + (i * 64 + lane) * 16));
    }
  }
}

int main(int argc, char** argv) {
  const int nruns = 262144;
  std::vector<int> perm(1 << 20);
  std::mt19937 rng(1);
  printf("[\n");
  bool first = true;
  for (int a = 1; a < argc; ++a) {
    const long NB = atol(argv[a]);
    uint8_t* buf;
    if (hipMalloc(&buf, (size_t)2 * NB * 4096) != hipSuccess) { fprintf(stderr, "alloc %ld failed\n", NB); continue; }
    (void)hipMemset(buf, 1, (size_t)2 * NB * 4096);
    uint8_t* k = buf; uint8_t* v = buf + (size_t)NB * 4096;
    const int used = (int)std::min<long>(NB, 524288 + 10000);     // the sequence's blocks: shuffled ids below `used`
    std::vector<int> ids(used);
    for (int i = 0; i < used; ++i) ids[i] = i;
    std::shuffle(ids.begin(), ids.end(), rng);
    std::vector<int2> h(nruns);
    for (int i = 0; i < nruns; ++i) h[i] = int2{ids[2 * i], ids[2 * i + 1]};
    int2* runs; (void)hipMalloc(&runs, sizeof(int2) * nruns);
    (void)hipMemcpy(runs, h.data(), sizeof(int2) * nruns, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms_both, ms_split;
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k_<true>, dim3(512), dim3(256), 0, 0, k, v, runs, nruns);
    (void)hipEventRecord(e0);
    for (int it = 0; it < 6; ++it) hipLaunchKernelGGL(k_<true>, dim3(512), dim3(256), 0, 0, k, v, runs, nruns);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms_both, e0, e1); ms_both /= 6;
    for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(k_<false>, dim3(512), dim3(256), 0, 0, k, v, runs, nruns); hipLaunchKernelGGL(k_<false>, dim3(512), dim3(256), 0, 0, v, k, runs, nruns); }
    (void)hipEventRecord(e0);
    for (int it = 0; it < 6; ++it) { hipLaunchKernelGGL(k_<false>, dim3(512), dim3(256), 0, 0, k, v, runs, nruns); hipLaunchKernelGGL(k_<false>, dim3(512), dim3(256), 0, 0, v, k, runs, nruns); }
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms_split, e0, e1); ms_split /= 6;
    const double bytes = 24576.0 * nruns;
    printf("%s {\"num_blocks\": %ld, \"plane_distance_GiB\": %.4f, \"both_planes_GBps\": %.0f, \"one_plane_at_a_time_GBps\": %.0f}",
           first ? "" : ",\n", NB, NB * 4096.0 / (1 << 30), bytes / ms_both / 1e6, bytes / ms_split / 1e6);
    first = false;
    (void)hipFree(buf); (void)hipFree(runs);
  }
  printf("\n]\n");
  return 0;
}
