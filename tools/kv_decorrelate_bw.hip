// Is the 5.1 TB/s of a "tight" cache caused by the K and V images of one block being touched
// together at a fixed distance (the plane distance)?  (profiling aid)
// Pattern of compact_runs_kernel (read destination K+V, read source K+V, write destination K+V);
// "same": K and V of the SAME block ids (what the kernel does); "decor": the V images come from an
// independent permutation of the block ids (the same traffic, K/V addresses uncorrelated).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_(uint8_t* __restrict__ k, uint8_t* __restrict__ v, const int2* __restrict__ kruns,
                                          const int2* __restrict__ vruns, int nruns) {
  __shared__ __attribute__((aligned(16))) uint8_t lds_s[4][8192];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint8_t* lds = lds_s[wib];
  const int nw = gridDim.x * 4, wid = blockIdx.x * 4 + wib;
  const int r0 = (int)((int64_t)nruns * wid / nw), r1 = (int)((int64_t)nruns * (wid + 1) / nw);
  u32x4 kd[4], vd[4];
  for (int r = r0; r < r1; ++r) {
    const int2 kr = kruns[r], vr = vruns[r];
    uint8_t* kdp = k + (int64_t)kr.x * 4096; uint8_t* vdp = v + (int64_t)vr.x * 4096;
    const uint8_t* ksp = k + (int64_t)kr.y * 4096; const uint8_t* vsp = v + (int64_t)vr.y * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      kd[i] = __builtin_nontemporal_load((const u32x4*)(kdp + (i * 64 + lane) * 16));
      vd[i] = __builtin_nontemporal_load((const u32x4*)(vdp + (i * 64 + lane) * 16));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksp + (i * 64 + lane) * 16),
                                       (__attribute__((address_space(3))) void*)(lds + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsp + (i * 64 + lane) * 16),
                                       (__attribute__((address_space(3))) void*)(lds + 4096 + i * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4 ks = *(const u32x4*)(lds + i * 1024 + (lane ^ 1) * 16);
      const u32x4 vs = *(const u32x4*)(lds + 4096 + i * 1024 + (lane ^ 1) * 16);
      kd[i].x = ks.x; kd[i].z = ks.z; vd[i].y = vs.y; vd[i].w = vs.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_nontemporal_store(kd[i], (u32x4*)(kdp + (i * 64 + lane) * 16));
      __builtin_nontemporal_store(vd[i], (u32x4*)(vdp + (i * 64 + lane) * 16));
    }
  }
}

static float run(uint8_t* k, uint8_t* v, int2* a, int2* b, int nruns) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k_, dim3(512), dim3(256), 0, 0, k, v, a, b, nruns);
  (void)hipEventRecord(e0);
  for (int it = 0; it < 6; ++it) hipLaunchKernelGGL(k_, dim3(512), dim3(256), 0, 0, k, v, a, b, nruns);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 6;
}

int main(int argc, char** argv) {
  const int nruns = 262144;
  std::mt19937 rng(1);
  printf("[\n");
  bool first = true;
  for (int a = 1; a < argc; ++a) {
    const long NB = atol(argv[a]);
    uint8_t* buf;
    if (hipMalloc(&buf, (size_t)2 * NB * 4096) != hipSuccess) { fprintf(stderr, "alloc %ld failed\n", NB); continue; }
    (void)hipMemset(buf, 1, (size_t)2 * NB * 4096);
    uint8_t* k = buf; uint8_t* v = buf + (size_t)NB * 4096;
    const int used = (int)std::min<long>(NB, 524288 + 10000);
    std::vector<int> ids(used), ids2(used);
    for (int i = 0; i < used; ++i) ids[i] = ids2[i] = i;
    std::shuffle(ids.begin(), ids.end(), rng);
    std::shuffle(ids2.begin(), ids2.end(), rng);
    std::vector<int2> h(nruns), h2(nruns), h3(nruns);
    for (int i = 0; i < nruns; ++i) {
      h[i] = int2{ids[2 * i], ids[2 * i + 1]};
      h2[i] = int2{ids2[2 * i], ids2[2 * i + 1]};
      h3[i] = h[(i + 1) % nruns];                   // the next run's blocks (a software skew of one run)
    }
    int2 *r1, *r2, *r3;
    (void)hipMalloc(&r1, sizeof(int2) * nruns); (void)hipMalloc(&r2, sizeof(int2) * nruns); (void)hipMalloc(&r3, sizeof(int2) * nruns);
    (void)hipMemcpy(r1, h.data(), sizeof(int2) * nruns, hipMemcpyHostToDevice);
    (void)hipMemcpy(r2, h2.data(), sizeof(int2) * nruns, hipMemcpyHostToDevice);
    (void)hipMemcpy(r3, h3.data(), sizeof(int2) * nruns, hipMemcpyHostToDevice);
    const double bytes = 24576.0 * nruns;
    const float same = run(k, v, r1, r1, nruns), decor = run(k, v, r1, r2, nruns), skew = run(k, v, r1, r3, nruns);
    const float same2 = run(k, v, r1, r1, nruns);
    printf("%s {\"num_blocks\": %ld, \"plane_distance_GiB\": %.4f, \"same_GBps\": %.0f, \"decorrelated_GBps\": %.0f, \"skewed_by_one_run_GBps\": %.0f, \"same_again_GBps\": %.0f}",
           first ? "" : ",\n", NB, NB * 4096.0 / (1 << 30), bytes / same / 1e6, bytes / decor / 1e6, bytes / skew / 1e6, bytes / same2 / 1e6);
    first = false;
    (void)hipFree(buf); (void)hipFree(r1); (void)hipFree(r2); (void)hipFree(r3);
  }
  printf("\n]\n");
  return 0;
}
