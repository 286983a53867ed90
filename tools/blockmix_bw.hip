// Ceiling of the compaction kernel's ACCESS PATTERN on MI355X (profiling aid, no patch logic):
// randomly placed 4 KiB K + 4 KiB V block images, per "run" read the destination images (registers),
// read the source images (LDS via global_load_lds, or registers) and write the destination images
// back.  Sweeps resident waves per CU, the read:write mix (2:1 = random evictions, 1:1 = clustered
// evictions where the destination is only written) and non-temporal hints.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool READ_DST, bool SRC_LDS, bool NT>
__global__ __launch_bounds__(64) void mix_k(uint8_t* __restrict__ k, uint8_t* __restrict__ v,
                                            const int2* __restrict__ runs, int nruns) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x;
  const int nw = gridDim.x;
  const int r0 = (int)((int64_t)nruns * blockIdx.x / nw), r1 = (int)((int64_t)nruns * (blockIdx.x + 1) / nw);
  u32x4 kd[4], vd[4], ks[4], vs[4];
  for (int i = 0; i < 4; ++i) { kd[i] = vd[i] = ks[i] = vs[i] = u32x4{0, 0, 0, 0}; }
  for (int r = r0; r < r1; ++r) {
    const int2 run = runs[r];
    uint8_t* kdp = k + (int64_t)run.x * 4096; uint8_t* vdp = v + (int64_t)run.x * 4096;
    const uint8_t* ksp = k + (int64_t)run.y * 4096; const uint8_t* vsp = v + (int64_t)run.y * 4096;
    if (SRC_LDS) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksp + (i * 64 + lane) * 16),
                                         (__attribute__((address_space(3))) void*)(lds + i * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsp + (i * 64 + lane) * 16),
                                         (__attribute__((address_space(3))) void*)(lds + 4096 + i * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ks[i] = NT ? __builtin_nontemporal_load((const u32x4*)(ksp + (i * 64 + lane) * 16)) : *(const u32x4*)(ksp + (i * 64 + lane) * 16);
        vs[i] = NT ? __builtin_nontemporal_load((const u32x4*)(vsp + (i * 64 + lane) * 16)) : *(const u32x4*)(vsp + (i * 64 + lane) * 16);
      }
    }
    if (READ_DST) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        kd[i] = NT ? __builtin_nontemporal_load((const u32x4*)(kdp + (i * 64 + lane) * 16)) : *(const u32x4*)(kdp + (i * 64 + lane) * 16);
        vd[i] = NT ? __builtin_nontemporal_load((const u32x4*)(vdp + (i * 64 + lane) * 16)) : *(const u32x4*)(vdp + (i * 64 + lane) * 16);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (SRC_LDS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ks[i] = *(const u32x4*)(lds + i * 1024 + (lane ^ 1) * 16);
        vs[i] = *(const u32x4*)(lds + 4096 + i * 1024 + (lane ^ 1) * 16);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // stand-in for the patch: half the dwords come from the source
      kd[i].x = ks[i].x; kd[i].z = ks[i].z; vd[i].y = vs[i].y; vd[i].w = vs[i].w;
      if (!READ_DST) { kd[i].y = ks[i].y; kd[i].w = ks[i].w; vd[i].x = vs[i].x; vd[i].z = vs[i].z; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (NT) {
        __builtin_nontemporal_store(kd[i], (u32x4*)(kdp + (i * 64 + lane) * 16));
        __builtin_nontemporal_store(vd[i], (u32x4*)(vdp + (i * 64 + lane) * 16));
      } else {
        *(u32x4*)(kdp + (i * 64 + lane) * 16) = kd[i];
        *(u32x4*)(vdp + (i * 64 + lane) * 16) = vd[i];
      }
    }
  }
}

template <bool READ_DST, bool SRC_LDS, bool NT>
float run(uint8_t* k, uint8_t* v, const int2* runs, int nruns, int waves_per_cu) {
  const size_t lds = 160 * 1024 / waves_per_cu / 512 * 512;     // LDS sized so that exactly this many waves fit a CU
  (void)hipFuncSetAttribute((const void*)mix_k<READ_DST, SRC_LDS, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int grid = 256 * waves_per_cu;
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((mix_k<READ_DST, SRC_LDS, NT>), dim3(grid), dim3(64), lds, 0, k, v, runs, nruns);
  (void)hipEventRecord(a);
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((mix_k<READ_DST, SRC_LDS, NT>), dim3(grid), dim3(64), lds, 0, k, v, runs, nruns);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}

int main(int argc, char** argv) {
  const int NB = 1 << (argc > 1 ? atoi(argv[1]) : 18);   // default 1 GiB of K + 1 GiB of V
  uint8_t *k, *v; int2* runs;
  (void)hipMalloc(&k, (size_t)NB * 4096); (void)hipMalloc(&v, (size_t)NB * 4096);
  (void)hipMemset(k, 1, (size_t)NB * 4096); (void)hipMemset(v, 2, (size_t)NB * 4096);
  std::vector<int> perm(NB);
  for (int i = 0; i < NB; ++i) perm[i] = i;
  std::mt19937 rng(1);
  std::shuffle(perm.begin(), perm.end(), rng);
  const int nruns = NB / 2;
  std::vector<int2> h(nruns);
  for (int i = 0; i < nruns; ++i) h[i] = int2{perm[2 * i], perm[2 * i + 1]};
  (void)hipMalloc(&runs, sizeof(int2) * nruns);
  (void)hipMemcpy(runs, h.data(), sizeof(int2) * nruns, hipMemcpyHostToDevice);
  printf("{\"_note\": \"tools/blockmix_bw.hip: random 4+4 KiB block images, %d runs over %d blocks (%.1f GiB of K + V); GB/s of total traffic (reads + writes)\",\n \"rows\": [\n", nruns, NB, NB * 8192.0 / (1 << 30));
  bool first = true;
  for (int w : {8, 16}) {
    const double rmw = 24576.0 * nruns, cp = 16384.0 * nruns;
    float t;
#define ROW(name, RD, SL, NT_, bytes)                                                              \
    if (SL && w > 20) {} else {                                                                    \
      t = run<RD, SL, NT_>(k, v, runs, nruns, w);                                                  \
      printf("%s  {\"pattern\": \"%s\", \"waves_per_cu\": %d, \"ms\": %.4f, \"GBps\": %.0f}", first ? "" : ",\n", name, w, t, bytes / t / 1e6); \
      first = false; }
    ROW("rmw 2R:1W src->LDS nt", true, true, true, rmw)
    ROW("rmw 2R:1W src->LDS plain", true, true, false, rmw)
    ROW("rmw 2R:1W src->regs nt", true, false, true, rmw)
    ROW("copy 1R:1W src->LDS nt", false, true, true, cp)
    ROW("copy 1R:1W src->regs nt", false, false, true, cp)
    ROW("copy 1R:1W src->regs plain", false, false, false, cp)
  }
  printf("\n]}\n");
  return 0;
}
