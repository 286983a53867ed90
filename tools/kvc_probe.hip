// libkvc_probe.so -- measurement aid, NOT part of the drop-in library: the memory traffic of
// execute_cache_moves' compaction kernel with no logic in it, so that bench.py can put the
// kernel's time next to what the SAME box sustains for the bare access pattern (the rate differs
// by 20 % between MI355X boxes, profiles/r2_compact_variants.md).
// Per run: read the destination block's K + V images into registers (rmw mode only), stream one
// source block's images into LDS (global_load_lds), write the destination images back; randomly
// placed images, one round trip per run, non-temporal loads / stores -- the structure of
// kvc::compact_runs_kernel.  tools/blockmix_bw.hip is the stand-alone sweep of the same pattern.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NPL, bool READ_DST>
__global__ __launch_bounds__(256) void probe_block_stream_kernel(uint8_t* __restrict__ k, uint8_t* __restrict__ v,
                                                                 const int2* __restrict__ runs, int nruns) {
  constexpr int IMG = NPL * 1024;
  __shared__ __attribute__((aligned(16))) uint8_t lds_s[4][2 * IMG];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint8_t* lds = lds_s[wib];
  const int nw = gridDim.x * 4, wid = blockIdx.x * 4 + wib;
  const int r0 = (int)((int64_t)nruns * wid / nw), r1 = (int)((int64_t)nruns * (wid + 1) / nw);
  u32x4 kd[NPL], vd[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) { kd[i] = u32x4{0, 0, 0, 0}; vd[i] = u32x4{0, 0, 0, 0}; }
  for (int r = r0; r < r1; ++r) {
    const int2 run = runs[r];
    uint8_t* kdp = k + (int64_t)run.x * IMG; uint8_t* vdp = v + (int64_t)run.x * IMG;
    const uint8_t* ksp = k + (int64_t)run.y * IMG; const uint8_t* vsp = v + (int64_t)run.y * IMG;
    if (READ_DST) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        kd[i] = __builtin_nontemporal_load((const u32x4*)(kdp + (i * 64 + lane) * 16));
        vd[i] = __builtin_nontemporal_load((const u32x4*)(vdp + (i * 64 + lane) * 16));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksp + (i * 64 + lane) * 16),
                                       (__attribute__((address_space(3))) void*)(lds + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsp + (i * 64 + lane) * 16),
                                       (__attribute__((address_space(3))) void*)(lds + IMG + i * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NPL; ++i) {     // stand-in for the patch: half of every piece comes from the source
      const u32x4 ks = *(const u32x4*)(lds + i * 1024 + (lane ^ 1) * 16);
      const u32x4 vs = *(const u32x4*)(lds + IMG + i * 1024 + (lane ^ 1) * 16);
      kd[i].x = ks.x; kd[i].z = ks.z; vd[i].y = vs.y; vd[i].w = vs.w;
      if (!READ_DST) { kd[i].y = ks.y; kd[i].w = ks.w; vd[i].x = vs.x; vd[i].z = vs.z; }
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      __builtin_nontemporal_store(kd[i], (u32x4*)(kdp + (i * 64 + lane) * 16));
      __builtin_nontemporal_store(vd[i], (u32x4*)(vdp + (i * 64 + lane) * 16));
    }
  }
}

// k, v: device buffers of at least (max block id + 1) * image_bytes bytes; runs: [nruns, 2] int32
// (destination block, source block), all ids distinct; image_bytes 4096 or 8192; read_dst != 0:
// read-modify-write (2 reads : 1 write), else copy (1 : 1).  Asynchronous on `stream`.
// Returns 0, or 1 for an unsupported image size.
extern "C" int kvc_probe_block_stream(void* k, void* v, const int32_t* runs, int32_t nruns,
                                      int32_t image_bytes, int32_t read_dst, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  uint8_t* kp = (uint8_t*)k; uint8_t* vp = (uint8_t*)v;
  const int2* rp = (const int2*)runs;
#define PROBE(NPL, RD, WGS) hipLaunchKernelGGL((probe_block_stream_kernel<NPL, RD>), dim3(256 * WGS), dim3(256), 0, s, kp, vp, rp, nruns)
  if (image_bytes == 4096) { if (read_dst) PROBE(4, true, 2); else PROBE(4, false, 2); }
  else if (image_bytes == 8192) { if (read_dst) PROBE(8, true, 2); else PROBE(8, false, 2); }
  else return 1;
#undef PROBE
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// Test aid (tests/test_gpu_fallback_residency.py): hold compute units for a while -- `workgroups`
// workgroups of 256 threads with `lds_bytes` of LDS each (<= 64 KiB) that spin until `usec`
// microseconds of the 100 MHz wall clock have passed.  With 2 x 64 KiB per CU on every CU a kernel
// launched next to it on another stream gets a fraction of the chip.  Asynchronous on `stream`.
__global__ __launch_bounds__(256) void probe_occupy_kernel(unsigned long long ticks, uint32_t* sink) {
  extern __shared__ uint32_t occ_s[];
  occ_s[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  if (sink != nullptr && occ_s[(threadIdx.x + 1) & 255] == 0xFFFFFFFFu) *sink = 1u;
}
extern "C" int kvc_probe_occupy(int32_t workgroups, int32_t lds_bytes, int64_t usec, void* stream) {
  if (workgroups < 1 || lds_bytes < 1024 || lds_bytes > 65536 || usec < 0 || usec > 2000000) return 1;
  hipLaunchKernelGGL(probe_occupy_kernel, dim3((unsigned)workgroups), dim3(256), (size_t)lds_bytes, (hipStream_t)stream,
                     (unsigned long long)usec * 100ull, (uint32_t*)nullptr);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
