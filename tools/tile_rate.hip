// Calibration for the F4 kernels: time of one 32x32 logit tile (8 x v_mfma_f32_32x32x16_f16 +
// the 16-element softmax epilogue per lane) per SIMD with operands held in registers, at 2 and
// 4 waves per SIMD, to separate compute from the operand streaming of the real kernels.
// Build: hipcc --offload-arch=gfx950 -O3 tools/tile_rate.hip -o tools/bin/tile_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>   // 1 = MFMA only, 2 = epilogue only, 3 = both
__global__ __launch_bounds__(256) void k(float* out, int iters, float sc, float ls) {
  h8 a[8], b[8];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[s][e] = (_Float16)(0.01f * (float)((threadIdx.x + s + e) % 7));
      b[s][e] = (_Float16)(0.02f * (float)((threadIdx.x * 3 + s + e) % 5));
    }
  f16v acc = {0.f};
  f16v c = {0.f};
  for (int it = 0; it < iters; ++it) {
    if (MODE & 1) {
      c = f16v{0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b[s], c, 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) c[i] = c[i] * 0.5f + 0.125f;
    }
    if (MODE & 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf((float)(_Float16)c[i], sc, -ls));
        acc[i] = __builtin_fmaf(p, p, acc[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] += c[i];
    }
    a[0][0] += (_Float16)1e-4f;               // keep the tiles from being hoisted
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// two tiles in flight per wave: tile i + 1's eight MFMAs are issued before tile i's epilogue (no
// dependence between them), SCHED = 1 additionally pins the interleave (one MFMA, then an eighth
// of the epilogue) with sched_group_barrier -- the transformation VERDICT r2 asked to try on F4
template <int SCHED>
__global__ __launch_bounds__(256) void k2(float* out, int iters, float sc, float ls) {
  h8 a[8], b[8];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[s][e] = (_Float16)(0.01f * (float)((threadIdx.x + s + e) % 7));
      b[s][e] = (_Float16)(0.02f * (float)((threadIdx.x * 3 + s + e) % 5));
    }
  f16v acc = {0.f};
  f16v c0 = {0.f}, c1 = {0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b[s], c0, 0, 0, 0);
  auto step = [&](f16v& cur, f16v& nxt) {
    nxt = f16v{0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) nxt = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b[s], nxt, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = __builtin_amdgcn_exp2f(__builtin_fmaf((float)(_Float16)cur[i], sc, -ls));
      acc[i] = __builtin_fmaf(p, p, acc[i]);
    }
    if (SCHED) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);      // VALU of two elements
        __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);      // their two exps
      }
    }
    a[0][0] += (_Float16)1e-4f;
  };
  for (int it = 0; it < iters; it += 2) { step(c0, c1); step(c1, c0); }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i] + c0[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SCHED>
void run2(const char* name, int wg_per_cu) {
  float* d;
  hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
  const int iters = 20000, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k2<SCHED>, dim3(grid), dim3(256), 0, 0, d, 10, 1.1f, 0.3f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k2<SCHED>, dim3(grid), dim3(256), 0, 0, d, iters, 1.1f, 0.3f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double tiles = (double)wg_per_cu * iters;
  printf("%-34s %d waves/SIMD %8.3f ms -> %7.1f ns per tile per SIMD\n", name, wg_per_cu, ms, ms * 1e6 / tiles);
  hipFree(d);
}

template <int MODE>
void run(const char* name, int wg_per_cu) {
  float* d;
  hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
  const int iters = 20000, grid = 256 * wg_per_cu;        // 4 waves per WG: wg_per_cu waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 10, 1.1f, 0.3f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 1.1f, 0.3f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double tiles = (double)wg_per_cu * iters;          // tiles per SIMD
  printf("%-34s %d waves/SIMD %8.3f ms -> %7.1f ns per tile per SIMD\n", name, wg_per_cu, ms, ms * 1e6 / tiles);
  hipFree(d);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<1>("8 MFMA 32x32x16 only", w);
    run<2>("epilogue only (16 elem/lane)", w);
    run<3>("MFMA + epilogue", w);
    run2<0>("two tiles in flight", w);
    run2<1>("two tiles, pinned interleave", w);
  }
  return 0;
}
