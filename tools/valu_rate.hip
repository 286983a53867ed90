// Calibration for the F4 epilogue: cost per wave64 instruction of the four VALU ops of the
// softmax epilogue (v_cvt_f16_f32, v_fma_mix_f32, v_exp_f32, v_pk_fma_f32) on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float sc, float ls) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x + i) * 1e-3f;
  float acc[16] = {0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float x = v[i];
      if (MODE & 1) x = (float)(_Float16)x;                 // cvt + (fused into fma_mix below)
      if (MODE & 2) x = __builtin_fmaf(x, sc, -ls);
      if (MODE & 4) x = __builtin_amdgcn_exp2f(x);
      if (MODE & 8) acc[i] = __builtin_fmaf(x, x, acc[i]); else acc[i] += x;
      v[i] = x * 0.5f + 0.25f;                              // keep a dependency chain per element
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i] + v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name) {
  float* d;
  hipMalloc(&d, 256 * 1024 * 8 * sizeof(float));
  const int iters = 4000, grid = 256 * 8;                   // 8 WGs of 4 waves per CU: 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 10, 1.1f, 0.3f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 1.1f, 0.3f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions of the "element body" executed per SIMD: 8 waves x iters x 16
  const double bodies = 8.0 * iters * 16;
  printf("%-28s %8.3f ms  -> %6.1f ns per element body per SIMD (x clock GHz = cycles)\n", name, ms,
         ms * 1e6 / bodies);
  hipFree(d);
}

int main() {
  run<0>("add + mul-add only");
  run<1>("+ cvt f16 round trip");
  run<3>("+ cvt + fma");
  run<4>("+ exp2 only");
  run<7>("cvt + fma + exp2");
  run<15>("cvt + fma + exp2 + fma(p,p)");
  return 0;
}
