// tests/test_gpu_device_helpers.py compiles and runs this: the wave-level helpers of csrc/kvc_common.h against a host loop.
//   wave_reduce_sum / wave_inclusive_scan           (shuffles: any set of active lanes)
//   wave_reduce_sum_full / wave_inclusive_scan_full (DPP row operations: all 64 lanes active)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../vllm_kvcompress_amd/csrc/kvc_common.h"

__global__ void k(uint32_t* out, const uint32_t* in) {
  const uint32_t v = in[blockIdx.x * 64 + threadIdx.x];
  uint32_t* o = out + (size_t)blockIdx.x * 256;
  o[threadIdx.x] = kvc::wave_reduce_sum(v);
  o[64 + threadIdx.x] = kvc::wave_inclusive_scan(v);
  o[128 + threadIdx.x] = kvc::wave_reduce_sum_full(v);
  o[192 + threadIdx.x] = kvc::wave_inclusive_scan_full(v);
}

int main() {
  const int W = 64;                                   // waves, each with its own values
  std::vector<uint32_t> h(W * 64), r(W * 256);
  uint32_t x = 12345u;
  for (auto& e : h) { x = x * 1664525u + 1013904223u; e = (x >> 8) & 0xFFFFFu; }
  for (int i = 0; i < 64; ++i) h[i] = 0u;            // a wave of zeros
  for (int i = 0; i < 64; ++i) h[64 + i] = 0xFFFFFFFFu;   // and one that wraps around
  uint32_t *d, *o;
  if (hipMalloc(&d, h.size() * 4) != hipSuccess || hipMalloc(&o, r.size() * 4) != hipSuccess) return 2;
  if (hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return 2;
  hipLaunchKernelGGL(k, dim3(W), dim3(64), 0, 0, o, d);
  if (hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  int bad = 0;
  for (int w = 0; w < W; ++w) {
    uint32_t s = 0;
    for (int i = 0; i < 64; ++i) {
      s += h[w * 64 + i];
      bad += r[w * 256 + 64 + i] != s;
      bad += r[w * 256 + 192 + i] != s;
    }
    for (int i = 0; i < 64; ++i) {
      bad += r[w * 256 + i] != s;
      bad += r[w * 256 + 128 + i] != s;
    }
  }
  printf(bad ? "WAVE_HELPERS_BAD %d\n" : "WAVE_HELPERS_OK %d\n", bad);
  return bad != 0;
}
