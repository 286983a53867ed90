// tests/test_gpu_device_helpers.py compiles and runs this: the wave-level helpers of csrc/kvc_common.h against a host loop.
//   wave_reduce_sum / wave_inclusive_scan           (shuffles: any set of active lanes)
//   wave_reduce_sum_full / wave_inclusive_scan_full (DPP row operations: all 64 lanes active)
//   rank_in_halves (csrc/kvc_schedule_fused.h): a lane's rank among the n first entries of its half of the wave
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../vllm_kvcompress_amd/csrc/kvc_common.h"
#include "../../vllm_kvcompress_amd/csrc/kvc_schedule_fused.h"      // rank_in_halves: ranks through DPP row broadcasts

__global__ void k(uint32_t* out, const uint32_t* in) {
  const uint32_t v = in[blockIdx.x * 64 + threadIdx.x];
  uint32_t* o = out + (size_t)blockIdx.x * 256;
  o[threadIdx.x] = kvc::wave_reduce_sum(v);
  o[64 + threadIdx.x] = kvc::wave_inclusive_scan(v);
  o[128 + threadIdx.x] = kvc::wave_reduce_sum_full(v);
  o[192 + threadIdx.x] = kvc::wave_inclusive_scan_full(v);
}

// two lists per wave (lanes 0 .. 31 and 32 .. 63), n entries each at most; the lanes beyond a list's end hold 0xFFFFFFFF
__global__ void kr(uint32_t* out, const uint32_t* in, const int* lens) {
  const int ca = lens[2 * blockIdx.x], cb = lens[2 * blockIdx.x + 1];
  const int e = threadIdx.x & 31, c = threadIdx.x < 32 ? ca : cb;
  const uint32_t v = e < c ? in[blockIdx.x * 64 + threadIdx.x] : 0xFFFFFFFFu;
  out[blockIdx.x * 64 + threadIdx.x] = kvc::rank_in_halves(v, ca > cb ? ca : cb);
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
  {                                                   // ranks within halves: every pair of lengths 0 .. 32 (distinct keys)
    std::vector<int> lens;
    for (int a = 0; a <= 32; ++a) for (int b = 0; b <= 32; b += (a % 4 == 0 ? 1 : 5)) { lens.push_back(a); lens.push_back(b); }
    const int P = (int)lens.size() / 2;
    std::vector<uint32_t> hv(P * 64), rv(P * 64);
    for (auto& e : hv) { x = x * 1664525u + 1013904223u; e = x >> 1; }
    for (size_t t = 0; t < hv.size(); ++t) hv[t] = (hv[t] & 0xFFFFFF00u) | (uint32_t)(t & 63);   // distinct within a wave
    uint32_t *dv, *ov; int* dl;
    if (hipMalloc(&dv, hv.size() * 4) != hipSuccess || hipMalloc(&ov, rv.size() * 4) != hipSuccess || hipMalloc(&dl, lens.size() * 4) != hipSuccess) return 2;
    if (hipMemcpy(dv, hv.data(), hv.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return 2;
    if (hipMemcpy(dl, lens.data(), lens.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return 2;
    hipLaunchKernelGGL(kr, dim3(P), dim3(64), 0, 0, ov, dv, dl);
    if (hipMemcpy(rv.data(), ov, rv.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    for (int pq = 0; pq < P; ++pq)
      for (int half = 0; half < 2; ++half) {
        const int c = lens[2 * pq + half];
        for (int e = 0; e < c; ++e) {
          uint32_t want = 0;
          for (int j = 0; j < c; ++j) want += hv[pq * 64 + half * 32 + j] < hv[pq * 64 + half * 32 + e];
          bad += rv[pq * 64 + half * 32 + e] != want;
        }
      }
  }
  printf(bad ? "WAVE_HELPERS_BAD %d\n" : "WAVE_HELPERS_OK %d\n", bad);
  return bad != 0;
}
