// Where in HBM do randomly placed 4 KiB images stream fast?  (profiling aid)
// One 200 GiB allocation; the rmw pattern of compact_runs_kernel (read dst, read src, write dst;
// one image per "plane" here: both streams draw from the same block set).  Block ids are drawn
// from one window [O, O+W) or from two windows of W/2 at offsets O and O+X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_(uint8_t* __restrict__ base, const int2* __restrict__ runs, int nruns) {
  __shared__ __attribute__((aligned(16))) uint8_t lds_s[4][8192];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint8_t* lds = lds_s[wib];
  const int nw = gridDim.x * 4, wid = blockIdx.x * 4 + wib;
  const int r0 = (int)((int64_t)nruns * wid / nw), r1 = (int)((int64_t)nruns * (wid + 1) / nw);
  u32x4 kd[8];
  for (int r = r0; r < r1; ++r) {
    const int2 kr = runs[r];
    uint8_t* dp = base + (int64_t)kr.x * 8192;
    const uint8_t* sp = base + (int64_t)kr.y * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i) kd[i] = __builtin_nontemporal_load((const u32x4*)(dp + (i * 64 + lane) * 16));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sp + (i * 64 + lane) * 16),
                                       (__attribute__((address_space(3))) void*)(lds + i * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const u32x4 ks = *(const u32x4*)(lds + i * 1024 + (lane ^ 1) * 16);
      kd[i].x = ks.x; kd[i].z = ks.z;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(kd[i], (u32x4*)(dp + (i * 64 + lane) * 16));
  }
}

static uint8_t* buf;
static int2* druns;
static const int nruns = 262144;
static std::mt19937_64 rng(7);

// ids of 8 KiB units: two windows of `half` units each at unit offsets o1, o2
static double measure(int64_t o1, int64_t o2, int64_t half) {
  std::vector<int2> h(nruns);
  // distinct ids: random sample without replacement via shuffle of a strided subset when the window is small
  std::vector<int64_t> ids(2 * (size_t)nruns);
  const int64_t total = 2 * half;
  if (total >= 4 * (int64_t)ids.size()) {
    std::vector<int64_t> tmp;
    tmp.reserve(ids.size() * 2);
    while (tmp.size() < ids.size() * 2) tmp.push_back((int64_t)(rng() % (uint64_t)total));
    std::sort(tmp.begin(), tmp.end()); tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    std::shuffle(tmp.begin(), tmp.end(), rng);
    if (tmp.size() < ids.size()) return -1;
    std::copy(tmp.begin(), tmp.begin() + ids.size(), ids.begin());
  } else {
    if (total < (int64_t)ids.size()) return -1;
    std::vector<int64_t> all(total);
    for (int64_t i = 0; i < total; ++i) all[i] = i;
    std::shuffle(all.begin(), all.end(), rng);
    std::copy(all.begin(), all.begin() + ids.size(), ids.begin());
  }
  auto map = [&](int64_t i) { return i < half ? o1 + i : o2 + (i - half); };
  for (int i = 0; i < nruns; ++i) h[i] = int2{(int)map(ids[2 * i]), (int)map(ids[2 * i + 1])};
  (void)hipMemcpy(druns, h.data(), sizeof(int2) * nruns, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k_, dim3(512), dim3(256), 0, 0, buf, druns, nruns);
  (void)hipEventRecord(e0);
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(k_, dim3(512), dim3(256), 0, 0, buf, druns, nruns);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return 24576.0 * nruns / (ms / 5) / 1e6;
}

int main() {
  const int64_t GiB = 1ll << 30, U = GiB / 8192;        // units per GiB
  const int64_t total_gib = 200;
  if (hipMalloc(&buf, (size_t)total_gib * GiB) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
  (void)hipMemset(buf, 1, (size_t)total_gib * GiB);
  (void)hipMalloc(&druns, sizeof(int2) * nruns);
  printf("{\"one_window\": [\n");
  const double ws[] = {4.5, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192};
  bool first = true;
  for (double w : ws) {
    const int64_t half = (int64_t)(w * U / 2);
    printf("%s {\"window_GiB\": %.1f, \"offset_GiB\": 0, \"GBps\": %.0f}", first ? "" : ",\n", w, measure(0, half, half));
    first = false;
  }
  for (double o : {1.0, 3.0, 7.0, 20.0, 100.0})
    printf(",\n {\"window_GiB\": 4.5, \"offset_GiB\": %.0f, \"GBps\": %.0f}", o, measure((int64_t)(o * U), (int64_t)(o * U) + (int64_t)(2.25 * U), (int64_t)(2.25 * U)));
  printf("\n],\n\"two_windows_of_2.25_GiB\": [\n");
  first = true;
  for (double x : {2.25, 2.5, 3.0, 3.5, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0, 12.0, 16.0, 20.0, 24.0, 32.0, 33.0, 48.0, 62.0, 64.0, 96.0, 100.0, 128.0, 160.0}) {
    printf("%s {\"second_window_at_GiB\": %.2f, \"GBps\": %.0f}", first ? "" : ",\n", x, measure(0, (int64_t)(x * U), (int64_t)(2.25 * U)));
    first = false;
  }
  printf("\n]}\n");
  return 0;
}
