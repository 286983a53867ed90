// What limits aggregate_decode (metrics[s] += sum_q temp[s,q]^2; 24 B per slot at qpk 4) at config 3's size
// (275 M slots: 4.4 GB of temp + 1.1 GB of metrics read and written)?  Variants of the access pattern, timed
// on hipMalloc'd buffers (profiling aid for csrc/kvc_aggregate.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/agg_bw.hip -o /tmp/agg_bw && /tmp/agg_bw [slots_millions]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sum_sq(f32x4 v) {
  v.x = __fmul_rn(v.x, v.x); v.y = __fmul_rn(v.y, v.y); v.z = __fmul_rn(v.z, v.z); v.w = __fmul_rn(v.w, v.w);
  return __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.0f, v.x), v.y), v.z), v.w);
}

// V0: the library's kernel -- a lane per slot, U rows a grid stride apart
template <int U, bool NT>
__global__ __launch_bounds__(256) void agg_stride(float* __restrict__ metrics, const float* __restrict__ temp, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t s0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; s0 + (U - 1) * stride < n; s0 += U * stride) {
    f32x4 t[U];
    float m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      t[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(temp) + (s0 + u * stride));
      m[u] = NT ? __builtin_nontemporal_load(metrics + s0 + u * stride) : metrics[s0 + u * stride];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float r = __fadd_rn(m[u], sum_sq(t[u]));
      if (NT) __builtin_nontemporal_store(r, metrics + s0 + u * stride); else metrics[s0 + u * stride] = r;
    }
  }
  for (int64_t s = s0; s < n; s += stride) metrics[s] = __fadd_rn(metrics[s], sum_sq(reinterpret_cast<const f32x4*>(temp)[s]));
}

// V1: a workgroup takes contiguous tiles of 256 * U slots (the U rows adjacent), tiles a grid stride apart
template <int U, bool NT>
__global__ __launch_bounds__(256) void agg_tile(float* __restrict__ metrics, const float* __restrict__ temp, int64_t n) {
  const int64_t tile = 256 * U, ntiles = n / tile;
  for (int64_t tI = blockIdx.x; tI < ntiles; tI += gridDim.x) {
    const int64_t base = tI * tile + threadIdx.x;
    f32x4 t[U];
    float m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      t[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(temp) + (base + u * 256));
      m[u] = NT ? __builtin_nontemporal_load(metrics + base + u * 256) : metrics[base + u * 256];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float r = __fadd_rn(m[u], sum_sq(t[u]));
      if (NT) __builtin_nontemporal_store(r, metrics + base + u * 256); else metrics[base + u * 256] = r;
    }
  }
}

// V2: four slots per lane: metrics as one 16-byte access, temp as four 16-byte loads 64 B apart per lane
template <int U>
__global__ __launch_bounds__(256) void agg_vec4(float* __restrict__ metrics, const float* __restrict__ temp, int64_t n) {
  const int64_t n4 = n / 4, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q0 + (U - 1) * stride < n4; q0 += U * stride) {
    f32x4 t[U][4], m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t q = q0 + u * stride;
#pragma unroll
      for (int k = 0; k < 4; ++k) t[u][k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(temp) + (q * 4 + k));
      m[u] = reinterpret_cast<const f32x4*>(metrics)[q];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f32x4 r = m[u];
      r.x = __fadd_rn(r.x, sum_sq(t[u][0])); r.y = __fadd_rn(r.y, sum_sq(t[u][1]));
      r.z = __fadd_rn(r.z, sum_sq(t[u][2])); r.w = __fadd_rn(r.w, sum_sq(t[u][3]));
      reinterpret_cast<f32x4*>(metrics)[q0 + u * stride] = r;
    }
  }
}

// with the fused clear of temp (40 B per slot): one slot per trip (the library's), tiles of U rows, the grid-stride loop
template <int U, int MODE, bool NT>   // MODE 0: tile, 1: stride
__global__ __launch_bounds__(256) void agg_clear(float* __restrict__ metrics, float* __restrict__ temp, int64_t n) {
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4* temp4 = reinterpret_cast<f32x4*>(temp);
  const int64_t span = MODE == 0 ? 256 * U : (int64_t)gridDim.x * 256 * U;
  const int64_t step = MODE == 0 ? 256 : (int64_t)gridDim.x * 256;
  const int64_t nspan = n / span;
  for (int64_t si = MODE == 0 ? blockIdx.x : 0; si < nspan; si += MODE == 0 ? gridDim.x : 1) {
    const int64_t base = si * span + (MODE == 0 ? threadIdx.x : (int64_t)blockIdx.x * 256 + threadIdx.x);
    f32x4 t[U];
    float m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      t[u] = temp4[base + u * step];
      m[u] = NT ? __builtin_nontemporal_load(metrics + base + u * step) : metrics[base + u * step];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float r = __fadd_rn(m[u], sum_sq(t[u]));
      if (NT) { __builtin_nontemporal_store(r, metrics + base + u * step); __builtin_nontemporal_store(zero, temp4 + base + u * step); }
      else { metrics[base + u * step] = r; temp4[base + u * step] = zero; }
    }
  }
}

// reference points: a pure read of the same bytes, and a float4 copy
__global__ __launch_bounds__(256) void read_only(const float* __restrict__ metrics, const float* __restrict__ temp, int64_t n, float* sink) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += stride) {
    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(temp) + s);
    acc += t.x + t.y + t.z + t.w + __builtin_nontemporal_load(metrics + s);
  }
  if (acc == 123.456f) *sink = acc;
}

template <typename F>
static float timed(F launch, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main(int argc, char** argv) {
  const int64_t n = (int64_t)(argc > 1 ? atof(argv[1]) : 275.0) * 1000000 / 4096 * 4096;
  float *metrics, *temp, *sink;
  if (hipMalloc(&metrics, n * 4) != hipSuccess || hipMalloc(&temp, n * 16) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) {
    printf("alloc failed\n");
    return 1;
  }
  hipMemset(metrics, 0, n * 4);
  hipMemset(temp, 0, n * 16);
  const double gb = n * 24.0 / 1e9;
  auto report = [&](const char* name, float ms) { printf("%-44s %8.3f ms  %6.2f TB/s\n", name, ms, gb / ms); };
  for (unsigned grid : {2048u, 4096u, 8192u, 16384u}) {
    printf("grid %u\n", grid);
    report("stride U4 (library)", timed([&] { hipLaunchKernelGGL((agg_stride<4, false>), dim3(grid), dim3(256), 0, 0, metrics, temp, n); }, 5));
    report("stride U8", timed([&] { hipLaunchKernelGGL((agg_stride<8, false>), dim3(grid), dim3(256), 0, 0, metrics, temp, n); }, 5));
    report("stride U4, metrics nontemporal", timed([&] { hipLaunchKernelGGL((agg_stride<4, true>), dim3(grid), dim3(256), 0, 0, metrics, temp, n); }, 5));
    report("tile U4", timed([&] { hipLaunchKernelGGL((agg_tile<4, false>), dim3(grid), dim3(256), 0, 0, metrics, temp, n); }, 5));
    report("tile U8", timed([&] { hipLaunchKernelGGL((agg_tile<8, false>), dim3(grid), dim3(256), 0, 0, metrics, temp, n); }, 5));
    report("tile U8, metrics nontemporal", timed([&] { hipLaunchKernelGGL((agg_tile<8, true>), dim3(grid), dim3(256), 0, 0, metrics, temp, n); }, 5));
    {
      const double gb40 = n * 40.0 / 1e9;
      auto rep40 = [&](const char* name, float ms) { printf("%-44s %8.3f ms  %6.2f TB/s (40 B)\n", name, ms, gb40 / ms); };
      rep40("clear: one slot per trip (library)", timed([&] { hipLaunchKernelGGL((agg_clear<1, 1, false>), dim3(grid), dim3(256), 0, 0, metrics, temp, n); }, 5));
      rep40("clear: stride U4", timed([&] { hipLaunchKernelGGL((agg_clear<4, 1, false>), dim3(grid), dim3(256), 0, 0, metrics, temp, n); }, 5));
      rep40("clear: stride U4 nontemporal", timed([&] { hipLaunchKernelGGL((agg_clear<4, 1, true>), dim3(grid), dim3(256), 0, 0, metrics, temp, n); }, 5));
      rep40("clear: tile U4", timed([&] { hipLaunchKernelGGL((agg_clear<4, 0, false>), dim3(grid), dim3(256), 0, 0, metrics, temp, n); }, 5));
      rep40("clear: tile U8", timed([&] { hipLaunchKernelGGL((agg_clear<8, 0, false>), dim3(grid), dim3(256), 0, 0, metrics, temp, n); }, 5));
      rep40("clear: tile U8 nontemporal", timed([&] { hipLaunchKernelGGL((agg_clear<8, 0, true>), dim3(grid), dim3(256), 0, 0, metrics, temp, n); }, 5));
      rep40("clear: tile U2", timed([&] { hipLaunchKernelGGL((agg_clear<2, 0, false>), dim3(grid), dim3(256), 0, 0, metrics, temp, n); }, 5));
    }
    report("vec4 U1", timed([&] { hipLaunchKernelGGL((agg_vec4<1>), dim3(grid), dim3(256), 0, 0, metrics, temp, n); }, 5));
    report("vec4 U2", timed([&] { hipLaunchKernelGGL((agg_vec4<2>), dim3(grid), dim3(256), 0, 0, metrics, temp, n); }, 5));
  }
  {
    const float ms = timed([&] { hipLaunchKernelGGL(read_only, dim3(8192), dim3(256), 0, 0, metrics, temp, n, sink); }, 5);
    printf("%-44s %8.3f ms  %6.2f TB/s (20 B per slot)\n", "read only (temp + metrics)", ms, n * 20.0 / 1e9 / ms);
  }
  return 0;
}
