// What does ONE coalesced pass over the metric store in physical-block order cost, and what do the
// per-head atomics of a candidate harvest add?  (profiling aid for the stream schedule of
// kvc_schedule.hip; the shape is config 3: 16.8 M blocks of 16 slots, 256 sequences x 256 heads.)
//
//   rows   : metrics[NB,16] f32 + token_positions[NB,16] i32, one 16 B load of each per lane,
//            4 lanes per block, U blocks-of-16 in flight per wave
//   meta   : seq / layer / head / logical block number per block, one coalesced load of 64
//            entries each per 64 blocks, handed to the row lanes by shuffles
//   harvest: keys <= pivot[seq] are candidates; a block that has any takes ONE returning atomicAdd
//            on its head's counter and stores (key, slot) pairs into the head's list
//   deficit: a block with masked slots adds their number to a second per-head counter
//            (non-returning)
// Swept: candidate fraction q, share of blocks with masked slots, U, grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int BS = 16;
constexpr int CAP = 256;

struct P {
  const float* metrics; const int* pos; const int* seq; const int* layer; const int* head; const int* lbn;
  const int* seq_slot; const int* ctx; const int* seq_pos; const int* prot; const uint32_t* pivot;
  uint32_t* cnt; uint32_t* deficit; unsigned long long* cand; uint32_t* claimed;
  int64_t nb; int B, L, H;
};

__device__ __forceinline__ uint32_t f2k(float v) {
  uint32_t b = __float_as_uint(v + 0.0f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// MODE bit 0: harvest (returning atomics + stores); bit 1: deficit atomics; bit 2: an unconditional
// non-returning atomic per block (counting finite keys the direct way)
template <int U, int MODE>
__global__ __launch_bounds__(256) void pass(P p) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  constexpr int BPW = 16 * U;                     // blocks per wave iteration
  static_assert(BPW <= 64, "one metadata load covers 64 blocks");
  uint32_t claimed = 0;
  for (int64_t b0 = wave * BPW; b0 < p.nb; b0 += nwaves * BPW) {
    f32x4 m[U]; i32x4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t blk = b0 + u * 16 + (lane >> 2);
      if (blk >= p.nb) blk = p.nb - 1;
      m[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.metrics + blk * BS) + (lane & 3));
      q[u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(p.pos + blk * BS) + (lane & 3));
    }
    const int64_t mb = b0 + lane;
    int s = -1, l = 0, h = 0, lb = 0;
    if (lane < BPW && mb < p.nb) { s = p.seq[mb]; l = p.layer[mb]; h = p.head[mb]; lb = p.lbn[mb]; }
    int i = -1;
    if (s >= 0) i = p.seq_slot[s];
    int ctx = 0, sp = 0, pr = 0; uint32_t pv = 0;
    if (i >= 0) { ctx = p.ctx[(l * p.B + i) * p.H + h]; sp = p.seq_pos[i]; pr = p.prot[i]; pv = p.pivot[i]; }
    const bool ok = i >= 0 && lb >= 0 && lb < (ctx + BS - 1) / BS;
    const int g = ok ? (i * p.L + l) * p.H + h : -1;
    claimed += (uint32_t)__popcll(__ballot(ok));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int src = u * 16 + (lane >> 2);
      const int gg = __shfl(g, src, 64);
      const int spp = __shfl(sp, src, 64), prr = __shfl(pr, src, 64);
      const uint32_t pvv = (uint32_t)__shfl((int)pv, src, 64);
      const float mm[4] = {m[u].x, m[u].y, m[u].z, m[u].w};
      const int qq[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
      uint32_t key[4]; int nc = 0, nmask = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool in = qq[k] <= spp - prr && qq[k] >= 0;
        key[k] = in ? f2k(mm[k]) : 0xFF800000u;
        nmask += !in;
        nc += key[k] <= pvv && in;
      }
      if (gg < 0) { nc = 0; nmask = 0; }
      // block totals over the 4 lanes of a block
      int t1 = nc + __shfl_xor(nc, 1, 64); int tot = t1 + __shfl_xor(t1, 2, 64);
      int m1 = nmask + __shfl_xor(nmask, 1, 64); int mtot = m1 + __shfl_xor(m1, 2, 64);
      int up1 = __shfl_up(nc, 1, 64), up2 = __shfl_up(nc, 2, 64), up3 = __shfl_up(nc, 3, 64);
      const int sub = lane & 3;
      const int pre = (sub > 0 ? up1 : 0) + (sub > 1 ? up2 : 0) + (sub > 2 ? up3 : 0);
      uint32_t base = 0;
      if constexpr (MODE & 1) {
        if (sub == 0 && tot > 0) base = atomicAdd(&p.cnt[gg], (uint32_t)tot);
        base = (uint32_t)__shfl((int)base, lane & ~3, 64);
        if (nc > 0) {
          uint32_t o = base + pre;
          const int64_t blk = b0 + src;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (key[k] <= pvv && key[k] < 0xFF800000u) {
              if (o < CAP) p.cand[(int64_t)gg * CAP + o] = ((unsigned long long)key[k] << 32) | (uint32_t)(blk * BS + sub * 4 + k);
              ++o;
            }
        }
      }
      if constexpr (MODE & 2) { if (sub == 0 && mtot > 0) atomicAdd(&p.deficit[gg], (uint32_t)mtot); }
      if constexpr (MODE & 4) { if (sub == 0 && gg >= 0) atomicAdd(&p.deficit[gg], (uint32_t)(BS - mtot)); }
      if constexpr (MODE == 0) { if (tot == 12345) p.cnt[0] = 1; }
    }
  }
  if (lane == 0 && claimed) atomicAdd(p.claimed, claimed);
}


// The harvest decoupled from the stream: candidates go into a per-wave LDS queue (ballot
// compaction, no memory traffic) and are drained 64 at a time -- one returning atomic per lane,
// all in flight together -- so the wave waits for an atomic round trip once per 64 candidates, not
// once per iteration.  PF: the next iteration's rows are requested before this one is processed.
template <int U, bool PF>
__global__ __launch_bounds__(256) void pass_q(P p) {
  __shared__ uint32_t qk[4][128], qs[4][128], qg[4][128];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4;
  constexpr int BPW = 16 * U;
  uint32_t claimed = 0;
  int qn = 0;
  auto drain = [&](int n) {                        // pops the top n (<= 64) entries
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < n) {
      const int e = qn - n + lane;
      const uint32_t g = qg[w][e];
      const uint32_t pos = atomicAdd(&p.cnt[g], 1u);
      if (pos < CAP) p.cand[(int64_t)g * CAP + pos] = ((unsigned long long)qk[w][e] << 32) | qs[w][e];
    }
    qn -= n;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  f32x4 m[U], mn[U]; i32x4 q[U], qx[U];
  int s = -1, l = 0, h = 0, lb = 0, sn = -1, ln = 0, hn = 0, lbn_ = 0;
  auto issue = [&](int64_t b0, f32x4* mm, i32x4* qq, int& s_, int& l_, int& h_, int& lb_) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t blk = b0 + u * 16 + (lane >> 2);
      if (blk >= p.nb) blk = p.nb - 1;
      mm[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.metrics + blk * BS) + (lane & 3));
      qq[u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(p.pos + blk * BS) + (lane & 3));
    }
    const int64_t mb = b0 + lane;
    s_ = -1; l_ = 0; h_ = 0; lb_ = 0;
    if (lane < BPW && mb < p.nb) { s_ = p.seq[mb]; l_ = p.layer[mb]; h_ = p.head[mb]; lb_ = p.lbn[mb]; }
  };
  int64_t b0 = wave * BPW;
  if (PF && b0 < p.nb) issue(b0, mn, qx, sn, ln, hn, lbn_);
  for (; b0 < p.nb; b0 += nwaves * BPW) {
    if constexpr (PF) {
#pragma unroll
      for (int u = 0; u < U; ++u) { m[u] = mn[u]; q[u] = qx[u]; }
      s = sn; l = ln; h = hn; lb = lbn_;
      if (b0 + nwaves * BPW < p.nb) issue(b0 + nwaves * BPW, mn, qx, sn, ln, hn, lbn_);
    } else {
      issue(b0, m, q, s, l, h, lb);
    }
    int i = -1;
    if (s >= 0) i = p.seq_slot[s];
    int ctx = 0, sp = 0, pr = 0; uint32_t pv = 0;
    if (i >= 0) { ctx = p.ctx[(l * p.B + i) * p.H + h]; sp = p.seq_pos[i]; pr = p.prot[i]; pv = p.pivot[i]; }
    const bool ok = i >= 0 && lb >= 0 && lb < (ctx + BS - 1) / BS;
    const int g = ok ? (i * p.L + l) * p.H + h : -1;
    claimed += (uint32_t)__popcll(__ballot(ok));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int src = u * 16 + (lane >> 2);
      const int gg = __shfl(g, src, 64);
      const int lim = __shfl(sp - pr, src, 64);
      const uint32_t pvv = (uint32_t)__shfl((int)pv, src, 64);
      const float mm[4] = {m[u].x, m[u].y, m[u].z, m[u].w};
      const int qq[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
      int nmask = 0;
      const uint32_t slot0 = (uint32_t)((b0 + src) * BS + (lane & 3) * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool in = qq[k] <= lim && qq[k] >= 0 && gg >= 0;
        const uint32_t key = f2k(mm[k]);
        nmask += !in;
        const bool c = in && key <= pvv && key < 0xFF800000u;
        const unsigned long long bal = __ballot(c);
        if (bal) {
          if (c) {
            const int pos = qn + __popcll(bal & ((1ull << lane) - 1ull));
            qk[w][pos] = key; qs[w][pos] = slot0 + k; qg[w][pos] = (uint32_t)gg;
          }
          qn += __popcll(bal);
          if (qn >= 64) drain(64);
        }
      }
      if (gg < 0) nmask = 0;
      int m1 = nmask + __shfl_xor(nmask, 1, 64); int mtot = m1 + __shfl_xor(m1, 2, 64);
      if ((lane & 3) == 0 && mtot > 0) atomicAdd(&p.deficit[gg], (uint32_t)mtot);
    }
  }
  if (qn > 0) drain(qn);
  if (lane == 0 && claimed) atomicAdd(p.claimed, claimed);
}

template <int U, bool PF>
static double run_q(const P& p, int grid, int G) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float tot = 0;
  for (int it = 0; it < 6; ++it) {
    (void)hipMemsetAsync(p.cnt, 0, (size_t)G * 4, 0);
    (void)hipMemsetAsync(p.deficit, 0, (size_t)G * 4, 0);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((pass_q<U, PF>), dim3(grid), dim3(256), 0, 0, p);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (it >= 2) tot += ms;
  }
  return tot / 4;
}

template <int U, int MODE>
static double run(const P& p, int grid, int G) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float tot = 0;
  for (int it = 0; it < 6; ++it) {
    (void)hipMemsetAsync(p.cnt, 0, (size_t)G * 4, 0);
    (void)hipMemsetAsync(p.deficit, 0, (size_t)G * 4, 0);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((pass<U, MODE>), dim3(grid), dim3(256), 0, 0, p);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (it >= 2) tot += ms;
  }
  return tot / 4;
}

int main(int argc, char** argv) {
  const int B = 256, L = 32, H = 8, LH = L * H, G = B * LH;
  const int nblk = argc > 1 ? atoi(argv[1]) : 257;
  const int64_t NB = (int64_t)G * nblk;
  std::vector<int> seq(NB), layer(NB), head(NB), lbn(NB);
  {
    std::vector<int> perm(NB);
    for (int64_t i = 0; i < NB; ++i) perm[i] = (int)i;
    std::mt19937_64 rng(5);
    for (int64_t i = NB - 1; i > 0; --i) { int64_t j = rng() % (i + 1); std::swap(perm[i], perm[j]); }
    int64_t c = 0;
    for (int b = 0; b < B; ++b) for (int l = 0; l < L; ++l) for (int h = 0; h < H; ++h) for (int n = 0; n < nblk; ++n) {
      const int blk = perm[c++]; seq[blk] = b; layer[blk] = l; head[blk] = h; lbn[blk] = n;
    }
  }
  P p{};
  p.nb = NB; p.B = B; p.L = L; p.H = H;
  float* dm; int* dp;
  (void)hipMalloc(&dm, NB * BS * 4); (void)hipMalloc(&dp, NB * BS * 4);
  // metrics: uniform in [0,1) from a device-side hash; positions: 0 except masked blocks
  std::vector<float> hm(NB * BS); std::vector<int> hp(NB * BS, 0);
  { std::mt19937 r2(9); std::uniform_real_distribution<float> ud(0.f, 1.f); for (auto& v : hm) v = ud(r2); }
  (void)hipMemcpy(dm, hm.data(), NB * BS * 4, hipMemcpyHostToDevice);
  auto up = [](const std::vector<int>& v) { int* d; (void)hipMalloc(&d, v.size() * 4); (void)hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice); return d; };
  p.metrics = dm; p.seq = up(seq); p.layer = up(layer); p.head = up(head); p.lbn = up(lbn);
  std::vector<int> slot(B), ctx(G, nblk * BS - 15), spos(B, 1000000), prot(B, 32);
  for (int i = 0; i < B; ++i) slot[i] = i;
  p.seq_slot = up(slot); p.ctx = up(ctx); p.seq_pos = up(spos); p.prot = up(prot);
  uint32_t* piv; (void)hipMalloc(&piv, B * 4); p.pivot = piv;
  (void)hipMalloc(&p.cnt, G * 4); (void)hipMalloc(&p.deficit, G * 4); (void)hipMalloc(&p.claimed, 4);
  (void)hipMalloc(&p.cand, (size_t)G * CAP * 8);
  printf("{\"blocks\": %lld, \"bytes_per_pass\": %lld, \"sweep\": [\n", (long long)NB, (long long)NB * (2 * BS * 4 + 16));
  const double gb = (double)NB * (2 * BS * 4 + 16) / 1e6;
  bool first = true;
  for (int maskden : {64, 8}) {              // one block in maskden has masked slots (0: none)
    for (int64_t b = 0; b < NB; ++b) {
      const bool mk = maskden && (b % maskden) == 3;
      for (int o = 0; o < BS; ++o) hp[b * BS + o] = mk && o >= 12 ? 2000000 : 0;
    }
    (void)hipMemcpy(dp, hp.data(), NB * BS * 4, hipMemcpyHostToDevice);
    p.pos = dp;
    for (float q : {0.0f, 0.004f, 0.01f, 0.0175f, 0.10f}) {
      std::vector<uint32_t> pv(B);
      uint32_t kb; { float qq = q; uint32_t b; memcpy(&b, &qq, 4); kb = b | 0x80000000u; }
      for (auto& v : pv) v = q > 0 ? kb : 0u;
      (void)hipMemcpy(piv, pv.data(), B * 4, hipMemcpyHostToDevice);
      printf("%s {\"masked_1_in\": %d, \"cand_frac\": %.4f", first ? "" : ",\n", maskden, q);
      first = false;
      double t;
      t = run<4, 0>(p, 4096, G); printf(", \"read_only_ms\": %.3f, \"read_only_GBps\": %.0f", t, gb / t);
      t = run<4, 1>(p, 4096, G); printf(", \"harvest_ms\": %.3f", t);
      t = run<4, 3>(p, 4096, G); printf(", \"harvest_deficit_ms\": %.3f", t);
      t = run<4, 5>(p, 4096, G); printf(", \"harvest_count_all_ms\": %.3f", t);
      t = run<4, 3>(p, 2048, G); printf(", \"grid2048_ms\": %.3f", t);
      t = run_q<4, false>(p, 4096, G); printf(", \"queue_ms\": %.3f", t);
      t = run_q<4, true>(p, 4096, G); printf(", \"queue_prefetch_ms\": %.3f", t);
      t = run_q<2, true>(p, 4096, G); printf(", \"queue_prefetch_U2_ms\": %.3f", t);
      t = run_q<4, true>(p, 2048, G); printf(", \"queue_prefetch_grid2048_ms\": %.3f", t);
      t = run_q<4, false>(p, 2048, G); printf(", \"queue_grid2048_ms\": %.3f", t);
      printf("}");
      fflush(stdout);
    }
  }
  printf("]}\n");
  return 0;
}
