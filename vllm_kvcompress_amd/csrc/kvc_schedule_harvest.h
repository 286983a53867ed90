// kvc_schedule_harvest.h -- A2a + A3: the decode step's aggregation that harvests the small-eviction schedule's candidates
// (one translation unit: included by kvc_schedule.hip behind kvc_schedule_small.h; see the overview there)
#pragma once
#include "kvc_common.h"
#include "kvc_schedule_common.h"
#include "kvc_schedule_small.h"
#include "../../include/kvc_mi355x.h"

namespace kvc {

// ------------------------------------------------------------------ 10. harvest-ahead
// With compression_interval = 1 every decode step sweeps the metric store twice: aggregate_decode
// (reference metrics.py:429-439: metrics += sum_q temp^2, the whole cache, 4 * qpk + 8 B per slot)
// and, right behind it, stream_collect_kernel (section 7: 4 + 1 B per candidate slot, to find the
// ~1 % of the keys below each sequence's pivot).  The sums pass through registers in the first
// sweep: this kernel is aggregate_decode_kernel with section 7's harvest behind the add -- the
// second sweep disappears.  What it cannot know is the pivot, a quantile of the sums it is making;
// it uses the one the PREVIOUS schedule call left in the harvest buffer (stream_pivot_kernel,
// aimed past what that call evicts at what this one will need).  Exactness does not depend on how
// good that guess is: the records made from these lists hold EVERY evictable key below the pivot
// that was used, so the selection is exact as soon as they list k' thresholds -- or the flag is
// raised, as for any record that falls short (section 7).
//
// Layout: the plain kernel's -- a lane per slot, rows of 64 consecutive slots, the temp row as
// qpk / 4 16-byte loads per lane -- so that the arithmetic (and its order) is the plain kernel's.
// A wave iteration covers 64 blocks = BS rows: lane b looks after the metadata of block b (one
// coalesced load per table, as in section 7) and hands pivot / head / position bound to the rows'
// lanes by shuffles; U = 4 rows are in flight at a time.
struct HvLayout { size_t pivot, claimed, cnt, rec64, total; };
inline HvLayout hv_layout(int32_t G, int32_t B) {
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  HvLayout l;
  size_t o = 256;                                    // (header: reserved)
  l.pivot = o;    o = up(o + (size_t)B * 4);
  l.claimed = o;  o = up(o + (size_t)CLAIM_SHARDS * 128);   // claimed | cnt: one fill per harvest
  l.cnt = o;      o = up(o + (size_t)G * 4);
  l.rec64 = o;    o = up(o + (size_t)G * KREC * 8);
  l.total = o;
  return l;
}

template <int BS, int QV>
__global__ __launch_bounds__(256) void aggregate_harvest_kernel(kvc_schedule_params p, SchedWs ws, float* __restrict__ temp,
                                                                const uint32_t* __restrict__ hv_pivot, int use_l2,
                                                                int clear_temp) {
  constexpr int ROWS = BS;                           // 64 blocks x BS slots = BS rows of 64 slots
  constexpr int U = 4;                               // rows in flight
  constexpr int BPR = 64 / BS;                       // blocks per row
  static_assert(ROWS % U == 0, "block sizes 8 / 16 / 32");
  __shared__ uint32_t qk[4][128], qs[4][128], qg[4][128];
  __shared__ int32_t ql[4][128];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int L = p.num_layers, H = p.num_kv_heads;
  float* __restrict__ metrics = const_cast<float*>(p.metrics);
  f32x4* __restrict__ temp4 = reinterpret_cast<f32x4*>(temp);
  const int64_t num_slots = p.num_blocks * BS;
  unsigned long long* lists = reinterpret_cast<unsigned long long*>(ws.rec64);
  uint32_t claimed = 0;
  int qn = 0;
  auto drain = [&](int n) {                          // pops the top n (<= 64) queue entries (section 7, LAZY)
    wave_lds_sync();
    if (lane < n) {
      const int e = qn - n + lane;
      const uint32_t g = qg[w][e];
      const int tp = p.token_positions[qs[w][e]];    // metrics.py:539-544, for the few that matter
      if (tp <= ql[w][e] && tp >= p.num_sinks) {
        const uint32_t pos = atomicAdd(&ws.st_cnt[g], 1u);
        if (pos < (uint32_t)KREC) lists[(int64_t)g * KREC + pos] = ((unsigned long long)qk[w][e] << 32) | qs[w][e];
      }
    }
    qn -= n;
    wave_lds_sync();
  };
  const int64_t wave = (int64_t)blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4;
  for (int64_t b0 = wave * 64; b0 < p.num_blocks; b0 += nwaves * 64) {
    const int64_t slot0 = b0 * BS;
    f32x4 t[U][QV];
    float m[U];
    auto load_rows = [&](int r0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t s = slot0 + (int64_t)(r0 + u) * 64 + lane;
        if (s < num_slots) {
#pragma unroll
          for (int v = 0; v < QV; ++v) t[u][v] = __builtin_nontemporal_load(temp4 + s * QV + v);
          m[u] = metrics[s];
        } else {
#pragma unroll
          for (int v = 0; v < QV; ++v) t[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
          m[u] = 0.f;
        }
      }
    };
    load_rows(0);                                    // the rows do not wait for the metadata
    const int64_t mb = b0 + lane;
    const bool have = mb < p.num_blocks;
    const BlockMeta mt = load_meta(p, mb, have);
    bool ok = have && mt.s >= 0 && mt.s < p.seq_slot_len;
    int i = p.seq_slot_of_seq[ok ? mt.s : 0];
    ok = ok && i >= 0 && mt.l >= 0 && mt.l < L && mt.h >= 0 && mt.h < H;
    const int l = ok ? mt.l : 0, h = ok ? mt.h : 0;
    if (!ok) i = 0;
    const int ctx = p.context_lens[(l * p.num_seqs + i) * H + h];
    ok = ok && mt.lbn >= 0 && mt.lbn < (ctx + BS - 1) / BS;
    claimed += (uint32_t)__popcll(__ballot(ok));
    const int g = ok ? (i * L + l) * H + h : 0;
    const uint32_t pex = ok ? hv_pivot[i] : 0u;      // (0: no key lies below it)
    const int bound = p.seq_positions[i] - p.num_protected[i];
#pragma unroll
    for (int r0 = 0; r0 < ROWS; r0 += U) {
      if (r0 > 0) load_rows(r0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t s = slot0 + (int64_t)(r0 + u) * 64 + lane;
        const bool in = s < num_slots;
        float acc = 0.0f;
#pragma unroll
        for (int v = 0; v < QV; ++v) {
          f32x4 x = t[u][v];
          if (use_l2) { x.x = __fmul_rn(x.x, x.x); x.y = __fmul_rn(x.y, x.y); x.z = __fmul_rn(x.z, x.z); x.w = __fmul_rn(x.w, x.w); }
          acc = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(acc, x.x), x.y), x.z), x.w);
        }
        const float mn = __fadd_rn(m[u], acc);
        if (in) {
          metrics[s] = mn;
          if (clear_temp) {
#pragma unroll
            for (int v = 0; v < QV; ++v) temp4[s * QV + v] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
        const int src = (r0 + u) * BPR + lane / BS;  // the lane that looks after this slot's block
        const uint32_t pvv = (uint32_t)__shfl((int)pex, src, 64);
        const uint32_t key = float_to_key(mn);
        const bool c = in && key < pvv;              // (pvv <= KEY_INF)
        const unsigned long long bal = __ballot(c);
        if (bal) {                                   // wave-uniform
          const int gg = __shfl(g, src, 64);
          const int bb = __shfl(bound, src, 64);
          if (c) {
            const int pos = qn + __popcll(bal & ((1ull << lane) - 1ull));
            qk[w][pos] = key; qs[w][pos] = (uint32_t)s; qg[w][pos] = (uint32_t)gg; ql[w][pos] = bb;
          }
          qn += __popcll(bal);
          if (qn >= 64) drain(64);
        }
      }
    }
  }
  if (qn > 0) drain(qn);
  // blocks that are logical blocks of the batch (stream_records_kernel wants every one): as in section 7
  __shared__ uint32_t claimed_s;
  if (threadIdx.x == 0) claimed_s = 0;
  __syncthreads();
  if (lane == 0 && claimed) atomicAdd(&claimed_s, claimed);
  __syncthreads();
  if (threadIdx.x == 0 && claimed_s) atomicAdd(&ws.st_claimed[(blockIdx.x % CLAIM_SHARDS) * 32], claimed_s);
}

}  // namespace kvc
