// A4 count_block_evictions and A5 schedule_t1_cache_moves for gfx950.
//
// Both reference kernels run ONE THREAD PER HEAD with a serial loop
// (csrc/kvcompress_eviction_kernels.cu:190-221, 223-289).  Here every loop iteration is
// an independent work item:
//
//  * count: one 64-lane wave per head; lanes test 64 chunk heads per step, the first
//    null chunk is found with a ballot.
//  * moves: the serial two-pointer walk is put in closed form (derivation in DESIGN.md):
//    with E the head's ascending evicted indices, cnt = |E|, R_k = E[cnt-1-k],
//    the k-th "skip" iteration is t_k = max(k, ctx-1-R_k) (strictly increasing), the
//    j-th emitted move comes from iteration i = j + k*, k* = min{k : t_k - k > j}
//    (binary search), and it exists iff i < cnt and E[j] < ctx-1-i.  One thread per ROW
//    of the [rows,2] move table, so the kernel also produces the zero fill of the
//    reference wrapper (vllm/_custom_ops.py:1168) in the same pass, writing each row once.
#include "kvc_common.h"
#include <atomic>
#include "../../include/kvc_mi355x.h"

namespace kvc {

// ------------------------------------------------------------------------------- A4
__global__ __launch_bounds__(256) void count_block_evictions_kernel(
    int32_t* __restrict__ evicted_block_count, int32_t* __restrict__ idx,
    const int32_t* __restrict__ offs, const int32_t* __restrict__ hang, int total_heads,
    int64_t total_kvs, int bs, int null_value) {
  const int g = blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x / WAVE);
  if (g >= total_heads) return;
  const int lane = lane_id();
  const int64_t start = offs[g];
  const int64_t end = (g + 1 >= total_heads) ? total_kvs : (int64_t)offs[g + 1];
  const int64_t nchunks = (end - start + bs - 1) / bs;   // loop "i < end; i += bs"
  int64_t run = nchunks;                                 // leading evicted chunks
  for (int64_t c0 = 0; c0 < nchunks; c0 += WAVE) {
    const int64_t c = c0 + lane;
    bool is_null = false;
    if (c < nchunks) is_null = idx[start + c * bs] == null_value;
    const unsigned long long m = __ballot(is_null);
    if (m) { run = c0 + __ffsll((long long)m) - 1; break; }
  }
  if (lane == 0) evicted_block_count[g] = (int)run;
  if (run > 0) {
    const int64_t last_end = start + run * bs;
    for (int64_t i = last_end - bs + hang[g] + lane; i < last_end; i += WAVE) idx[i] = null_value;
  }
}

// ------------------------------------------------------------------------------- A5
struct MoveHead {
  const int32_t* E;   // evicted indices of this head (ascending), cnt entries
  int cnt;
  int ctx;
};

// iteration index of the j-th emitted move, or -1
__device__ __forceinline__ int move_iteration(const MoveHead& h, int j) {
  // smallest k in [0,cnt] with (k == cnt) or max(0, ctx-1-E[cnt-1-k]-k) > j
  int lo = 0, hi = h.cnt;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    int d = h.ctx - 1 - h.E[h.cnt - 1 - mid] - mid;
    d = d > 0 ? d : 0;
    if (d > j) hi = mid; else lo = mid + 1;
  }
  const int i = j + lo;
  if (i >= h.cnt) return -1;
  if (h.E[j] >= h.ctx - 1 - i) return -1;   // "dst >= src": the walk has stopped
  return i;
}

// What the kernel does about the rows of the [rows, 2] table that hold no move (the reference's
// wrapper clears the WHOLE table before the op, vllm/_custom_ops.py:1168 -- 8 B per candidate slot,
// 2.2 GB at 256 resident sequences, every decode step):
//   ZF_NONE  nothing (the bare op)
//   ZF_ALL   every row without a move <- (0, 0)
//   ZF_DIRTY the table is known to be zero except where the dirty map says otherwise: only those
//            rows are cleared.  The map holds one bit per chunk of `bs` rows (head segments start
//            at multiples of bs) and is rewritten to describe the table after this call; with
//            ZF_ALL it is written as well (from all zeros: the caller cleared it), so that the next
//            call can be a ZF_DIRTY one.  Observable result: identical to ZF_ALL.
constexpr int ZF_NONE = 0, ZF_ALL = 1, ZF_DIRTY = 2;

struct MovesArgs {
  int32_t* moves; int64_t rows; int32_t* count;
  const int32_t* evicted; const int32_t* ekc; const int32_t* offs;
  const int32_t* block_tables; const int32_t* context_lens;
  int B, L, H, M, bs, zero_fill;
  uint32_t* dirty;       // [ceil(rows / bs / 32) + 1] or nullptr
  int32_t* plan;         // [2 * MOVES_PLAN_WGS] or nullptr: tiles per workgroup, (64 - bs)-move and 32-move tiles
  int nwg, heads_per_wg; // moves_plan_shape(G)
};

// rows [b, e) of the move workspace <- (0, 0): 16 B stores
__device__ __forceinline__ void zero_rows(int2* mv2, int64_t b, int64_t e, int tid, int nthreads) {
  typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
  if (b >= e) return;
  if ((reinterpret_cast<uintptr_t>(mv2 + b) & 15u) != 0) {      // odd leading row
    if (tid == 0) mv2[b] = make_int2(0, 0);
    ++b;
  }
  const int64_t pairs = (e - b) >> 1;
  i32x4* m4 = reinterpret_cast<i32x4*>(mv2 + b);
  const i32x4 z = {0, 0, 0, 0};
  // (plain stores: streaming ones made this op 0.13 ms slower at 256 resident sequences and the
  // compaction behind it 0.06 ms faster -- a loss)
  for (int64_t i = tid; i < pairs; i += nthreads) m4[i] = z;
  if (((e - b) & 1) && tid == 0) mv2[e - 1] = make_int2(0, 0);
}

// The dirty map over the chunks [c0, c1) of one owner (a head's segment, or a piece of the rows
// behind the last head): thread `tid` of `nthreads` takes every nthreads-th word.  Old bits (ZF_DIRTY)
// name chunks whose rows from `keep_from` on are cleared; the first `new_chunks` chunks are marked
// for the next call.  Words that straddle the owner's ends are shared with the neighbours, who do
// the same to THEIR bits at the same time: those go through atomics, the others are plain.
__device__ __forceinline__ void dirty_update(uint32_t* map, int2* mv2, int64_t c0, int64_t c1, int64_t new_chunks,
                                             int64_t keep_from, int64_t rows, int bs, bool read_old, int tid, int nthreads) {
  if (c0 >= c1) return;
  const int64_t w0 = c0 >> 5, w1 = (c1 - 1) >> 5;
  const int64_t cn = c0 + new_chunks;                // chunks [c0, cn) hold moves after this call
  for (int64_t w = w0 + tid; w <= w1; w += nthreads) {
    const int64_t lo = max(c0, w << 5), hi = min(c1, (w + 1) << 5);         // this word's chunks of ours
    const uint32_t mask = (hi - lo >= 32) ? 0xFFFFFFFFu : (((1u << (hi - lo)) - 1u) << (lo & 31));
    const int64_t nh = min(hi, cn);
    const uint32_t fresh = nh > lo ? ((nh - lo >= 32) ? 0xFFFFFFFFu : (((1u << (nh - lo)) - 1u) << (lo & 31))) : 0u;
    uint32_t old = 0u;
    if (mask == 0xFFFFFFFFu) {
      if (read_old) old = map[w];
      if (old != fresh || !read_old) map[w] = fresh;
    } else {
      // (only this owner changes these bits, so a plain look at them is safe next to the neighbours'
      // atomics on theirs; in the steady state they already are what they should be: no atomic at all)
      const uint32_t mine = map[w] & mask;
      if (read_old) old = mine;
      if (mine & ~fresh) atomicAnd(&map[w], ~(mine & ~fresh));
      if (fresh & ~mine) atomicOr(&map[w], fresh & ~mine);
    }
    while (old) {                                    // chunks that held moves before this call
      const int bit = __ffs((int)old) - 1;
      old &= old - 1u;
      const int64_t rb = max(((w << 5) + bit) * (int64_t)bs, keep_from);
      const int64_t re = min((((w << 5) + bit) + 1) * (int64_t)bs, rows);
      zero_rows(mv2, rb, re, 0, 1);
    }
  }
}

// first index in [0, cnt) with E[idx] >= x (cnt if none), by the 64 lanes of a wave: two or three
// rounds of 64 probes instead of log2(cnt) dependent loads
__device__ __forceinline__ int wave_lower_bound(const int32_t* __restrict__ E, int cnt, int x, int lane) {
  int lo = 0, len = cnt;                             // the answer lies in [lo, lo + len]
  while (len > WAVE) {
    const int step = (len + WAVE - 1) / WAVE;        // piece p = [lo + p*step, lo + min(len, (p+1)*step))
    const int npieces = (len + step - 1) / step;
    bool below = false;
    if (lane < npieces) below = E[lo + min(len, (lane + 1) * step) - 1] < x;   // the piece's last element
    const int full = __popcll(__ballot(below));      // pieces entirely below x: a prefix (E ascends)
    if (full == npieces) return lo + len;
    const int end = lo + len;
    lo += full * step;
    len = min(step, end - lo);                       // inside piece `full`, whose last element is >= x
  }
  const bool below = lane < len && E[lo + lane] < x;
  return lo + __popcll(__ballot(below));
}

// One workgroup per `heads_per_wg` consecutive heads (plus a few that see to the rows behind the
// last head).  Heads that evict little -- the continual-compression steady state: tens of
// thousands of heads with a move or two each -- take ONE LANE each (up to LANE_HEAD_MAX evictions)
// or ONE WAVE each, the lanes / waves of a workgroup working on different heads at once (a
// workgroup per head was 0.11 ms of dependent-load latency for 65 536 heads); a head beyond
// WAVE_HEAD_MAX evictions takes the whole workgroup.
//
// Regular heads (every evicted index is a real slot, E[cnt-1] < ctx -- always true when
// protected_window >= 1): the walk degenerates to "the j-th hole below new_len = ctx-cnt
// receives the j-th surviving slot of the tail [new_len, ctx), counted from the top".  The
// tail's evicted slots are marked in an LDS bitmap, the tail is scanned top-down with
// ballot prefix sums, and M = #holes below new_len moves are written, coalesced.
// Irregular heads and tails beyond the bitmap use the closed form above.
//
// The same launch leaves behind what execute_cache_moves needs to spread this move list over its
// waves without a pass of its own: the number of move tiles of every workgroup's heads (`plan`).
constexpr int MOVE_BITMAP_WORDS = 15000;             // 480 k tail slots per head in LDS (next to the waves' own bitmaps: < 64 KiB)
constexpr int WAVE_HEAD_MAX = 2048;                  // evictions a single wave takes (64 words of bitmap)
constexpr int LANE_HEAD_MAX = 32;                    // evictions a single lane walks serially

// -DKVC_S2_STAMPS (experiment builds): workgroup 0 prints the 100 MHz wall clock of its phases
#ifdef KVC_S2_STAMPS
#define S2_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) s2_t[k] = wall_clock64(); } while (0)
#else
#define S2_STAMP(k) do { } while (0)
#endif
// compute units of the current device (asked once per device)
static int cu_count() {
  static std::atomic<int> cus[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  int c = cus[dev].load(std::memory_order_relaxed);
  if (c == 0) {
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 1) c = 256;
    cus[dev].store(c, std::memory_order_relaxed);
  }
  return c;
}

// WIDE: the form for a launch of at most one workgroup per CU (a few hundred heads of thousands of evictions each:
// configs[1], [4]) -- compiled for four waves per SIMD, its main loop keeps four chains of dependent loads in flight per
// thread (95 registers).  With more workgroups than CUs two of them share a CU at eight waves per SIMD (64 registers), and
// that occupancy is worth more than the chains (measured: configs[3] 1.13 ms against 1.41 in the wide form).
template <int THREADS, int BITMAP_WORDS, bool WIDE = false>
__global__ __launch_bounds__(THREADS, (WIDE ? 4 : 8)) void schedule_moves_heads_kernel(MovesArgs a) {
#ifdef KVC_S2_STAMPS
  unsigned long long s2_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  S2_STAMP(0);
  constexpr int NWAVES = THREADS / WAVE;
  __shared__ uint32_t bitmap[BITMAP_WORDS];       // 64 KiB for long heads, 4 KiB for short ones (occupancy)
  __shared__ uint32_t wbitmap[NWAVES][WAVE_HEAD_MAX / 32];
  __shared__ uint32_t wave_tot[2][NWAVES];
  __shared__ uint32_t tiles_s[2];
  const int B = a.B, L = a.L, H = a.H, M = a.M, bs = a.bs;
  const BlockDiv bd(bs);
  const int G = B * L * H;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int2* mv2 = reinterpret_cast<int2*>(a.moves);
  const int64_t rows = a.rows;
  const bool with_map = a.dirty != nullptr && a.zero_fill != ZF_NONE;
  // slots of the last head -> total number of rows that belong to heads
  const int lastg = G - 1;
  const int lb = lastg / (L * H), llh = lastg % (L * H);
  const int last_ctx = a.context_lens[((llh / H) * B + lb) * H + (llh % H)];
  const int64_t n_total = (int64_t)a.offs[lastg] + (int64_t)((last_ctx + bs - 1) / bs) * bs;
  if ((int)blockIdx.x >= a.nwg) {                   // rows behind the last head
    if (a.zero_fill == ZF_NONE || n_total >= rows) return;
    const int64_t nt = gridDim.x - a.nwg;
    if (a.zero_fill == ZF_ALL) {
      const int64_t part = (rows - n_total + nt - 1) / nt;        // one contiguous piece per workgroup
      const int64_t b = n_total + (int64_t)(blockIdx.x - a.nwg) * part;
      zero_rows(mv2, b, min(rows, b + part), tid, THREADS);
    } else {                                         // only what the map marks (n_total is a multiple of bs)
      const int64_t c0 = n_total / bs, c1 = (rows + bs - 1) / bs;
      const int64_t part = ((c1 - c0 + nt - 1) / nt + 31) / 32 * 32;
      const int64_t b = c0 + (int64_t)(blockIdx.x - a.nwg) * part;
      dirty_update(a.dirty, mv2, b, min(c1, b + part), 0, 0, rows, bs, true, tid, THREADS);
    }
    return;
  }
  if (tid < 2) tiles_s[tid] = 0;
  __syncthreads();
  const int g_begin = blockIdx.x * a.heads_per_wg, g_end = min(G, g_begin + a.heads_per_wg);
  const int tm_fast = bs <= 32 ? 64 - bs : 32;
  auto head_geometry = [&](int g, int& lbh, int64_t& off, int64_t& seg_end) {
    const int b = g / (L * H), lh = g % (L * H), l = lh / H, hh = lh % H;
    lbh = (l * B + b) * H + hh;
    off = a.offs[g];
    seg_end = min(rows, (g + 1 < G) ? (int64_t)a.offs[g + 1] : n_total);
  };
  auto finish_head = [&](int g, int64_t off, int64_t seg_end, int cnt, int nmoves, int t, int nt) {
    // (t of nt threads: a wave's lanes or the whole workgroup)
    if (t == 0) {
      a.count[g] = nmoves;
      if (a.plan != nullptr && nmoves > 0) {
        atomicAdd(&tiles_s[0], (uint32_t)((nmoves + tm_fast - 1) / tm_fast));
        atomicAdd(&tiles_s[1], (uint32_t)((nmoves + 31) / 32));
      }
    }
    if (a.zero_fill == ZF_ALL) zero_rows(mv2, off + nmoves, min(seg_end, off + cnt), t, nt);
    if (with_map)
      dirty_update(a.dirty, mv2, off / bs, (seg_end + bs - 1) / bs, (min((int64_t)nmoves, max(seg_end - off, (int64_t)0)) + bs - 1) / bs,
                   off + nmoves, rows, bs, a.zero_fill == ZF_DIRTY, t, nt);
  };

  // the zeros behind every head's cnt entries depend on nothing below: they go out first, by the
  // whole workgroup, so that the stores fly while the heads' dependent loads (E, the block table)
  // are still coming in
  if (a.zero_fill == ZF_ALL) {
    for (int g = g_begin; g < g_end; ++g) {
      int lbh; int64_t off, seg_end;
      head_geometry(g, lbh, off, seg_end);
      zero_rows(mv2, off + a.ekc[g], seg_end, tid, THREADS);
    }
  }

  S2_STAMP(1);
  // ---- heads a single lane takes: the reference's own serial walk (kvcompress_eviction_kernels.cu:
  // 256-272), one head per lane -- in the continual-compression steady state a head evicts a block's
  // hanging tokens (a handful of indices) and what a step costs is the chain of dependent loads,
  // which the heads of a workgroup now walk side by side
  for (int g = g_begin + tid; g < g_end; g += THREADS) {
    const int cnt = a.ekc[g];
    if (cnt > LANE_HEAD_MAX) continue;
    int lbh; int64_t off, seg_end;
    head_geometry(g, lbh, off, seg_end);
    const int ctx = a.context_lens[lbh];
    const int32_t* E = a.evicted + off;
    const int32_t* bt = a.block_tables + (int64_t)lbh * M;
    int mc = 0, ec = 0;
    for (int i = 0; i < cnt; ++i) {
      const int src = ctx - 1 - i;
      const int stop = E[cnt - 1 - ec];
      const int dst = E[mc];
      if (dst >= src) break;
      if (src <= stop) { ++ec; continue; }
      if (off + mc < rows)
        mv2[off + mc] = make_int2(bd.phys(bt, dst), bd.phys(bt, src));
      ++mc;
    }
    finish_head(g, off, seg_end, cnt, mc, 0, 1);
  }

  S2_STAMP(2);
  // ---- heads a wave takes
  for (int g = g_begin + w; g < g_end; g += NWAVES) {
    const int cnt = a.ekc[g];
    if (cnt <= LANE_HEAD_MAX || cnt > WAVE_HEAD_MAX) continue;     // (wave-uniform)
    int lbh; int64_t off, seg_end;
    head_geometry(g, lbh, off, seg_end);
    const int ctx = a.context_lens[lbh];
    const int32_t* E = a.evicted + off;
    const int32_t* bt = a.block_tables + (int64_t)lbh * M;
    int nmoves = 0;
    if (cnt > 0 && E[cnt - 1] < ctx) {               // regular
      const int new_len = ctx - cnt;
      nmoves = wave_lower_bound(E, cnt, new_len, lane);
      if (nmoves > 0) {
        uint32_t* bm = wbitmap[w];
        bm[lane] = 0u;                               // WAVE_HEAD_MAX / 32 = 64 words: one per lane
        wave_lds_sync();
        for (int k = nmoves + lane; k < cnt; k += WAVE) {           // evicted slots inside the tail
          const int t = E[k] - new_len;
          atomicOr(&bm[t >> 5], 1u << (t & 31));
        }
        wave_lds_sync();
        int carry = 0;
        for (int base = 0; base < cnt && carry < nmoves; base += WAVE) {   // i-th slot from the top
          const int i = base + lane;
          bool surv = false;
          int slot = 0;
          if (i < cnt) {
            slot = ctx - 1 - i;
            const int t = slot - new_len;
            surv = !((bm[t >> 5] >> (t & 31)) & 1u);
          }
          const unsigned long long bal = __ballot(surv);
          const int j = carry + __popcll(bal & ((1ull << lane) - 1ull));
          if (surv && j < nmoves) {
            const int dst = E[j];
            if (off + j < rows)
              mv2[off + j] = make_int2(bd.phys(bt, dst), bd.phys(bt, slot));
          }
          carry += __popcll(bal);
        }
      }
    } else if (cnt > 0) {
      MoveHead h{E, cnt, ctx};
      for (int j = lane; j < cnt; j += WAVE) {
        const int i = move_iteration(h, j);
        if (i >= 0) {
          const int src = ctx - 1 - i, dst = E[j];
          if (off + j < rows)
            mv2[off + j] = make_int2(bd.phys(bt, dst), bd.phys(bt, src));
        }
      }
      int lo = 0, hi = cnt;                                    // first j that is not a move
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (move_iteration(h, mid) >= 0) lo = mid + 1; else hi = mid; }
      nmoves = lo;
    }
    finish_head(g, off, seg_end, cnt, nmoves, lane, WAVE);
  }

  S2_STAMP(3);
  // ---- heads the whole workgroup takes, one after the other
  for (int g = g_begin; g < g_end; ++g) {
    const int cnt = a.ekc[g];
    if (cnt <= WAVE_HEAD_MAX) continue;              // (uniform)
    int lbh; int64_t off, seg_end;
    head_geometry(g, lbh, off, seg_end);
    const int ctx = a.context_lens[lbh];
    const int32_t* E = a.evicted + off;
    const int32_t* bt = a.block_tables + (int64_t)lbh * M;
    int nmoves = 0;
    const bool regular = E[cnt - 1] < ctx && (cnt + 31) / 32 <= BITMAP_WORDS;
    if (regular) {
      const int new_len = ctx - cnt;
      // holes below new_len = lower_bound(E, new_len): 64 probes per step by every wave (the same answer in each) --
      // three dependent loads for 16 Ki entries where the bisection took fourteen
      S2_STAMP(4);
      nmoves = wave_lower_bound(E, cnt, new_len, lane);
      S2_STAMP(5);
      const int words = (cnt + 31) / 32;
      __syncthreads();                               // (the bitmap of the head before)
      for (int i = tid; i < words; i += THREADS) bitmap[i] = 0;
      __syncthreads();
      for (int k0 = nmoves + tid; k0 < cnt; k0 += THREADS * 4) {   // evicted slots inside the tail, four requested at once
        int t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int k = k0 + u * THREADS; t[u] = k < cnt ? E[k] - new_len : -1; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (t[u] >= 0) atomicOr(&bitmap[t[u] >> 5], 1u << (t[u] & 31));
      }
      __syncthreads();
      S2_STAMP(6);
      uint32_t carry = 0;
      int buf = 0;
      // Q consecutive slots from the top per thread and step.  A step is one barrier and, per move, a chain of two
      // dependent loads (E[j], then the block table): the wave executes in order, so a step costs both round trips
      // (1.45 us x 16 steps of the kernel's 39 us at 16 Ki evictions per head, by its stamps) -- Q chains in flight per
      // thread instead of one, 1 / Q of the steps: the loop 24 -> 14 us at Q = 4 (which needs more than the 64 registers of
      // eight waves per SIMD: the WIDE form only).
      constexpr int Q = WIDE ? 4 : 1;
      for (int base = 0; base < cnt; base += THREADS * Q) {
        const int i0 = base + tid * Q;                          // my slots: the i0-th .. (i0 + Q - 1)-th from the top
        uint32_t sm = 0;                                        // ... and which of them survive
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const int i = i0 + q;
          if (i < cnt) {
            const int t = ctx - 1 - i - new_len;
            if (!((bitmap[t >> 5] >> (t & 31)) & 1u)) sm |= 1u << q;
          }
        }
        const uint32_t c = (uint32_t)__popc(sm);
        const uint32_t inc = wave_inclusive_scan(c);
        if (lane == WAVE - 1) wave_tot[buf][w] = inc;
        lds_barrier();                                          // (the moves of the step before stay in flight)
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < NWAVES; ++q) { const uint32_t cq = wave_tot[buf][q]; if (q < w) woff += cq; tot += cq; }
        const int j0 = (int)(carry + woff + inc - c);           // my first survivor is the j0-th of the head
        int dst[Q], pd[Q], ps[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q)
          if ((sm >> q) & 1u) dst[q] = E[j0 + __popc(sm & ((1u << q) - 1u))];
#pragma unroll
        for (int q = 0; q < Q; ++q)
          if ((sm >> q) & 1u) { pd[q] = bt[bd.blk(dst[q])]; ps[q] = bt[bd.blk(ctx - 1 - i0 - q)]; }
#pragma unroll
        for (int q = 0; q < Q; ++q)
          if ((sm >> q) & 1u) {
            const int j = j0 + __popc(sm & ((1u << q) - 1u)), slot = ctx - 1 - i0 - q;
            if (off + j < rows) mv2[off + j] = make_int2(pd[q] * bs + bd.off(dst[q]), ps[q] * bs + bd.off(slot));
          }
        carry += tot;
        buf ^= 1;
      }
    } else {
      MoveHead h{E, cnt, ctx};
      for (int j = tid; j < cnt; j += THREADS) {
        const int i = move_iteration(h, j);
        if (i >= 0) {
          const int src = ctx - 1 - i, dst = E[j];
          if (off + j < rows)
            mv2[off + j] = make_int2(bd.phys(bt, dst), bd.phys(bt, src));
        }
      }
      int lo = 0, hi = cnt;                                    // first j that is not a move
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (move_iteration(h, mid) >= 0) lo = mid + 1; else hi = mid; }
      nmoves = lo;
    }
    S2_STAMP(7);
    finish_head(g, off, seg_end, cnt, nmoves, tid, THREADS);
    S2_STAMP(8);
  }
  if (a.plan != nullptr) {
    __syncthreads();
    if (tid < 2) a.plan[tid * MOVES_PLAN_WGS + blockIdx.x] = (int32_t)tiles_s[tid];
  }
#ifdef KVC_S2_STAMPS
  S2_STAMP(9);
  if (blockIdx.x == 0 && threadIdx.x == 0)
    printf("S2 stamps (us): zero %.2f lane %.2f wave %.2f geom %.2f lb %.2f mark %.2f loop %.2f finish %.2f plan %.2f\n",
           (s2_t[1] - s2_t[0]) * 0.01, (s2_t[2] - s2_t[1]) * 0.01, (s2_t[3] - s2_t[2]) * 0.01, (s2_t[4] - s2_t[3]) * 0.01,
           (s2_t[5] - s2_t[4]) * 0.01, (s2_t[6] - s2_t[5]) * 0.01, (s2_t[7] - s2_t[6]) * 0.01, (s2_t[8] - s2_t[7]) * 0.01,
           (s2_t[9] - s2_t[8]) * 0.01);
#endif
}

}  // namespace kvc

extern "C" int kvc_count_block_evictions(int32_t* evicted_block_count,
                                         int32_t* evicted_logical_indices,
                                         const int32_t* evicted_kv_offsets,
                                         const int32_t* hanging_token_count,
                                         int32_t total_heads, int64_t total_kvs,
                                         int32_t block_size, int32_t null_value,
                                         kvc_stream_t stream) {
  if (block_size < 1) return kvc::fail_invalid("Unsupported block size: " + std::to_string(block_size));
  if (total_heads <= 0) return KVC_OK;
  const int waves_per_block = 4;
  dim3 grid((total_heads + waves_per_block - 1) / waves_per_block), block(waves_per_block * kvc::WAVE);
  hipLaunchKernelGGL(kvc::count_block_evictions_kernel, grid, block, 0, (hipStream_t)stream,
                     evicted_block_count, evicted_logical_indices, evicted_kv_offsets,
                     hanging_token_count, total_heads, total_kvs, block_size, null_value);
  return kvc::check_launch("count_block_evictions");
}

extern "C" size_t kvc_cache_moves_dirty_map_bytes(int64_t cache_moves_rows, int32_t block_size) {
  if (block_size < 1 || cache_moves_rows < 0) return 0;
  const int64_t chunks = (cache_moves_rows + block_size - 1) / block_size;
  return (size_t)((chunks + 31) / 32 + 1) * 4;
}

extern "C" int kvc_schedule_t1_cache_moves_ex(
    int32_t* cache_moves_idx, int64_t cache_moves_rows, int32_t* cache_moves_count,
    const int32_t* evicted_logical_indices, const int32_t* evicted_kv_count,
    const int32_t* evicted_kv_offsets, const int32_t* block_tables,
    const int32_t* context_lens, int32_t num_seqs, int32_t num_layers, int32_t num_kv_heads,
    int32_t max_num_blocks_per_seq, int32_t block_size, int32_t zero_fill,
    uint32_t* dirty_map, size_t dirty_map_bytes, int32_t* plan_out, kvc_stream_t stream) {
  if (block_size < 1) return kvc::fail_invalid("Unsupported block size: " + std::to_string(block_size));
  if (zero_fill < 0 || zero_fill > 2) return kvc::fail_invalid("schedule_t1_cache_moves: zero_fill must be 0, 1 or 2");
  if (zero_fill == kvc::ZF_DIRTY && dirty_map == nullptr)
    return kvc::fail_invalid("schedule_t1_cache_moves: zero_fill 2 needs the table's dirty map");
  if (dirty_map != nullptr && dirty_map_bytes < kvc_cache_moves_dirty_map_bytes(cache_moves_rows, block_size))
    return kvc::fail_invalid("schedule_t1_cache_moves: dirty map too small");
  const int G = num_seqs * num_layers * num_kv_heads;
  if (G <= 0) return KVC_OK;
  hipStream_t s = (hipStream_t)stream;
  kvc::MovesArgs a;
  a.moves = cache_moves_idx; a.rows = cache_moves_rows; a.count = cache_moves_count;
  a.evicted = evicted_logical_indices; a.ekc = evicted_kv_count; a.offs = evicted_kv_offsets;
  a.block_tables = block_tables; a.context_lens = context_lens;
  a.B = num_seqs; a.L = num_layers; a.H = num_kv_heads; a.M = max_num_blocks_per_seq; a.bs = block_size;
  a.zero_fill = zero_fill; a.dirty = dirty_map; a.plan = plan_out;
  kvc::moves_plan_shape(G, a.nwg, a.heads_per_wg);
  // extra workgroups see to the rows behind the last head's segment
  const int tail_wgs = zero_fill ? 64 : 0;
  const int64_t rows_per_head = cache_moves_rows / G;
  if (rows_per_head >= 8192 && a.nwg <= kvc::cu_count())
    hipLaunchKernelGGL((kvc::schedule_moves_heads_kernel<1024, kvc::MOVE_BITMAP_WORDS, true>), dim3(a.nwg + tail_wgs), dim3(1024), 0, s, a);
  else if (rows_per_head >= 8192)
    hipLaunchKernelGGL((kvc::schedule_moves_heads_kernel<1024, kvc::MOVE_BITMAP_WORDS>), dim3(a.nwg + tail_wgs), dim3(1024), 0, s, a);
  else
    hipLaunchKernelGGL((kvc::schedule_moves_heads_kernel<256, 1024>), dim3(a.nwg + tail_wgs), dim3(256), 0, s, a);
  return kvc::check_launch("schedule_t1_cache_moves");
}

extern "C" int kvc_schedule_t1_cache_moves(
    int32_t* cache_moves_idx, int64_t cache_moves_rows, int32_t* cache_moves_count,
    const int32_t* evicted_logical_indices, const int32_t* evicted_kv_count,
    const int32_t* evicted_kv_offsets, const int32_t* block_tables,
    const int32_t* context_lens, int32_t num_seqs, int32_t num_layers, int32_t num_kv_heads,
    int32_t max_num_blocks_per_seq, int32_t block_size, int32_t zero_fill,
    kvc_stream_t stream) {
  return kvc_schedule_t1_cache_moves_ex(cache_moves_idx, cache_moves_rows, cache_moves_count, evicted_logical_indices,
                                        evicted_kv_count, evicted_kv_offsets, block_tables, context_lens, num_seqs,
                                        num_layers, num_kv_heads, max_num_blocks_per_seq, block_size,
                                        zero_fill ? 1 : 0, nullptr, 0, nullptr, stream);
}
