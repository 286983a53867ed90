// A6 execute_cache_moves: paged K/V compaction for gfx950 (MI355X).
//
// Reference kernel (csrc/kvcompress_eviction_kernels.cu:359-435): 128 threads per
// (seq,layer), each thread walks whole KVs serially and copies 2-byte elements one at a
// time at a 32 B stride.  Measured on MI355X (tools/gather_bw.hip): randomly placed
// chunks copy at full HBM speed only from 128 B upward; 16 B chunks reach a tenth of it.
// A KV slot is 16 B pieces in K ([hd*e/16 rows][bs slots][16 B]) and single elements in V
// ([hd rows][bs slots * e B]), so ANY slot-granular copy is in the slow regime.  This
// file therefore never touches HBM in less than 1 KiB per wave instruction:
//
//   * the unit of work is a "run": the consecutive moves of one head into ONE destination
//     block (<= bs moves).  One 64-lane wave owns a run.  It streams the whole K and V
//     destination block into registers (lane l holds the 16 B pieces l, l+64, ... of the
//     block image -> perfectly contiguous 1 KiB per load), streams each source block the
//     run reads the same way, patches the moved slots register-to-register, and streams
//     the destination block back.
//   * V: a slot is one element of every row; rows are RB = bs*e bytes, i.e. RB/16 lanes
//     per row.  The element is extracted from the source piece (scalar-selected dword),
//     handed to the lane that owns the destination piece with a DPP quad permute and
//     inserted with one v_bfi.
//   * K: a slot is one whole 16 B piece per K row; the piece moves between lanes of the
//     same bs-lane group (ds_bpermute with a uniform source lane).
//   * slot numbers are wave-uniform (v_readlane), all control flow is scalar.
//
// Rewriting whole blocks needs the wave to be the only writer of that block.  A planning
// pass counts the runs that target each physical block (one byte per block, zeroed per
// call); a run takes the block path only if it is the sole claimant -- always the case for
// schedules produced by A5 (ascending dst, a block belongs to one head).  Any other run
// (unsorted lists, blocks shared between heads) is copied slot-wise with 16 B / element
// accesses, which is correct for every independent move list, the only case the
// reference defines (kvcompress_eviction_kernels.cu:358).
//
// Work distribution: a tile is 64 - bs consecutive moves of one head; a wave handles, in
// order, the runs that START in its tile.  Move counts live in device memory, so
// tiny planning kernels (tiles per head, exclusive scan, claims) let a fixed persistent
// grid walk the tiles without host synchronisation.
#include "kvc_common.h"
#include <atomic>
#include "../../include/kvc_mi355x.h"

#ifndef KVC_RUNS_WGS
#define KVC_RUNS_WGS 4     // persistent workgroups per CU of the compaction kernel
#endif

namespace kvc {

constexpr int KVC_TM_GENERIC = 32;    // moves per tile of the generic (byte-wise) kernel

// native 16 B vector: HIP's uint4 is a struct whose copies lower to memcpy through a
// private alloca, which the compiler then parks in LDS
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------- planning
// A move list's plan (kvc_common.h: moves_plan_shape) = the number of move tiles held by the heads
// of each of <= MOVES_PLAN_WGS groups of consecutive heads.  schedule_t1_cache_moves leaves it behind
// for its own lists; for any other list this kernel makes it: a thread per group, the other
// workgroups zero a slice of the claim table (one launch instead of a memset + two kernels).
__global__ __launch_bounds__(256) void compact_plan_kernel(int32_t* __restrict__ wg_tiles,
                                                           const int32_t* __restrict__ count, int G,
                                                           int tm, int nwg, int heads_per_wg, int plan_blocks,
                                                           u32x4* __restrict__ claims16, int64_t claim_vecs) {
  if ((int)blockIdx.x >= plan_blocks) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int64_t i = (int64_t)(blockIdx.x - plan_blocks) * 256 + threadIdx.x; i < claim_vecs;
         i += (int64_t)(gridDim.x - plan_blocks) * 256)
      claims16[i] = z;
    return;
  }
  // a wave per group when the groups are wide (coalesced reads of `count`), else a thread per group
  if (heads_per_wg >= 16) {
    const int lane = lane_id();
    for (int wg = blockIdx.x * 4 + threadIdx.x / WAVE; wg < nwg; wg += plan_blocks * 4) {
      const int gb = wg * heads_per_wg, ge = min(G, gb + heads_per_wg);
      uint32_t t = 0;
      for (int g = gb + lane; g < ge; g += WAVE) t += (uint32_t)((count[g] + tm - 1) / tm);
      t = wave_reduce_sum(t);
      if (lane == 0) wg_tiles[wg] = (int32_t)t;
    }
  } else {
    for (int wg = blockIdx.x * 256 + threadIdx.x; wg < nwg; wg += plan_blocks * 256) {
      const int gb = wg * heads_per_wg, ge = min(G, gb + heads_per_wg);
      int t = 0;
      for (int g = gb; g < ge; ++g) t += (count[g] + tm - 1) / tm;
      wg_tiles[wg] = t;
    }
  }
}

// The tiles [t_begin, t_end) of wave `wid` of `nw`, the head g that holds tile t_begin and the
// index g_first of that head's first tile -- found by the wave's 64 lanes with three scans (the
// groups' sums, 64 at a time per lane; the groups of the lane that holds t_begin; the heads of the
// group) instead of a binary search over a prefix array somebody had to scan first.
// false: no tiles for this wave.
__device__ __forceinline__ bool locate_tiles(const int32_t* __restrict__ wg_tiles, int nwg, int heads_per_wg,
                                             const int32_t* __restrict__ count, int G, int tm, int wid, int nw,
                                             int lane, int& t_begin, int& t_end, int& g_out, int& g_first_out) {
  const int per = (nwg + WAVE - 1) / WAVE;
  const int e0 = lane * per, e1 = min(nwg, e0 + per);
  uint32_t s1 = 0;
  for (int e = e0; e < e1; ++e) s1 += (uint32_t)wg_tiles[e];
  const uint32_t inc1 = wave_inclusive_scan(s1);
  const int total = (int)__shfl((int)inc1, 63, 64);
  t_begin = (int)((int64_t)total * wid / nw);
  t_end = (int)((int64_t)total * (wid + 1) / nw);
  if (t_begin >= t_end) return false;
  const int l1 = __ffsll((long long)__ballot((int)inc1 > t_begin)) - 1;
  int base = __shfl((int)(inc1 - s1), l1, 64);
  const int eb = l1 * per;
  const uint32_t s2 = (lane < per && eb + lane < nwg) ? (uint32_t)wg_tiles[eb + lane] : 0u;
  const uint32_t inc2 = wave_inclusive_scan(s2);
  const unsigned long long m2 = __ballot(base + (int)inc2 > t_begin);
  // a plan that does not belong to these counts (the caller's contract, include/kvc_mi355x.h: both bindings make it
  // impossible through version checks; a raw C-ABI caller vouches): dropping the wave's moves quietly would leave
  // a cache that is neither the old nor the new one -- the launch faults instead (the next call on the stream
  // reports the error)
  if (!m2) __builtin_trap();
  const int l2 = __ffsll((long long)m2) - 1;
  base += __shfl((int)(inc2 - s2), l2, 64);
  const int gb0 = (eb + l2) * heads_per_wg, ge = min(G, gb0 + heads_per_wg);
  for (int gb = gb0; gb < ge; gb += WAVE) {
    const uint32_t s3 = gb + lane < ge ? (uint32_t)((count[gb + lane] + tm - 1) / tm) : 0u;
    const uint32_t inc3 = wave_inclusive_scan(s3);
    const unsigned long long m3 = __ballot(base + (int)inc3 > t_begin);
    if (m3) {
      const int l3 = __ffsll((long long)m3) - 1;
      g_out = gb + l3;
      g_first_out = base + __shfl((int)(inc3 - s3), l3, 64);
      return true;
    }
    base += __shfl((int)inc3, 63, 64);
  }
  __builtin_trap();                                  // (the plan lists more tiles than the counts hold: see above)
  return false;
}

// one wave per contiguous range of tiles: every run start claims its destination block
// (claims: 4 one-byte counters per word; a block has at most bs <= 255 claimants)
__global__ __launch_bounds__(256) void compact_plan_claims_kernel(
    uint32_t* __restrict__ claims, const int32_t* __restrict__ moves,
    const int32_t* __restrict__ count, const int32_t* __restrict__ offs,
    const int32_t* __restrict__ wg_tiles, int nwg, int heads_per_wg, int G, int bs, int tm) {
  const int lane = lane_id();
  const int nw = gridDim.x * (blockDim.x / WAVE);
  const int wv = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  int tb, te, g, g_first;
  if (!locate_tiles(wg_tiles, nwg, heads_per_wg, count, G, tm, wv, nw, lane, tb, te, g, g_first)) return;
  int cnt = count[g];
  int g_next = g_first + (cnt + tm - 1) / tm;
  for (int t = tb; t < te; ++t) {
    while (t >= g_next && g + 1 < G) { ++g; g_first = g_next; cnt = count[g]; g_next = g_first + (cnt + tm - 1) / tm; }
    const int j = (t - g_first) * tm + lane;
    const int2* mv = reinterpret_cast<const int2*>(moves) + offs[g];
    const bool in = lane < tm && j < cnt;
    const int dblk = in ? mv[j].x / bs : -1;
    // destination block of the previous move: lane - 1 holds it, except for the tile's first lane
    int prev = __shfl_up(dblk, 1, 64);
    if (lane == 0) prev = (in && j > 0) ? mv[j - 1].x / bs : -1;
    if (in && prev != dblk) atomicAdd(&claims[dblk >> 2], 1u << (8 * (dblk & 3)));
  }
}

// ------------------------------------------------------------------------- block path
// v_bfi_b32: (mask & a) | (~mask & b)
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) {
  return (mask & a) | (~mask & b);
}

// NPL = 16 B pieces per lane of one block image
template <int NPL>
struct BlockImg { u32x4 p[NPL]; };

template <int NPL>
__device__ __forceinline__ void img_load(BlockImg<NPL>& b, const uint8_t* base, int lane) {
#pragma unroll
  for (int i = 0; i < NPL; ++i)   // block images are touched once: non-temporal (measured -15 % kernel time)
    b.p[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + ((int64_t)i * 64 + lane) * 16));
}
template <int NPL>
__device__ __forceinline__ void img_store(const BlockImg<NPL>& b, uint8_t* base, int lane) {
#pragma unroll
  for (int i = 0; i < NPL; ++i)
    __builtin_nontemporal_store(b.p[i], reinterpret_cast<u32x4*>(base + ((int64_t)i * 64 + lane) * 16));
}

// asynchronous copy of one block image HBM -> this wave's LDS buffer (global_load_lds_dwordx4:
// 64 lanes x 16 B = 1 KiB per instruction, no VGPRs, LDS destination = uniform base + lane*16)
template <int NPL>
__device__ __forceinline__ void img_load_lds(uint8_t* lds_base, const uint8_t* base, int lane) {
#pragma unroll
  for (int i = 0; i < NPL; ++i)
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(base + ((int64_t)i * 64 + lane) * 16),
        (__attribute__((address_space(3))) void*)(lds_base + i * 1024), 16, 0, 0);
}

template <int E>
__device__ __forceinline__ uint32_t lds_elem(const uint8_t* p) {
  if constexpr (E == 1) return *p;
  else if constexpr (E == 2) return *reinterpret_cast<const uint16_t*>(p);
  else return *reinterpret_cast<const uint32_t*>(p);
}

// HD = head size, BS = block size, E = element bytes, WPB = waves per workgroup.  Waves are
// independent (no workgroup barrier); every wave owns a contiguous range of tiles of
// tm = 64 - bs consecutive moves and handles the runs that START in its tiles, in order.
//
// What bounds this kernel is the number of DEPENDENT memory round trips per run, not its
// instruction count: with the chip's queues full a round trip costs 5-15 us whatever its size
// (tools/blockmix_bw.hip: the bare access pattern -- read destination + source images, write the
// destination back, one round trip per run -- sustains 6.0-6.2 TB/s from 8 waves per CU).  So a
// run is ONE round trip: the destination K / V images are streamed into REGISTERS (lane l holds
// pieces l, l+64, ...) and, in the same breath, the source images of up to two source blocks
// straight into this wave's two LDS slots (global_load_lds, asynchronous, no registers); a
// source block that is already resident (it fed the previous run) is not fetched again; the next
// tile's moves and the claim bytes of their destination blocks are fetched one tile ahead, and
// the head walk is incremental.  Then the patch is a handful of lane-masked LDS reads -- there is
// no per-move loop:
//  * slot table: the moves of a (run, source block) segment live one per lane; ONE ds_permute
//    scatters "source slot" to the lane numbered "destination slot", two shuffles pack 4
//    consecutive entries per lane, and every lane fetches the entries of the slots it owns
//    (crossbar only, no LDS memory, built while the loads are in flight);
//  * K: a slot is a whole 16 B piece per K row, owned by the lanes with lane % BS == slot: one
//    masked ds_read_b128 per image piece from [row][table slot];
//  * V: a slot is one element of every row; a lane owns EP = 16/E consecutive slots of one row
//    per piece: per destination dword, 4/E ds_reads of the elements + one v_bfi under the
//    lane's mask;
//  * metrics / positions: 16-lane rows, ds_bpermute + select.
// A source block that contributes <= KVC_CHUNK_MAX slots (high compression: survivors are
// sparse) has only those 16 B K pieces fetched, straight from HBM.
// Tried and dropped (profiles/r2_compact_variants.md): not reading the destination K image and
// storing only the moved 16 B pieces under the lane mask -- FETCH_SIZE falls by 24 %, the kernel
// gets 11 % SLOWER (partial-line writes are read-modify-writes further down).
#ifndef KVC_CHUNK_MAX
#define KVC_CHUNK_MAX 3
#endif
// A run of <= KVC_DSTK_SPARSE_MAX moves (the continual-compression steady state: ONE slot moved into
// a block) does not read the destination's K image at all and stores only the moved 16 B pieces
// (16 x hd*e/16 of them per slot) -- the V image, where every 32 B sector holds an element of every
// slot, is still rewritten whole.  0 = off.  Measured at 256 resident sequences in the steady state
// (66 k single-move runs per step): compaction kernel 0.278 -> 0.216 ms.  (For the 8-move runs of a
// bulk eviction the same idea lost: 128 partial-line writes per run against one 4 KiB image,
// profiles/r2_compact_variants.md; runs of two and more keep the image path.)
#ifndef KVC_DSTK_SPARSE_MAX
#define KVC_DSTK_SPARSE_MAX 1
#endif
template <int HD, int BS, int E, int WPB>
__global__ __launch_bounds__(64 * WPB) void compact_runs_kernel(
    uint8_t* __restrict__ k_cache, uint8_t* __restrict__ v_cache, float* __restrict__ metrics,
    int32_t* __restrict__ positions, const int32_t* __restrict__ moves,
    const int32_t* __restrict__ count, const int32_t* __restrict__ offs,
    const uint32_t* __restrict__ claims, const int32_t* __restrict__ wg_tiles, int nwg, int heads_per_wg,
    int G, int tm) {
  static_assert(BS <= 32 && (BS & (BS - 1)) == 0, "block size must be a power of two <= 32");
  constexpr int BLOCK_BYTES = HD * BS * E;
  constexpr int NPL = BLOCK_BYTES / 16 / 64;                 // 16 B pieces per lane
  static_assert(BLOCK_BYTES % (16 * 64) == 0, "block image must be a multiple of 1 KiB");
  constexpr int KR = HD * E / 16;                            // K rows (16 B pieces per slot)
  constexpr int RB = BS * E;                                 // bytes per V row
  constexpr int PR = RB / 16;                                // pieces (lanes) per V row
  constexpr int EP = 16 / E;                                 // slots per V piece
  constexpr int PER = 4 / E;                                 // elements per dword
  constexpr int TW = EP / 4;                                 // packed table dwords per lane
  static_assert(PR >= 1 && PR * EP == BS, "V rows must be whole pieces");
  constexpr uint32_t EMASK = E == 4 ? 0xFFFFFFFFu : ((1u << (8 * E)) - 1u);
  constexpr int SLOT_BYTES = 2 * BLOCK_BYTES;                // one source block: [K image | V image]
  __shared__ __attribute__((aligned(16))) uint8_t lds_s[WPB][2 * SLOT_BYTES];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint8_t* const lds_w = lds_s[wib];
  const int slot_k = lane & (BS - 1);                        // K / metric slot this lane owns
  const int kgrp16 = (lane & ~(BS - 1)) * 16;                // byte offset of this lane's K row group
  const int vrow = (lane / PR) * RB;                         // byte offset of this lane's V row
  const int vfirst = (lane & (PR - 1)) * EP;                 // first slot of this lane's V pieces
  const int nw = gridDim.x * WPB;
  const int wid = blockIdx.x * WPB + wib;
  int t_begin, t_end, g, g_first;
  if (!locate_tiles(wg_tiles, nwg, heads_per_wg, count, G, tm, wid, nw, lane, t_begin, t_end, g, g_first)) return;

  // ---- tile fetch: incremental head walk + this lane's move (lane q <-> move jbase + q: one
  // look-behind, tm moves of the tile, BS-1 look-ahead)
  int g_cnt = count[g];
  int g_next = g_first + (g_cnt + tm - 1) / tm;
  auto fetch_moves = [&](int t, int& jbase_o, int& j1_o, int& mvx_o, int& mvy_o) {
    while (t >= g_next && g + 1 < G) { ++g; g_first = g_next; g_cnt = count[g]; g_next = g_first + (g_cnt + tm - 1) / tm; }
    const int cnt = g_cnt;
    const int j0 = (t - g_first) * tm;
    const int2* __restrict__ mv = reinterpret_cast<const int2*>(moves) + offs[g];
    jbase_o = j0 - 1;
    j1_o = min(cnt, j0 + tm);
    const int jl = j0 - 1 + lane;
    mvx_o = -1; mvy_o = -1;
    if (jl >= 0 && jl < cnt) { const int2 m = mv[jl]; mvx_o = m.x; mvy_o = m.y; }
  };
  // claim word of a move's destination block (4 one-byte run counters; 1 = the run is the block's
  // only writer).  The byte is extracted at the point of use so that issuing the load never waits.
  // claims == nullptr: the list comes from schedule_t1_cache_moves, which writes every destination
  // block from ONE run (ascending destinations, a block belongs to one head) -- no table to read.
  const bool counted = claims != nullptr;
  auto claim_word = [&](int mx) -> uint32_t {
    uint32_t w = 0u;
    if (counted && mx >= 0) w = claims[(mx / BS) >> 2];
    return w;
  };
  auto claim_of = [&](uint32_t w, int mx) -> int { return counted ? (int)((w >> (8 * ((mx / BS) & 3))) & 0xFFu) : 1; };

  BlockImg<NPL> kd, vd;
#pragma unroll
  for (int i = 0; i < NPL; ++i) { kd.p[i] = u32x4{0u, 0u, 0u, 0u}; vd.p[i] = u32x4{0u, 0u, 0u, 0u}; }
  float md = 0.f;
  int pd = 0;
  // the two LDS source slots: block held, whether its K image is there too (a "chunky" block has
  // only its V image fetched), its metric / position row (lanes < BS)
  int held0 = -1, held1 = -1;
  bool kval0 = false, kval1 = false;
  float ms0 = 0.f, ms1 = 0.f;
  int ps0 = 0, ps1 = 0;
  int last_slot = 1;

  int jbase, j1, mvx, mvy;
  fetch_moves(t_begin, jbase, j1, mvx, mvy);
  uint32_t clw = claim_word(mvx);
  for (int t = t_begin; t < t_end; ++t) {
    // the next tile's moves are requested now and land during this tile's first round trip
    int n_jbase = 0, n_j1 = 0, n_mvx = -1, n_mvy = -1;
    uint32_t n_clw = 0u;
    const int clm = claim_of(clw, mvx);
    const bool has_next = t + 1 < t_end;
    if (has_next) fetch_moves(t + 1, n_jbase, n_j1, n_mvx, n_mvy);
    bool clm_pending = has_next;

    const int jl = jbase + lane;
    const int myblk = mvx >= 0 ? mvx / BS : -2;            // destination block of this lane's move
    const int mysb = mvy >= 0 ? mvy / BS : -2;             // source block of this lane's move
    const int left = __shfl_up(myblk, 1, 64);
    const int sleft = __shfl_up(mysb, 1, 64);
    // run starts inside the tile
    unsigned long long starts = __ballot(jl > jbase && jl < j1 && (jl == 0 || myblk != left));

    while (starts) {
      const int qr = __ffsll((long long)starts) - 1;        // lane of the run's first move
      starts &= starts - 1;
      const int dblk = __builtin_amdgcn_readlane(myblk, qr);
      // run end (exclusive): first later lane whose destination block differs (lanes past
      // the head's last move hold block -2)
      int qe;
      {
        const unsigned long long diff = __ballot(myblk != dblk) & ~((2ull << qr) - 1ull);
        qe = diff ? __ffsll((long long)diff) - 1 : 64;
      }
      const bool sole = __builtin_amdgcn_readlane(clm, qr) == 1;
      uint8_t* kd_p = k_cache + (int64_t)dblk * BLOCK_BYTES;
      uint8_t* vd_p = v_cache + (int64_t)dblk * BLOCK_BYTES;
      if (sole) {
        // a run that overwrites all BS slots (distinct dst slots of one block) leaves nothing
        // of the old block alive: no read-modify-write, the block is only written
        bool want_dst = (qe - qr) != BS;
        const bool sparse_k = (qe - qr) <= KVC_DSTK_SPARSE_MAX;     // (wave-uniform)
        bool ktaken = false;
        // segments = maximal stretches of the run fed by one source block
        unsigned long long segs = __ballot(lane >= qr && lane < qe && (lane == qr || mysb != sleft));
        while (segs) {
          // ---- up to two segments per round trip
          const int qa = __ffsll((long long)segs) - 1;
          segs &= segs - 1;
          const int qa_end = segs ? __ffsll((long long)segs) - 1 : qe;
          const int sa = __builtin_amdgcn_readlane(mysb, qa);
          const bool hasb = segs != 0ull;
          int qb = qe, qb_end = qe, sb = -3;
          if (hasb) {
            qb = qa_end;
            segs &= segs - 1;
            qb_end = segs ? __ffsll((long long)segs) - 1 : qe;
            sb = __builtin_amdgcn_readlane(mysb, qb);
          }
          // LDS slot of each: the one that already holds the block, else the one not used last
          const int slot_a = sa == held0 ? 0 : (sa == held1 ? 1 : 1 - last_slot);
          const int slot_b = 1 - slot_a;
          const bool new_a = (slot_a == 0 ? held0 : held1) != sa;
          const bool kv_a = !new_a && (slot_a == 0 ? kval0 : kval1);
          const bool chunky_a = !kv_a && (qa_end - qa) <= KVC_CHUNK_MAX;
          const bool new_b = hasb && (slot_b == 0 ? held0 : held1) != sb;
          const bool kv_b = hasb && !new_b && (slot_b == 0 ? kval0 : kval1);
          const bool chunky_b = hasb && !kv_b && (qb_end - qb) <= KVC_CHUNK_MAX;
          uint8_t* const la = lds_w + slot_a * SLOT_BYTES;
          uint8_t* const lb = lds_w + slot_b * SLOT_BYTES;
          const uint8_t* ka_p = k_cache + (int64_t)sa * BLOCK_BYTES;
          const uint8_t* kb_p = k_cache + (int64_t)sb * BLOCK_BYTES;
          // ---- issue every load of the round trip
          if (want_dst) {
            if (!sparse_k) img_load<NPL>(kd, kd_p, lane);
            img_load<NPL>(vd, vd_p, lane);
            if (lane < BS) { md = metrics[(int64_t)dblk * BS + lane]; pd = positions[(int64_t)dblk * BS + lane]; }
            want_dst = false;
          }
          // the LDS reads of earlier patches have returned before a slot is refilled
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          // per physical slot (no value loaded here is consumed before the common wait below; the
          // metric / position rows go straight into the slot's own registers)
          const bool a0 = slot_a == 0;
          const int blk0 = a0 ? sa : sb, blk1 = a0 ? sb : sa;
          const bool new0 = a0 ? new_a : new_b, new1 = a0 ? new_b : new_a;
          const bool ldk_a = !chunky_a && !kv_a, ldk_b = hasb && !chunky_b && !kv_b;
          const bool ldk0 = a0 ? ldk_a : ldk_b, ldk1 = a0 ? ldk_b : ldk_a;
          if (new0) {
            img_load_lds<NPL>(lds_w + BLOCK_BYTES, v_cache + (int64_t)blk0 * BLOCK_BYTES, lane);
            if (lane < BS) { ms0 = metrics[(int64_t)blk0 * BS + lane]; ps0 = positions[(int64_t)blk0 * BS + lane]; }
            held0 = blk0; kval0 = false;
          }
          if (ldk0) { img_load_lds<NPL>(lds_w, k_cache + (int64_t)blk0 * BLOCK_BYTES, lane); kval0 = true; }
          if (new1) {
            img_load_lds<NPL>(lds_w + SLOT_BYTES + BLOCK_BYTES, v_cache + (int64_t)blk1 * BLOCK_BYTES, lane);
            if (lane < BS) { ms1 = metrics[(int64_t)blk1 * BS + lane]; ps1 = positions[(int64_t)blk1 * BS + lane]; }
            held1 = blk1; kval1 = false;
          }
          if (ldk1) { img_load_lds<NPL>(lds_w + SLOT_BYTES, k_cache + (int64_t)blk1 * BLOCK_BYTES, lane); kval1 = true; }
          // ---- slot tables (crossbar only; overlap the loads).  Lane s < BS receives the entry
          // of destination slot s: 0x80 | source slot, or 0; then every lane fetches the entry of
          // its K slot and the packed entries of its V slots.
          uint32_t tk_a, tk_b = 0u, tw_a[TW], tw_b[TW];
          auto slot_table = [&](int q0, int q1, uint32_t& tk, uint32_t (&tw)[TW]) {
            const bool in_seg = lane >= q0 && lane < q1;
            const int tblv = __builtin_amdgcn_ds_permute((in_seg ? (mvx & (BS - 1)) : 63) * 4,
                                                         in_seg ? (0x80 | (mvy & (BS - 1))) : 0);
            tk = (uint32_t)__builtin_amdgcn_ds_bpermute(slot_k * 4, tblv);
            const uint32_t t1 = (uint32_t)tblv | ((uint32_t)__shfl_down(tblv, 1, 64) << 8);
            const uint32_t t2 = t1 | ((uint32_t)__shfl_down((int)t1, 2, 64) << 16);   // entries s .. s+3
#pragma unroll
            for (int w = 0; w < TW; ++w)
              tw[w] = (uint32_t)__builtin_amdgcn_ds_bpermute((vfirst + 4 * w) * 4, (int)t2);
          };
          slot_table(qa, qa_end, tk_a, tw_a);
#pragma unroll
          for (int w = 0; w < TW; ++w) tw_b[w] = 0u;
          if (hasb) slot_table(qb, qb_end, tk_b, tw_b);
          // sparse source blocks: just the needed 16 B K pieces, from HBM into registers of their own
          // (merged after the wait: a load on top of the in-flight destination image would be
          // serialised behind it)
          BlockImg<NPL> kc_a, kc_b;
#pragma unroll
          for (int i = 0; i < NPL; ++i) { kc_a.p[i] = u32x4{0u, 0u, 0u, 0u}; kc_b.p[i] = u32x4{0u, 0u, 0u, 0u}; }
          if (chunky_a && (tk_a & 0x80u)) {
#pragma unroll
            for (int i = 0; i < NPL; ++i)
              kc_a.p[i] = *reinterpret_cast<const u32x4*>(ka_p + i * 1024 + kgrp16 + (int)(tk_a & 0x7Fu) * 16);
          }
          if (chunky_b && (tk_b & 0x80u)) {
#pragma unroll
            for (int i = 0; i < NPL; ++i)
              kc_b.p[i] = *reinterpret_cast<const u32x4*>(kb_p + i * 1024 + kgrp16 + (int)(tk_b & 0x7Fu) * 16);
          }
          // ---- everything has landed
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          // ---- patch
          auto patch = [&](const uint8_t* ls, uint32_t tk, const uint32_t (&tw)[TW], bool chunky,
                           const BlockImg<NPL>& kc, float ms, int ps) {
            const bool take = (tk & 0x80u) != 0u;
            ktaken = ktaken || take;
            const int ksl = kgrp16 + (int)(tk & 0x7Fu) * 16;
            if (chunky) {
              if (take) {
#pragma unroll
                for (int i = 0; i < NPL; ++i) kd.p[i] = kc.p[i];
              }
            } else if (take) {
#pragma unroll
              for (int i = 0; i < NPL; ++i)
                kd.p[i] = *reinterpret_cast<const u32x4*>(ls + i * 1024 + ksl);
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) {                     // destination dword w of every V piece
              uint32_t mask = 0u;
              int addr[PER];
#pragma unroll
              for (int q = 0; q < PER; ++q) {
                const int e = w * PER + q;                    // slot vfirst + e
                const uint32_t tb = (tw[e >> 2] >> (8 * (e & 3))) & 0xFFu;
                addr[q] = vrow + (int)(tb & 0x7Fu) * E;
                mask |= (tb & 0x80u) ? (EMASK << (8 * E * q)) : 0u;
              }
              if (__ballot(mask != 0u) == 0ull) continue;     // nobody patches this dword
#pragma unroll
              for (int i = 0; i < NPL; ++i) {
                uint32_t val = 0u;
#pragma unroll
                for (int q = 0; q < PER; ++q)
                  val |= lds_elem<E>(ls + BLOCK_BYTES + i * 1024 + addr[q]) << (8 * E * q);
                vd.p[i][w] = bfi(mask, val, vd.p[i][w]);
              }
            }
            const int mval = __builtin_amdgcn_ds_bpermute((int)(tk & 0x7Fu) * 4, __builtin_bit_cast(int, ms));
            const int pval = __builtin_amdgcn_ds_bpermute((int)(tk & 0x7Fu) * 4, ps);
            md = take ? __builtin_bit_cast(float, mval) : md;
            pd = take ? pval : pd;
          };
          patch(la, tk_a, tw_a, chunky_a, kc_a, slot_a == 0 ? ms0 : ms1, slot_a == 0 ? ps0 : ps1);
          if (hasb) patch(lb, tk_b, tw_b, chunky_b, kc_b, slot_b == 0 ? ms0 : ms1, slot_b == 0 ? ps0 : ps1);
          last_slot = hasb ? slot_b : slot_a;
        }
        // the next tile's moves have landed with the round trip above: request the claim words of
        // their destination blocks (behind the last LDS read of the patch: the compiler drains
        // every outstanding load in front of the first LDS read that follows an LDS-DMA)
        if (clm_pending) { n_clw = claim_word(n_mvx); clm_pending = false; }
        if (sparse_k) {
          if (ktaken) {
#pragma unroll
            for (int i = 0; i < NPL; ++i)
              *reinterpret_cast<u32x4*>(kd_p + ((int64_t)i * 64 + lane) * 16) = kd.p[i];
          }
        } else {
          img_store<NPL>(kd, kd_p, lane);
        }
        img_store<NPL>(vd, vd_p, lane);
        if (lane < BS) { metrics[(int64_t)dblk * BS + lane] = md; positions[(int64_t)dblk * BS + lane] = pd; }
      } else {
        // shared destination block: slot-wise, correct for any independent move list
        for (int q = qr; q < qe; ++q) {
          const int sy = __builtin_amdgcn_readlane(mvy, q), dx = __builtin_amdgcn_readlane(mvx, q);
          const int64_t sb = (int64_t)(sy / BS) * BLOCK_BYTES, db = (int64_t)dblk * BLOCK_BYTES;
          if (lane == 0) { metrics[dx] = metrics[sy]; positions[dx] = positions[sy]; }
          for (int r = lane; r < KR; r += 64)
            *reinterpret_cast<u32x4*>(k_cache + db + ((int64_t)r * BS + dx % BS) * 16) =
                *reinterpret_cast<const u32x4*>(k_cache + sb + ((int64_t)r * BS + sy % BS) * 16);
          for (int dd = lane; dd < HD; dd += 64) {
            const int64_t so = sb + (int64_t)dd * RB + (sy % BS) * E;
            const int64_t dof = db + (int64_t)dd * RB + (dx % BS) * E;
            if constexpr (E == 1) v_cache[dof] = v_cache[so];
            else if constexpr (E == 2) *reinterpret_cast<uint16_t*>(v_cache + dof) = *reinterpret_cast<const uint16_t*>(v_cache + so);
            else *reinterpret_cast<uint32_t*>(v_cache + dof) = *reinterpret_cast<const uint32_t*>(v_cache + so);
          }
        }
      }
    }
    if (clm_pending) n_clw = claim_word(n_mvx);
    jbase = n_jbase; j1 = n_j1; mvx = n_mvx; mvy = n_mvy; clw = n_clw;
  }
}

// ------------------------------------------------------------------------- generic path
// any block_size / head_size / element size / K vector width x: byte-granular copies,
// one thread per byte.
__global__ __launch_bounds__(256) void compact_generic_kernel(
    uint8_t* __restrict__ k_cache, uint8_t* __restrict__ v_cache, float* __restrict__ metrics,
    int32_t* __restrict__ positions, const int32_t* __restrict__ moves,
    const int32_t* __restrict__ count, const int32_t* __restrict__ offs,
    const int32_t* __restrict__ wg_tiles, int nwg, int heads_per_wg, int G, int bs, int hd, int e, int x) {
  __shared__ int loc_s[5];
  const int tid = threadIdx.x;
  const int kgroups = hd / x;                 // K vectors per slot
  const int kvec_bytes = x * e;
  const int64_t block_bytes = (int64_t)hd * bs * e;
  // a contiguous range of tiles per workgroup, found by its first wave
  if (tid < WAVE) {
    int tb = 0, te = 0, g0 = 0, gf = 0;
    const bool any = locate_tiles(wg_tiles, nwg, heads_per_wg, count, G, KVC_TM_GENERIC, blockIdx.x, gridDim.x, tid,
                                  tb, te, g0, gf);
    if (tid == 0) { loc_s[0] = any ? 1 : 0; loc_s[1] = tb; loc_s[2] = te; loc_s[3] = g0; loc_s[4] = gf; }
  }
  __syncthreads();
  if (!loc_s[0]) return;
  int g = loc_s[3], g_first = loc_s[4];
  int cnt = count[g];
  int g_next = g_first + (cnt + KVC_TM_GENERIC - 1) / KVC_TM_GENERIC;
  for (int t = loc_s[1]; t < loc_s[2]; ++t) {
    while (t >= g_next && g + 1 < G) { ++g; g_first = g_next; cnt = count[g]; g_next = g_first + (cnt + KVC_TM_GENERIC - 1) / KVC_TM_GENERIC; }
    const int j0 = (t - g_first) * KVC_TM_GENERIC;
    const int j1 = min(cnt, j0 + KVC_TM_GENERIC);
    const int2* __restrict__ mv = reinterpret_cast<const int2*>(moves) + offs[g];
    for (int j = j0 + tid; j < j1; j += blockDim.x) {
      const int2 m = mv[j];
      metrics[m.x] = metrics[m.y];
      positions[m.x] = positions[m.y];
    }
    const int per_move = kgroups * kvec_bytes + hd * e;      // bytes per move (K then V)
    const int n = (j1 - j0) * per_move;
    for (int idx = tid; idx < n; idx += blockDim.x) {
      const int2 m = mv[j0 + idx / per_move];
      const int bsel = idx % per_move;
      const int sb = m.y / bs, so = m.y % bs, db = m.x / bs, dof = m.x % bs;
      if (bsel < kgroups * kvec_bytes) {
        const int r = bsel / kvec_bytes, byte = bsel % kvec_bytes;
        k_cache[(int64_t)db * block_bytes + ((int64_t)r * bs + dof) * kvec_bytes + byte] =
            k_cache[(int64_t)sb * block_bytes + ((int64_t)r * bs + so) * kvec_bytes + byte];
      } else {
        const int vb = bsel - kgroups * kvec_bytes;
        const int d = vb / e, byte = vb % e;
        v_cache[(int64_t)db * block_bytes + ((int64_t)d * bs + dof) * e + byte] =
            v_cache[(int64_t)sb * block_bytes + ((int64_t)d * bs + so) * e + byte];
      }
    }
  }
}

// ------------------------------------------------------------------------- slot-major layout
// KVC_LAYOUT_SLOT_MAJOR (include/kvc_mi355x.h): K and V of physical slot s are the SB = hd * e bytes at
// byte s * SB of their planes.  A move is two contiguous copies of SB bytes + 8 B of metric / position --
// randomly placed chunks of >= 128 B copy at the speed of a linear copy (tools/gather_bw.hip), so this
// kernel is the algorithmic traffic and nothing else: no destination image is read, no claim table.
//   * LPS = SB / 16 lanes hold one slot image; a wave instruction moves 64 / LPS slots (1 KiB);
//   * a wave owns a contiguous range of the plan's 32-move tiles (the byte-wise half of the plan that
//     schedule_t1_cache_moves leaves behind, or compact_plan_kernel's).  Its moves go through a queue in
//     LDS so that heads with one or two moves (the continual steady state: 65 536 heads, ~1 move each)
//     fill whole batches: a batch = MB moves, ALL of whose loads (K, V, metric, position: <= 18 per lane)
//     are requested before the first store;
//   * loads and stores are non-temporal: every byte is touched once.
template <int LPS>
__global__ __launch_bounds__(256) void compact_slots_kernel(
    uint8_t* __restrict__ k_cache, uint8_t* __restrict__ v_cache, float* __restrict__ metrics,
    int32_t* __restrict__ positions, const int32_t* __restrict__ moves,
    const int32_t* __restrict__ count, const int32_t* __restrict__ offs,
    const int32_t* __restrict__ wg_tiles, int nwg, int heads_per_wg, int G) {
  constexpr int TM = KVC_TM_GENERIC;                      // moves per tile of the plan
  constexpr int MB = LPS <= 16 ? 32 : 512 / LPS;          // moves per batch
  constexpr int SPI = 64 / LPS;                           // slots per wave instruction
  constexpr int NIT = MB / SPI;                           // instructions per plane and batch (<= 8)
  constexpr int64_t SB = (int64_t)LPS * 16;               // bytes of a slot image
  __shared__ int2 queue[4][TM + MB];
  const int lane = lane_id(), w = threadIdx.x / WAVE;
  const int nw = gridDim.x * 4, wv = blockIdx.x * 4 + w;
  int tb, te, g, g_first;
  if (!locate_tiles(wg_tiles, nwg, heads_per_wg, count, G, TM, wv, nw, lane, tb, te, g, g_first)) return;
  int2* q = queue[w];
  const int sub = lane / LPS, piece = lane % LPS;
  int qn = 0;
  auto copy_batch = [&](int n) {                          // q[0, n), n <= MB (wave-uniform)
    u32x4 kr[NIT], vr[NIT];
    int2 mv[NIT];
    float m = 0.0f;
    int32_t pos = 0;
    const int2 mine = lane < n ? q[lane] : int2{0, 0};
    if (lane < n) { m = __builtin_nontemporal_load(metrics + mine.y); pos = __builtin_nontemporal_load(positions + mine.y); }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = it * SPI + sub;
      mv[it] = idx < n ? q[idx] : int2{-1, -1};
      if (it * SPI < n && mv[it].x >= 0) {
        kr[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(k_cache + (int64_t)mv[it].y * SB) + piece);
        vr[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(v_cache + (int64_t)mv[it].y * SB) + piece);
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (it * SPI < n && mv[it].x >= 0) {
        __builtin_nontemporal_store(kr[it], reinterpret_cast<u32x4*>(k_cache + (int64_t)mv[it].x * SB) + piece);
        __builtin_nontemporal_store(vr[it], reinterpret_cast<u32x4*>(v_cache + (int64_t)mv[it].x * SB) + piece);
      }
    }
    if (lane < n) { metrics[mine.x] = m; positions[mine.x] = pos; }
  };
  int cnt = count[g];
  int g_next = g_first + (cnt + TM - 1) / TM;
  for (int t = tb; t < te; ++t) {
    while (t >= g_next && g + 1 < G) { ++g; g_first = g_next; cnt = count[g]; g_next = g_first + (cnt + TM - 1) / TM; }
    const int j = (t - g_first) * TM + lane;
    const bool in = lane < TM && j < cnt;
    int2 m = {0, 0};
    if (in) m = reinterpret_cast<const int2*>(moves)[(int64_t)offs[g] + j];
    const unsigned long long mask = __ballot(in);
    if (in) q[qn + __popcll(mask & ((1ull << lane) - 1ull))] = m;
    qn += __popcll(mask);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    while (qn >= MB) {
      copy_batch(MB);
      // what is left moves to the front (< TM entries: one read and one write per lane)
      const int rest = qn - MB;
      const int2 keep = lane < rest ? q[MB + lane] : int2{0, 0};
      __builtin_amdgcn_wave_barrier();
      if (lane < rest) q[lane] = keep;
      qn = rest;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (qn > 0) copy_batch(qn);
}

// slot images that are not a power-of-two number of 16 B pieces <= 64 (e.g. head size 96): 16 B pieces one by one
__global__ __launch_bounds__(256) void compact_slots_generic_kernel(
    uint8_t* __restrict__ k_cache, uint8_t* __restrict__ v_cache, float* __restrict__ metrics,
    int32_t* __restrict__ positions, const int32_t* __restrict__ moves,
    const int32_t* __restrict__ count, const int32_t* __restrict__ offs,
    const int32_t* __restrict__ wg_tiles, int nwg, int heads_per_wg, int G, int pieces) {
  __shared__ int loc_s[5];
  const int tid = threadIdx.x;
  if (tid < WAVE) {
    int tb = 0, te = 0, g0 = 0, gf = 0;
    const bool any = locate_tiles(wg_tiles, nwg, heads_per_wg, count, G, KVC_TM_GENERIC, blockIdx.x, gridDim.x, tid,
                                  tb, te, g0, gf);
    if (tid == 0) { loc_s[0] = any ? 1 : 0; loc_s[1] = tb; loc_s[2] = te; loc_s[3] = g0; loc_s[4] = gf; }
  }
  __syncthreads();
  if (!loc_s[0]) return;
  int g = loc_s[3], g_first = loc_s[4];
  int cnt = count[g];
  int g_next = g_first + (cnt + KVC_TM_GENERIC - 1) / KVC_TM_GENERIC;
  const int64_t sb = (int64_t)pieces * 16;
  for (int t = loc_s[1]; t < loc_s[2]; ++t) {
    while (t >= g_next && g + 1 < G) { ++g; g_first = g_next; cnt = count[g]; g_next = g_first + (cnt + KVC_TM_GENERIC - 1) / KVC_TM_GENERIC; }
    const int j0 = (t - g_first) * KVC_TM_GENERIC;
    const int j1 = min(cnt, j0 + KVC_TM_GENERIC);
    const int2* __restrict__ mv = reinterpret_cast<const int2*>(moves) + offs[g];
    for (int j = j0 + tid; j < j1; j += blockDim.x) {
      const int2 m = mv[j];
      metrics[m.x] = metrics[m.y];
      positions[m.x] = positions[m.y];
    }
    const int n = (j1 - j0) * 2 * pieces;
    for (int idx = tid; idx < n; idx += blockDim.x) {
      const int2 m = mv[j0 + idx / (2 * pieces)];
      const int r = idx % (2 * pieces);
      uint8_t* base = r < pieces ? k_cache : v_cache;
      const int pc = r < pieces ? r : r - pieces;
      reinterpret_cast<u32x4*>(base + (int64_t)m.x * sb)[pc] = reinterpret_cast<const u32x4*>(base + (int64_t)m.y * sb)[pc];
    }
  }
}

}  // namespace kvc

// one byte per block, in whole 16 B vectors
static size_t claims_bytes(int64_t num_blocks) { return (size_t)((num_blocks + 15) / 16 + 1) * 16; }
// [plan: 2 x MOVES_PLAN_WGS int32][claims]
static size_t claims_offset() { return (size_t)2 * kvc::MOVES_PLAN_WGS * sizeof(int32_t); }

extern "C" size_t kvc_cache_moves_plan_bytes(void) { return claims_offset(); }

extern "C" size_t kvc_execute_cache_moves_workspace_bytes(int32_t total_heads, int64_t num_blocks) {
  (void)total_heads;
  return claims_offset() + claims_bytes(num_blocks);
}

namespace kvc {

// block-path instantiations (head size, block size, element bytes)
static bool compact_shape_fast(int head_size, int block_size, int elem_bytes, int vec_size) {
  const int combo = head_size * 10000 + block_size * 100 + elem_bytes;
  return vec_size * elem_bytes == 16 &&
      (combo == 1281602 || combo == 1283201 || combo == 1283202 || combo == 1281601 ||
       combo == 1281604 || combo == 641602 || combo == 2561602);
}

static int compact_check_args(int32_t total_heads, int64_t num_blocks, int32_t block_size,
                              int32_t head_size, int32_t elem_bytes, int32_t vec_size,
                              const void* workspace, size_t workspace_bytes) {
  if (block_size < 1) return fail_invalid("Unsupported block size: " + std::to_string(block_size));
  if (head_size < 1) return fail_invalid("Unsupported head size: " + std::to_string(head_size));
  if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4)
    return fail_invalid("Unsupported cache element size: " + std::to_string(elem_bytes));
  if (vec_size < 1 || head_size % vec_size != 0)
    return fail_invalid("Unsupported vec size: " + std::to_string(vec_size));
  if (total_heads <= 0) return KVC_OK;
  if (workspace_bytes < kvc_execute_cache_moves_workspace_bytes(total_heads, num_blocks))
    return fail_invalid("execute_cache_moves: workspace too small");
  if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0)
    return fail_invalid("execute_cache_moves: workspace must be 16-byte aligned");
  return KVC_OK;
}

// persistent grid of the compaction kernel = what is resident at once (asked once per
// instantiation and device: LDS-bound, 2 block images per wave)
template <int HD, int BS, int E, int WPB>
static int compact_runs_grid() {
  static std::atomic<int> grid[64];                  // (zero-initialised; any thread, any device)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (grid[dev].load(std::memory_order_relaxed) == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, compact_runs_kernel<HD, BS, E, WPB>,
                                                     64 * WPB, 0) != hipSuccess || per_cu < 1)
      per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
      cus = 256;
    grid[dev].store(per_cu * cus, std::memory_order_relaxed);
  }
  return grid[dev].load(std::memory_order_relaxed);
}

}  // namespace kvc

#ifndef KVC_COMPACT_WPB
#define KVC_COMPACT_WPB 4   // waves per workgroup of the compaction kernel (waves are independent; measured:
                            // 4-wave workgroups, one wave per SIMD, beat single-wave workgroups by 5-7 %)
#endif

// planning half (move lists of unknown origin): tiles per group of heads + destination-block claim
// counts (2 small launches)
extern "C" int kvc_execute_cache_moves_plan(const int32_t* cache_moves_idx,
                                            const int32_t* cache_moves_count,
                                            const int32_t* evicted_kv_offsets, int32_t total_heads,
                                            int64_t num_blocks, int32_t block_size,
                                            int32_t head_size, int32_t elem_bytes, int32_t vec_size,
                                            void* workspace, size_t workspace_bytes,
                                            kvc_stream_t stream) {
  using namespace kvc;
  if (int rc = compact_check_args(total_heads, num_blocks, block_size, head_size, elem_bytes, vec_size,
                                  workspace, workspace_bytes)) return rc;
  if (total_heads <= 0) return KVC_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool fast = compact_shape_fast(head_size, block_size, elem_bytes, vec_size);
  int32_t* wg_tiles = reinterpret_cast<int32_t*>(workspace) + (fast ? 0 : MOVES_PLAN_WGS);
  uint32_t* claims = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(workspace) + claims_offset());
  const int G = total_heads;
  int nwg, hpw;
  moves_plan_shape(G, nwg, hpw);
  // a wave holds one move per lane: tile + look-behind + (bs-1) look-ahead <= 64 lanes
  const int tm = fast ? 64 - block_size : KVC_TM_GENERIC;
  const int64_t claim_vecs = (int64_t)(claims_bytes(num_blocks) / 16);
  const int64_t zb = (claim_vecs + 1023) / 1024;                 // 4 stores per thread
  const int plan_blocks = hpw >= 16 ? (nwg + 3) / 4 : (nwg + 255) / 256;
  hipLaunchKernelGGL(compact_plan_kernel, dim3(plan_blocks + (unsigned)(zb < 1 ? 1 : (zb > 4096 ? 4096 : zb))),
                     dim3(256), 0, s, wg_tiles, cache_moves_count, G, tm, nwg, hpw, plan_blocks,
                     reinterpret_cast<u32x4*>(claims), claim_vecs);
  hipLaunchKernelGGL(compact_plan_claims_kernel, dim3(256 * 8), dim3(256), 0, s, claims, cache_moves_idx,
                     cache_moves_count, evicted_kv_offsets, wg_tiles, nwg, hpw, G, block_size, tm);
  return check_launch("execute_cache_moves (plan)");
}

namespace kvc {
// the compaction kernel over a plan; claims == nullptr: every run is the only writer of its block
static int launch_compaction(void* k_cache, void* v_cache, float* kv_metrics, int32_t* kv_position,
                             const int32_t* cache_moves_idx, const int32_t* cache_moves_count,
                             const int32_t* evicted_kv_offsets, int32_t total_heads, int32_t block_size,
                             int32_t head_size, int32_t elem_bytes, int32_t vec_size, const int32_t* plan,
                             const uint32_t* claims, hipStream_t s) {
  const int G = total_heads;
  int nwg, hpw;
  moves_plan_shape(G, nwg, hpw);
  uint8_t* k = reinterpret_cast<uint8_t*>(k_cache);
  uint8_t* v = reinterpret_cast<uint8_t*>(v_cache);
  // a wave's two LDS source slots take 4 block images: 4 waves per workgroup for 4 KiB images
  // (two workgroups per CU), half as many for 8 KiB images
#define KVC_WPB(HD, BS, E) ((HD) * (BS) * (E) <= 4096 ? KVC_COMPACT_WPB : (KVC_COMPACT_WPB + 1) / 2)
#define KVC_RUNS(HD, BS, E)                                                                        \
  hipLaunchKernelGGL((compact_runs_kernel<HD, BS, E, KVC_WPB(HD, BS, E)>),                          \
                     dim3(compact_runs_grid<HD, BS, E, KVC_WPB(HD, BS, E)>()),                      \
                     dim3(64 * KVC_WPB(HD, BS, E)), 0, s, k, v, kv_metrics, kv_position,            \
                     cache_moves_idx, cache_moves_count, evicted_kv_offsets, claims, plan, nwg, hpw, G, 64 - BS)
  bool fast = compact_shape_fast(head_size, block_size, elem_bytes, vec_size);
  if (!fast) {}
  else if (head_size == 128 && block_size == 16 && elem_bytes == 2) KVC_RUNS(128, 16, 2);
  else if (head_size == 128 && block_size == 32 && elem_bytes == 1) KVC_RUNS(128, 32, 1);
  else if (head_size == 128 && block_size == 32 && elem_bytes == 2) KVC_RUNS(128, 32, 2);
  else if (head_size == 128 && block_size == 16 && elem_bytes == 1) KVC_RUNS(128, 16, 1);
  else if (head_size == 128 && block_size == 16 && elem_bytes == 4) KVC_RUNS(128, 16, 4);
  else if (head_size == 64 && block_size == 16 && elem_bytes == 2) KVC_RUNS(64, 16, 2);
  else if (head_size == 256 && block_size == 16 && elem_bytes == 2) KVC_RUNS(256, 16, 2);
  else fast = false;
#undef KVC_RUNS
#undef KVC_WPB
  if (!fast) {
    hipLaunchKernelGGL(compact_generic_kernel, dim3(256 * 8), dim3(256), 0, s, k, v, kv_metrics,
                       kv_position, cache_moves_idx, cache_moves_count, evicted_kv_offsets,
                       plan + MOVES_PLAN_WGS, nwg, hpw, G, block_size, head_size, elem_bytes, vec_size);
  }
  return check_launch("execute_cache_moves");
}
}  // namespace kvc

// data half: the compaction kernel itself, on a workspace filled by _plan for the same move list
extern "C" int kvc_execute_cache_moves_apply(void* k_cache, void* v_cache, float* kv_metrics,
                                             int32_t* kv_position, const int32_t* cache_moves_idx,
                                             const int32_t* cache_moves_count,
                                             const int32_t* evicted_kv_offsets, int32_t total_heads,
                                             int64_t num_blocks, int32_t block_size,
                                             int32_t head_size, int32_t elem_bytes, int32_t vec_size,
                                             void* workspace, size_t workspace_bytes,
                                             kvc_stream_t stream) {
  using namespace kvc;
  if (int rc = compact_check_args(total_heads, num_blocks, block_size, head_size, elem_bytes, vec_size,
                                  workspace, workspace_bytes)) return rc;
  if (total_heads <= 0) return KVC_OK;
  const int32_t* plan = reinterpret_cast<const int32_t*>(workspace);
  const uint32_t* claims = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(workspace) + claims_offset());
  return launch_compaction(k_cache, v_cache, kv_metrics, kv_position, cache_moves_idx, cache_moves_count,
                           evicted_kv_offsets, total_heads, block_size, head_size, elem_bytes, vec_size, plan,
                           claims, (hipStream_t)stream);
}

// execute_cache_moves for a move list that kvc_schedule_t1_cache_moves_ex produced together with `plan`
// (and that nobody has touched since): ONE launch, no planning pass, no claim table
extern "C" int kvc_execute_cache_moves_planned(void* k_cache, void* v_cache, float* kv_metrics,
                                               int32_t* kv_position, const int32_t* cache_moves_idx,
                                               const int32_t* cache_moves_count,
                                               const int32_t* evicted_kv_offsets, int32_t total_heads,
                                               int64_t num_blocks, int32_t block_size,
                                               int32_t head_size, int32_t elem_bytes, int32_t vec_size,
                                               const int32_t* plan, kvc_stream_t stream) {
  using namespace kvc;
  if (plan == nullptr) return fail_invalid("execute_cache_moves: no plan");
  if ((reinterpret_cast<uintptr_t>(plan) & 15) != 0) return fail_invalid("execute_cache_moves: plan must be 16-byte aligned");
  if (int rc = compact_check_args(total_heads, num_blocks, block_size, head_size, elem_bytes, vec_size,
                                  plan, kvc_execute_cache_moves_workspace_bytes(total_heads, num_blocks))) return rc;
  if (total_heads <= 0) return KVC_OK;
  return launch_compaction(k_cache, v_cache, kv_metrics, kv_position, cache_moves_idx, cache_moves_count,
                           evicted_kv_offsets, total_heads, block_size, head_size, elem_bytes, vec_size, plan,
                           nullptr, (hipStream_t)stream);
}

extern "C" int kvc_execute_cache_moves(void* k_cache, void* v_cache, float* kv_metrics,
                                       int32_t* kv_position, const int32_t* cache_moves_idx,
                                       const int32_t* cache_moves_count,
                                       const int32_t* evicted_kv_offsets, int32_t total_heads,
                                       int64_t num_blocks, int32_t block_size,
                                       int32_t head_size, int32_t elem_bytes, int32_t vec_size,
                                       void* workspace, size_t workspace_bytes,
                                       kvc_stream_t stream) {
  if (int rc = kvc_execute_cache_moves_plan(cache_moves_idx, cache_moves_count, evicted_kv_offsets,
                                            total_heads, num_blocks, block_size, head_size, elem_bytes,
                                            vec_size, workspace, workspace_bytes, stream)) return rc;
  return kvc_execute_cache_moves_apply(k_cache, v_cache, kv_metrics, kv_position, cache_moves_idx,
                                       cache_moves_count, evicted_kv_offsets, total_heads, num_blocks,
                                       block_size, head_size, elem_bytes, vec_size, workspace,
                                       workspace_bytes, stream);
}

// ---- KVC_LAYOUT_SLOT_MAJOR (ABI version 7)
namespace kvc {
template <int LPS>
static int compact_slots_grid() {
  static std::atomic<int> grid[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (grid[dev].load(std::memory_order_relaxed) == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, compact_slots_kernel<LPS>, 256, 0) != hipSuccess || per_cu < 1)
      per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
      cus = 256;
    grid[dev].store(per_cu * cus, std::memory_order_relaxed);
  }
  return grid[dev].load(std::memory_order_relaxed);
}
}  // namespace kvc

extern "C" int kvc_execute_cache_moves_slot_major_plan(const int32_t* cache_moves_count, int32_t total_heads,
                                                       void* workspace, size_t workspace_bytes, kvc_stream_t stream) {
  using namespace kvc;
  if (total_heads <= 0) return KVC_OK;
  if (cache_moves_count == nullptr || workspace == nullptr || workspace_bytes < claims_offset() ||
      (reinterpret_cast<uintptr_t>(workspace) & 15) != 0)
    return fail_invalid("execute_cache_moves (slot-major plan): bad arguments");
  int nwg, hpw;
  moves_plan_shape(total_heads, nwg, hpw);
  const int plan_blocks = hpw >= 16 ? (nwg + 3) / 4 : (nwg + 255) / 256;
  // (the byte-wise half of the plan: 32-move tiles; no claim table in this layout)
  hipLaunchKernelGGL(compact_plan_kernel, dim3(plan_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<int32_t*>(workspace) + MOVES_PLAN_WGS, cache_moves_count, total_heads,
                     KVC_TM_GENERIC, nwg, hpw, plan_blocks, (u32x4*)nullptr, (int64_t)0);
  return check_launch("execute_cache_moves (slot-major plan)");
}

extern "C" int kvc_execute_cache_moves_slot_major(void* k_cache, void* v_cache, float* kv_metrics,
                                                  int32_t* kv_position, const int32_t* cache_moves_idx,
                                                  const int32_t* cache_moves_count,
                                                  const int32_t* evicted_kv_offsets, int32_t total_heads,
                                                  int64_t num_blocks, int32_t block_size, int32_t head_size,
                                                  int32_t elem_bytes, const int32_t* plan, void* workspace,
                                                  size_t workspace_bytes, kvc_stream_t stream) {
  using namespace kvc;
  (void)num_blocks;
  if (block_size < 1) return fail_invalid("Unsupported block size: " + std::to_string(block_size));
  if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4)
    return fail_invalid("Unsupported cache element size: " + std::to_string(elem_bytes));
  if (head_size < 1 || (head_size * elem_bytes) % 16 != 0)
    return fail_invalid("Unsupported head size: " + std::to_string(head_size) + " (slot-major blocks hold slots of whole 16-byte pieces)");
  if (total_heads <= 0) return KVC_OK;
  if (plan == nullptr) {
    if (int rc = kvc_execute_cache_moves_slot_major_plan(cache_moves_count, total_heads, workspace, workspace_bytes, stream))
      return rc;
    plan = reinterpret_cast<const int32_t*>(workspace);
  } else if ((reinterpret_cast<uintptr_t>(plan) & 15) != 0) {
    return fail_invalid("execute_cache_moves: plan must be 16-byte aligned");
  }
  const int32_t* tiles = plan + MOVES_PLAN_WGS;
  int nwg, hpw;
  moves_plan_shape(total_heads, nwg, hpw);
  hipStream_t s = (hipStream_t)stream;
  uint8_t* k = reinterpret_cast<uint8_t*>(k_cache);
  uint8_t* v = reinterpret_cast<uint8_t*>(v_cache);
  const int pieces = head_size * elem_bytes / 16;
#define KVC_SLOTS(LPS)                                                                               \
  hipLaunchKernelGGL(compact_slots_kernel<LPS>, dim3(compact_slots_grid<LPS>()), dim3(256), 0, s, k, v, \
                     kv_metrics, kv_position, cache_moves_idx, cache_moves_count, evicted_kv_offsets, tiles, nwg, hpw, total_heads)
  switch (pieces) {
    case 4: KVC_SLOTS(4); break;
    case 8: KVC_SLOTS(8); break;
    case 16: KVC_SLOTS(16); break;
    case 32: KVC_SLOTS(32); break;
    case 64: KVC_SLOTS(64); break;
    default:
      hipLaunchKernelGGL(compact_slots_generic_kernel, dim3(256 * 8), dim3(256), 0, s, k, v, kv_metrics, kv_position,
                         cache_moves_idx, cache_moves_count, evicted_kv_offsets, tiles, nwg, hpw, total_heads, pieces);
  }
#undef KVC_SLOTS
  return check_launch("execute_cache_moves (slot-major)");
}
