// A3 CompressionMetrics.schedule_evictions for gfx950 -- without a single sort.
//
// The reference (vllm/kvcompress/metrics.py:441-847) masks the candidate metrics, sorts
// them globally (twice), gathers one threshold per block-sized chunk, sorts those
// (twice), walks the sequences on the host, counts leading evicted chunks per head and
// sorts the surviving logical indices (twice) -- six device sorts over N slots and ~8N
// of temporaries.  What it computes is an order-statistics problem:
//
//   * per head g the chunk thresholds are every bs-th order statistic of the head's
//     masked metrics, thr[g,c] = (c*bs + hang_g)-th smallest;
//   * per sequence the k smallest thresholds are selected; because thr[g,.] increases
//     with c, head g frees n_g = #{c : thr[g,c] <= T*} chunks where T* is the k-th
//     smallest threshold of the sequence, i.e. n_g = floor((R_g(T*) - hang_g)/bs) + 1
//     with R_g(T) = #{finite keys of head g that are <= T};
//   * the evicted slots of head g are its cnt_g = (n_g-1)*bs + hang_g smallest keys,
//     listed by ascending logical index.
//
// So: one pass builds order-preserving 32-bit keys in head-contiguous *logical* order;
// T* per sequence is found by an MSB-first radix select (4 rounds of per-head 256-bin
// histograms, a per-head scan that turns cumulative counts into chunk counts, and a
// per-sequence pick of the digit); a per-head radix select finds the cnt_g-th smallest
// key, and a flag + prefix-sum pass emits the logical indices already in ascending order.
// Everything is a streaming pass over N x 4 B; ties are resolved exactly in the canonical
// order of DESIGN.md ((metric, physical block, offset) within a head, (threshold, head,
// chunk) within a sequence).  The reference's batch>1 quirk (metrics.py:718-721) only
// changes how many chunks each sequence may free, which is computed on device in
// seq_prepare_kernel for mode 0.
#include "kvc_common.h"
#include "../../include/kvc_mi355x.h"
#include <atomic>
#include <cstdlib>

// The device code lives in five headers, included here in dependency order (ONE translation unit: the
// single-launch fallback calls the bodies of every section, and HIP has no cross-module device calls
// without relocatable device code):
//   kvc_schedule_common.h    scratch layout (SchedWs), key order, dirty-map and histogram helpers
//   kvc_schedule_general.h   sections 0-6: keys, digit rounds, scan + pick, select + emit
//   kvc_schedule_small.h     section 7: the small-eviction schedule (sample, pivot, collect, records, select, emit)
//   kvc_schedule_harvest.h   section 10: aggregate_decode that harvests section 7's candidate lists on its way
//   kvc_schedule_fused.h     section 7b: records + selection + emission + next pivots of section 7 as one launch
//   kvc_schedule_bracket.h   section 9: the bracket schedule (bracket, count + collect, records, select)
//   kvc_schedule_fallback.h  section 8: the general pipeline as one gated launch, phases ordered by work counters
// This file: the host side -- workspace layout, which schedule a call takes (and why), the launches.
#include "kvc_schedule_common.h"
#include "kvc_schedule_general.h"
#include "kvc_schedule_small.h"
#include "kvc_schedule_harvest.h"
#include "kvc_schedule_fused.h"
#include "kvc_schedule_bracket.h"
#include "kvc_schedule_fallback.h"

// --------------------------------------------------------------------------- host side
#ifndef KVC_COLLECT_GRID_CAP
#define KVC_COLLECT_GRID_CAP 4096                  // (experiment builds: tools/, profiles/DESIGN_history_r1_r4.md section 6)
#endif
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// The null padding of the output list is 4 B per candidate slot of pure HBM writes with nobody
// waiting for it until the emission: on the small-eviction schedule it runs on a side stream,
// forked off the caller's stream behind the sampling kernel and joined in front of emit_topk, so
// that it runs next to the pivot kernel (registers and LDS: the HBM is idle) and into the start
// of the collecting pass.  Measured at 256 x 1 M slots (step = S1 + S2 + S3): inline 1.92 ms;
// forked at the very start 1.84 (the sampling pass lives on random accesses and is slowed down
// as much as the fill is hidden); forked behind the sampling kernel 1.76; split with a part next
// to the record / selection kernels 1.78-1.81 (S1 itself is 20 us shorter, but the list is then
// still dirty in the caches when schedule_cache_moves starts, which pays 50 us for it).
// One non-blocking stream and two events per (host thread, device), made on first use and kept for
// the life of the thread (they are not destroyed at thread exit: by then the HIP runtime may be
// unloading); never created under stream capture (a call that is being captured before any other
// gets the fill inline).  Any failure of the fork / join calls is answered by doing without the
// side stream (the fill inline, or waited for on the host) -- never by an unordered fill.
struct SideStream { hipStream_t s2 = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool failed = false; };
static SideStream* side_stream(hipStream_t main) {
  thread_local SideStream tab[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  SideStream& t = tab[dev & 63];
  if (t.failed) return nullptr;
  if (t.s2 == nullptr) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      return nullptr;
    }
    if (hipStreamCreateWithFlags(&t.s2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&t.join, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      t.failed = true;
      return nullptr;
    }
  }
  return &t;
}

// > 64 KiB of dynamic LDS needs an opt-in per function AND per device (gfx950 has 160 KiB per CU):
// done once per (function, device), whatever thread or device the caller is on
static void allow_dynamic_lds(const void* fn, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done.fetch_or(bit, std::memory_order_release);
}

struct WsLayout {
  size_t keys, zero_begin, chunk_phys, bsample, hist, less, eq, seq_prefix, seq_k, zero_end, cum,
      seq_tmp, tz_begin, fallback, st_claimed, st_cnt, st_def, st_samp, tz_end, st_seqrec, head_fc, rec64, blist, bthr, total;
};

// where the bracket schedule's claim counters were last left zero (per device; speed only: see count_claims)
static std::atomic<uintptr_t>& claims_clean_at() {
  static std::atomic<uintptr_t> at[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  return at[(dev >= 0 && dev < 64) ? dev : 0];
}

static WsLayout ws_layout(int64_t N, int32_t G, int32_t B, int32_t bs) {
  WsLayout l;
  size_t o = 0;
  l.keys = o;        o = align_up(o + (size_t)N * 4, 256);      // keys + chunk_phys (+ bsample on the bracket schedule): ONE 0xFF memset
  l.chunk_phys = o;  o = align_up(o + (size_t)(N / bs + 1) * 4, 256);
  l.bsample = o;     o = align_up(o + (size_t)B * kvc::BR_CELLS * 4, 256);   // (bracket schedule only; the memset's tail)
  l.zero_begin = o;  // everything up to zero_end is cleared by build_keys' tail workgroups
  l.hist = o;        o = align_up(o + (size_t)G * kvc::RADIX * 4, 256);
  l.less = o;        o = align_up(o + (size_t)G * 4, 256);
  l.eq = o;          o = align_up(o + (size_t)G * 4, 256);
  l.seq_prefix = o;  o = align_up(o + (size_t)B * 4, 256);
  l.seq_k = o;       o = align_up(o + (size_t)B * 4, 256);
  l.zero_end = o;
  l.cum = o;         o = align_up(o + (size_t)4 * G * kvc::RADIX * 4, 256);
  l.seq_tmp = o;     o = align_up(o + (size_t)B * 12, 256);
  l.tz_begin = o;    // one memset at the head of the small-eviction schedule
  l.fallback = o;    o = align_up(o + 512, 256);   // flag word | +128: stamps | +256: claim / done counters of the fallback's phases
  l.st_claimed = o;  o = align_up(o + (size_t)kvc::CLAIM_SHARDS * 128, 256);
  l.st_cnt = o;      o = align_up(o + (size_t)G * 4, 256);
  l.st_def = o;      o = align_up(o + (size_t)G * 4, 256);
  l.st_samp = o;     o = align_up(o + (size_t)G * 4, 256);
  l.tz_end = o;
  l.st_seqrec = o;   o = align_up(o + (size_t)B * 16, 256);
  l.head_fc = o;     o = align_up(o + (size_t)G * 8, 256);
  l.rec64 = o;       o = align_up(o + (size_t)G * kvc::KREC * 8, 256);
  l.blist = o;       o = align_up(o + (size_t)(N / kvc::BR_DIV + (int64_t)G * kvc::BR_PAD + 4) * 4, 256);
  l.bthr = o;        o = align_up(o + (size_t)(N / bs + 1) * 4, 256);
  l.total = o;
  return l;
}

// grid of the single-launch fallback (section 8): what is resident at once on an idle device, less
// one workgroup per CU (the occupancy query can be one too high where SGPRs decide,
// MI355X_MICROARCH.md), at most 3 per CU (4 measured slower: the cost of a phase end grows with the
// number of waiters); asked once per device.  Only speed depends on it: the kernel's phases wait for
// work, not for workgroups (kvc_schedule_params.fallback_grid launches any other number: tests)
static int fallback_grid() {
  static std::atomic<int> grid[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (grid[dev].load(std::memory_order_relaxed) == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kvc::fallback_general_kernel, 256, 0) != hipSuccess)
      per_cu = 1;
    per_cu = per_cu - 1 < 1 ? 1 : (per_cu - 1 > 3 ? 3 : per_cu - 1);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    grid[dev].store(per_cu * cus, std::memory_order_relaxed);
  }
  return grid[dev].load(std::memory_order_relaxed);
}

// small-eviction schedule (section 7) or not: the host knows how many blocks a sequence frees at
// most (the reference passes a Python list); eligible when that is on average <= 1/8 of what a
// head's record covers and a sequence's thresholds fit one workgroup's LDS.
// p2 = padded threshold count per sequence (0: not eligible), sshift = log2 of the sample stride;
// returns why not (KVC_WHY_*, include/kvc_mi355x.h; KVC_WHY_TAKEN = eligible)
static int topk_plan(const kvc_schedule_params& p, int& p2_out, int& sshift) {
  p2_out = 0; sshift = 0;
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  if (G < 1 || p.total_slots <= 0) return KVC_WHY_EMPTY;
  const int LH = p.num_layers * p.num_kv_heads;
  const int bsz = p.block_size;
  if (p.uniform_evict) return KVC_WHY_UNIFORM;
  if (p.schedule_path == 1 || p.schedule_path == 4) return KVC_WHY_FORCED_PATH;
  if (!(bsz == 8 || bsz == 16 || bsz == 32)) return KVC_WHY_BLOCK_SIZE;
  // (per-head tables of the pivot kernel in LDS; a record entry packs the physical slot into 32 bits)
  if (LH > kvc::PIV_MAXLH) return KVC_WHY_HEADS_PER_SEQ;
  if (p.num_seqs > 65535 || p.num_blocks * (int64_t)bsz >= (int64_t)1 << 32) return KVC_WHY_INDEX_RANGE;
  const int mch = kvc::KREC / bsz;
  int p2 = 128;
  while (p2 < LH * mch && p2 <= 16384) p2 <<= 1;
  if (p2 > 16384) return KVC_WHY_THRESHOLDS_LDS;
  if (p.schedule_path != 2 && p.schedule_path != 3) {
    if (p.max_evicted_blocks_hint < 0) return KVC_WHY_HINT_UNKNOWN;
    if ((int64_t)p.max_evicted_blocks_hint * 8 > (int64_t)mch * LH) return KVC_WHY_BULK;
  }
  p2_out = p2;
  // sample stride: about 16 Ki sampled keys per sequence (a sequence's pivot is a low quantile of
  // its sample, held in the registers of one workgroup; a sampled block costs ~9 random accesses,
  // a candidate ~3: at 256 x 1 M slots stride 64 is where the two meet); small sequences are
  // sampled whole
  int64_t stride = p.total_slots / p.num_seqs / 16384;
  if (p.sample_stride > 0) stride = p.sample_stride;
  while (sshift < 8 && (2ll << sshift) <= stride) ++sshift;
  return KVC_WHY_TAKEN;
}

// the small-eviction schedule's position-lazy form (stream_collect_kernel, LAZY): keys that do not
// depend on the position, sequences that do not need each other's inf counts
static bool lazy_plan(const kvc_schedule_params& p) {
  return !p.use_average && p.bias == nullptr && !(p.mode == 0 && p.num_seqs > 1) && p.schedule_path != 3;
}
// harvest-ahead (section 10): any call that takes the small-eviction schedule (its position-lazy form costs the
// aggregation pass nothing but the harvest; the full form streams the position rows as well)
static bool harvest_plan(const kvc_schedule_params& p) {
  int p2 = 0, sshift = 0;
  return topk_plan(p, p2, sshift) == KVC_WHY_TAKEN;
}
extern "C" int32_t kvc_harvest_eligible(const kvc_schedule_params* p, int32_t num_queries_per_kv) {
  if (p == nullptr || num_queries_per_kv < 1) return 0;
  return harvest_plan(*p) ? 1 : 0;
}
namespace kvc {
__global__ __launch_bounds__(256) void harvest_seen_seq_kernel(const int32_t* __restrict__ seq_positions,
                                                               const int32_t* __restrict__ num_protected, int B,
                                                               int32_t* __restrict__ seen, int delta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < B) { seen[2 * i] = seq_positions[i] + delta; seen[2 * i + 1] = num_protected[i]; }
}
}  // namespace kvc

// harvest in the decode attention's fused-metric epilogue (kvc_attention_kernels.h): keys that are the sum alone (no
// averaged metrics, no position bias: the epilogue does not have the bias table) -- the position-lazy form, or the
// reference's batch > 1 rule, for which the epilogue counts every head's masked slots as the full collecting pass does
static bool attention_harvest_plan(const kvc_schedule_params& p) {
  return harvest_plan(p) && !p.use_average && p.bias == nullptr && p.schedule_path != 3;
}
extern "C" int32_t kvc_attention_harvest_eligible(const kvc_schedule_params* p) {
  if (p == nullptr) return 0;
  return attention_harvest_plan(*p) ? 1 : 0;
}
extern "C" int kvc_attention_harvest_begin(const kvc_schedule_params* pp, kvc_stream_t stream) {
  using namespace kvc;
  if (pp == nullptr) return fail_invalid("attention_harvest_begin: null argument");
  const kvc_schedule_params& p = *pp;
  if (p.harvest_buf == nullptr || (reinterpret_cast<uintptr_t>(p.harvest_buf) & 15) != 0)
    return fail_invalid("attention_harvest_begin: harvest_buf must be a 16-byte aligned buffer of kvc_harvest_buffer_bytes()");
  if (!kvc_attention_harvest_eligible(pp))
    return fail_invalid("attention_harvest_begin: the call that follows is not eligible (kvc_attention_harvest_eligible)");
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  const HvLayout hl = hv_layout(G, p.num_seqs);
  uint8_t* hb = reinterpret_cast<uint8_t*>(p.harvest_buf);
  hipStream_t s = (hipStream_t)stream;
  fill32_async(hb + hl.claimed, 0u, hl.rec64 - hl.claimed, s);                          // claimed | cnt | def
  fill32_async(hb + hl.seen_ctx, 0xFFFFFFFFu, hl.seen_seq - hl.seen_ctx, s);           // no head has been walked yet
  hipLaunchKernelGGL(harvest_seen_seq_kernel, dim3((p.num_seqs + 255) / 256), dim3(256), 0, s, p.seq_positions, p.num_protected,
                     p.num_seqs, reinterpret_cast<int32_t*>(hb + hl.seen_seq), 0);
  return check_launch("attention_harvest_begin");
}

// pivot memory (harvest bits 1 and 2 without bit 0) needs no more than the small-eviction schedule itself:
// whatever its collecting pass streams, its lists are all the evictable keys below the pivots
extern "C" int32_t kvc_pivot_memory_eligible(const kvc_schedule_params* p) {
  if (p == nullptr) return 0;
  int p2 = 0, sshift = 0;
  return topk_plan(*p, p2, sshift) == KVC_WHY_TAKEN ? 1 : 0;
}
extern "C" size_t kvc_harvest_pivot_bytes(int32_t num_seqs) {
  if (num_seqs < 1) return 0;
  return kvc::hv_layout(1, num_seqs).claimed;
}
extern "C" size_t kvc_harvest_buffer_bytes(int32_t total_heads, int32_t num_seqs) {
  if (total_heads < 1 || num_seqs < 1) return 0;
  return kvc::hv_layout(total_heads, num_seqs).total;
}

// bracket schedule (section 9) or the digit rounds, for calls the small-eviction schedule does not
// take: a head per thread of one workgroup, list indices in 32 bits.
// Chosen by itself from 64 Ki slots per sequence and 64 blocks per head on: below that the digit
// rounds are as fast, and a head's list (an eighth of its slots) gets too short for the bracket.
// Returns why not (KVC_WHY_TAKEN = the bracket schedule).
static int bracket_why(const kvc_schedule_params& p) {
  const int LH = p.num_layers * p.num_kv_heads;
  const int64_t G = (int64_t)p.num_seqs * LH;
  if (G < 1 || p.total_slots <= 0 || p.block_size < 1) return KVC_WHY_EMPTY;
  if (p.uniform_evict) return KVC_WHY_UNIFORM;
  if (p.schedule_path != 0 && p.schedule_path != 4) return KVC_WHY_FORCED_PATH;
  // (the reference's batch > 1 rule: as many sequences as the single-launch fallback has tables for)
  if (p.mode == 0 && p.num_seqs > kvc::FB_MAX_COUPLED) return KVC_WHY_COUPLED_BATCH;
  if (LH > kvc::PIV_MAXLH) return KVC_WHY_HEADS_PER_SEQ;
  if (p.total_slots >= (int64_t)1 << 32) return KVC_WHY_INDEX_RANGE;
  if (p.schedule_path == 4) return KVC_WHY_TAKEN;
  if (!(p.total_slots / p.num_seqs >= 65536 && p.total_slots / G >= 64 * (int64_t)p.block_size)) return KVC_WHY_SMALL_BATCH;
  return KVC_WHY_TAKEN;
}
static bool bracket_plan(const kvc_schedule_params& p) { return bracket_why(p) == KVC_WHY_TAKEN; }

// introspection for tests and bench.py: which schedule a call with these parameters enqueues
// (0 = the digit rounds, 1 = small-eviction, 2 = bracket)
extern "C" int32_t kvc_schedule_evictions_plan(const kvc_schedule_params* p) {
  if (p == nullptr) return 0;
  int p2 = 0, sshift = 0;
  topk_plan(*p, p2, sshift);
  if (p2 > 0) return 1;
  return bracket_plan(*p) ? 2 : 0;
}

// ... and why: bits 0-7 why not the small-eviction schedule, bits 8-15 why not the bracket schedule
// (looked at only when the small-eviction one is not taken), bits 16-23 the form of the fallback
// behind a taken schedule (0 = the single launch of section 8, KVC_WHY_COUPLED_BATCH = the gated
// launch chain: the reference's batch > 1 rule over more sequences than its tables hold)
extern "C" int32_t kvc_schedule_evictions_plan_reason(const kvc_schedule_params* p) {
  if (p == nullptr) return KVC_WHY_EMPTY | (KVC_WHY_EMPTY << 8);
  int p2 = 0, sshift = 0;
  const int w1 = topk_plan(*p, p2, sshift);
  if (w1 == KVC_WHY_TAKEN)
    return (p->mode == 0 && p->num_seqs > kvc::FB_MAX_COUPLED) ? (KVC_WHY_COUPLED_BATCH << 16) : 0;
  return w1 | (bracket_why(*p) << 8);
}

// the key pass through the caller's block tables (build_keys_tables_kernel) or from the per-block
// metadata: tables given, a batch that takes less than half of the cache, 16-byte rows
static bool tables_plan(const kvc_schedule_params& p) {
  return p.block_tables != nullptr && p.seq_index_of_slot != nullptr && p.block_tables_width > 0 && p.max_num_seqs > 0 &&
         p.block_size >= 4 && p.block_size % 4 == 0 && p.total_slots < (int64_t)p.num_blocks * p.block_size / 2;
}
extern "C" int32_t kvc_schedule_evictions_uses_block_tables(const kvc_schedule_params* p) {
  if (p == nullptr || !tables_plan(*p)) return 0;
  int p2 = 0, sshift = 0;
  topk_plan(*p, p2, sshift);
  return p2 > 0 ? 0 : 1;                             // (the small-eviction schedule streams the store: no tables)
}

// introspection for tests and bench.py: 1 if a call with these parameters enqueues the
// small-eviction schedule; byte offset of its `fallback` word inside the workspace (non-zero
// after the call = the general pipeline behind it recomputed the result)
extern "C" int32_t kvc_schedule_evictions_uses_small_eviction_schedule(const kvc_schedule_params* p) {
  int p2 = 0, sshift = 0;
  if (p != nullptr) topk_plan(*p, p2, sshift);
  return p2 > 0 ? 1 : 0;
}
static WsLayout ws_layout(int64_t N, int32_t G, int32_t B, int32_t bs);
extern "C" size_t kvc_schedule_evictions_fallback_offset(int64_t total_slots, int32_t total_heads,
                                                         int32_t num_seqs, int32_t block_size) {
  if (block_size < 1) return 0;
  return ws_layout(total_slots, total_heads, num_seqs, block_size).fallback;
}

#ifdef KVC_BR_STAMPS
extern "C" size_t kvc_br_stamps_offset(int64_t total_slots, int32_t total_heads, int32_t num_seqs, int32_t block_size) {
  return ws_layout(total_slots, total_heads, num_seqs, block_size).head_fc;
}
#endif

extern "C" size_t kvc_schedule_evictions_workspace_bytes(int64_t total_slots, int32_t total_heads,
                                                         int32_t num_seqs, int32_t block_size) {
  if (block_size < 1) return 0;
  return ws_layout(total_slots, total_heads, num_seqs, block_size).total;
}

// ABI version 6: N and the eviction counts of a batch for a host that holds both as device tensors (the
// fork's call, vllm/kvcompress/scheduler.py:245-247, 491-499).  One workgroup: the [L,B,H] context lengths
// are at most a few hundred KiB; what the call costs is the launch and the wait, not the sum.
namespace kvc {
__global__ __launch_bounds__(1024) void batch_summary_kernel(const int32_t* __restrict__ context_lens, int total_heads,
                                                             int bs, const int32_t* __restrict__ k_per_seq, int num_seqs,
                                                             int64_t* __restrict__ out, int64_t ticket,
                                                             int64_t* __restrict__ n_dev = nullptr, int64_t n_bound = 0) {
  __shared__ unsigned long long part[16];
  unsigned long long blocks = 0;
  const int n4 = (reinterpret_cast<uintptr_t>(context_lens) & 15) == 0 ? total_heads / 4 : 0;
  const int4* c4 = reinterpret_cast<const int4*>(context_lens);
  const int sh = (bs & (bs - 1)) == 0 ? __ffs(bs) - 1 : -1;        // (a 32-bit division is ~40 instructions: 64 of them per thread)
  auto nb = [&](int c) -> unsigned { return sh >= 0 ? (unsigned)((c + bs - 1) >> sh) : (unsigned)((c + bs - 1) / bs); };
  for (int i0 = threadIdx.x; i0 < n4; i0 += 1024 * 8) {             // eight rows requested before the first is used
    int4 c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = i0 + u * 1024 < n4 ? c4[i0 + u * 1024] : int4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 8; ++u) blocks += nb(c[u].x) + nb(c[u].y) + nb(c[u].z) + nb(c[u].w);
  }
  for (int i = n4 * 4 + threadIdx.x; i < total_heads; i += 1024) blocks += nb(context_lens[i]);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) blocks += __shfl_xor(blocks, d, 64);
  if (lane_id() == 0) part[threadIdx.x >> 6] = blocks;
  for (int i = threadIdx.x; i < num_seqs; i += 1024) out[1 + i] = (int64_t)k_per_seq[i];
  if (ticket) __threadfence_system();             // (the counts are on their way before the ticket is)
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += part[w];
    out[0] = (int64_t)t * bs;
    // ABI version 8: N for the schedule's kernels behind this launch (and whether the host's bound holds)
    if (n_dev != nullptr) { n_dev[0] = (int64_t)t * bs; n_dev[1] = (int64_t)t * bs > n_bound ? 1 : 0; }
    if (ticket) {
      // a host that polls the ticket word sees the numbers a memory round trip after this store, without waiting
      // for the kernel's end-of-grid signal to travel through the runtime
      __threadfence_system();
      __hip_atomic_store(out + 1 + num_seqs, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
}  // namespace kvc

extern "C" int kvc_schedule_batch_summary(const int32_t* context_lens, int32_t total_heads, int32_t block_size,
                                          const int32_t* evicted_blocks_per_seq, int32_t num_seqs, int64_t* host_out,
                                          int32_t host_mapped, int32_t wait, void* workspace, size_t workspace_bytes,
                                          kvc_stream_t stream) {
  using namespace kvc;
  if (block_size < 1) return fail_invalid("Unsupported block size: " + std::to_string(block_size));
  if (context_lens == nullptr || total_heads < 0 || num_seqs < 0 || host_out == nullptr ||
      (num_seqs > 0 && evicted_blocks_per_seq == nullptr))
    return fail_invalid("schedule_batch_summary: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const size_t bytes = (size_t)(1 + num_seqs) * 8;
  int64_t* dst = host_out;
  if (!host_mapped) {
    if (workspace == nullptr || workspace_bytes < bytes || (reinterpret_cast<uintptr_t>(workspace) & 7) != 0)
      return fail_invalid("schedule_batch_summary: workspace too small or misaligned");
    dst = reinterpret_cast<int64_t*>(workspace);
  }
  batch_summary_kernel<<<1, 1024, 0, s>>>(context_lens, total_heads, block_size, evicted_blocks_per_seq, num_seqs, dst, (int64_t)0);
  if (int rc = check_launch("schedule_batch_summary")) return rc;
  if (!host_mapped && hipMemcpyAsync(host_out, dst, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) {
    set_error(std::string("schedule_batch_summary: ") + hipGetErrorString(hipGetLastError()));
    return KVC_ERR_HIP;
  }
  return wait ? kvc_schedule_batch_summary_wait(stream) : KVC_OK;
}

extern "C" int kvc_schedule_batch_summary_ticket(const int32_t* context_lens, int32_t total_heads, int32_t block_size,
                                                 const int32_t* evicted_blocks_per_seq, int32_t num_seqs,
                                                 int64_t* host_mapped_out, int64_t ticket, kvc_stream_t stream) {
  using namespace kvc;
  if (block_size < 1) return fail_invalid("Unsupported block size: " + std::to_string(block_size));
  if (context_lens == nullptr || total_heads < 0 || num_seqs < 0 || host_mapped_out == nullptr || ticket == 0 ||
      (num_seqs > 0 && evicted_blocks_per_seq == nullptr))
    return fail_invalid("schedule_batch_summary: bad arguments");
  batch_summary_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(context_lens, total_heads, block_size, evicted_blocks_per_seq,
                                                            num_seqs, host_mapped_out, ticket);
  return check_launch("schedule_batch_summary");
}

extern "C" int kvc_schedule_batch_summary_deferred(const int32_t* context_lens, int32_t total_heads, int32_t block_size,
                                                   const int32_t* evicted_blocks_per_seq, int32_t num_seqs,
                                                   int64_t* host_mapped_out, int64_t ticket, int64_t* total_slots_dev,
                                                   int64_t total_slots_bound, kvc_stream_t stream) {
  using namespace kvc;
  if (block_size < 1) return fail_invalid("Unsupported block size: " + std::to_string(block_size));
  if (context_lens == nullptr || total_heads < 0 || num_seqs < 0 || host_mapped_out == nullptr || ticket == 0 ||
      total_slots_dev == nullptr || total_slots_bound < 0 || (num_seqs > 0 && evicted_blocks_per_seq == nullptr))
    return fail_invalid("schedule_batch_summary: bad arguments");
  batch_summary_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(context_lens, total_heads, block_size, evicted_blocks_per_seq,
                                                            num_seqs, host_mapped_out, ticket, total_slots_dev, total_slots_bound);
  return check_launch("schedule_batch_summary");
}

extern "C" int kvc_schedule_batch_summary_wait(kvc_stream_t stream) {
  const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) {
    kvc::set_error(std::string("schedule_batch_summary: ") + hipGetErrorString(e));
    return KVC_ERR_HIP;
  }
  return KVC_OK;
}

extern "C" int kvc_schedule_evictions(const kvc_schedule_params* pp, void* workspace,
                                      size_t workspace_bytes, kvc_stream_t stream) {
  using namespace kvc;
  const kvc_schedule_params p = *pp;
  if (p.block_size < 1) return fail_invalid("Unsupported block size: " + std::to_string(p.block_size));
  // (the per-sequence tables of seq_prepare live in LDS: 20 B per sequence of the 160 KiB)
  if (p.num_seqs < 1 || p.num_seqs > 6500)
    return fail_invalid("schedule_evictions: num_seqs must be in [1,6500]");
  if (p.mode != 0 && p.mode != 1) return fail_invalid("schedule_evictions: mode must be 0 or 1");
  if (p.total_slots < 0 || p.total_slots >= (int64_t)2147483647)
    return fail_invalid("schedule_evictions: total slots must stay below 2^31 (int32 offsets)");
  if (p.total_slots % p.block_size != 0)
    return fail_invalid("schedule_evictions: total_slots must be a multiple of block_size");
  if (p.total_slots_dev != nullptr) {
    // ABI version 8: total_slots is a bound, N is on the device.  The schedules that take their decisions from N on the
    // device take it (digit rounds, bracket); the rest is the host's to avoid.
    if (p.max_evicted_blocks_hint >= 0 || (p.mode == 0 && p.num_seqs > 1) || p.uniform_evict || p.block_tables != nullptr ||
        (p.harvest & 5) != 0 || p.eli_dirty_map != nullptr || p.total_slots == 0)
      return fail_invalid("schedule_evictions: total_slots_dev goes with max_evicted_blocks_hint = -1, mode 1 or one sequence, "
                          "no uniform_evict, no block_tables, no harvested lists / remembered pivots, no dirty map, a bound > 0");
  }
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  const int B = p.num_seqs;
  const WsLayout l = ws_layout(p.total_slots, G, B, p.block_size);
  if (workspace_bytes < l.total) return fail_invalid("schedule_evictions: workspace too small");
  if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0)
    return fail_invalid("schedule_evictions: workspace must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  uint8_t* wb = reinterpret_cast<uint8_t*>(workspace);
  SchedWs ws;
  ws.keys = reinterpret_cast<uint32_t*>(wb + l.keys);
  ws.chunk_phys = reinterpret_cast<int32_t*>(wb + l.chunk_phys);
  ws.bsample = nullptr;
  ws.bnonfin = nullptr;
  ws.bclaim = nullptr;
  ws.bk = p.evicted_blocks_per_seq;
  ws.hist = reinterpret_cast<uint32_t*>(wb + l.hist);
  ws.cum = reinterpret_cast<uint32_t*>(wb + l.cum);
  ws.less = reinterpret_cast<uint32_t*>(wb + l.less);
  ws.eq = reinterpret_cast<uint32_t*>(wb + l.eq);
  ws.seq_prefix = reinterpret_cast<uint32_t*>(wb + l.seq_prefix);
  ws.seq_k = reinterpret_cast<int32_t*>(wb + l.seq_k);
  ws.seq_tmp = reinterpret_cast<int32_t*>(wb + l.seq_tmp);
  ws.rec64 = reinterpret_cast<uint64_t*>(wb + l.rec64);
  ws.st_cnt = reinterpret_cast<uint32_t*>(wb + l.st_cnt);
  ws.st_def = reinterpret_cast<uint32_t*>(wb + l.st_def);
  ws.st_samp = reinterpret_cast<uint32_t*>(wb + l.st_samp);
  ws.st_claimed = reinterpret_cast<uint32_t*>(wb + l.st_claimed);
  ws.st_seqrec = reinterpret_cast<SeqRec*>(wb + l.st_seqrec);
  ws.fallback = reinterpret_cast<uint32_t*>(wb + l.fallback);
  ws.bar = reinterpret_cast<uint32_t*>(wb + l.fallback + 128);     // (stamps; +128 / +256 the phase counters: same zeroed region)
  ws.head_fc = reinterpret_cast<uint32_t*>(wb + l.head_fc);
  ws.blist = reinterpret_cast<uint32_t*>(wb + l.blist);
  ws.bthr = reinterpret_cast<uint32_t*>(wb + l.bthr);
  ws.gate = nullptr;
  ws.n_dev = p.total_slots_dev;
  ws.hv_seen_ctx = nullptr;
  ws.hv_seen_seq = nullptr;
  if (p.total_slots == 0) {
    fill32_async(p.evicted_kv_count, 0u, (size_t)G * 4, s);
    fill32_async(p.evicted_block_count, 0u, (size_t)G * 4, s);
    return check_launch("schedule_evictions(empty)");
  }
  // keys default to "not evictable" (0xFFFFFFFF > KEY_INF) and the chunk table to -1 for slots
  // no physical block claims (inconsistent metadata); histograms and counters are zeroed by
  // the tail workgroups of build_keys
  const int LH = p.num_layers * p.num_kv_heads;
  int topk_p2 = 0, sshift = 0;
  topk_plan(p, topk_p2, sshift);
  const bool topk = topk_p2 > 0;
  const size_t prep_lds = (size_t)B * 24;
  if (prep_lds > 64 * 1024) {
    static std::atomic<uint64_t> prep_done{0};
    allow_dynamic_lds(reinterpret_cast<const void*>(seq_prepare_kernel), 160 * 1024 - 1024, prep_done);
  }
  if (topk) {
    // ---- small-eviction schedule (section 7): one fill of its counters, the output fill (side
    // stream), 6 launches (+ 2 for the reference's batch > 1 rule; + 1 that leaves the next call's
    // pivots, section 10); the general pipeline is enqueued behind it -- one gated launch (section 8)
    // -- and runs only if the flag was raised.  With the pivots of the call before (harvest bit 2)
    // the sampling pass and the pivot kernel are not launched; on lists the aggregation pass made
    // (bit 0) the collecting pass is not either: records | selection | next pivots | emission.
    // (on harvested lists the claims, counts and deficits are the harvest buffer's, zeroed where they were made, and
    // nothing is sampled: only the flag block is this call's to clear)
    fill32_async(wb + l.tz_begin, 0u, (p.harvest & 1) != 0 && p.harvest_buf != nullptr ? l.st_claimed - l.tz_begin : l.tz_end - l.tz_begin, s);
    SideStream* side = nullptr;
    if (!(p.lean & 1) && p.eli_dirty_map == nullptr) {
      if (p.total_slots >= (1 << 22)) side = side_stream(s);
      if (side == nullptr)
        fill32_async(p.evicted_logical_indices, (uint32_t)p.null_value, (size_t)p.total_slots * 4, s);
    }
    static std::atomic<uint64_t> sel_done{0};
    allow_dynamic_lds(reinterpret_cast<const void*>(seq_select_topk_kernel), 156 * 1024, sel_done);   // + its static tables
    // positions only for the slots whose metric lies below the pivot (stream_collect_kernel, LAZY):
    // keys that do not depend on the position, sequences that do not need each other's inf counts
    const bool lazy = lazy_plan(p);
    // harvest-ahead (section 10): the lists were made by the aggregation pass / a pivot is wanted for the next one
    const bool hv_ok = p.harvest_buf != nullptr;
    if ((p.harvest & 5) && !hv_ok)
      return fail_invalid("schedule_evictions: harvested lists / remembered pivots without a harvest buffer");
    const bool harvested = (p.harvest & 1) != 0;
    // the collecting pass with the pivots the previous call left behind instead of a sample's (bit 2)
    const bool remembered = !harvested && (p.harvest & 4) != 0;
    uint32_t* hv_pivot = nullptr;
    const uint32_t* hv_pivot_in = nullptr;
    if (hv_ok && (p.harvest & 7)) {
      if ((reinterpret_cast<uintptr_t>(p.harvest_buf) & 15) != 0) return fail_invalid("schedule_evictions: harvest_buf must be 16-byte aligned");
      uint8_t* hb = reinterpret_cast<uint8_t*>(p.harvest_buf);
      const HvLayout hl = hv_layout(G, B);
      if (p.harvest & 2) hv_pivot = reinterpret_cast<uint32_t*>(hb + hl.pivot);
      hv_pivot_in = reinterpret_cast<const uint32_t*>(hb + hl.pivot);
      if (harvested) {
        ws.st_claimed = reinterpret_cast<uint32_t*>(hb + hl.claimed);
        ws.st_cnt = reinterpret_cast<uint32_t*>(hb + hl.cnt);
        ws.st_def = reinterpret_cast<uint32_t*>(hb + hl.def);
        ws.rec64 = reinterpret_cast<uint64_t*>(hb + hl.rec64);
        if (p.harvest & 8) {                         // made by the attention's epilogue: verified against this call's batch
          // (lists of the attention's epilogue cannot serve averaged or biased metrics; the host does not offer them
          // there -- kvc_attention_harvest_eligible -- and lists of the aggregation pass, which may, are full keys)
          ws.hv_seen_ctx = reinterpret_cast<const int32_t*>(hb + hl.seen_ctx);
          ws.hv_seen_seq = reinterpret_cast<const int32_t*>(hb + hl.seen_seq);
        }
      }
    }
    const float hv_widen = p.harvest_widen > 0.0f ? p.harvest_widen : 0.25f;
    if (!harvested && !remembered) {
      int64_t sb = (p.num_blocks + 2047) / 2048;     // >= 8 steps of 64 block indices per wave
      sb = sb < 1 ? 1 : (sb > 4096 ? 4096 : sb);
      const dim3 grid((unsigned)sb), blk(256);
      if (p.block_size == 8) hipLaunchKernelGGL(stream_sample_kernel<8>, grid, blk, 0, s, p, ws, sshift);
      else if (p.block_size == 16) hipLaunchKernelGGL(stream_sample_kernel<16>, grid, blk, 0, s, p, ws, sshift);
      else hipLaunchKernelGGL(stream_sample_kernel<32>, grid, blk, 0, s, p, ws, sshift);
    }
    if (side != nullptr) {
      if (hipEventRecord(side->fork, s) != hipSuccess || hipStreamWaitEvent(side->s2, side->fork, 0) != hipSuccess) {
        (void)hipGetLastError();                     // (no side stream after all: inline)
        side = nullptr;
      }
      fill32_async(p.evicted_logical_indices, (uint32_t)p.null_value, (size_t)p.total_slots * 4, side != nullptr ? side->s2 : s);
      if (side != nullptr && hipEventRecord(side->join, side->s2) != hipSuccess) {
        // no event to wait for: the fill is waited for here and now (a later wait on `join` would
        // refer to an earlier call's record), and this thread fills inline from now on
        (void)hipGetLastError();
        (void)hipStreamSynchronize(side->s2);
        side->failed = true;
        side = nullptr;
      }
    }
    if (remembered) hipLaunchKernelGGL(seqrec_from_pivots_kernel, dim3((B + 255) / 256), dim3(256), 0, s, p, ws, hv_pivot_in);
    else if (!harvested) hipLaunchKernelGGL(stream_pivot_kernel, dim3(B), dim3(1024), 0, s, p, ws, sshift);
    if (!harvested) {
      // blocks of the batch / blocks of the cache: a dense cache requests the rows before it has
      // looked at the metadata, a sparse one (engine-sized cache, small batch) only the batch's rows
      const bool dense = p.total_slots >= (int64_t)p.num_blocks * p.block_size / 2;
      int64_t cb = dense ? (p.num_blocks + 255) / 256 : (p.num_blocks + kvc::SPARSE_CHUNK - 1) / kvc::SPARSE_CHUNK;
      cb = cb < 1 ? 1 : (cb > KVC_COLLECT_GRID_CAP ? KVC_COLLECT_GRID_CAP : cb);
      const dim3 grid((unsigned)cb), blk(256);
#define KVC_COLLECT(BSV)                                                                                  \
      if (lazy) {                                                                                         \
        if (dense) hipLaunchKernelGGL((stream_collect_kernel<BSV, true, true>), grid, blk, 0, s, p, ws);  \
        else hipLaunchKernelGGL((stream_collect_kernel<BSV, false, true>), grid, blk, 0, s, p, ws);       \
      } else {                                                                                            \
        if (dense) hipLaunchKernelGGL((stream_collect_kernel<BSV, true, false>), grid, blk, 0, s, p, ws); \
        else hipLaunchKernelGGL((stream_collect_kernel<BSV, false, false>), grid, blk, 0, s, p, ws);      \
      }
      if (p.block_size == 8) { KVC_COLLECT(8); }
      else if (p.block_size == 16) { KVC_COLLECT(16); }
      else { KVC_COLLECT(32); }
#undef KVC_COLLECT
    }
    const int coupled_tk = (p.mode == 0 && B > 1) ? 1 : (lazy ? 2 : 0);
    // records + selection + emission as ONE launch (section 7b) where the sequences do not need each other's counts
    // and a sequence's heads fit one workgroup's waves; KVC_TOPK_CHAIN=1 keeps the launch chain (tests compare the two)
    static const bool chain_env = [] { const char* e = getenv("KVC_TOPK_CHAIN"); return e != nullptr && e[0] != '\0' && e[0] != '0'; }();
    const bool fused_tk = coupled_tk != 1 && LH <= 256 && topk_p2 <= 8192 && !chain_env;
    if (fused_tk) {
      if (side != nullptr && hipStreamWaitEvent(s, side->join, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(side->s2);        // (the emission must not race the fill)
        side->failed = true;
      }
      const int hpw = LH <= 64 ? 4 : 16;
      // the pivots for the next decode step's harvest, from what is left of this call's lists (section 10): the kernel's
      // last phase (KVC_TOPK_PIVOT_LAUNCH=1: harvest_pivot_kernel behind it, as on the launch chain -- tests compare)
      static const bool piv_env = [] { const char* e = getenv("KVC_TOPK_PIVOT_LAUNCH"); return e != nullptr && e[0] != '\0' && e[0] != '0'; }();
      const bool piv_in = hv_pivot != nullptr && !piv_env;
      const int lds_keys = piv_in ? (LH * WAVE < kvc::HVP_LDS_KEYS ? LH * WAVE : kvc::HVP_LDS_KEYS) : 0;
      const size_t fl_sel = (size_t)topk_p2 * 8 + (size_t)16 * hpw * 4;
      const size_t fl = fl_sel > (size_t)lds_keys * 4 ? fl_sel : (size_t)lds_keys * 4;
      const int fh = (harvested || remembered) ? 1 : 0;
      static std::atomic<uint64_t> f4_done{0}, f16_done{0}, f4p_done{0}, f16p_done{0};
#define KVC_FUSED(HPWV, PIVV, DONE)                                                                                        \
      do {                                                                                                                 \
        allow_dynamic_lds(reinterpret_cast<const void*>(topk_fused_kernel<HPWV, PIVV>), 100 * 1024, DONE);                 \
        hipLaunchKernelGGL((topk_fused_kernel<HPWV, PIVV>), dim3(B), dim3(1024), fl, s, p, ws, topk_p2, lazy ? 1 : 0,      \
                           coupled_tk, lds_keys, piv_in ? hv_pivot : nullptr, fh, hv_widen);                               \
      } while (0)
      if (hpw == 4) { if (piv_in) KVC_FUSED(4, true, f4p_done); else KVC_FUSED(4, false, f4_done); }
      else { if (piv_in) KVC_FUSED(16, true, f16p_done); else KVC_FUSED(16, false, f16_done); }
#undef KVC_FUSED
      if (hv_pivot != nullptr && !piv_in)
        hipLaunchKernelGGL(harvest_pivot_kernel, dim3(B), dim3(1024), 0, s, p, ws, hv_pivot, fh, hv_widen);
    } else {
      hipLaunchKernelGGL((stream_records_kernel<4, 4>), dim3((G + 15) / 16), dim3(256), 0, s, p, ws, lazy ? 1 : 0);
      if (coupled_tk == 1) {
        hipLaunchKernelGGL(seq_sums_topk_kernel, dim3(B), dim3(256), 0, s, p, ws);
        hipLaunchKernelGGL(seq_prepare_kernel, dim3(1), dim3(1024), prep_lds, s, p, ws);
      }
      hipLaunchKernelGGL(seq_select_topk_kernel, dim3(B), dim3(1024), (size_t)topk_p2 * 8 + (size_t)LH * 4, s, p, ws, topk_p2, coupled_tk);
      // the pivots for the next decode step's harvest, from what is left of this call's lists (section 10)
      if (hv_pivot != nullptr)
        hipLaunchKernelGGL(harvest_pivot_kernel, dim3(B), dim3(1024), 0, s, p, ws, hv_pivot, (harvested || remembered) ? 1 : 0, hv_widen);
      if (side != nullptr && hipStreamWaitEvent(s, side->join, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(side->s2);          // (the emission below must not race the fill)
        side->failed = true;
      }
      hipLaunchKernelGGL(emit_topk_kernel<4>, dim3((G + 3) / 4), dim3(256), 0, s, p, ws);
    }
    ws.gate = ws.fallback;
    claims_clean_at().store(0, std::memory_order_release);   // (this schedule counts its claimed blocks in the same words)
  }
  if (topk && !(p.mode == 0 && B > kvc::FB_MAX_COUPLED)) {
    // ---- the general pipeline as ONE gated launch (section 8)
    uint4* z16 = reinterpret_cast<uint4*>(wb + l.zero_begin);
    const int64_t zv = (int64_t)((l.zero_end - l.zero_begin) / 16);
    const int sparse = p.total_slots < (int64_t)p.num_blocks * p.block_size / 2 ? 1 : 0;
    const unsigned vgrid = (unsigned)fallback_grid();
    hipLaunchKernelGGL(fallback_general_kernel, dim3(p.fallback_grid > 0 ? (unsigned)p.fallback_grid : vgrid), dim3(256), 0, s,
                       p, ws, sparse, z16, zv, 0, vgrid,
                       (p.harvest_buf != nullptr && (p.harvest & 2))
                           ? reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(p.harvest_buf) + hv_layout(G, B).pivot) : nullptr);
    return check_launch("schedule_evictions");
  }
  // ---- general pipeline
  // keys default to "not evictable" (0xFFFFFFFF > KEY_INF) and the chunk table to -1 for slots
  // no physical block claims (inconsistent metadata); histograms and counters are zeroed by
  // the tail workgroups of build_keys.  Behind the small-eviction schedule (gated) the clear is a
  // gated kernel instead of a memset.
  // bulk evictions: T* from a bracket around a sample's quantile
  // instead of four digit rounds (section 9, bracket_plan)
  const bool bracket = !topk && bracket_plan(p);
  // a batch that is sparse in its cache, with the caller's block tables at hand: the keys in logical
  // order through the tables (build_keys_tables_kernel; every slot is written: no clearing)
  const bool by_tables = !topk && tables_plan(p);
  // (the bracket's sample -- 128 KiB per sequence -- lies behind the keys and the chunk table: cleared with them
  // only when that schedule runs)
  const bool bracket_coupled = bracket && p.mode == 0 && B > 1;
  // the bracket schedule without that fill (33 MB at 8 M slots, 12-17 us in front of everything): the key pass counts
  // the logical blocks it finds a physical block for, bracket_kernel compares with N / bs (and leaves the counters
  // zero).  They are clean when this workspace's last call was such a call; otherwise 8 KiB are cleared first.
  const bool count_claims = bracket && !bracket_coupled && !(p.lean & 2) && !by_tables && p.block_size % 4 == 0;
  if (count_claims) {
    std::atomic<uintptr_t>& slot = claims_clean_at();
    const uintptr_t here = reinterpret_cast<uintptr_t>(wb + l.st_claimed);
    if (slot.exchange(here, std::memory_order_acq_rel) != here)
      fill32_async(wb + l.st_claimed, 0u, (size_t)kvc::CLAIM_SHARDS * 128, s);
    ws.bclaim = ws.st_claimed;
  } else {
    claims_clean_at().store(0, std::memory_order_release);   // (other schedules use the counters their own way)
    if (!topk && !(p.lean & 2) && !by_tables) fill32_async(ws.keys, 0xFFFFFFFFu, (bracket ? l.zero_begin : l.bsample) - l.keys, s);
  }
  if (bracket) ws.bsample = reinterpret_cast<uint32_t*>(wb + l.bsample);   // build_keys leaves the sample behind
  if (bracket_coupled) {                             // ... and counts the keys that are not evictable
    fill32_async(wb + l.tz_begin, 0u, l.tz_end - l.tz_begin, s);
    ws.bnonfin = ws.st_samp;
    ws.bk = ws.seq_k;
  }
  if (topk && !(p.lean & 2)) hipLaunchKernelGGL(clear_chunk_table_kernel, dim3(1024), dim3(256), 0, s, p, ws);
  {
    uint4* z16 = reinterpret_cast<uint4*>(wb + l.zero_begin);
    const int64_t zv = (int64_t)((l.zero_end - l.zero_begin) / 16);
    const int64_t zb64 = (zv + 1023) / 1024;
    const unsigned zb = (unsigned)(zb64 < 1 ? 1 : (zb64 > 2048 ? 2048 : zb64));
    const int64_t threads = p.block_size % 4 == 0 ? p.num_blocks * (p.block_size / 4) : p.num_blocks * p.block_size;
    // grid-stride: at most 16 Ki workgroups (an engine-sized cache has tens of millions of blocks,
    // most of them outside the batch: one workgroup per 64 of them cost 0.5 ms in dispatch alone);
    // half of that behind the small-eviction schedule, where the launch is a no-op unless the flag was raised
    int64_t db64 = (threads + 255) / 256;
    const int64_t cap = topk ? 8192 : 16384;
    if (db64 > cap) db64 = cap;
    const unsigned db = (unsigned)db64;
    const int bsz = p.block_size;
    // blocks of the batch / blocks of the cache: a dense cache is faster with one independent thread
    // per 4 slots (build_keys_kernel), a sparse one with the compacting sweep
    const bool sparse = p.total_slots < (int64_t)p.num_blocks * bsz / 2;
    if (by_tables) {
      int64_t tb64 = (p.total_slots + 1023) / 1024;
      if (tb64 > cap) tb64 = cap;
      const unsigned tbk = (unsigned)(tb64 < 1 ? 1 : tb64);
      hipLaunchKernelGGL(build_keys_tables_kernel, dim3(tbk + zb), dim3(256), 0, s, p, ws, tbk, z16, zv);
    } else if (sparse && (bsz == 4 || bsz == 8 || bsz == 16 || bsz == 32 || bsz == 64)) {
      // one workgroup per SPARSE_CHUNK blocks (grid-stride when capped)
      int64_t wb64 = (p.num_blocks + kvc::SPARSE_CHUNK - 1) / kvc::SPARSE_CHUNK;
      if (wb64 > cap) wb64 = cap;
      const unsigned wbk = (unsigned)(wb64 < 1 ? 1 : wb64);
      hipLaunchKernelGGL(build_keys_sparse_kernel, dim3(wbk + zb), dim3(256), 0, s, p, ws, wbk, z16, zv);
    } else if (p.block_size % 4 == 0)
      hipLaunchKernelGGL(build_keys_kernel<4>, dim3(db + zb), dim3(256), 0, s, p, ws, db, z16, zv);
    else
      hipLaunchKernelGGL(build_keys_kernel<1>, dim3(db + zb), dim3(256), 0, s, p, ws, db, z16, zv);
  }
  if (topk && !(p.lean & 2)) hipLaunchKernelGGL(fix_unclaimed_kernel, dim3(1024), dim3(256), 0, s, p, ws);
  const int64_t htiles_all = (p.total_slots + HTILE - 1) / HTILE;
  // persistent grid: 4 workgroups per CU up to 4M keys per round-pass, growing to 16 per CU for
  // very large batches (measured: 1024 is best at 8M keys, 4096 is 18 % faster at 270M)
  int64_t hgrid = htiles_all / 32;
  hgrid = hgrid < 1024 ? 1024 : (hgrid > 4096 ? 4096 : hgrid);
  const unsigned htiles = (unsigned)(htiles_all < hgrid ? htiles_all : hgrid);
  if (bracket) {
    // ---- bracket schedule (section 9) over the keys just built; the digit rounds behind it as the
    // single gated launch of section 8
    static std::atomic<uint64_t> bsel_done{0};
    allow_dynamic_lds(reinterpret_cast<const void*>(bracket_select_kernel), 140 * 1024, bsel_done);   // (+ 17.7 KiB static)
    int p2 = 1024;
    while (p2 < LH * 128 && p2 < 32768) p2 <<= 1;      // room for ~128 listed thresholds per head
    const size_t sel_lds = (size_t)p2 * 4 + (size_t)(2 * LH + 1) * 4;
    if (bracket_coupled) {                           // k' of every sequence first (the batch > 1 rule)
      hipLaunchKernelGGL(bracket_totals_kernel, dim3(B), dim3(256), 0, s, p, ws);
      hipLaunchKernelGGL(seq_prepare_kernel, dim3(1), dim3(1024), prep_lds, s, p, ws);
    }
    hipLaunchKernelGGL(bracket_kernel, dim3(B), dim3(1024), 0, s, p, ws);
    // (1024 workgroups at 8 M keys: 512 take 23 us, 2048 34, 4096 65 -- what a workgroup costs is its returning adds on
    // the heads' list counters, 16 per address at this grid, not its keys)
    hipLaunchKernelGGL(count_collect_kernel, dim3(htiles), dim3(256), 0, s, p, ws);
    hipLaunchKernelGGL(bracket_records_kernel, dim3(G), dim3(512), 0, s, p, ws);
    hipLaunchKernelGGL(bracket_select_kernel, dim3(B), dim3(1024), sel_lds, s, p, ws, p2);
    {
      const int64_t avg = p.total_slots / G;
      int64_t want = (avg + avg / 4 + 2047) / 2048 * 2048;
      const int lds_cap = (int)(want < 2048 ? 2048 : (want > 32768 ? 32768 : want));
      if (avg <= 8192) {
        hipLaunchKernelGGL(select_emit_kernel<256>, dim3(G), dim3(256), (size_t)lds_cap * 4, s, p, ws, lds_cap, 1);
      } else {
        static std::atomic<uint64_t> long_done_b{0};
        allow_dynamic_lds(reinterpret_cast<const void*>(select_emit_kernel<1024>), 32768 * 4, long_done_b);
        hipLaunchKernelGGL(select_emit_kernel<1024>, dim3(G), dim3(1024), (size_t)lds_cap * 4, s, p, ws, lds_cap, 1);
      }
    }
    ws.gate = ws.fallback;
    uint4* z16 = reinterpret_cast<uint4*>(wb + l.zero_begin);
    const int64_t zv = (int64_t)((l.zero_end - l.zero_begin) / 16);
    const unsigned vgrid = (unsigned)fallback_grid();
    // (FB_HOLES_BIT: the keys are made anew, by the pass the launch above used)
    const int bsz = p.block_size;
    const int fb_sparse = (p.total_slots < (int64_t)p.num_blocks * bsz / 2 && (bsz == 4 || bsz == 8 || bsz == 16 || bsz == 32 || bsz == 64)) ? 1 : 0;
    hipLaunchKernelGGL(fallback_general_kernel, dim3(p.fallback_grid > 0 ? (unsigned)p.fallback_grid : vgrid), dim3(256), 0, s,
                       p, ws, fb_sparse, z16, zv, 1, vgrid, (uint32_t*)nullptr);
    return check_launch("schedule_evictions");
  }
  if (p.uniform_evict) {
    // ---- the reference's uniform_evict rule (section 5b): keys, one histogram round (the heads'
    // evictable keys), the per-head counts, select + emit with every head's own select
    hipLaunchKernelGGL(hist_round_kernel, dim3(htiles), dim3(256), 0, s, p, ws, 0);
    hipLaunchKernelGGL(uniform_counts_kernel, dim3((G + 3) / 4), dim3(256), 0, s, p, ws);
    const int64_t avg = p.total_slots / G;
    int64_t want = (avg + avg / 4 + 2047) / 2048 * 2048;
    const int lds_cap = (int)(want < 2048 ? 2048 : (want > 32768 ? 32768 : want));
    if (avg <= 8192) {
      hipLaunchKernelGGL(select_emit_kernel<256>, dim3(G), dim3(256), (size_t)lds_cap * 4, s, p, ws, lds_cap, 2);
    } else {
      static std::atomic<uint64_t> long_done_u{0};
      allow_dynamic_lds(reinterpret_cast<const void*>(select_emit_kernel<1024>), 32768 * 4, long_done_u);
      hipLaunchKernelGGL(select_emit_kernel<1024>, dim3(G), dim3(1024), (size_t)lds_cap * 4, s, p, ws, lds_cap, 2);
    }
    return check_launch("schedule_evictions");
  }
  // per round: the histograms, then ONE launch for scan + pick (round 0: + the chunk totals and k',
  // round 3: + the per-head counts).  Only the reference's batch > 1 rule (mode 0, B > 1) couples
  // the sequences in round 0 and takes three launches there (totals | k' | pick).  10 (12) launches
  // + the memset.
  const bool coupled = p.mode == 0 && B > 1;
  for (int round = 0; round < 4; ++round) {
    hipLaunchKernelGGL(hist_round_kernel, dim3(htiles), dim3(256), 0, s, p, ws, round);
    if (round == 0 && coupled) {
      hipLaunchKernelGGL(scan_pick_kernel, dim3(B), dim3(1024), 0, s, p, ws, 0, 1);
      hipLaunchKernelGGL(seq_prepare_kernel, dim3(1), dim3(1024), prep_lds, s, p, ws);
      hipLaunchKernelGGL(scan_pick_kernel, dim3(B), dim3(1024), 0, s, p, ws, 0, 2);
    } else {
      hipLaunchKernelGGL(scan_pick_kernel, dim3(B), dim3(1024), 0, s, p, ws, round, 0);
    }
  }
  {
    // stage a head's keys in LDS when the average head fits with 25 % slack (ragged heads
    // that do not fit read from L2).  Small heads (the continual-compression steady state:
    // thousands of heads of a few thousand slots) take 256-thread workgroups and a small
    // buffer so that 6+ heads share a CU; the per-head passes are barrier-latency bound.
    const int64_t avg = p.total_slots / G;
    int64_t want = (avg + avg / 4 + 2047) / 2048 * 2048;
    // (heads beyond 32k slots read their keys from L2: measured faster than one 144 KiB
    // staging workgroup per CU)
    const int lds_cap = (int)(want < 2048 ? 2048 : (want > 32768 ? 32768 : want));
    const unsigned sel_grid = (unsigned)((topk && G > 2048) ? 2048 : G);   // (gated: see select_emit_kernel)
    // > 64 KiB of dynamic LDS needs the opt-in (gfx950 has 160 KiB per CU); per device, cheap
    if (avg <= 8192) {
      hipLaunchKernelGGL(select_emit_kernel<256>, dim3(sel_grid), dim3(256), (size_t)lds_cap * 4, s, p, ws, lds_cap, 0);
    } else {
      static std::atomic<uint64_t> long_done{0};
      allow_dynamic_lds(reinterpret_cast<const void*>(select_emit_kernel<1024>), 32768 * 4, long_done);
      hipLaunchKernelGGL(select_emit_kernel<1024>, dim3(sel_grid), dim3(1024), (size_t)lds_cap * 4, s, p, ws, lds_cap, 0);
    }
  }
  return check_launch("schedule_evictions");
}

// A2a with section 10's harvest: `p` describes the schedule call that will follow
extern "C" int kvc_aggregate_decode_harvest(const kvc_schedule_params* pp, float* temp_metrics, int32_t num_queries_per_kv,
                                            int32_t use_l2, int32_t clear_temp, kvc_stream_t stream) {
  using namespace kvc;
  if (pp == nullptr || temp_metrics == nullptr) return fail_invalid("aggregate_decode_harvest: null argument");
  const kvc_schedule_params p = *pp;
  if (p.harvest_buf == nullptr || (reinterpret_cast<uintptr_t>(p.harvest_buf) & 15) != 0)
    return fail_invalid("aggregate_decode_harvest: harvest_buf must be a 16-byte aligned buffer of kvc_harvest_buffer_bytes()");
  if (!kvc_harvest_eligible(pp, num_queries_per_kv))
    return fail_invalid("aggregate_decode_harvest: the call that follows is not eligible (kvc_harvest_eligible)");
  if (p.num_blocks <= 0) return KVC_OK;
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  const HvLayout hl = hv_layout(G, p.num_seqs);
  uint8_t* hb = reinterpret_cast<uint8_t*>(p.harvest_buf);
  hipStream_t s = (hipStream_t)stream;
  SchedWs ws{};
  ws.st_claimed = reinterpret_cast<uint32_t*>(hb + hl.claimed);
  ws.st_cnt = reinterpret_cast<uint32_t*>(hb + hl.cnt);
  ws.st_def = reinterpret_cast<uint32_t*>(hb + hl.def);
  ws.rec64 = reinterpret_cast<uint64_t*>(hb + hl.rec64);
  const uint32_t* hv_pivot = reinterpret_cast<const uint32_t*>(hb + hl.pivot);
  const bool lazy = lazy_plan(p);
  fill32_async(hb + hl.claimed, 0u, hl.rec64 - hl.claimed, s);       // claimed | cnt | def
  // what the lists are made with, for a schedule call that takes them with harvest bit 3 (verified on the device):
  // positions (+ delta) and protected windows; the context lengths are not recorded (-2: the walked blocks are counted)
  fill32_async(hb + hl.seen_ctx, 0xFFFFFFFEu, hl.seen_seq - hl.seen_ctx, s);
  hipLaunchKernelGGL(harvest_seen_seq_kernel, dim3((p.num_seqs + 255) / 256), dim3(256), 0, s, p.seq_positions, p.num_protected,
                     p.num_seqs, reinterpret_cast<int32_t*>(hb + hl.seen_seq), p.harvest_position_delta);
  // a wave iteration covers 64 blocks; at most 16 Ki workgroups of 4 waves (grid-stride beyond)
  int64_t cb = (p.num_blocks + 255) / 256;
#ifndef KVC_HV_GRID
#define KVC_HV_GRID 16384
#endif
  cb = cb < 1 ? 1 : (cb > KVC_HV_GRID ? KVC_HV_GRID : cb);
  const dim3 grid((unsigned)cb), blk(256);
  const bool big = p.num_blocks * (int64_t)p.block_size >= (int64_t)1 << 28;        // >= 1 GiB of metrics
#define KVC_HARVEST3(BSV, QVV, LZ)                                                                                  \
  if (big)                                                                                                           \
    hipLaunchKernelGGL((aggregate_harvest_kernel<BSV, QVV, true, LZ>), grid, blk, 0, s, p, ws, temp_metrics, hv_pivot, num_queries_per_kv, use_l2, clear_temp); \
  else                                                                                                               \
    hipLaunchKernelGGL((aggregate_harvest_kernel<BSV, QVV, false, LZ>), grid, blk, 0, s, p, ws, temp_metrics, hv_pivot, num_queries_per_kv, use_l2, clear_temp);
#define KVC_HARVEST2(BSV, QVV)                                                                                      \
  if (lazy) { KVC_HARVEST3(BSV, QVV, true) } else { KVC_HARVEST3(BSV, QVV, false) }
#define KVC_HARVEST(BSV)                                                                                             \
  if (num_queries_per_kv == 4) { KVC_HARVEST2(BSV, 1) } else if (num_queries_per_kv == 8) { KVC_HARVEST2(BSV, 2) } else { KVC_HARVEST2(BSV, 0) }
  if (p.block_size == 8) { KVC_HARVEST(8); }
  else if (p.block_size == 16) { KVC_HARVEST(16); }
  else { KVC_HARVEST(32); }
#undef KVC_HARVEST2
#undef KVC_HARVEST3
#undef KVC_HARVEST
  return check_launch("aggregate_decode_harvest");
}
