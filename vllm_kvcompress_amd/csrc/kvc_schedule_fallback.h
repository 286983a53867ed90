// kvc_schedule_fallback.h -- A3 schedule_evictions: the general pipeline as ONE gated launch (what a schedule that cannot finish exactly falls back to)
// (one translation unit: included by kvc_schedule.hip in this order; see the overview there)
#pragma once
#include "kvc_common.h"
#include "kvc_schedule_common.h"
#include "../../include/kvc_mi355x.h"

namespace kvc {

// ------------------------------------------------------------------ 8. the fallback in ONE launch
// HIP has no conditional enqueue: behind the small-eviction schedule the general pipeline used to
// be 13 launches that read the flag and return, ~4.6 us each -- 60 us of a 150 us schedule at 16
// resident sequences.  This kernel is the whole general pipeline (for sequences that do not couple:
// mode 1 or a single one; the batch > 1 rule up to FB_MAX_COUPLED sequences) in ONE launch; with
// the flag down it is one launch that returns.
//
// Its phases depend on each other across workgroups, and nothing guarantees that a grid is resident
// at once (another stream, another process or a CU mask may hold compute units whatever the
// occupancy query says): a barrier that waits for every WORKGROUP to arrive can wait for one that
// has not started.  So the phases do not wait for workgroups, they wait for WORK: a phase is cut
// into V virtual workgroups (the bodies take their index and count as arguments), the real
// workgroups claim them from a counter until none is left and then wait until V of them are done.
// Whatever is resident does all of the work; a workgroup that starts late finds the counters of
// the finished phases full and falls through them.  Every claimed piece is being executed by a
// workgroup that runs, so every wait ends: correctness does not depend on co-residency, only speed
// does (the host still sizes the grid to what the occupancy query says is resident at once).
// Publishing a piece is the release / acquire recipe of cdna_hip_programming.md Guideline 16: every
// wave's stores are complete at the workgroup barrier, lane 0 writes the XCD's L2 back (release,
// agent scope) and adds to the phase's done counter; a waiter polls it with relaxed loads and a
// sleep, invalidates the CU's L1 (acquire), and the workgroup barrier hands that to the other waves.
// A wait that does not end within ten seconds of the 100 MHz wall clock (a device that lost a
// workgroup: must not happen) raises bit 1 of the flag word, which is sticky: every workgroup that
// sees it stops and overwrites the outputs with the schedule that evicts NOTHING (zero counts, a
// null list) -- never a partial one -- and the host raises when it reads the bit (metrics.py).
constexpr int FB_PHASES = 32;                        // claim / done counters (14 phases at most)
constexpr uint32_t FB_TIMEOUT_BIT = 2u;
struct FbSync {
  uint32_t* claim;       // [FB_PHASES] virtual workgroups handed out
  uint32_t* done;        // [FB_PHASES] ... finished
  uint32_t* flag;        // the schedule's flag word (bit 1: a wait timed out, results void)
  uint32_t* mirror;      // kvc_schedule_params.flag_mirror (page-locked host word) or nullptr
  uint32_t ticket;
};
__device__ __forceinline__ void fb_mirror(uint32_t* mirror, uint32_t ticket, uint32_t flag) {
  if (mirror != nullptr) __hip_atomic_store(mirror, (ticket << 8) | (flag & 0xFFu), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// runs body(v, V) for the virtual workgroups v this workgroup can claim, then waits for all V;
// false = the wait was given up (or somebody else gave up): stop
template <typename F>
__device__ __forceinline__ bool fb_phase(const FbSync& fs, uint32_t phase, uint32_t V, uint32_t* word_s, F&& body) {
  uint32_t* claim = fs.claim + phase;
  uint32_t* done = fs.done + phase;
  for (;;) {
    __syncthreads();                                 // (word_s and the body's LDS are free again)
    if (threadIdx.x == 0) *word_s = __hip_atomic_fetch_add(claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t v = *word_s;
    if (v >= V) break;
    body(v, V);
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t ok = 1u;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < V) {
      __builtin_amdgcn_s_sleep(16);
      if (__hip_atomic_load(fs.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & FB_TIMEOUT_BIT) { ok = 0u; break; }
      if (wall_clock64() - t0 > 1000000000ull) {
        const uint32_t was = atomicOr(fs.flag, FB_TIMEOUT_BIT);
        fb_mirror(fs.mirror, fs.ticket, was | FB_TIMEOUT_BIT);      // (the host hears of a fault whoever notices it)
        ok = 0u;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *word_s = ok;
  }
  __syncthreads();
  return *word_s != 0u;
}

// the schedule that evicts nothing (what a call leaves behind when a wait was given up)
__device__ __forceinline__ void fb_void_outputs(const kvc_schedule_params& p, const SchedWs& ws, unsigned bid, unsigned nb) {
  const int G = p.num_seqs * p.num_layers * p.num_kv_heads;
  for (int64_t g = (int64_t)bid * 256 + threadIdx.x; g < G; g += (int64_t)nb * 256) {
    p.evicted_kv_count[g] = 0;
    p.evicted_block_count[g] = 0;
  }
  const int64_t N = true_n(p, ws);
  for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < N; i += (int64_t)nb * 256)
    p.evicted_logical_indices[i] = p.null_value;
}

constexpr int FB_MAX_COUPLED = 256;                  // sequences whose batch > 1 rule fits the static tables below
// have_keys: the key pass ran already (the bracket schedule's) -- straight to the digit rounds
// vgrid: virtual workgroups of the streaming phases (the grid the host would like to be resident)
__global__ __launch_bounds__(256, 2) void fallback_general_kernel(kvc_schedule_params p, SchedWs ws, int sparse,
                                                               uint4* zero16, int64_t zero_vecs, int have_keys,
                                                               unsigned vgrid, uint32_t* hv_pivot) {
  if (voided(ws)) return;
  const uint32_t flag0 = __hip_atomic_load(ws.fallback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // the call's flag word for the host, in page-locked memory (what the schedule in front raised is final here)
  if (blockIdx.x == 0 && threadIdx.x == 0) fb_mirror(p.flag_mirror, p.flag_ticket, flag0);
  if (flag0 == 0u) return;                           // flag down: this launch is all the fallback costs
  // The pivots this call left for the next one were computed from lists that just turned out not to be trustworthy
  // (a workgroup of topk_fused_kernel reads the flag once, before its emission: it may have written next pivots while
  // another one was raising it).  The host drops them when it sees the flag, up to a ring of calls later; until then
  // the next call would start from them.  Pivot 0 lists nothing: that call falls short, is redone here, and stays exact.
  if (hv_pivot != nullptr && blockIdx.x == 0)
    for (int i = threadIdx.x; i < p.num_seqs; i += 256) hv_pivot[i] = 0u;
  __shared__ __attribute__((aligned(16))) uint8_t prep_s[FB_MAX_COUPLED * 24];
  __shared__ uint32_t word_s;
  const bool coupled = p.mode == 0 && p.num_seqs > 1;
  const unsigned bid = blockIdx.x, nb = gridDim.x;
  if (flag0 & FB_TIMEOUT_BIT) { fb_void_outputs(p, ws, bid, nb); return; }
  FbSync fs;
  fs.claim = ws.bar + 32;
  fs.done = ws.bar + 32 + FB_PHASES;
  fs.flag = ws.fallback;
  fs.mirror = p.flag_mirror;
  fs.ticket = p.flag_ticket;
  uint32_t phase = 0;
  bool alive = true;
  // (workgroup 0 leaves the 100 MHz wall clock of every phase end behind the counter: tools/fallback_cost.py)
  auto run = [&](uint32_t V, auto&& body) {
    if (!alive) return;
    alive = fb_phase(fs, phase, V, &word_s, body);
    ++phase;
    if (bid == 0 && threadIdx.x == 0 && phase < 15) ws.bar[1 + phase] = (uint32_t)wall_clock64();
  };
  if (bid == 0 && threadIdx.x == 0) ws.bar[1] = (uint32_t)wall_clock64();
  const int B = p.num_seqs, G = B * p.num_layers * p.num_kv_heads;
  const uint32_t VS = (uint32_t)B < 4u * vgrid ? (uint32_t)B : 4u * vgrid;     // per-sequence phases
  const uint32_t VH = (uint32_t)G < 8u * vgrid ? (uint32_t)G : 8u * vgrid;     // the per-head phase
  if (have_keys && !(flag0 & FB_HOLES_BIT)) {
    run(vgrid, [&](unsigned v, unsigned V) { zero_body(zero16, zero_vecs, v, V); });
  } else {
    // every logical block of the batch has a physical block (the collecting pass counted them:
    // the same for all workgroups) -> nothing to clear, nothing to fix: two phases less
    uint32_t claimed = 0;
    for (int q = 0; q < CLAIM_SHARDS; ++q) claimed += ws.st_claimed[q * 32];
    const bool holes = (int64_t)claimed != true_n(p, ws) / p.block_size && !(p.lean & 2);
    auto keys = [&](unsigned v, unsigned V) {
      if (sparse) build_keys_sparse_body(p, ws, v, V);
      else build_keys_body<4>(p, ws, v, V);
    };
    if (holes) {
      run(vgrid, [&](unsigned v, unsigned V) { zero_body(zero16, zero_vecs, v, V); clear_chunk_table_body(p, ws, v, V); });
      run(vgrid, keys);
      run(vgrid, [&](unsigned v, unsigned V) { fix_unclaimed_body(p, ws, v, V); });
    } else {
      run(vgrid, [&](unsigned v, unsigned V) { zero_body(zero16, zero_vecs, v, V); keys(v, V); });
    }
  }
  for (int round = 0; round < 4; ++round) {
    run(vgrid, [&](unsigned v, unsigned V) { hist_round_body(p, ws, round, v, V); });
    if (round == 0 && coupled) {                       // the reference's batch > 1 rule: totals, k', then the pick
      run(VS, [&](unsigned v, unsigned V) {
        for (int i = (int)v; i < B; i += (int)V) { scan_pick_body<4, 4>(p, ws, 0, i, 1); __syncthreads(); }
      });
      run(1u, [&](unsigned, unsigned) { seq_prepare_tables(p, ws, prep_s); });
      run(VS, [&](unsigned v, unsigned V) {
        for (int i = (int)v; i < B; i += (int)V) { scan_pick_body<4, 4>(p, ws, 0, i, 2); __syncthreads(); }
      });
    } else {
      run(VS, [&](unsigned v, unsigned V) {
        for (int i = (int)v; i < B; i += (int)V) { scan_pick_body<4, 4>(p, ws, round, i); __syncthreads(); }
      });
    }
  }
  if (!alive) { fb_void_outputs(p, ws, bid, nb); return; }
  // the last phase: nobody waits for it (the kernel's end does)
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) word_s = __hip_atomic_fetch_add(fs.claim + phase, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t v = word_s;
    if (v >= VH) break;
    for (int g = (int)v; g < G; g += (int)VH) {
      select_emit_head<256>(p, ws, 0, g, nullptr);
      __syncthreads();
    }
  }
  if (bid == 0 && threadIdx.x == 0) ws.bar[17] = (uint32_t)wall_clock64();     // (workgroup 0's own end)
}


}  // namespace kvc
