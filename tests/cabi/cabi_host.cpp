// A host with no Python and no torch driving libkvc_mi355x.so through its C ABI
// (include/kvc_mi355x.h) and checking the result against the C oracle (oracle/kvc_oracle.c).
// Built and run by tests/test_gpu_cabi.py:
//   gcc -c oracle/kvc_oracle.c; hipcc --offload-arch=gfx950 tests/cabi/cabi_host.cpp kvc_oracle.o -Iinclude \
//         -Lvllm_kvcompress_amd -lkvc_mi355x -o <tmp>/cabi_host
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "kvc_mi355x.h"

extern "C" {
void orc_count_block_evictions(int32_t*, int32_t*, const int32_t*, const int32_t*, int32_t, int64_t, int32_t, int32_t);
void orc_schedule_t1_cache_moves(int32_t*, int64_t, int32_t*, const int32_t*, const int32_t*, const int32_t*,
                                 const int32_t*, const int32_t*, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t);
void orc_execute_cache_moves(uint8_t*, uint8_t*, float*, int32_t*, const int32_t*, const int32_t*, const int32_t*,
                             int32_t, int32_t, int32_t, int32_t, int32_t);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define KV(x) do { int rc_ = (x); if (rc_ != 0) { printf("kvc error %d: %s\n", rc_, kvc_last_error()); return 3; } } while (0)

template <typename T> T* to_dev(const std::vector<T>& v) {
  T* d = nullptr;
  if (hipMalloc(&d, v.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
  hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}
template <typename T> bool same(const T* dev, const std::vector<T>& want, const char* name) {
  std::vector<T> got(want.size());
  hipMemcpy(got.data(), dev, want.size() * sizeof(T), hipMemcpyDeviceToHost);
  if (memcmp(got.data(), want.data(), want.size() * sizeof(T)) != 0) { printf("MISMATCH %s\n", name); return false; }
  return true;
}

int main() {
  // one sequence, L=2, H=2, bs=16, hd=128, fp16; every head: ctx = 160 (10 blocks), 4 blocks freed
  const int B = 1, L = 2, H = 2, bs = 16, hd = 128, e = 2, x = 8, G = B * L * H, M = 10;
  const int ctx = 160, nblk = 10, NB = G * nblk + 3, seg = nblk * bs, N = G * seg;
  const int NUL = 2147483000;
  uint32_t rs = 12345u;
  auto rnd = [&]() { rs = rs * 1664525u + 1013904223u; return rs >> 8; };
  std::vector<int32_t> offs(G), hang(G, bs), ctxs(L * B * H, ctx), bt(L * B * H * M), perm(NB);
  for (int i = 0; i < NB; ++i) perm[i] = i;
  for (int i = NB - 1; i > 0; --i) std::swap(perm[i], perm[rnd() % (i + 1)]);
  for (int g = 0; g < G; ++g) offs[g] = g * seg;
  for (int i = 0; i < L * B * H * M; ++i) bt[i] = perm[i];
  // evicted logical indices per head: 64 distinct random slots, ascending, rest null
  std::vector<int32_t> eli(N, NUL), ekc(G, 4 * bs);
  for (int g = 0; g < G; ++g) {
    std::vector<char> pick(ctx, 0);
    int c = 0;
    while (c < 4 * bs) { int s = rnd() % ctx; if (!pick[s]) { pick[s] = 1; ++c; } }
    int o = 0;
    for (int s = 0; s < ctx; ++s) if (pick[s]) eli[g * seg + o++] = s;
  }
  std::vector<uint8_t> k((size_t)NB * hd * bs * e), v((size_t)NB * hd * bs * e);
  for (auto& b : k) b = (uint8_t)rnd();
  for (auto& b : v) b = (uint8_t)rnd();
  std::vector<float> met((size_t)NB * bs);
  std::vector<int32_t> pos((size_t)NB * bs);
  for (size_t i = 0; i < met.size(); ++i) { met[i] = (float)(rnd() % 100000); pos[i] = (int32_t)(rnd() % 100000); }

  // ---- oracle
  std::vector<int32_t> w_eli = eli, w_ebc(G), w_moves((size_t)N * 2, 77), w_cnt(G);
  orc_count_block_evictions(w_ebc.data(), w_eli.data(), offs.data(), hang.data(), G, N, bs, NUL);
  orc_schedule_t1_cache_moves(w_moves.data(), N, w_cnt.data(), w_eli.data(), ekc.data(), offs.data(), bt.data(),
                              ctxs.data(), B, L, H, M, bs, 1);
  std::vector<uint8_t> wk = k, wv = v;
  std::vector<float> wm = met;
  std::vector<int32_t> wp = pos;
  orc_execute_cache_moves(wk.data(), wv.data(), wm.data(), wp.data(), w_moves.data(), w_cnt.data(), offs.data(), G,
                          bs, hd, e, x);

  // ---- device through the C ABI
  printf("kvc abi %d\n", kvc_abi_version());
  if (kvc_abi_version() != KVC_ABI_VERSION) { printf("library ABI %d, header %d\n", kvc_abi_version(), KVC_ABI_VERSION); return 4; }
  hipStream_t s;
  CK(hipStreamCreate(&s));
  int32_t *d_eli = to_dev(eli), *d_offs = to_dev(offs), *d_hang = to_dev(hang), *d_ekc = to_dev(ekc),
          *d_bt = to_dev(bt), *d_ctx = to_dev(ctxs), *d_pos = to_dev(pos);
  uint8_t *d_k = to_dev(k), *d_v = to_dev(v);
  float* d_met = to_dev(met);
  int32_t *d_ebc, *d_moves, *d_cnt;
  CK(hipMalloc(&d_ebc, G * 4)); CK(hipMalloc(&d_moves, (size_t)N * 8)); CK(hipMalloc(&d_cnt, G * 4));
  CK(hipMemset(d_moves, 0x7f, (size_t)N * 8));
  KV(kvc_count_block_evictions(d_ebc, d_eli, d_offs, d_hang, G, N, bs, NUL, s));
  KV(kvc_schedule_t1_cache_moves(d_moves, N, d_cnt, d_eli, d_ekc, d_offs, d_bt, d_ctx, B, L, H, M, bs, 1, s));
  const size_t wsb = kvc_execute_cache_moves_workspace_bytes(G, NB);
  void* ws;
  CK(hipMalloc(&ws, wsb));
  KV(kvc_execute_cache_moves(d_k, d_v, d_met, d_pos, d_moves, d_cnt, d_offs, G, NB, bs, hd, e, x, ws, wsb, s));
  CK(hipStreamSynchronize(s));
  bool ok = same(d_ebc, w_ebc, "evicted_block_count") & same(d_eli, w_eli, "evicted_logical_indices") &
            same(d_moves, w_moves, "cache_moves_idx") & same(d_cnt, w_cnt, "cache_moves_count") &
            same(d_k, wk, "k_cache") & same(d_v, wv, "v_cache") & same(d_met, wm, "kv_metrics") &
            same(d_pos, wp, "kv_position");
  int rc = 0;
  // ---- ABI version 4: the move table with its dirty map (full fill, then only what the map marks) and the
  // compaction on the plan the move scheduler leaves behind (one launch, no workspace)
  {
    const size_t mapb = kvc_cache_moves_dirty_map_bytes(N, bs), planb = kvc_cache_moves_plan_bytes();
    uint32_t* d_map; int32_t* d_plan;
    CK(hipMalloc(&d_map, mapb)); CK(hipMalloc(&d_plan, planb));
    CK(hipMemset(d_map, 0, mapb));
    CK(hipMemset(d_moves, 0x55, (size_t)N * 8));                    // junk: zero_fill 1 clears all of it
    KV(kvc_schedule_t1_cache_moves_ex(d_moves, N, d_cnt, d_eli, d_ekc, d_offs, d_bt, d_ctx, B, L, H, M, bs, 1, d_map, mapb,
                                      d_plan, s));
    CK(hipStreamSynchronize(s));
    ok = ok & same(d_moves, w_moves, "cache_moves_idx (zero_fill 1 + map)");
    KV(kvc_schedule_t1_cache_moves_ex(d_moves, N, d_cnt, d_eli, d_ekc, d_offs, d_bt, d_ctx, B, L, H, M, bs, 2, d_map, mapb,
                                      d_plan, s));
    CK(hipMemcpyAsync(d_k, k.data(), k.size(), hipMemcpyHostToDevice, s));
    CK(hipMemcpyAsync(d_v, v.data(), v.size(), hipMemcpyHostToDevice, s));
    CK(hipMemcpyAsync(d_met, met.data(), met.size() * 4, hipMemcpyHostToDevice, s));
    CK(hipMemcpyAsync(d_pos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice, s));
    KV(kvc_execute_cache_moves_planned(d_k, d_v, d_met, d_pos, d_moves, d_cnt, d_offs, G, NB, bs, hd, e, x, d_plan, s));
    CK(hipStreamSynchronize(s));
    ok = ok & same(d_moves, w_moves, "cache_moves_idx (zero_fill 2)") & same(d_cnt, w_cnt, "cache_moves_count (ex)") &
         same(d_k, wk, "k_cache (planned)") & same(d_v, wv, "v_cache (planned)") &
         same(d_met, wm, "kv_metrics (planned)") & same(d_pos, wp, "kv_position (planned)");
    rc = kvc_schedule_t1_cache_moves_ex(d_moves, N, d_cnt, d_eli, d_ekc, d_offs, d_bt, d_ctx, B, L, H, M, bs, 2, nullptr, 0,
                                        nullptr, s);
    ok = ok && rc == 1 && strstr(kvc_last_error(), "dirty map") != nullptr;
  }
  // ---- ABI version 5: A2a + A3 with the harvest protocol.  Two decode steps of the same batch: the second one's
  // schedule once the usual way (aggregate_decode, then schedule_evictions' own pass on remembered pivots) and once on
  // the lists kvc_aggregate_decode_harvest made -- the same store and the same schedule, no flag raised; the sums
  // against a plain float loop (metrics.py:429-439: metrics += sum_q temp^2, in index order)
  {
    const int qpk = 4, kfree = 4, seq_pos = 400, prot = 32;
    const int64_t slots = (int64_t)NB * bs;
    std::vector<int32_t> seq_by(NB, -1), lay_by(NB, 0), head_by(NB, 0), lbn_by(NB, 0), pos2((size_t)slots, 0);
    for (int l = 0; l < L; ++l)
      for (int h = 0; h < H; ++h)
        for (int j = 0; j < nblk; ++j) {
          const int blk = bt[((l * B + 0) * H + h) * M + j];
          seq_by[blk] = 0; lay_by[blk] = l; head_by[blk] = h; lbn_by[blk] = j;
          for (int o = 0; o < bs; ++o) pos2[(size_t)blk * bs + o] = j * bs + o;
        }
    std::vector<float> m0((size_t)slots), temp((size_t)slots * qpk);
    for (size_t i = 0; i < m0.size(); ++i) m0[i] = (float)perm[i / bs] * 64.0f + (float)(rnd() % 64) + (float)(i % bs) * 0.001f;
    for (auto& t : temp) t = (float)(rnd() % 1000) * 1e-3f;
    auto aggregate = [&](std::vector<float>& m) {
      for (int64_t i = 0; i < slots; ++i) {
        float acc = 0.0f;
        for (int q = 0; q < qpk; ++q) { volatile float sq = temp[i * qpk + q] * temp[i * qpk + q]; acc = acc + sq; }
        m[i] = m[i] + acc;
      }
    };
    std::vector<float> m1 = m0, m2;
    aggregate(m1);
    m2 = m1;
    aggregate(m2);
    std::vector<int32_t> slot_of_seq = {0}, seqpos = {seq_pos}, protv = {prot}, kper = {kfree};
    std::vector<int32_t> hang2(G), ctx2(L * B * H, ctx);
    for (int g = 0; g < G; ++g) hang2[g] = ctx % bs == 0 ? bs : ctx % bs;
    float *d_m = to_dev(m0), *d_mr = to_dev(m1), *d_temp = to_dev(temp);
    int32_t *d_seq = to_dev(seq_by), *d_lay = to_dev(lay_by), *d_head = to_dev(head_by), *d_lbn = to_dev(lbn_by),
            *d_pos2 = to_dev(pos2), *d_sos = to_dev(slot_of_seq), *d_sp = to_dev(seqpos), *d_pr = to_dev(protv),
            *d_kper = to_dev(kper), *d_hang2 = to_dev(hang2), *d_ctx2 = to_dev(ctx2);
    const size_t wsb2 = kvc_schedule_evictions_workspace_bytes(N, G, B, bs), hvb = kvc_harvest_buffer_bytes(G, B);
    void *ws2, *hbuf;
    CK(hipMalloc(&ws2, wsb2)); CK(hipMalloc(&hbuf, hvb)); CK(hipMemset(hbuf, 0, hvb));
    int32_t* out[3][3];
    for (auto& o : out) { CK(hipMalloc(&o[0], (size_t)N * 4)); CK(hipMalloc(&o[1], G * 4)); CK(hipMalloc(&o[2], G * 4)); }
    kvc_schedule_params sp;
    memset(&sp, 0, sizeof(sp));
    sp.token_positions = d_pos2; sp.seq_index_by_block = d_seq; sp.layer_index_by_block = d_lay;
    sp.head_index_by_block = d_head; sp.logical_block_num_by_block = d_lbn; sp.num_blocks = NB;
    sp.block_size = bs; sp.num_layers = L; sp.num_kv_heads = H; sp.num_seqs = B;
    sp.seq_slot_of_seq = d_sos; sp.seq_slot_len = 1; sp.seq_positions = d_sp; sp.num_protected = d_pr;
    sp.evicted_blocks_per_seq = d_kper; sp.context_lens = d_ctx2; sp.hanging_token_count = d_hang2;
    sp.evicted_kv_offsets = d_offs; sp.total_slots = N; sp.mode = 1; sp.null_value = NUL;
    sp.max_evicted_blocks_hint = kfree; sp.harvest_buf = hbuf; sp.harvest_widen = 0.25f;
    auto schedule = [&](float* metrics, int harvest, int32_t** o) {
      sp.metrics = metrics; sp.harvest = harvest;
      sp.evicted_logical_indices = o[0]; sp.evicted_kv_count = o[1]; sp.evicted_block_count = o[2];
      return kvc_schedule_evictions(&sp, ws2, wsb2, s);
    };
    auto flag = [&]() {
      uint32_t w = 99;
      hipStreamSynchronize(s);
      hipMemcpy(&w, (uint8_t*)ws2 + kvc_schedule_evictions_fallback_offset(N, G, B, bs), 4, hipMemcpyDeviceToHost);
      return w;
    };
    bool hok = kvc_harvest_eligible(&sp, qpk) == 1 && kvc_pivot_memory_eligible(&sp) == 1 &&
               kvc_schedule_evictions_plan(&sp) == 1;
    // ABI version 6: N and the eviction counts for a host that holds them as device tensors only (the fork's call form,
    // vllm/kvcompress/scheduler.py:245-247) -- into page-locked memory the kernel writes itself, and into plain host
    // memory through the workspace
    {
      int64_t *pinned = nullptr, plain[2] = {-1, -1};
      CK(hipHostMalloc(&pinned, 16, hipHostMallocDefault));
      pinned[0] = pinned[1] = -1;
      KV(kvc_schedule_batch_summary(d_ctx2, L * B * H, bs, d_kper, B, pinned, 1, 0, nullptr, 0, s));
      KV(kvc_schedule_batch_summary_wait(s));
      KV(kvc_schedule_batch_summary(d_ctx2, L * B * H, bs, d_kper, B, plain, 0, 1, ws2, wsb2, s));
      hok = hok && pinned[0] == N && pinned[1] == kfree && plain[0] == N && plain[1] == kfree;
      if (!hok) printf("batch summary: %ld %ld / %ld %ld, want %d %d\n", (long)pinned[0], (long)pinned[1], (long)plain[0],
                       (long)plain[1], N, kfree);
      CK(hipHostFree(pinned));
    }
    // step 1: the usual way; the call leaves pivots (bit 1)
    KV(kvc_aggregate_decode(d_m, d_temp, slots, qpk, 1, 0, s));
    KV(schedule(d_m, 2, out[0]));
    hok = hok && flag() == 0 && same(d_m, m1, "metrics after step 1");
    // step 2, reference run: aggregate_decode on a copy of the store, the call's own pass on the remembered pivots (bits 1 | 2)
    KV(kvc_aggregate_decode(d_mr, d_temp, slots, qpk, 1, 0, s));
    void* hbuf2;
    CK(hipMalloc(&hbuf2, hvb));
    CK(hipMemcpyAsync(hbuf2, hbuf, hvb, hipMemcpyDeviceToDevice, s));
    sp.harvest_buf = hbuf2;
    KV(schedule(d_mr, 2 | 4, out[1]));
    hok = hok && flag() == 0 && same(d_mr, m2, "metrics after step 2 (aggregate_decode)");
    // step 2, harvested: the aggregation pass makes the lists, the schedule call runs on them (bits 0 | 1)
    sp.harvest_buf = hbuf; sp.metrics = d_m;
    KV(kvc_aggregate_decode_harvest(&sp, d_temp, qpk, 1, 0, s));
    KV(schedule(d_m, 1 | 2, out[2]));
    hok = hok && flag() == 0 && same(d_m, m2, "metrics after step 2 (aggregate_decode_harvest)");
    std::vector<int32_t> r_eli(N), r_cnt(G), r_blk(G);
    CK(hipMemcpy(r_eli.data(), out[1][0], (size_t)N * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r_cnt.data(), out[1][1], G * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r_blk.data(), out[1][2], G * 4, hipMemcpyDeviceToHost));
    long freed = 0;
    for (int g = 0; g < G; ++g) freed += r_blk[g];
    hok = hok && freed == kfree && same(out[2][0], r_eli, "evicted_logical_indices (harvested vs own pass)") &&
          same(out[2][1], r_cnt, "evicted_kv_count (harvested vs own pass)") &&
          same(out[2][2], r_blk, "evicted_block_count (harvested vs own pass)");
    // ---- ABI version 8: the same call without a wait in front of the launches -- N on the device, a bound on the host
    // (digit rounds / bracket schedule: max_evicted_blocks_hint = -1), the flag word stored by the call's last launch
    // itself; and a bound that does not hold: voided on the device, nothing written
    {
      int64_t* pinned = nullptr;
      uint32_t* mirror = nullptr;
      int64_t* n_dev = nullptr;
      CK(hipHostMalloc(&pinned, 8 * (2 + B), hipHostMallocDefault));
      CK(hipHostMalloc(&mirror, 4, hipHostMallocDefault));
      CK(hipMalloc(&n_dev, 16));
      const int64_t bound = N + 64 * bs;
      const size_t wsb8 = kvc_schedule_evictions_workspace_bytes(bound, G, B, bs);
      void* ws8;
      CK(hipMalloc(&ws8, wsb8));
      int32_t *o_ref[3], *o_def[3];
      for (auto& o : {o_ref, o_def}) { CK(hipMalloc(&o[0], (size_t)bound * 4)); CK(hipMalloc(&o[1], G * 4)); CK(hipMalloc(&o[2], G * 4)); }
      kvc_schedule_params p8 = sp;
      p8.metrics = d_m; p8.harvest = 0; p8.harvest_buf = nullptr; p8.max_evicted_blocks_hint = -1; p8.schedule_path = 4;
      // the waiting way (reference for this leg): N known, the bracket schedule forced
      p8.total_slots = N; p8.total_slots_dev = nullptr;
      p8.evicted_logical_indices = o_ref[0]; p8.evicted_kv_count = o_ref[1]; p8.evicted_block_count = o_ref[2];
      KV(kvc_schedule_evictions(&p8, ws8, wsb8, s));
      std::vector<int32_t> w_eli(N), w_cnt(G), w_blk(G);
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(w_eli.data(), o_ref[0], (size_t)N * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(w_cnt.data(), o_ref[1], G * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(w_blk.data(), o_ref[2], G * 4, hipMemcpyDeviceToHost));
      // the deferred way: summary + schedule back to back, then the host reads
      *mirror = 0; pinned[1 + B] = 0;
      KV(kvc_schedule_batch_summary_deferred(d_ctx2, L * B * H, bs, d_kper, B, pinned, 7, n_dev, bound, s));
      p8.total_slots = bound; p8.total_slots_dev = n_dev; p8.flag_mirror = mirror; p8.flag_ticket = 41;
      p8.evicted_logical_indices = o_def[0]; p8.evicted_kv_count = o_def[1]; p8.evicted_block_count = o_def[2];
      KV(kvc_schedule_evictions(&p8, ws8, wsb8, s));
      while (__atomic_load_n(&pinned[1 + B], __ATOMIC_ACQUIRE) != 7) { }
      bool ok8 = pinned[0] == N && pinned[1] == kfree;
      CK(hipStreamSynchronize(s));
      ok8 = ok8 && (*mirror >> 8) == 41u && (*mirror & 2u) == 0u;
      std::vector<int32_t> g_eli(N);
      CK(hipMemcpy(g_eli.data(), o_def[0], (size_t)N * 4, hipMemcpyDeviceToHost));
      ok8 = ok8 && g_eli == w_eli && same(o_def[1], w_cnt, "evicted_kv_count (N on the device)") &&
            same(o_def[2], w_blk, "evicted_block_count (N on the device)");
      // a bound below N: the summary sets the void word, every kernel returns, outputs and mirror stay as they were
      CK(hipMemsetAsync(o_def[1], 0x5A, G * 4, s));
      *mirror = 0;
      KV(kvc_schedule_batch_summary_deferred(d_ctx2, L * B * H, bs, d_kper, B, pinned, 8, n_dev, (int64_t)N - bs, s));
      p8.total_slots = N - bs; p8.flag_ticket = 42;
      KV(kvc_schedule_evictions(&p8, ws8, wsb8, s));
      CK(hipStreamSynchronize(s));
      std::vector<int32_t> v_cnt(G);
      CK(hipMemcpy(v_cnt.data(), o_def[1], G * 4, hipMemcpyDeviceToHost));
      bool untouched = true;
      for (int g = 0; g < G; ++g) untouched = untouched && v_cnt[g] == 0x5A5A5A5A;
      ok8 = ok8 && pinned[1 + B] == 8 && pinned[0] == N && untouched && *mirror == 0u;
      if (!ok8) printf("ABI 8 leg failed: N %ld mirror %08x untouched %d\n", (long)pinned[0], *mirror, (int)untouched);
      printf("schedule with N on the device (ABI 8): %s\n", ok8 ? "ok" : "MISMATCH");
      hok = hok && ok8;
      CK(hipHostFree(pinned)); CK(hipHostFree(mirror)); CK(hipFree(n_dev)); CK(hipFree(ws8));
    }
    // ---- ABI version 6: the decode step without a sweep of the store.  Step 3 twice from the same store: (a) the fused-metric
    // attention of both layers, then the schedule's own pass on remembered pivots; (b) the same attention with the harvest
    // fields set (kvc_attention_harvest_begin in front), then the schedule on the lists the epilogues made (bits 0 | 3).
    // The same store and the same schedule, no flag raised; and lists of ANOTHER batch (other positions) are redone, not trusted.
    {
      const int Hq = H * qpk;
      std::vector<_Float16> kc((size_t)NB * hd * bs), vc((size_t)NB * hd * bs), qv((size_t)B * Hq * hd), outv((size_t)B * Hq * hd);
      for (auto& x : kc) x = (_Float16)(((int)(rnd() % 201) - 100) * 0.004f);
      for (auto& x : vc) x = (_Float16)(((int)(rnd() % 201) - 100) * 0.004f);
      for (auto& x : qv) x = (_Float16)(((int)(rnd() % 201) - 100) * 0.01f);
      _Float16 *d_kc = to_dev(kc), *d_vc = to_dev(vc), *d_q = to_dev(qv), *d_out = to_dev(outv);
      std::vector<int32_t> lastp = {seq_pos - 1}, bufl = {0}, slot_of_att = {0};
      int32_t *d_last = to_dev(lastp), *d_bufl = to_dev(bufl), *d_slot = to_dev(slot_of_att);
      float* d_ma; float* d_mb;
      CK(hipMalloc(&d_ma, slots * 4)); CK(hipMalloc(&d_mb, slots * 4));
      CK(hipMemcpyAsync(d_ma, d_m, slots * 4, hipMemcpyDeviceToDevice, s));
      CK(hipMemcpyAsync(d_mb, d_m, slots * 4, hipMemcpyDeviceToDevice, s));
      void* hbuf3;
      CK(hipMalloc(&hbuf3, hvb));
      CK(hipMemcpyAsync(hbuf3, hbuf, hvb, hipMemcpyDeviceToDevice, s));       // (the pivots step 2 left behind)
      auto attention = [&](float* metrics, void* harvest) {
        for (int l = 0; l < L; ++l) {
          kvc_attention_params ap;
          memset(&ap, 0, sizeof(ap));
          ap.out = d_out; ap.fused_metrics = metrics; ap.fused_use_l2 = 1; ap.query = d_q; ap.key_cache = d_kc; ap.value_cache = d_vc;
          ap.block_tables = d_bt + (size_t)l * B * H * M; ap.context_lens = d_ctx2 + (size_t)l * B * H; ap.kv_position = d_pos2;
          ap.last_position = d_last; ap.kv_metric_buffer_len = d_bufl; ap.q_stride = (int64_t)Hq * hd;
          ap.kv_block_stride = (int64_t)hd * bs; ap.scale = 0.088f; ap.k_scale = ap.v_scale = 1.0f; ap.num_seqs = B;
          ap.num_heads = Hq; ap.num_kv_heads = H; ap.head_size = hd; ap.block_size = bs; ap.max_num_blocks_per_seq = M;
          ap.max_context_len = ctx; ap.dtype = 0; ap.kv_cache_dtype = 0; ap.record_kv_metrics = 1;
          if (harvest != nullptr) {
            ap.harvest_buf = harvest; ap.harvest_seq_slot = d_slot; ap.harvest_seq_positions = d_sp; ap.harvest_num_protected = d_pr;
            ap.harvest_num_seqs = B; ap.harvest_layer = l; ap.harvest_num_layers = L; ap.harvest_num_sinks = 0;
          }
          int rc_ = kvc_paged_attention_decode(&ap, s);
          if (rc_ != 0) return rc_;
        }
        return 0;
      };
      bool aok = kvc_attention_harvest_eligible(&sp) == 1;
      // (a) the fused attention alone, the schedule's own pass
      KV(attention(d_ma, nullptr));
      sp.harvest_buf = hbuf3;
      KV(schedule(d_ma, 2 | 4, out[1]));
      aok = aok && flag() == 0;
      // (b) ... with the harvest
      sp.harvest_buf = hbuf; sp.metrics = d_mb;
      KV(kvc_attention_harvest_begin(&sp, s));
      KV(attention(d_mb, hbuf));
      KV(schedule(d_mb, 1 | 2 | 8, out[2]));
      aok = aok && flag() == 0;
      std::vector<float> ma((size_t)slots);
      std::vector<int32_t> a_eli(N), a_cnt(G), a_blk(G);
      CK(hipMemcpy(ma.data(), d_ma, slots * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(a_eli.data(), out[1][0], (size_t)N * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(a_cnt.data(), out[1][1], G * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(a_blk.data(), out[1][2], G * 4, hipMemcpyDeviceToHost));
      long afreed = 0;
      for (int g = 0; g < G; ++g) afreed += a_blk[g];
      aok = aok && afreed == kfree && same(d_mb, ma, "metrics (attention with harvest vs without)") &&
            memcmp(ma.data(), m2.data(), slots * 4) != 0 &&                                      // (the attention did add something)
            same(out[2][0], a_eli, "evicted_logical_indices (epilogue's lists vs own pass)") &&
            same(out[2][1], a_cnt, "evicted_kv_count (epilogue's lists vs own pass)") &&
            same(out[2][2], a_blk, "evicted_block_count (epilogue's lists vs own pass)");
      // (c) the aggregation pass harvesting for a call it PREDICTS (the fork's flow: aggregate_decode() at the end of an
      // iteration, the schedule call at the start of the next): the last call's positions + 1, context lengths not known
      // (NULL) -- against the plain pass + the schedule's own pass on the same pivots
      {
        std::vector<int32_t> seqpos_prev = {seq_pos - 1};
        int32_t* d_sp_prev = to_dev(seqpos_prev);
        CK(hipMemcpyAsync(hbuf3, hbuf, hvb, hipMemcpyDeviceToDevice, s));     // (the pivots (b)'s call left behind)
        KV(kvc_aggregate_decode(d_ma, d_temp, slots, qpk, 1, 0, s));
        sp.harvest_buf = hbuf3;
        KV(schedule(d_ma, 2 | 4, out[1]));
        aok = aok && flag() == 0;
        sp.harvest_buf = hbuf; sp.metrics = d_mb; sp.seq_positions = d_sp_prev; sp.context_lens = nullptr;
        sp.harvest_position_delta = 1;
        KV(kvc_aggregate_decode_harvest(&sp, d_temp, qpk, 1, 0, s));
        sp.seq_positions = d_sp; sp.context_lens = d_ctx2; sp.harvest_position_delta = 0;
        KV(schedule(d_mb, 1 | 2 | 8, out[2]));
        aok = aok && flag() == 0;
        CK(hipMemcpy(ma.data(), d_ma, slots * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(a_eli.data(), out[1][0], (size_t)N * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(a_cnt.data(), out[1][1], G * 4, hipMemcpyDeviceToHost));
        aok = aok && same(d_mb, ma, "metrics (predicting aggregation pass vs plain)") &&
              same(out[2][0], a_eli, "evicted_logical_indices (predicted lists vs own pass)") &&
              same(out[2][1], a_cnt, "evicted_kv_count (predicted lists vs own pass)");
        if (!aok) printf("predicted harvest failed\n");
      }
      // lists made for other positions: the call sees it on the device (bit 3), raises its flag and redoes the work
      std::vector<int32_t> seqpos_other = {seq_pos + 1};
      int32_t* d_sp_other = to_dev(seqpos_other);
      CK(hipMemcpyAsync(d_mb, d_ma, slots * 4, hipMemcpyDeviceToDevice, s));
      sp.seq_positions = d_sp_other;
      KV(schedule(d_mb, 1 | 8, out[0]));
      aok = aok && (flag() & 1u) == 1u;
      sp.seq_positions = d_sp;
      if (!aok) printf("attention harvest protocol failed\n");
      hok = hok && aok;
    }
    // lists without the buffer they would be in: an error
    sp.harvest_buf = nullptr;
    rc = schedule(d_m, 1, out[2]);
    hok = hok && rc == 1 && strstr(kvc_last_error(), "without a harvest buffer") != nullptr;
    if (!hok) printf("harvest protocol failed\n");
    ok = ok && hok;
  }
  // error convention: unsupported block size -> rc 1 + message
  rc = kvc_count_block_evictions(d_ebc, d_eli, d_offs, d_hang, G, N, 0, NUL, s);
  ok = ok && rc == 1 && strstr(kvc_last_error(), "Unsupported block size") != nullptr;
  // ---- F3 through the struct ABI: one sequence, 2 KV heads x 4 query heads, 200 / 37 keys
  {
    const int S = 1, Hkv = 2, Hq = 8, qpk = 4, ahd = 128, abs_ = 16, ANB = 20, AM = 14;
    const int actx[2] = {200, 37};
    std::vector<_Float16> q((size_t)S * Hq * ahd), akc((size_t)ANB * ahd * abs_), avc((size_t)ANB * ahd * abs_);
    auto frand = [&]() { return (float)((int)(rnd() % 2001) - 1000) * 1e-3f; };
    for (auto& t : q) t = (_Float16)frand();
    for (auto& t : akc) t = (_Float16)frand();
    for (auto& t : avc) t = (_Float16)frand();
    std::vector<int32_t> abt((size_t)S * Hkv * AM, 0), actxv = {actx[0], actx[1]}, apos((size_t)ANB * abs_, 0);
    std::vector<int32_t> alast = {1000}, abuf = {0};
    for (int h = 0, nb = 0; h < Hkv; ++h)
      for (int b = 0; b < (actx[h] + abs_ - 1) / abs_; ++b) abt[h * AM + b] = (nb++ * 7) % ANB;   // distinct: 7 coprime to 20
    const float scale = 0.088388f;
    // plain float reference: logits, softmax with the reference's 1/(sum + 1e-6), fp16 weights for P.V
    std::vector<float> w_out((size_t)Hq * ahd, 0.f), w_met((size_t)ANB * abs_ * qpk, -1.f);
    for (int qh = 0; qh < Hq; ++qh) {
      const int h = qh / qpk, n = actx[h];
      std::vector<float> lg(n);
      float mx = -1e30f;
      for (int i = 0; i < n; ++i) {
        const int blk = abt[h * AM + i / abs_], off = i % abs_;
        float acc = 0.f;
        for (int d = 0; d < ahd; ++d)
          acc += (float)q[(size_t)qh * ahd + d] * (float)akc[((size_t)blk * (ahd / 8) + d / 8) * abs_ * 8 + off * 8 + d % 8];
        lg[i] = acc * scale;
        mx = lg[i] > mx ? lg[i] : mx;
      }
      float sum = 0.f;
      for (int i = 0; i < n; ++i) { lg[i] = expf(lg[i] - mx); sum += lg[i]; }
      for (int i = 0; i < n; ++i) {
        const int blk = abt[h * AM + i / abs_], off = i % abs_;
        const float pr = lg[i] / (sum + 1e-6f);
        w_met[((size_t)blk * abs_ + off) * qpk + qh % qpk] = pr;
        const float ph = (float)(_Float16)pr;
        for (int d = 0; d < ahd; ++d) w_out[(size_t)qh * ahd + d] += ph * (float)avc[((size_t)blk * ahd + d) * abs_ + off];
      }
    }
    _Float16 *dq = to_dev(q), *dkc = to_dev(akc), *dvc = to_dev(avc), *dout = nullptr;
    float* dmet = to_dev(std::vector<float>(w_met.size(), -1.f));
    int32_t *dbt = to_dev(abt), *dctx = to_dev(actxv), *dpos = to_dev(apos), *dlast = to_dev(alast), *dbuf = to_dev(abuf);
    CK(hipMalloc(&dout, q.size() * sizeof(_Float16)));
    kvc_attention_params ap;
    memset(&ap, 0, sizeof(ap));
    ap.out = dout; ap.kv_metric_out = dmet; ap.query = dq; ap.key_cache = dkc; ap.value_cache = dvc;
    ap.block_tables = dbt; ap.context_lens = dctx; ap.kv_position = dpos; ap.last_position = dlast;
    ap.kv_metric_buffer_len = dbuf; ap.q_stride = (int64_t)Hq * ahd; ap.kv_block_stride = (int64_t)ahd * abs_;
    ap.scale = scale; ap.k_scale = 1.f; ap.v_scale = 1.f; ap.num_seqs = S; ap.num_heads = Hq; ap.num_kv_heads = Hkv;
    ap.head_size = ahd; ap.block_size = abs_; ap.max_num_blocks_per_seq = AM; ap.max_context_len = 200;
    ap.dtype = 0; ap.kv_cache_dtype = 0; ap.record_kv_metrics = 1;
    KV(kvc_paged_attention_decode(&ap, s));
    CK(hipStreamSynchronize(s));
    std::vector<_Float16> g_out(q.size());
    std::vector<float> g_met(w_met.size());
    CK(hipMemcpy(g_out.data(), dout, g_out.size() * sizeof(_Float16), hipMemcpyDeviceToHost));
    CK(hipMemcpy(g_met.data(), dmet, g_met.size() * sizeof(float), hipMemcpyDeviceToHost));
    double eo = 0, em = 0;
    for (size_t i = 0; i < g_out.size(); ++i) { const double d = fabs((double)(float)g_out[i] - w_out[i]); eo = d > eo ? d : eo; }
    for (size_t i = 0; i < g_met.size(); ++i) {
      if ((w_met[i] < 0) != (g_met[i] < 0)) em = 1.0;                       // same slots written
      else if (w_met[i] >= 0) { const double d = fabs((double)g_met[i] - w_met[i]) / (w_met[i] + 1e-9); em = d > em ? d : em; }
    }
    // tolerances of the reference's own test (test_kvcompress_attention.py:145, 356-357), weights
    // relaxed to 1e-4 for the float summation order of this plain loop
    const bool aok = eo < 1e-3 && em < 1e-4;
    if (!aok) printf("attention mismatch: out %.3g metrics %.3g\n", eo, em);
    ok = ok && aok;
  }
  long moves = 0;
  for (int g = 0; g < G; ++g) moves += w_cnt[g];
  printf("%s (%ld moves)\n", ok ? "CABI_OK" : "CABI_FAIL", moves);
  return ok ? 0 : 1;
}
