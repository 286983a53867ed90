/*
 * kvc_oracle.c -- C restatement of the serial reference kernels, for parity checks at
 * sizes where the Python loops of oracle/kvc_oracle.py are too slow, and for the
 * cpu_baseline leg of bench.py.
 *
 * TEST INFRASTRUCTURE ONLY: nothing under vllm_kvcompress_amd/ links or loads this.
 * Each function follows the reference kernel it names line by line (one "thread" = one
 * loop iteration of the outer head loop); it is pinned to the golden vectors through
 * tests/test_oracle_golden.py (C oracle == NumPy oracle == reference fixtures).
 *
 * The outer head loops are independent (one "thread" each in the reference), so they run under
 * OpenMP: orc_set_threads(n) picks how many host cores the cpu_baseline leg uses (1 = the scalar
 * port); results do not depend on it.
 *
 * build: gcc -O2 -fopenmp -shared -fPIC -o oracle/_build/libkvc_oracle.so oracle/kvc_oracle.c
 */
#include <omp.h>
#include <stdint.h>
#include <string.h>

static int g_threads = 1;
void orc_set_threads(int32_t n) { g_threads = n < 1 ? 1 : n; }
int32_t orc_max_threads(void) { return omp_get_num_procs(); }

/* count_block_evictions_kernel   csrc/kvcompress_eviction_kernels.cu:190-221 */
void orc_count_block_evictions(int32_t* evicted_block_count, int32_t* idx, const int32_t* offs,
                               const int32_t* hang, int32_t total_heads, int64_t total_kvs,
                               int32_t bs, int32_t null_value) {
#pragma omp parallel for schedule(static) num_threads(g_threads)
  for (int32_t g = 0; g < total_heads; ++g) {
    const int64_t start = offs[g];
    const int64_t end = (g + 1 >= total_heads) ? total_kvs : offs[g + 1];
    int32_t n = 0;
    for (int64_t i = start; i < end; i += bs) {
      if (idx[i] != null_value) ++n; else break;
    }
    evicted_block_count[g] = n;
    if (n > 0) {
      const int64_t last_end = start + (int64_t)n * bs;
      for (int64_t i = last_end - bs + hang[g]; i < last_end; ++i) idx[i] = null_value;
    }
  }
}

/* single_tier_schedule_cache_moves_kernel   csrc/kvcompress_eviction_kernels.cu:223-289
 * (+ the wrapper's fill_(0), vllm/_custom_ops.py:1168, when zero_fill != 0) */
void orc_schedule_t1_cache_moves(int32_t* moves, int64_t rows, int32_t* moves_count,
                                 const int32_t* evicted, const int32_t* ekc, const int32_t* offs,
                                 const int32_t* block_tables, const int32_t* context_lens,
                                 int32_t B, int32_t L, int32_t H, int32_t M, int32_t bs,
                                 int32_t zero_fill) {
  if (zero_fill) memset(moves, 0, (size_t)rows * 2 * sizeof(int32_t));
#pragma omp parallel for schedule(dynamic, 4) num_threads(g_threads)
  for (int32_t bl = 0; bl < B * L; ++bl) {
      const int32_t b = bl / L, l = bl % L;
      for (int32_t h = 0; h < H; ++h) {
        const int32_t slh = (b * L + l) * H + h;
        const int32_t lsh = (l * B + b) * H + h;
        const int32_t* bt = block_tables + (int64_t)lsh * M;
        const int64_t off = offs[slh];
        const int32_t cnt = ekc[slh];
        int32_t mc = 0, ec = 0;
        for (int32_t i = 0; i < cnt; ++i) {
          const int32_t src = context_lens[lsh] - 1 - i;
          const int32_t stop = evicted[off + cnt - 1 - ec];
          const int32_t dst = evicted[off + mc];
          if (dst >= src) break;
          if (src <= stop) { ++ec; continue; }
          moves[(off + mc) * 2] = bt[dst / bs] * bs + dst % bs;
          moves[(off + mc) * 2 + 1] = bt[src / bs] * bs + src % bs;
          ++mc;
        }
        moves_count[slh] = mc;
      }
  }
}

/* execute_cache_moves_kernel   csrc/kvcompress_eviction_kernels.cu:359-435
 * element size e bytes, K [NB, hd/x, bs, x], V [NB, hd, bs] */
void orc_execute_cache_moves(uint8_t* k, uint8_t* v, float* metrics, int32_t* positions,
                             const int32_t* moves, const int32_t* count, const int32_t* offs,
                             int32_t total_heads, int32_t bs, int32_t hd, int32_t e, int32_t x) {
  const int64_t block_stride = (int64_t)bs * hd;      /* elements */
  const int64_t k_stride = (int64_t)bs * x;
  /* moves are independent (kvcompress_eviction_kernels.cu:358), heads touch disjoint slots */
#pragma omp parallel for schedule(dynamic, 4) num_threads(g_threads)
  for (int32_t g = 0; g < total_heads; ++g) {
    for (int64_t i = offs[g]; i < (int64_t)offs[g] + count[g]; ++i) {
      const int32_t dst = moves[i * 2], src = moves[i * 2 + 1];
      const int64_t sbs = (int64_t)(src / bs) * block_stride, dbs = (int64_t)(dst / bs) * block_stride;
      const int32_t so = src % bs, dof = dst % bs;
      metrics[dst] = metrics[src];
      positions[dst] = positions[src];
      for (int64_t j = 0; j < block_stride; j += k_stride)
        memcpy(k + (dbs + (int64_t)dof * x + j) * e, k + (sbs + (int64_t)so * x + j) * e, (size_t)x * e);
      for (int64_t j = 0; j < block_stride; j += bs)
        memcpy(v + (dbs + dof + j) * e, v + (sbs + so + j) * e, (size_t)e);
    }
  }
}
