/*
 * kvc_mi355x.h -- C ABI of libkvc_mi355x.so
 *
 * MI355X (gfx950) implementation of the KV-Compress eviction + compaction hot path.
 * Every entry point takes plain device pointers, sizes and a HIP stream (passed as
 * void*, i.e. hipStream_t); no torch types.  All calls are asynchronous on `stream`,
 * never synchronise (one exception, named as such: kvc_schedule_batch_summary / _wait), never allocate: scratch memory is passed in by the caller (query
 * the size with the matching *_workspace_bytes function).  All index tensors are
 * int32 (reference: vllm/kvcompress/README.md:27, kernels reinterpret_cast<int*>,
 * csrc/kvcompress_eviction_kernels.cu:527-542).
 *
 * Return value: 0 = ok, 1 = invalid argument / unsupported shape (the reference's
 * TORCH_CHECK(false, "Unsupported block size: ...") -> RuntimeError), 2 = HIP error.
 * kvc_last_error() returns the message for the calling thread.
 *
 * Each function cites the reference interface it replaces (paths relative to the
 * reference repo IsaacRe/vllm-kvcompress @ 2024-12-20).
 *
 * Head order: "g" always enumerates heads in (seq, layer, kv_head) order,
 * g = b*L*H + l*H + h, which is the order of evicted_kv_offsets / hanging_token_count /
 * evicted_kv_count / cache_moves_count ([B,L,H]).  context_lens and block_tables are
 * layer-major ([L,B,H] / [L,B,H,M]) exactly as in the reference.
 */
#ifndef KVC_MI355X_H
#define KVC_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* kvc_stream_t; /* hipStream_t */

/* The ABI this header describes; kvc_abi_version() of the library loaded at run time must return it
 * (kvc_schedule_params grew fields in the middle between versions: a host built against another
 * version must not call in).  The Python binding and tests/cabi/cabi_host.cpp check it at start-up. */
#define KVC_ABI_VERSION 8

int kvc_abi_version(void);
const char* kvc_last_error(void);

/* ---------------------------------------------------------------------------------
 * ABI version 7: the layout of the bytes INSIDE a cache block.
 * The fork allocates the KV cache as an opaque [2, num_blocks, block_size * head_size] tensor and hands the
 * ops views of it (vllm/attention/ops/paged_attn.py:262-284: K [NB, hd/x, bs, x], V [NB, hd, bs]); a block
 * belongs to one (sequence, layer, KV head); swap_blocks / copy_blocks move whole blocks.  What a block's
 * head_size * block_size elements MEAN is known to exactly three ops -- the cache write
 * (kvcompress_reshape_and_cache), the decode attention (kvcompress_paged_attention_v1 / _v2) and
 * execute_cache_moves -- and all three are this library's.  Two layouts, chosen per call:
 *   KVC_LAYOUT_REFERENCE   the reference's: K [hd/x][bs][x] (x = 16 B of elements), V [hd][bs]
 *                          (csrc/kvcompress_cache_kernels.cu:57-77).  A slot is 16 B pieces at a 16 * bs B
 *                          stride in K and single elements at a bs * e B stride in V: moving ONE slot
 *                          into a block rewrites the block's whole V image.
 *   KVC_LAYOUT_SLOT_MAJOR  K [bs][hd], V [bs][hd]: a slot's K (and V) is ONE contiguous run of
 *                          head_size * elem_bytes bytes at byte offset slot * head_size * elem_bytes of
 *                          the K (V) plane.  A move is two contiguous copies; nothing else is touched.
 * Tensor shapes, op signatures, block tables, slot numbers and the metric store are the same in both;
 * a cache must be written, attended to and compacted in ONE layout (the library cannot tell them apart).
 * --------------------------------------------------------------------------------- */
#define KVC_LAYOUT_REFERENCE 0
#define KVC_LAYOUT_SLOT_MAJOR 1

/* ---------------------------------------------------------------------------------
 * A4  count_block_evictions
 * replaces torch.ops._C_kvc_ops.count_block_evictions
 *   (csrc/torch_bindings.cpp:388-394, csrc/kvcompress_eviction_kernels.cu:190-221,640-664,
 *    Python wrapper vllm/_custom_ops.py:1065-1086)
 * In place: evicted_block_count[g] = length of the leading run of chunks whose first
 * entry != null_value; the [hanging..block_size) tail of the last such chunk is set to
 * null_value.  The segment of the last head ends at total_kvs.
 * Any block_size >= 1 is accepted (the reference: {1,2,4,16}).
 * --------------------------------------------------------------------------------- */
int kvc_count_block_evictions(int32_t* evicted_block_count,      /* [G] out */
                              int32_t* evicted_logical_indices,  /* [total_kvs] in/out */
                              const int32_t* evicted_kv_offsets, /* [G] */
                              const int32_t* hanging_token_count,/* [G] */
                              int32_t total_heads, int64_t total_kvs, int32_t block_size,
                              int32_t null_value, kvc_stream_t stream);

/* ---------------------------------------------------------------------------------
 * A5  schedule_t1_cache_moves  (+ the wrapper's zero fill of the whole workspace)
 * replaces torch.ops._C_kvc_ops.schedule_t1_cache_moves and the
 *   out_cache_moves_indices.fill_(0) in front of it
 *   (csrc/torch_bindings.cpp:396-402, csrc/kvcompress_eviction_kernels.cu:223-289,698-726,
 *    vllm/_custom_ops.py:1158-1179)
 * cache_moves_idx rows [off_g, off_g+count_g) = (dst_physical_slot, src_physical_slot);
 * if zero_fill != 0 every other row of the [cache_moves_rows,2] workspace is written 0
 * (the observable result of the reference wrapper); with zero_fill == 0 rows that hold
 * no move are left untouched (what the bare reference op does).
 * --------------------------------------------------------------------------------- */
int kvc_schedule_t1_cache_moves(int32_t* cache_moves_idx,            /* [rows,2] out */
                                int64_t cache_moves_rows,
                                int32_t* cache_moves_count,          /* [B,L,H] out */
                                const int32_t* evicted_logical_indices, /* [>=N] */
                                const int32_t* evicted_kv_count,     /* [B,L,H] */
                                const int32_t* evicted_kv_offsets,   /* [B,L,H] */
                                const int32_t* block_tables,         /* [L,B,H,M] */
                                const int32_t* context_lens,         /* [L,B,H] */
                                int32_t num_seqs, int32_t num_layers, int32_t num_kv_heads,
                                int32_t max_num_blocks_per_seq, int32_t block_size,
                                int32_t zero_fill, kvc_stream_t stream);
/* ABI version 4: the same op with two optional by-products.
 *
 * zero_fill = 2 + dirty_map: the wrapper's fill_(0) without writing the zeros again.  The reference clears
 *   the WHOLE table in front of every call (8 B per candidate slot -- 2.2 GB per decode step at 256
 *   resident sequences) although the table it clears is its own persistent workspace
 *   (vllm/kvcompress/scheduler.py:74-86), zero everywhere except where the previous call wrote moves.
 *   dirty_map (kvc_cache_moves_dirty_map_bytes(rows, block_size) bytes, owned by whoever owns the table)
 *   holds one bit per chunk of block_size rows that may be non-zero.  With zero_fill = 2 only marked
 *   chunks are cleared and the map is rewritten for the moves of this call: the table ends up exactly
 *   as after fill_(0) + the bare op.  With zero_fill = 1 and a map that the caller has cleared, every
 *   row without a move is written 0 and the map is set up (first use of a table, or after anybody
 *   else wrote to it).  dirty_map may be NULL with zero_fill 0 / 1.
 *
 * plan_out (kvc_cache_moves_plan_bytes() bytes, 16-byte aligned, or NULL): how the moves spread over
 *   the heads, in the form kvc_execute_cache_moves_planned takes: execute_cache_moves of exactly this
 *   move list is then ONE launch (no planning pass over the list, no per-block claim table: a list
 *   made here writes every destination block from one run of consecutive moves). */
size_t kvc_cache_moves_dirty_map_bytes(int64_t cache_moves_rows, int32_t block_size);
size_t kvc_cache_moves_plan_bytes(void);
int kvc_schedule_t1_cache_moves_ex(int32_t* cache_moves_idx, int64_t cache_moves_rows,
                                   int32_t* cache_moves_count,
                                   const int32_t* evicted_logical_indices,
                                   const int32_t* evicted_kv_count,
                                   const int32_t* evicted_kv_offsets,
                                   const int32_t* block_tables, const int32_t* context_lens,
                                   int32_t num_seqs, int32_t num_layers, int32_t num_kv_heads,
                                   int32_t max_num_blocks_per_seq, int32_t block_size,
                                   int32_t zero_fill, uint32_t* dirty_map, size_t dirty_map_bytes,
                                   int32_t* plan_out, kvc_stream_t stream);

/* ---------------------------------------------------------------------------------
 * A6  execute_cache_moves  (the K/V gather/scatter compaction)
 * replaces torch.ops._C_kvc_ops.execute_cache_moves
 *   (csrc/torch_bindings.cpp:410-418, csrc/kvcompress_eviction_kernels.cu:359-435,858-929,
 *    vllm/_custom_ops.py:1220-1256, caller vllm/worker/cache_engine.py:139-151)
 * k_cache [NB, head_size/x, block_size, x], v_cache [NB, head_size, block_size]; x =
 * vec_size = k_cache.size(3) (16/elem_bytes in the engine: KVCAttention.split_kv_cache,
 * vllm/attention/ops/paged_attn.py:272-284); pure byte copy, elem_bytes in {1,2,4}.
 * Any block_size/head_size/vec_size combination works (reference: bs {1,2,4,16}, hd
 * {1,2,4,128}, x {1,2,8}); hd 64/128/256 with x*elem_bytes == 16 take the row-wise path.  For every head g and
 * every j < cache_moves_count[g]: (dst,src) = cache_moves_idx[evicted_kv_offsets[g]+j];
 * K row, V row, kv_metrics and kv_position of slot src are copied to slot dst.  Source
 * slots are left untouched.  Moves must be independent (no dst is a src, dsts distinct),
 * the only case the reference defines (kvcompress_eviction_kernels.cu:358).
 * blocks_per_head / threads_per_head of the reference signature are launch hints of the
 * CUDA kernel and have no meaning here (the Python wrapper accepts and ignores them).
 * workspace: kvc_execute_cache_moves_workspace_bytes(total_heads, num_blocks) bytes of
 * device memory, 16-byte aligned (the list's plan + one claim byte per physical block);
 * no need to clear it, the planning kernel does.
 * The op is two halves that may also be called separately on the same stream with the same
 * move list and workspace (what bench.py does to time the data kernel alone with events on
 * the caller's stream): _plan = the two small planning launches (reads only the move list),
 * _apply = the compaction kernel.  kvc_execute_cache_moves == _plan followed by _apply.
 * _planned (ABI version 4) = the compaction kernel alone over the plan that
 * kvc_schedule_t1_cache_moves_ex left behind for exactly this (cache_moves_idx, cache_moves_count,
 * evicted_kv_offsets), none of them modified since -- the caller vouches for that (the Python
 * and C++ bindings check tensor identity and version counters and take the self-contained op
 * otherwise).
 * --------------------------------------------------------------------------------- */
size_t kvc_execute_cache_moves_workspace_bytes(int32_t total_heads, int64_t num_blocks);
int kvc_execute_cache_moves(void* k_cache, void* v_cache, float* kv_metrics,
                            int32_t* kv_position,
                            const int32_t* cache_moves_idx,    /* [rows,2] */
                            const int32_t* cache_moves_count,  /* [G] */
                            const int32_t* evicted_kv_offsets, /* [G] */
                            int32_t total_heads, int64_t num_blocks, int32_t block_size,
                            int32_t head_size, int32_t elem_bytes, int32_t vec_size,
                            void* workspace, size_t workspace_bytes, kvc_stream_t stream);
int kvc_execute_cache_moves_plan(const int32_t* cache_moves_idx, const int32_t* cache_moves_count,
                                 const int32_t* evicted_kv_offsets, int32_t total_heads,
                                 int64_t num_blocks, int32_t block_size, int32_t head_size,
                                 int32_t elem_bytes, int32_t vec_size, void* workspace,
                                 size_t workspace_bytes, kvc_stream_t stream);
int kvc_execute_cache_moves_apply(void* k_cache, void* v_cache, float* kv_metrics,
                                  int32_t* kv_position, const int32_t* cache_moves_idx,
                                  const int32_t* cache_moves_count,
                                  const int32_t* evicted_kv_offsets, int32_t total_heads,
                                  int64_t num_blocks, int32_t block_size, int32_t head_size,
                                  int32_t elem_bytes, int32_t vec_size, void* workspace,
                                  size_t workspace_bytes, kvc_stream_t stream);
int kvc_execute_cache_moves_planned(void* k_cache, void* v_cache, float* kv_metrics,
                                    int32_t* kv_position, const int32_t* cache_moves_idx,
                                    const int32_t* cache_moves_count,
                                    const int32_t* evicted_kv_offsets, int32_t total_heads,
                                    int64_t num_blocks, int32_t block_size, int32_t head_size,
                                    int32_t elem_bytes, int32_t vec_size, const int32_t* plan,
                                    kvc_stream_t stream);

/* execute_cache_moves on a KVC_LAYOUT_SLOT_MAJOR cache (ABI version 7): same move list, same semantics
 * (csrc/kvcompress_eviction_kernels.cu:359-435), but a move is 2 contiguous copies of head_size * elem_bytes
 * bytes + 8 B of metric / position: no destination image is read, nothing but the moved bytes is written.
 * head_size * elem_bytes must be a multiple of 16.  `plan`: the plan kvc_schedule_t1_cache_moves_ex left for
 * exactly this list (caller vouches, as for _planned), or the workspace after kvc_execute_cache_moves_slot_major_plan
 * on the same stream; NULL = the call plans for itself into `workspace` (kvc_cache_moves_plan_bytes() bytes,
 * 16-byte aligned; one small launch more).  No claim table: slots are copied independently, any independent
 * move list is handled the same way. */
int kvc_execute_cache_moves_slot_major_plan(const int32_t* cache_moves_count, int32_t total_heads,
                                            void* workspace, size_t workspace_bytes, kvc_stream_t stream);
int kvc_execute_cache_moves_slot_major(void* k_cache, void* v_cache, float* kv_metrics,
                                       int32_t* kv_position, const int32_t* cache_moves_idx,
                                       const int32_t* cache_moves_count,
                                       const int32_t* evicted_kv_offsets, int32_t total_heads,
                                       int64_t num_blocks, int32_t block_size, int32_t head_size,
                                       int32_t elem_bytes, const int32_t* plan, void* workspace,
                                       size_t workspace_bytes, kvc_stream_t stream);

/* ---------------------------------------------------------------------------------
 * A3  CompressionMetrics.schedule_evictions  (mask -> select -> count -> emit)
 * replaces the torch-op pipeline of vllm/kvcompress/metrics.py:441-847 (six device
 * sorts + a host loop) including its call of count_block_evictions (:774).
 * Outputs are bit-identical to the reference on tie-free metrics; ties are ordered
 * "stable by masked flat index" (physical block, offset) -- see DESIGN.md.
 *
 * mode 0 = "reference": reproduces the reference's batch>1 inf-count quirk
 *          (metrics.py:718-721); mode 1 = "per_sequence": every sequence scheduled as
 *          if alone (== reference called with B=1 per sequence).
 * seq_slot_of_seq[s] = batch slot of sequence index s, or -1 (size seq_slot_len).
 * bias may be NULL (== the reference's default zero bias, metrics.py:166-173).
 * total_slots N = sum over heads of ceil(ctx/bs)*bs (== size of evicted_logical_indices).
 * Asynchronous on `stream`.  The small-eviction schedule runs the null padding of
 * evicted_logical_indices on a side stream of the library's own (created on first use per host
 * thread and device, outside stream capture; forked off and joined back into `stream` inside
 * the call, so the call's results are ordered on `stream` like every other entry point's).
 * --------------------------------------------------------------------------------- */
typedef struct kvc_schedule_params {
  /* CompressionMetrics state (metrics.py:220-275) */
  const float* metrics;                       /* [NB, bs] */
  const int32_t* token_positions;             /* [NB, bs] */
  const int32_t* seq_index_by_block;          /* [NB] (-1 = unallocated) */
  const int32_t* layer_index_by_block;        /* [NB] */
  const int32_t* head_index_by_block;         /* [NB] */
  const int32_t* logical_block_num_by_block;  /* [NB] */
  int64_t num_blocks;                         /* NB */
  int32_t block_size, num_layers, num_kv_heads, num_seqs;
  /* per batch */
  const int32_t* seq_slot_of_seq;             /* [seq_slot_len] */
  int32_t seq_slot_len;
  const int32_t* seq_positions;               /* [B] */
  const int32_t* num_protected;               /* [B] */
  const int32_t* evicted_blocks_per_seq;      /* [B] */
  const int32_t* context_lens;                /* [L,B,H] */
  const int32_t* hanging_token_count;         /* [B,L,H] */
  const int32_t* evicted_kv_offsets;          /* [B,L,H] */
  int64_t total_slots;                        /* N */
  /* options */
  int32_t use_average;                        /* metrics.py:495-501 */
  int32_t num_sinks;                          /* metrics.py:542 */
  const float* bias;                          /* [L,H,num_bins] or NULL */
  const int32_t* position_bins;               /* [num_bins] */
  int32_t num_bins;
  float bias_weight;
  int32_t mode;                               /* 0 reference, 1 per_sequence */
  int32_t null_value;                         /* MAX_INT = 2147483000 (metrics.py:12) */
  int32_t lean;                               /* extension, 0 = reference-observable outputs.
                                               * bit 0: do not pad evicted_logical_indices with
                                               *   null_value behind the evicted_kv_count[g] entries
                                               *   of a head (consumers read only those);
                                               * bit 1: skip the defensive 0xFF clear of the key
                                               *   scratch (valid when every logical block below
                                               *   ceil(ctx/bs) of the selected sequences has
                                               *   block metadata, as the engine maintains) */
  int32_t max_evicted_blocks_hint;            /* host-known upper bound of evicted_blocks_per_seq (the
                                               * reference passes a Python list, scheduler.py:184-560),
                                               * or -1 if unknown.  Picks the schedule: when a step frees
                                               * on average <= 2 blocks per head (bs 16) -- the continual-
                                               * compression steady state -- the metric store is streamed
                                               * once, in physical order, instead of a key per slot being
                                               * written and read five times.  Results are
                                               * identical either way; a wrong hint only costs time. */
  const int32_t* block_tables;                /* optional (NULL: none): BlockState.block_tables
                                               * [L, max_num_seqs, H, block_tables_width].  ABI version 3: when
                                               * the batch takes less than half of the cache (an engine sizes
                                               * its cache to HBM) the keys are built in logical order through
                                               * these tables instead of by a sweep over every block's metadata
                                               * (3 scattered accesses per block of the batch instead of 7, no
                                               * sweep).  They must be the tables the per-block metadata was
                                               * written from; a listed block whose metadata no longer names
                                               * the sequence counts as not there, as in the sweep.  The
                                               * small-eviction schedule streams the store in physical order
                                               * and ignores them (version 2 ignored them everywhere, round
                                               * 2's schedule gathered rows through them). */
  const int32_t* seq_index_of_slot;           /* [B] with block_tables: the sequences' indices into dimension 1 */
  int32_t max_num_seqs, block_tables_width;   /* with block_tables */
  int32_t schedule_path;                      /* 0 = choose by the hint and the shapes, 1 = digit rounds only,
                                               * 2 = small-eviction schedule whenever the shapes allow
                                               * (falls back on device when it cannot finish exactly),
                                               * 3 = like 2, always streaming the position rows (tests),
                                               * 4 = bracket schedule whenever the shapes allow: bulk
                                               * evictions, T* from a bracket around a sample's quantile
                                               * and ONE counting pass instead of four digit rounds (0
                                               * takes it from 64 Ki slots per sequence and 64 blocks per
                                               * head on; falls back on device like 2) */
  int32_t sample_stride;                      /* small-eviction schedule: its pivots come from a sample
                                               * of one physical block in `sample_stride` (a power of
                                               * two <= 256); 0 = chosen from the batch size.  Results
                                               * do not depend on it (tests force several values). */
  int32_t fallback_grid;                      /* ABI version 4.  Workgroups of the single launch that redoes a
                                               * small-eviction / bracket call on the general pipeline when
                                               * its flag was raised; 0 = what the occupancy query says is
                                               * resident at once.  Results do not depend on it: the phases
                                               * of that launch wait for work, not for workgroups, so a grid
                                               * that is not resident at once (tests force one) only runs
                                               * slower. */
  int32_t uniform_evict;                      /* ABI version 4.  != 0: the reference's other selection rule
                                               * (metrics.py:639-666, `uniform_evict=True`; its scheduler never
                                               * passes it): every head of sequence i frees
                                               * evicted_blocks_per_seq[i] / (L*H) chunks -- its own lowest --
                                               * at most its finite-threshold chunks (the reference asserts
                                               * they are finite, and needs heads of equal length for a
                                               * reshape; neither is required here). */
  uint32_t* eli_dirty_map;                    /* ABI version 4, optional (NULL: none).  The small-eviction schedule
                                               * writes a handful of indices per head and then pads 4 B per
                                               * candidate slot with null_value -- 1.08 GB per decode step at 256
                                               * resident sequences, the largest single item of that schedule.  A
                                               * caller that keeps evicted_logical_indices between calls passes the
                                               * buffer's dirty map (one bit per block_size entries, (entries /
                                               * block_size + 31) / 32 + 1 words; all zero for a buffer that is
                                               * null_value everywhere): only what earlier calls left behind is
                                               * cleared, and the map is rewritten.  Contents after the call are
                                               * identical to the padded list.  Ignored by the other schedules
                                               * (they write every entry; a call that falls back from the
                                               * small-eviction schedule keeps the map right). */
  void* harvest_buf;                          /* ABI version 5, optional (NULL: none).  Harvest-ahead (DESIGN.md 3.1,
                                               * "one sweep per decode step"): kvc_harvest_buffer_bytes() bytes the
                                               * caller keeps between kvc_aggregate_decode_harvest and this call --
                                               * per-sequence pivots, per-head candidate lists. */
  int32_t harvest;                            /* bit 0: harvest_buf holds the candidate lists
                                               *   kvc_aggregate_decode_harvest made for EXACTLY this call (same
                                               *   batch, context_lens, positions, protected windows; the store
                                               *   untouched since): the small-eviction schedule does not stream
                                               *   the metric store again.  Exactness does not depend on the
                                               *   pivots the lists were made with: lists that do not cover the
                                               *   selection raise the `fallback` flag like any short record.
                                               * bit 1: leave in harvest_buf the pivots for the harvest of the
                                               *   NEXT decode step, made from what is left of this call's lists
                                               *   (its own collecting pass's or harvested ones: all keys below the
                                               *   pivot they were made with) once the selection has said what
                                               *   leaves: the (1 + harvest_widen) * Tgt-th smallest remaining key,
                                               *   Tgt = k * bs + sum(hang - 1) with this call's k and hang taken as
                                               *   the next step's; extrapolated upwards when fewer are left.
                                               * bit 2: this call's own collecting pass takes its pivots from
                                               *   harvest_buf (left there by the previous call's bit 1, for the same
                                               *   batch) instead of from a sample: no sampling pass, no pivot kernel,
                                               *   and half the candidates (an exact count needs no margin for a
                                               *   sample's error).  For callers that do not harvest; the pivots are
                                               *   one decode step of attention old either way, and a pass that lists
                                               *   too little raises the flag as always.  Only the first
                                               *   kvc_harvest_pivot_bytes() of the buffer are touched.
                                               * bit 3 (ABI version 6, with bit 0): the lists were made by the decode
                                               *   attention's fused-metric epilogue (kvc_attention_harvest_begin) and
                                               *   carry what they were made with -- every head's context length, every
                                               *   sequence's position and protected window; the call compares them with
                                               *   its own on the device and raises the flag on any difference (lists of
                                               *   another batch are redone, not trusted).  The caller still vouches for
                                               *   the store: nothing wrote to it since the last attention launch.
                                               * All of them are for calls that take the small-eviction schedule
                                               * (kvc_harvest_eligible / kvc_pivot_memory_eligible) and are ignored
                                               * by the other schedules; bits 0 and 2 without harvest_buf are an
                                               * error. */
  float harvest_widen;                        /* bit 1: allowance for keys that the next step's attention lifts over
                                               * the pivot, as a fraction of Tgt (<= 0: 0.25) */
  int32_t harvest_position_delta;             /* ABI version 6, kvc_aggregate_decode_harvest only: the schedule call
                                               * that will follow has seq_positions[i] + delta (a caller that harvests
                                               * for the NEXT iteration's call from the arguments of the last one: a
                                               * decode step adds one token per sequence).  With it, context_lens may be
                                               * NULL there (not known yet): a block belongs to the batch by its metadata
                                               * alone.  Lists made this way carry the positions and protected windows
                                               * they were made with; the schedule call takes them with harvest bits
                                               * 0 | 3 and verifies on the device. */
  /* outputs */
  int32_t* evicted_logical_indices;           /* [N] */
  int32_t* evicted_kv_count;                  /* [B,L,H] */
  int32_t* evicted_block_count;               /* [B,L,H] */
  const int64_t* total_slots_dev;             /* ABI version 8, optional (NULL: total_slots is N).  A host that holds its batch
                                               * as device tensors only (the fork's scheduler) learns N from the device; with
                                               * this field it does not have to WAIT for it before it launches: total_slots is
                                               * then an UPPER BOUND (a multiple of block_size) that sizes the scratch layout,
                                               * the grids and evicted_logical_indices ([total_slots]; entries from N on are
                                               * left untouched), and the kernels take the true N from total_slots_dev[0] --
                                               * device memory written EARLIER ON THE SAME STREAM by
                                               * kvc_schedule_batch_summary_deferred.  total_slots_dev[1] != 0 (N exceeded
                                               * the bound) voids the call: every kernel returns at once, the outputs are
                                               * unspecified and the caller repeats the call with the N it has read meanwhile.
                                               * For the digit rounds and the bracket schedule without the reference's batch > 1
                                               * rule: max_evicted_blocks_hint must be -1 (the small-eviction schedule is chosen
                                               * from host-side counts), mode 0 with num_seqs > 1, uniform_evict, block_tables
                                               * and harvest bits are refused. */
  uint32_t* flag_mirror;                      /* ABI version 8, optional (NULL: none).  One word of page-locked, device-mapped
                                               * host memory: the single gated launch at the end of a small-eviction / bracket
                                               * call stores (flag_ticket << 8) | (the call's flag word & 0xFF) there (system
                                               * scope) -- bit 0: the schedule could not finish exactly and the call was redone
                                               * on the device, bit 1: that launch gave up a wait (outputs void), bit 3: keys
                                               * and holes were made anew.  A host that wants to know reads the word some time
                                               * later instead of enqueueing a copy behind every call.  Not written by calls
                                               * that take the digit rounds directly (they have no flag) nor under the
                                               * reference's batch > 1 rule with more than 256 sequences (a launch chain). */
  uint32_t flag_ticket;                       /* (24 bits) */
} kvc_schedule_params;

size_t kvc_schedule_evictions_workspace_bytes(int64_t total_slots, int32_t total_heads,
                                              int32_t num_seqs, int32_t block_size);
/* ABI version 6.  What the HOST needs to know of a batch whose description it holds as device tensors
 * only -- the fork's scheduler makes evicted_blocks_per_seq a device int tensor and passes neither N
 * nor its maximum (vllm/kvcompress/scheduler.py:245-247, 491-499), and the reference method sizes its
 * outputs from device data (metrics.py:465-489: boolean-mask gathers, each a synchronisation):
 *   host_out[0]     = N = block_size * sum over the total_heads entries of context_lens of
 *                     ceil(ctx / block_size)   (size of evicted_logical_indices, total_slots above)
 *   host_out[1 + i] = evicted_blocks_per_seq[i], i < num_seqs   (evicted_blocks_per_seq may be NULL
 *                     with num_seqs 0: N only)
 * The entry point that may wait (the only one, with kvc_schedule_batch_summary_wait): a single launch; with
 * wait != 0 the calling thread then blocks until the numbers are in host memory (hipStreamSynchronize(stream)
 * -- everything enqueued on `stream` before has run by then as well; not allowed under stream capture), with
 * wait == 0 the call returns behind the launch and the caller prepares whatever does not need the numbers
 * before it calls kvc_schedule_batch_summary_wait(stream).  host_out: 1 + num_seqs int64.  With
 * host_mapped != 0 it must be page-locked host memory the device can write (hipHostMalloc /
 * torch's pin_memory(): the kernel stores into it directly, no copy is enqueued); with
 * host_mapped == 0 it may be any host memory and `workspace` (>= 8 * (1 + num_seqs) bytes of device
 * memory, 8-byte aligned) carries the numbers to a hipMemcpyAsync. */
int kvc_schedule_batch_summary(const int32_t* context_lens, int32_t total_heads, int32_t block_size,
                               const int32_t* evicted_blocks_per_seq, int32_t num_seqs,
                               int64_t* host_out, int32_t host_mapped, int32_t wait, void* workspace,
                               size_t workspace_bytes, kvc_stream_t stream);
int kvc_schedule_batch_summary_wait(kvc_stream_t stream);
/* ABI version 7: the same launch for a host that polls instead of waiting for the stream.  host_mapped_out:
 * 2 + num_seqs int64 of page-locked, device-mapped host memory; the kernel stores N and the counts, then
 * (system-scope release) `ticket` (any non-zero value the caller has not used for this buffer before) into
 * host_mapped_out[1 + num_seqs].  A host that reads the ticket (acquire) sees the numbers; it never has to
 * call _wait.  Never waits itself. */
int kvc_schedule_batch_summary_ticket(const int32_t* context_lens, int32_t total_heads, int32_t block_size,
                                      const int32_t* evicted_blocks_per_seq, int32_t num_seqs,
                                      int64_t* host_mapped_out, int64_t ticket, kvc_stream_t stream);
/* ABI version 8: the same launch, which also leaves N where the schedule's kernels can read it:
 * total_slots_dev[0] = N, total_slots_dev[1] = (N > total_slots_bound) -- two int64 of DEVICE memory, read by a
 * kvc_schedule_evictions call enqueued behind it with kvc_schedule_params.total_slots_dev.  The host enqueues both
 * back to back and polls the ticket while the device works. */
int kvc_schedule_batch_summary_deferred(const int32_t* context_lens, int32_t total_heads, int32_t block_size,
                                        const int32_t* evicted_blocks_per_seq, int32_t num_seqs,
                                        int64_t* host_mapped_out, int64_t ticket, int64_t* total_slots_dev,
                                        int64_t total_slots_bound, kvc_stream_t stream);
/* Harvest-ahead (ABI version 5).  Continual compression streams the whole metric store twice per
 * decode step: aggregate_decode adds the step's attention to it, and a moment later the
 * small-eviction schedule reads it all again to find the ~1 % of the keys that lie below each
 * sequence's pivot.  kvc_aggregate_decode_harvest is kvc_aggregate_decode (same arithmetic, bit
 * for bit, over the whole cache) that looks at every sum while it is in registers: keys of the
 * batch `p` describes (the schedule call that will follow: metadata, seq_slot_of_seq, seq_positions,
 * num_protected, context_lens, block_size ...; the outputs and evicted_blocks_per_seq are not read)
 * that lie below the pivot the PREVIOUS schedule call left in p->harvest_buf (harvest bit 1) are
 * appended to their head's candidate list there.  kvc_schedule_evictions with harvest bit 0 then
 * runs records -> selection -> emission on these lists.  The result is exact or the flag is raised
 * (fallback on device), exactly as when the lists come from the schedule's own collecting pass.
 * Eligible: every call that takes the small-eviction schedule (block size 8 / 16 / 32).  In its
 * position-lazy form (no use_average, no bias, mode 1 or one sequence) the pass reads a position only
 * for the keys below the pivot; otherwise (averaged metrics, a position bias, the reference's batch > 1
 * rule) it streams the position rows next to the sums, makes every key in full and counts the masked
 * slots per head, as that schedule's full collecting pass does.  num_queries_per_kv 4 or 8 read the
 * temp rows as 16-byte loads, any other value as scalars (as kvc_aggregate_decode does). */
size_t kvc_harvest_buffer_bytes(int32_t total_heads, int32_t num_seqs);
int32_t kvc_pivot_memory_eligible(const kvc_schedule_params* p);
size_t kvc_harvest_pivot_bytes(int32_t num_seqs);   /* its leading part: enough for harvest bits 1 and 2 without bit 0 */
int32_t kvc_harvest_eligible(const kvc_schedule_params* p, int32_t num_queries_per_kv);
int kvc_aggregate_decode_harvest(const kvc_schedule_params* p, float* temp_metrics,
                                 int32_t num_queries_per_kv, int32_t use_l2, int32_t clear_temp,
                                 kvc_stream_t stream);
int kvc_schedule_evictions(const kvc_schedule_params* p, void* workspace,
                           size_t workspace_bytes, kvc_stream_t stream);
/* introspection (tests, bench.py): 1 if a call with these parameters enqueues the small-eviction
 * schedule; and the byte offset inside the workspace of that schedule's `fallback` word -- non-zero
 * after the call if it could not finish exactly and the general pipeline recomputed the result. */
int32_t kvc_schedule_evictions_uses_small_eviction_schedule(const kvc_schedule_params* p);
/* which schedule a call with these parameters enqueues: 0 = the digit rounds (general pipeline),
 * 1 = small-eviction, 2 = bracket; 1 and 2 leave the `fallback` word behind: 0 = the schedule
 * finished exactly, bit 0 = it could not and the general pipeline behind it recomputed the result,
 * bit 1 (sticky) = that recomputation gave up a wait after ten seconds -- a device fault -- and the
 * outputs were overwritten with the schedule that evicts nothing; callers must treat bit 1 as an
 * error (CompressionMetrics raises RuntimeError). */
int32_t kvc_schedule_evictions_plan(const kvc_schedule_params* p);
/* ... and why (ABI version 4): bits 0-7 = why the call does not take the small-eviction schedule,
 * bits 8-15 = why it does not take the bracket schedule either (only meaningful when bits 0-7 are
 * non-zero), bits 16-23 = KVC_WHY_COUPLED_BATCH when a taken schedule keeps the gated launch chain
 * as its fallback instead of the single launch.  0 = the small-eviction schedule with the single
 * launch behind it. */
enum {
  KVC_WHY_TAKEN = 0,
  KVC_WHY_FORCED_PATH = 1,     /* schedule_path asks for another schedule */
  KVC_WHY_BLOCK_SIZE = 2,      /* small-eviction: block_size not in {8, 16, 32} */
  KVC_WHY_HINT_UNKNOWN = 3,    /* small-eviction: max_evicted_blocks_hint < 0 (counts passed as a device tensor) */
  KVC_WHY_BULK = 4,            /* small-eviction: the hint says a step frees more than 1/8 of what the heads' records hold */
  KVC_WHY_HEADS_PER_SEQ = 5,   /* num_layers * num_kv_heads > 1024 (per-head tables of one workgroup) */
  KVC_WHY_THRESHOLDS_LDS = 6,  /* small-eviction: L*H*(256/bs) recorded thresholds per sequence > 16384 */
  KVC_WHY_COUPLED_BATCH = 7,   /* mode 0 (the reference's batch > 1 rule) over more than 256 sequences */
  KVC_WHY_INDEX_RANGE = 8,     /* > 65535 sequences, or >= 2^32 slots */
  KVC_WHY_SMALL_BATCH = 9,     /* bracket: < 64 Ki slots per sequence or < 64 blocks per head (digit rounds are as fast) */
  KVC_WHY_EMPTY = 10,          /* nothing to schedule */
  KVC_WHY_UNIFORM = 11         /* uniform_evict: its own three launches */
};
int32_t kvc_schedule_evictions_plan_reason(const kvc_schedule_params* p);
/* 1 if a call with these parameters builds its keys through block_tables (see there) */
int32_t kvc_schedule_evictions_uses_block_tables(const kvc_schedule_params* p);
size_t kvc_schedule_evictions_fallback_offset(int64_t total_slots, int32_t total_heads,
                                              int32_t num_seqs, int32_t block_size);

/* ---------------------------------------------------------------------------------
 * A2a  aggregate_decode (+ clear_temp_metrics fused)
 * replaces CompressionMetrics.aggregate_decode / clear_temp_metrics
 *   (vllm/kvcompress/metrics.py:429-439, 337-342)
 * metrics[s] += sum_q temp[s,q]^2 (use_l2) or sum_q temp[s,q]; float32, q summed in
 * index order.  If clear_temp != 0 temp_metrics is zeroed in the same pass.
 * --------------------------------------------------------------------------------- */
int kvc_aggregate_decode(float* metrics, float* temp_metrics, int64_t num_slots,
                         int32_t num_queries_per_kv, int32_t use_l2, int32_t clear_temp,
                         kvc_stream_t stream);

/* A2b  aggregate_prefill  (vllm/kvcompress/metrics.py:396-427)
 * metrics.flat[slot_mapping[t,h]] += sum_q prefill_metrics[t, h*qpk + q]; slots unique. */
int kvc_aggregate_prefill(float* metrics, const float* prefill_metrics,
                          const int64_t* slot_mapping, int64_t num_tokens,
                          int32_t num_kv_heads, int32_t num_queries_per_kv,
                          kvc_stream_t stream);

/* A2c  prefill metric epilogue: square -> causal-buffer mask -> column sum -> (avg
 * scale) -> maxpool(7) -> accumulate, for one query block of softmax probabilities
 *   (vllm/attention/backends/flash_attn.py:1147-1161, 1189-1211)
 * out_kh [K, Hq] += f(probs [Hq, qb, K]);  q_offset = position of the tile's first query
 * row inside the sequence. */
int kvc_prefill_metric_epilogue(float* out_kh, const float* probs_hqk, int32_t num_q_heads,
                                int32_t q_block, int32_t num_keys, int32_t q_offset,
                                int32_t buffer_len, int32_t use_l2, int32_t use_average,
                                int32_t use_maxpool, void* workspace, size_t workspace_bytes,
                                kvc_stream_t stream);
size_t kvc_prefill_metric_epilogue_workspace_bytes(int32_t num_q_heads, int32_t num_keys);

/* F4  fused prefill metric collector: the whole of _naive_kvc_attention's inner loop for one
 * sequence, without materialising the probabilities
 *   (vllm/attention/backends/flash_attn.py:1143-1161 block loop, :1166-1211 per block)
 * The last num_observed query rows of the sequence (first one at position q_offset) are
 * processed in blocks of q_block rows; for every block
 *   out_kh [K, Hq] += maxpool7( sum over the block's rows r with k + buffer_len <= pos(r)
 *                               of P[r,k]^(2 if use_l2) [* (k+1)/rows_in_block] ),
 * P = softmax over keys k <= pos(r) of scale * (q_r . k_k), logits rounded to the input type
 * first like the reference's einsum.  query: first observed row, [num_observed, Hq, hd]; key:
 * first key of the sequence, [K, Hk, hd] (Hk = Hq or the un-repeated KV heads); strides in
 * elements between tokens; dtype 0 = fp16, 1 = bf16; head_size 64 or 128.
 * Floating point: agrees with the reference within the rounding of its fp16 logits. */
size_t kvc_prefill_metric_fused_workspace_bytes(int32_t num_q_heads, int32_t num_observed,
                                                int32_t num_keys);
int kvc_prefill_metric_fused(float* out_kh, const void* query, const void* key,
                             int32_t num_q_heads, int32_t num_k_heads, int32_t head_size,
                             int32_t num_observed, int32_t q_block, int32_t num_keys,
                             int32_t q_offset, int32_t buffer_len, int64_t q_stride,
                             int64_t k_stride, float scale, int32_t dtype, int32_t use_l2,
                             int32_t use_average, int32_t use_maxpool, void* workspace,
                             size_t workspace_bytes, kvc_stream_t stream);

/* ---------------------------------------------------------------------------------
 * A7  kvcompress_reshape_and_cache  ("auto" cache dtype: byte-identical store)
 * replaces torch.ops._C_cache_ops.kvcompress_reshape_and_cache
 *   (csrc/torch_bindings.cpp:353-362, csrc/kvcompress_cache_kernels.cu:27-139,
 *    vllm/_custom_ops.py:641-658)
 * key/value [T, H, head_size] with token strides key_stride/value_stride (elements);
 * slot_mapping [T*H] int64 (<0 = padding, skipped); kv_metrics[slot] = bias[h].
 * --------------------------------------------------------------------------------- */
int kvc_reshape_and_cache(const void* key, const void* value, void* key_cache,
                          void* value_cache, float* kv_metrics, const int64_t* slot_mapping,
                          const float* kv_metric_head_bias, int64_t num_tokens,
                          int32_t num_heads, int32_t head_size, int32_t block_size,
                          int32_t elem_bytes, int64_t key_stride, int64_t value_stride,
                          kvc_stream_t stream);

/* A7 with an fp8 cache: kv_cache_dtype "fp8"/"fp8_e4m3" (fp8_kind 0, OCP e4m3fn) or
 * "fp8_e5m2" (fp8_kind 1).  cache byte = fp8(float(x) / scale), round-to-nearest-even,
 * saturating to the largest finite value -- the reference's
 * __nv_cvt_float_to_fp8(x / scale, __NV_SATFINITE, type)
 *   (csrc/quantization/fp8/nvidia/quant_utils.cuh:456-489, dispatch :525-566).
 * src_dtype: 0 = fp16, 1 = bf16, 2 = fp32.  K vectors hold x = 16 fp8 elements. */
int kvc_reshape_and_cache_fp8(const void* key, const void* value, void* key_cache,
                              void* value_cache, float* kv_metrics, const int64_t* slot_mapping,
                              const float* kv_metric_head_bias, int64_t num_tokens,
                              int32_t num_heads, int32_t head_size, int32_t block_size,
                              int32_t src_dtype, int32_t fp8_kind, int64_t key_stride,
                              int64_t value_stride, float k_scale, float v_scale,
                              kvc_stream_t stream);

/* A7 into a cache of either block layout (ABI version 7; block_layout = KVC_LAYOUT_*): the two entry points
 * above are these with KVC_LAYOUT_REFERENCE. */
int kvc_reshape_and_cache_layout(const void* key, const void* value, void* key_cache,
                                 void* value_cache, float* kv_metrics, const int64_t* slot_mapping,
                                 const float* kv_metric_head_bias, int64_t num_tokens,
                                 int32_t num_heads, int32_t head_size, int32_t block_size,
                                 int32_t elem_bytes, int64_t key_stride, int64_t value_stride,
                                 int32_t block_layout, kvc_stream_t stream);
int kvc_reshape_and_cache_fp8_layout(const void* key, const void* value, void* key_cache,
                                     void* value_cache, float* kv_metrics, const int64_t* slot_mapping,
                                     const float* kv_metric_head_bias, int64_t num_tokens,
                                     int32_t num_heads, int32_t head_size, int32_t block_size,
                                     int32_t src_dtype, int32_t fp8_kind, int64_t key_stride,
                                     int64_t value_stride, float k_scale, float v_scale,
                                     int32_t block_layout, kvc_stream_t stream);

/* ---------------------------------------------------------------------------------
 * F2  the block-state side of a compression step
 * replaces BlockSpaceManagerKVC.free_compressed_blocks and what it calls
 *   (vllm/kvcompress/block_manager.py:466-530 -> block.py:367-379 last_n_allocated_block_mask,
 *    block.py:184-210 remove_trailing_blocks, block_manager.py:112-118 allocator.free,
 *    vllm/kvcompress/metrics.py:366-370 remove_metadata)
 * For batch position b (batch slot seq_slots[b]) and every (layer, head): the last
 * freed_block_count[b,l,h] allocated blocks (logical order) are freed: listed in
 * freed_blocks in the reference's order (layer, batch position, head, logical block
 * ascending), free_mask[blk] = 1 (may be NULL), seq_index_by_block[blk] = -1, and
 * context_lens[l, slot, h] -= clamp(n*bs - (bs - hanging), 0).
 * context_lens [L, max_num_seqs, H], block_tables [L, max_num_seqs, H, M] are the FULL
 * state tensors (BlockState, block.py:95-126); freed_block_count is [B, L, H] as produced
 * by schedule_evictions.  *freed_total = number of freed blocks (<= freed_capacity written).
 * --------------------------------------------------------------------------------- */
size_t kvc_free_compressed_blocks_workspace_bytes(int32_t num_layers, int32_t batch,
                                                  int32_t num_kv_heads);
int kvc_free_compressed_blocks(int32_t* context_lens, int32_t* seq_index_by_block,
                               uint8_t* free_mask, int32_t* freed_blocks, int32_t freed_capacity,
                               int32_t* freed_total, const int32_t* block_tables,
                               const int32_t* freed_block_count, const int32_t* seq_slots,
                               int32_t num_layers, int32_t batch, int32_t max_num_seqs,
                               int32_t num_kv_heads, int32_t max_num_blocks_per_seq,
                               int32_t block_size, void* workspace, size_t workspace_bytes,
                               kvc_stream_t stream);

/* ---------------------------------------------------------------------------------
 * F2  the append side of the block state: one decode step, token_count = 1
 * replaces BlockSpaceManagerKVC._append_to_sequence_batch and what it calls
 *   (vllm/kvcompress/block_manager.py:269-294 -> block.py:359-365 allocated_block_mask,
 *    block_manager.py:103-110 ParallelBlockAllocator.allocate, block.py:513-620
 *    get_batch_new_block_metadata, vllm/kvcompress/metrics.py:344-361 insert_metadata)
 * For batch position b (batch slot seq_slots[b]) and every (layer, head): a head whose context
 * length is a multiple of block_size gets logical block ctx / block_size; the new blocks are the
 * lowest-numbered free ones (free_mask != 0), handed out in (layer, batch position, head)
 * order; free_mask[blk] = 0, block_tables[l, slot, h, m] = blk, the four metadata rows of blk
 * are written and token_positions[blk, :] = last_token_position[b] + arange(block_size); then
 * context_lens[l, slot, h] += 1 for every head of the batch.  write_token_position != 0 (an
 * extension; the reference relies on the row written when the block was allocated) also stores
 * last_token_position[b] in the appended token's own slot.
 * status[0] = new blocks needed, status[1] = free blocks before the call, or -1 if some head's
 * block table has no room for its next block.  If there are fewer free blocks than needed (the
 * reference raises "Out of memory!") or a table is full, NOTHING is modified -- the caller
 * compares the two numbers.
 * --------------------------------------------------------------------------------- */
size_t kvc_append_slots_workspace_bytes(int32_t num_layers, int32_t batch, int32_t num_kv_heads,
                                        int64_t num_blocks);
int kvc_append_slots(int32_t* context_lens, int32_t* block_tables, uint8_t* free_mask,
                     int32_t* seq_index_by_block, int32_t* layer_index_by_block,
                     int32_t* head_index_by_block, int32_t* logical_block_num_by_block,
                     int32_t* token_positions, const int32_t* seq_slots,
                     const int32_t* last_token_position, int32_t* status, int32_t num_layers,
                     int32_t batch, int32_t max_num_seqs, int32_t num_kv_heads,
                     int32_t max_num_blocks_per_seq, int64_t num_blocks, int32_t block_size,
                     int32_t write_token_position, void* workspace, size_t workspace_bytes,
                     kvc_stream_t stream);

/* F2, the prefill side (ABI version 7): a new sequence's FIRST allocation
 * replaces BlockSpaceManagerKVC._add_sequence and what it calls
 *   (vllm/kvcompress/block_manager.py:196-222 -> :103-110 ParallelBlockAllocator.allocate,
 *    block.py:414-446 get_allocated_block_metadata, vllm/kvcompress/metrics.py:344-361 insert_metadata)
 * and BlockStateView.get_prefill_slot_mapping (block.py:275-303).
 * cnt = ceil(seq_len / block_size) blocks for every (layer, head): the L * H * cnt lowest-numbered free blocks
 * (free_mask, cleared for them), block r of that list to (l, h, j) = (r / (H cnt), r / cnt % H, r % cnt);
 * context_lens[l, seq_slot, h] = seq_len; block_tables[l, seq_slot, h, j]; the blocks' metadata rows
 * (sequence = seq_slot, layer, head, logical block j) and position rows j * bs + arange(bs); slot_mapping
 * (may be NULL) [L, seq_len, H] int64 = block[l, h, t / bs] * bs + t % bs -- what reshape_and_cache_kvc takes
 * for layer l.  status[0] = blocks needed, status[1] = free blocks before the call, or -1 when cnt exceeds
 * max_num_blocks_per_seq; when status[1] < status[0] NOTHING has been modified (the reference raises
 * "Out of memory!" before touching anything, block_manager.py:104-106). */
size_t kvc_add_sequence_workspace_bytes(int32_t num_layers, int32_t num_kv_heads, int32_t seq_len,
                                        int32_t block_size, int64_t num_blocks);
int kvc_add_sequence(int32_t* context_lens, int32_t* block_tables, uint8_t* free_mask,
                     int32_t* seq_index_by_block, int32_t* layer_index_by_block,
                     int32_t* head_index_by_block, int32_t* logical_block_num_by_block,
                     int32_t* token_positions, int64_t* slot_mapping, int32_t* status,
                     int32_t num_layers, int32_t max_num_seqs, int32_t num_kv_heads,
                     int32_t max_num_blocks_per_seq, int64_t num_blocks, int32_t block_size,
                     int32_t seq_slot, int32_t seq_len, void* workspace, size_t workspace_bytes,
                     kvc_stream_t stream);

/* ---------------------------------------------------------------------------------
 * F3  single-query paged attention with KV-metric output (decode step)
 * replaces torch.ops._C.kvcompress_paged_attention_v1 / _v2
 *   (csrc/attention/kvcompress_attention_kernels.cu:686-1056, kernels :97-455, :532-651;
 *    vllm/_custom_ops.py:135-201; caller vllm/attention/ops/paged_attn.py:312-407)
 * For every (sequence, query head): p = softmax(scale * q.K^T [+ ALiBi]) over the
 * context_lens[seq, kv_head] cached keys of that KV head (per-head paged cache),
 * out = p.V, and - if record_kv_metrics - kv_metric_out[block, offset, q % qpk] = p for
 * every key with kv_position <= last_position[seq] - kv_metric_buffer_len[seq]; other
 * entries of kv_metric_out are left untouched.  Softmax uses 1/(sum + 1e-6) like the
 * reference.  Floating point: results agree with the reference within its own test
 * tolerance (tests/kernels/test_kvcompress_attention.py:356-362), not bit for bit.
 * Contexts longer than 512 tokens are split into 512-token partitions (the reference's
 * v2); the four partition buffers have the v2 shapes and may be NULL when
 * max_context_len <= 512.
 * Extension (not in the reference, kvcompress/README.md:32,49 lists it as a to-do): with
 * fused_metrics != NULL nothing is written to kv_metric_out; instead, for every key inside the
 * metric window, fused_metrics[slot] += sum over the KV head's query heads of p^2 (or p) --
 * exactly what CompressionMetrics.aggregate_decode (metrics.py:429-439) would add from the
 * stored weights, bit for bit, without the [NB, bs, qpk] round trip.  Needs qpk <= 16.
 * dtype: 0 = fp16, 1 = bf16 (query, output and an "auto" cache).  kv_cache_dtype: 0 = "auto"
 * (cache elements of `dtype`, K vectors of x = 8), 1 = fp8 e4m3fn, 2 = fp8 e5m2 (OCP bytes,
 * K vectors of x = 16; dequantised as dtype(float(fp8) * k_scale / v_scale) like the
 * reference's fp8::scaled_convert, kvcompress_attention_kernels.cu:229-236, 369-377).
 * block_size 16 or 32 (for "auto" caches also 8 with head_size 64 / 128 and 1 with head_size
 * 128 -- the reference's launcher enables head size 128 with block sizes 1 / 8 / 16 / 32,
 * kvcompress_attention_kernels.cu:749-806); head_size 64, 96, 128 or 256 ("auto"), 64 or 128 (fp8).
 * --------------------------------------------------------------------------------- */
typedef struct kvc_attention_params {
  void* out;                            /* [num_seqs, num_heads, head_size] */
  float* kv_metric_out;                 /* [num_blocks, block_size, qpk] */
  float* exp_sums;                      /* [num_seqs, num_heads, max_parts] or NULL */
  float* max_logits;                    /* [num_seqs, num_heads, max_parts] or NULL */
  void* tmp_out;                        /* [num_seqs, num_heads, max_parts, head_size] or NULL */
  float* tmp_kv_metric_out;             /* [num_blocks, block_size, qpk] or NULL */
  float* fused_metrics;                 /* optional (may be NULL): [num_blocks, block_size]; see below */
  const void* query;                    /* [num_seqs, num_heads, head_size], seq stride q_stride */
  const void* key_cache;                /* [num_blocks, head_size/x, block_size, x] */
  const void* value_cache;              /* [num_blocks, head_size, block_size] */
  const int32_t* block_tables;          /* [num_seqs, num_kv_heads, max_num_blocks_per_seq] */
  const int32_t* context_lens;          /* [num_seqs, num_kv_heads] */
  const int32_t* kv_position;           /* [num_blocks, block_size] */
  const int32_t* last_position;         /* [num_seqs] */
  const int32_t* kv_metric_buffer_len;  /* [num_seqs] */
  const float* alibi_slopes;            /* [num_heads] or NULL */
  int64_t q_stride;                     /* elements between sequences of query */
  int64_t kv_block_stride;              /* elements between blocks of the caches */
  float scale, k_scale, v_scale;
  int32_t num_seqs, num_heads, num_kv_heads, head_size, block_size;
  int32_t max_num_blocks_per_seq, max_context_len;
  int32_t dtype, kv_cache_dtype, record_kv_metrics;
  int32_t fused_use_l2;                 /* with fused_metrics: accumulate p^2 (1) or p (0) */
  int32_t schedule;                     /* 0 = automatic, 1 = always partitioned (the reference's v2
                                           shape), 2 = single pass whenever the context fits in LDS */
  /* ABI version 6, with fused_metrics only: harvest in the epilogue (see kvc_attention_harvest_begin).
   * harvest_buf NULL = none; every field below is ignored then. */
  void* harvest_buf;                    /* the buffer kvc_attention_harvest_begin was called with */
  const int32_t* harvest_seq_slot;      /* [num_seqs]: position of this call's sequence in the compression
                                           batch the harvest was begun for, or -1 (not part of it) */
  const int32_t* harvest_seq_positions; /* [harvest_num_seqs]: the schedule call's seq_positions */
  const int32_t* harvest_num_protected; /* [harvest_num_seqs]: the schedule call's num_protected */
  int32_t harvest_num_seqs;             /* sequences of the compression batch */
  int32_t harvest_layer;                /* the layer this call computes */
  int32_t harvest_num_layers;           /* L of the schedule call (num_kv_heads above is its H) */
  int32_t harvest_num_sinks;            /* the schedule call's num_sinks */
  /* ABI version 7 */
  int32_t block_layout;                 /* KVC_LAYOUT_REFERENCE (0) / KVC_LAYOUT_SLOT_MAJOR: how key_cache and
                                           value_cache blocks are laid out inside (block_size 16 or 32 for slot-major) */
} kvc_attention_params;

/* The decode step without a sweep of the metric store (ABI version 6).  With fused_metrics the
 * attention adds a step's weights straight into the store -- no [NB, bs, qpk] round trip, no
 * aggregate_decode -- but the small-eviction schedule that follows still streams the whole store to
 * find the ~1 % of the keys below each sequence's pivot.  The epilogue has every new sum in a
 * register and the key's position next to it: with harvest_buf set it appends the keys of the NEXT
 * compression batch that lie below the pivots the previous kvc_schedule_evictions left there
 * (harvest bit 1) to their heads' candidate lists, as kvc_aggregate_decode_harvest does.  Protocol
 * of one decode step:
 *   kvc_attention_harvest_begin(p, stream)        p = the schedule call that will follow (as for
 *                                                 kvc_aggregate_decode_harvest); clears the counters
 *   kvc_paged_attention_decode x num_layers       one per layer, each with harvest_layer = its layer,
 *                                                 over every sequence of the compression batch
 *   kvc_schedule_evictions(harvest bits 0 | 3 [| 1])  runs records -> selection -> emission on the lists
 * Eligible (kvc_attention_harvest_eligible): calls that take the small-eviction schedule with keys
 * that are the sum alone (no use_average, no bias).  Under the reference's batch > 1 rule (mode 0,
 * more than one sequence) the epilogue also counts every head's masked slots, as the schedule's full
 * collecting pass does (that rule needs every head's count of evictable keys).  Exact or flagged, like every
 * list of that schedule: a layer that was not launched, a sequence of the batch the attention did not
 * see, a metric window that ends before the eviction bound (last_position - kv_metric_buffer_len <
 * seq_position - num_protected) or pivots that were too low leave lists that fall short, and the
 * schedule call redoes the work on the device.  The caller vouches that nothing else writes to the
 * store between begin and the schedule call (the Python binding checks version counters). */
int32_t kvc_attention_harvest_eligible(const kvc_schedule_params* p);
int kvc_attention_harvest_begin(const kvc_schedule_params* p, kvc_stream_t stream);

int kvc_paged_attention_decode(const kvc_attention_params* p, kvc_stream_t stream);
/* 1 if a call with these sizes goes through the partition buffers (exp_sums, max_logits, tmp_out,
 * tmp_kv_metric_out), 0 if it finishes in one kernel and they may be NULL. */
int32_t kvc_paged_attention_decode_uses_partitions(int32_t num_seqs, int32_t num_heads,
                                                   int32_t num_kv_heads, int32_t head_size,
                                                   int32_t max_context_len, int32_t schedule);
/* The same question for a cache of the given in-block layout (KVC_LAYOUT_*): slot-major blocks keep one K tile
 * per wave in the LDS, so the longest context that finishes in one kernel is slightly shorter.  The function above
 * answers for KVC_LAYOUT_REFERENCE. */
int32_t kvc_paged_attention_decode_uses_partitions_in(int32_t num_seqs, int32_t num_heads,
                                                      int32_t num_kv_heads, int32_t head_size,
                                                      int32_t max_context_len, int32_t schedule,
                                                      int32_t block_layout);

#ifdef __cplusplus
}
#endif
#endif /* KVC_MI355X_H */
