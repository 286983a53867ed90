// Decode paged attention with KV-metric output: the C-ABI entry points and the shape dispatch.
// The kernels live in kvc_attention_kernels.h and are instantiated in kvc_attention_inst_*.hip.
#include "kvc_attention_kernels.h"

namespace kvc {
#define KVC_X(HD, BS)                                                                          \
  extern template int launch_attention<_Float16, HD, BS, 0>(const AttnArgs&, int, hipStream_t); \
  extern template int launch_attention<__bf16, HD, BS, 0>(const AttnArgs&, int, hipStream_t);
KVC_ATT_SHAPES(KVC_X)
#undef KVC_X
#define KVC_X(HD, BS)                                                                          \
  extern template int launch_attention<_Float16, HD, BS, 1>(const AttnArgs&, int, hipStream_t); \
  extern template int launch_attention<_Float16, HD, BS, 2>(const AttnArgs&, int, hipStream_t); \
  extern template int launch_attention<_Float16, HD, BS, 3>(const AttnArgs&, int, hipStream_t); \
  extern template int launch_attention<_Float16, HD, BS, 4>(const AttnArgs&, int, hipStream_t); \
  extern template int launch_attention<__bf16, HD, BS, 1>(const AttnArgs&, int, hipStream_t);   \
  extern template int launch_attention<__bf16, HD, BS, 2>(const AttnArgs&, int, hipStream_t);   \
  extern template int launch_attention<__bf16, HD, BS, 3>(const AttnArgs&, int, hipStream_t);   \
  extern template int launch_attention<__bf16, HD, BS, 4>(const AttnArgs&, int, hipStream_t);
KVC_ATT_F8_SHAPES(KVC_X)
#undef KVC_X
}  // namespace kvc

extern "C" int32_t kvc_paged_attention_decode_uses_partitions(int32_t num_seqs, int32_t num_heads,
                                                              int32_t num_kv_heads, int32_t head_size,
                                                              int32_t max_context_len, int32_t schedule) {
  if (num_kv_heads < 1 || num_heads < num_kv_heads || max_context_len <= kvc::ATT_PART) return 0;
  return kvc::attention_plan(num_seqs, num_heads, num_kv_heads, head_size, max_context_len, schedule).whole ? 0 : 1;
}

extern "C" int32_t kvc_paged_attention_decode_uses_partitions_in(int32_t num_seqs, int32_t num_heads,
                                                                 int32_t num_kv_heads, int32_t head_size,
                                                                 int32_t max_context_len, int32_t schedule,
                                                                 int32_t block_layout) {
  if (num_kv_heads < 1 || num_heads < num_kv_heads || max_context_len <= kvc::ATT_PART) return 0;
  return kvc::attention_plan(num_seqs, num_heads, num_kv_heads, head_size, max_context_len, schedule,
                             kvc::attention_stage_bytes(head_size, block_layout)).whole ? 0 : 1;
}

extern "C" int kvc_paged_attention_decode(const kvc_attention_params* p, kvc_stream_t stream) {
  using namespace kvc;
  if (p == nullptr) return fail_invalid("paged_attention_decode: null params");
  if (p->num_seqs <= 0) return KVC_OK;
  if (p->num_kv_heads < 1 || p->num_heads % p->num_kv_heads != 0)
    return fail_invalid("paged_attention_decode: num_heads must be a multiple of num_kv_heads");
  if (p->kv_cache_dtype < 0 || p->kv_cache_dtype > 2)
    return fail_invalid("Unsupported data type of kv cache: " + std::to_string(p->kv_cache_dtype));
  if (p->dtype != 0 && p->dtype != 1) return fail_invalid("Unsupported data type of query");
  AttnArgs a;
  a.out = p->out; a.kv_metric_out = p->kv_metric_out; a.exp_sums = p->exp_sums;
  a.max_logits = p->max_logits; a.tmp_out = p->tmp_out; a.tmp_kv_metric_out = p->tmp_kv_metric_out;
  a.fused_metrics = p->fused_metrics; a.use_l2 = p->fused_use_l2 ? 1 : 0;
  if (a.fused_metrics != nullptr && p->num_heads / p->num_kv_heads > ATT_NQ)
    return fail_invalid("paged_attention_decode: fused metric aggregation needs <= 16 query heads per KV head");
  a.q = p->query; a.k_cache = p->key_cache; a.v_cache = p->value_cache;
  a.block_tables = p->block_tables; a.context_lens = p->context_lens; a.kv_position = p->kv_position;
  a.last_position = p->last_position; a.kv_metric_buffer_len = p->kv_metric_buffer_len;
  a.alibi_slopes = p->alibi_slopes; a.q_stride = p->q_stride; a.kv_block_stride = p->kv_block_stride;
  a.scale = p->scale; a.k_scale = p->k_scale; a.v_scale = p->v_scale; a.num_heads = p->num_heads; a.num_kv_heads = p->num_kv_heads;
  a.max_blocks = p->max_num_blocks_per_seq; a.record = p->record_kv_metrics ? 1 : 0;
  a.max_ctx = p->max_context_len > 0 ? p->max_context_len : 1;
  a.schedule = p->schedule;
  if (p->block_layout != KVC_LAYOUT_REFERENCE && p->block_layout != KVC_LAYOUT_SLOT_MAJOR)
    return fail_invalid("Unsupported block layout: " + std::to_string(p->block_layout));
  a.layout = p->block_layout;
  a.hv = AttnHarvest{};
  if (p->harvest_buf != nullptr) {
    if (a.fused_metrics == nullptr) return fail_invalid("paged_attention_decode: harvest_buf needs fused_metrics");
    if (p->harvest_seq_slot == nullptr || p->harvest_seq_positions == nullptr || p->harvest_num_protected == nullptr ||
        p->harvest_num_seqs < 1 || p->harvest_num_layers < 1 || p->harvest_layer < 0 || p->harvest_layer >= p->harvest_num_layers ||
        (reinterpret_cast<uintptr_t>(p->harvest_buf) & 15) != 0)
      return fail_invalid("paged_attention_decode: bad harvest arguments");
    if (p->block_size != 8 && p->block_size != 16 && p->block_size != 32)
      return fail_invalid("paged_attention_decode: harvest needs block size 8, 16 or 32 (the small-eviction schedule's)");
    const int G = p->harvest_num_seqs * p->harvest_num_layers * p->num_kv_heads;
    const HvLayout hl = hv_layout(G, p->harvest_num_seqs);
    uint8_t* hb = reinterpret_cast<uint8_t*>(p->harvest_buf);
    a.hv.pivot = reinterpret_cast<const uint32_t*>(hb + hl.pivot);
    a.hv.cnt = reinterpret_cast<uint32_t*>(hb + hl.cnt);
    a.hv.def = reinterpret_cast<uint32_t*>(hb + hl.def);
    a.hv.lists = reinterpret_cast<unsigned long long*>(hb + hl.rec64);
    a.hv.claimed = reinterpret_cast<uint32_t*>(hb + hl.claimed);
    a.hv.seen_ctx = reinterpret_cast<int32_t*>(hb + hl.seen_ctx);
    a.hv.seq_slot = p->harvest_seq_slot;
    a.hv.seq_positions = p->harvest_seq_positions;
    a.hv.num_protected = p->harvest_num_protected;
    a.hv.layer = p->harvest_layer; a.hv.num_layers = p->harvest_num_layers; a.hv.num_sinks = p->harvest_num_sinks;
  }
  a.max_parts = (p->max_context_len + ATT_PART - 1) / ATT_PART;
  if (a.max_parts < 1) a.max_parts = 1;
  hipStream_t s = (hipStream_t)stream;
  const int combo = p->head_size * 100 + p->block_size;
#define KVC_ATT_T(HD, BS, KVD)                                                             \
  (p->dtype == 0 ? launch_attention<_Float16, HD, BS, KVD>(a, p->num_seqs, s)             \
                 : launch_attention<__bf16, HD, BS, KVD>(a, p->num_seqs, s))
#define KVC_ATT(HD, BS) KVC_ATT_T(HD, BS, 0)
  if (p->kv_cache_dtype != 0) {                  // fp8 caches: head sizes 64 and 128
    // 1 / 2: e4m3 / e5m2 with scales; 3 / 4: both scales 1 (cheaper dequantisation)
    const int kvd = p->kv_cache_dtype + ((p->k_scale == 1.0f && p->v_scale == 1.0f) ? 2 : 0);
#define KVC_ATT_F8(HD, BS)                                                                 \
  (kvd == 1 ? KVC_ATT_T(HD, BS, 1) : kvd == 2 ? KVC_ATT_T(HD, BS, 2)                       \
            : kvd == 3 ? KVC_ATT_T(HD, BS, 3) : KVC_ATT_T(HD, BS, 4))
    switch (combo) {
      case 6416: return KVC_ATT_F8(64, 16);
      case 6432: return KVC_ATT_F8(64, 32);
      case 12816: return KVC_ATT_F8(128, 16);
      case 12832: return KVC_ATT_F8(128, 32);
      default:
        return fail_invalid("paged_attention_decode: fp8 caches support head sizes 64 and 128 with "
                            "block sizes 16 and 32");
    }
  }
  switch (combo) {
    case 12801: return KVC_ATT(128, 1);
    case 6408: return KVC_ATT(64, 8);
    case 12808: return KVC_ATT(128, 8);
    case 6416: return KVC_ATT(64, 16);
    case 6432: return KVC_ATT(64, 32);
    case 9616: return KVC_ATT(96, 16);
    case 9632: return KVC_ATT(96, 32);
    case 12816: return KVC_ATT(128, 16);
    case 12832: return KVC_ATT(128, 32);
    case 25616: return KVC_ATT(256, 16);
    case 25632: return KVC_ATT(256, 32);
    default: break;
  }
#undef KVC_ATT_F8
#undef KVC_ATT_T
#undef KVC_ATT
  if (p->block_size != 1 && p->block_size != 8 && p->block_size != 16 && p->block_size != 32)
    return fail_invalid("Unsupported block size: " + std::to_string(p->block_size));
  return fail_invalid("Unsupported head size: " + std::to_string(p->head_size));
}
