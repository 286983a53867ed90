// libkvc_torch.so -- the dispatcher binding of the drop-in boundary, compiled.
//
// The fork calls torch.ops._C_kvc_ops.{count_block_evictions, schedule_t1_cache_moves,
// execute_cache_moves}, torch.ops._C_cache_ops.kvcompress_reshape_and_cache and
// torch.ops._C.kvcompress_paged_attention_v1/_v2 (vllm/_custom_ops.py:1074, 1169, 1247, 649, 156,
// 192); its own extension registers them from C++ (csrc/torch_bindings.cpp:52-80, 353-362,
// 395-418).  This file does the same for MI355X: the schemas of those lines, and HIP-key kernels
// that unpack the tensors and call the C ABI of libkvc_mi355x.so (include/kvc_mi355x.h) on
// torch's current stream -- no Python frame between the dispatcher and the launch.
//
// Host code only (no device code here); built by csrc/build.sh, loaded with
// torch.ops.load_library by vllm_kvcompress_amd.torch_ops.register().
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <optional>
#include <string>
#include <tuple>

#include "../../include/kvc_mi355x.h"

namespace {

using at::Tensor;

void* current_stream(const Tensor& t) {
  return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream();
}

void check(int rc) {
  if (rc != 0) {
    const char* msg = kvc_last_error();
    TORCH_CHECK(false, (msg && *msg) ? msg : "kvc error");
  }
}

void require(const Tensor& t, const char* name) {
  TORCH_CHECK(t.defined() && t.is_cuda(), name, ": expected a tensor on a HIP device (no CPU fallback exists)");
}
void require(const Tensor& t, const char* name, at::ScalarType dt) {
  require(t, name);
  TORCH_CHECK(t.scalar_type() == dt, name, ": expected dtype ", dt, ", got ", t.scalar_type());
}

// The kernels write through raw pointers: neither torch nor the dispatcher counts such a write (a `Tensor!` in a
// schema is an annotation, not a hook).  Consumers that ask "has anybody written to this tensor since?" --
// CompressionMetrics' harvested lists and remembered pivots, the tracked move table, the plan a move list brings
// along -- must get the answer an in-place torch op would give them, whichever binding did the write: every kernel
// here bumps the version counter of every tensor its launch writes (the Python wrappers do the same,
// _custom_ops._written).  Tensors made under torch.inference_mode() keep no counter (and every such question
// about them is answered with "no").
void written(const Tensor& t) {
  if (t.defined() && !t.is_inference()) t.unsafeGetTensorImpl()->bump_version();
}

// persistent per-(device, stream, tag) scratch, grown geometrically; the C ABI never allocates
Tensor workspace(const Tensor& like, size_t nbytes, const char* tag) {
  static std::mutex mu;
  static std::map<std::tuple<int, void*, std::string>, Tensor> cache;
  std::lock_guard<std::mutex> g(mu);
  auto key = std::make_tuple((int)like.get_device(), current_stream(like), std::string(tag));
  auto it = cache.find(key);
  if (it == cache.end() || (size_t)it->second.numel() < nbytes) {
    const int64_t n = std::max<int64_t>((int64_t)(nbytes + nbytes / 4), 4096);
    Tensor buf = at::empty({n}, like.options().dtype(at::kByte));
    cache[key] = buf;
    return buf;
  }
  return it->second;
}

// kvc_attention_params.schedule of the attention ops registered here (0 = automatic); set through
// _kvc_mi355x::set_attention_schedule, like vllm_kvcompress_amd._custom_ops.set_attention_schedule
// does for the Python-registered ops
std::atomic<int> g_attention_schedule{0};

// KVC_LAYOUT_* of every cache op registered here (include/kvc_mi355x.h, ABI version 7): the environment's
// KVC_BLOCK_LAYOUT at load time, or _kvc_mi355x::set_block_layout (what vllm_kvcompress_amd.set_block_layout calls)
static int layout_from_env() {
  const char* e = std::getenv("KVC_BLOCK_LAYOUT");
  if (e == nullptr || *e == 0 || std::string(e) == "reference") return KVC_LAYOUT_REFERENCE;
  if (std::string(e) == "slot_major") return KVC_LAYOUT_SLOT_MAJOR;
  TORCH_CHECK(false, "KVC_BLOCK_LAYOUT=", e, ": expected reference or slot_major");
}
std::atomic<int> g_block_layout{layout_from_env()};
void set_block_layout_op(int64_t layout) {
  TORCH_CHECK(layout == KVC_LAYOUT_REFERENCE || layout == KVC_LAYOUT_SLOT_MAJOR, "block layout must be 0 (reference) or 1 (slot-major)");
  g_block_layout.store((int)layout);
}
int64_t block_layout_op() { return g_block_layout.load(); }

// start-up hooks of this binding (vllm_kvcompress_amd._custom_ops.reserve_workspace /
// reserve_attention_scratch / set_attention_schedule call them when this library is the one
// registered): size a scratch buffer of THIS binding's cache ahead of time, so that its ops never
// allocate while serving or inside a HIP-graph capture
void reserve_workspace_op(const Tensor& like, int64_t nbytes, std::string tag) {
  require(like, "like");
  c10::DeviceGuard guard(like.device());
  (void)workspace(like, (size_t)nbytes, tag.c_str());
}
void set_attention_schedule_op(int64_t schedule) {
  TORCH_CHECK(schedule >= 0 && schedule <= 2, "attention schedule must be 0, 1 or 2");
  g_attention_schedule.store((int)schedule);
}

// ------------------------------------------------------------------ _C_kvc_ops
void count_block_evictions(Tensor& evicted_block_count, Tensor& evicted_logical_indices,
                           const Tensor& evicted_kv_offsets, const Tensor& hanging_token_count,
                           int64_t block_size, int64_t null_value) {
  require(evicted_block_count, "evicted_block_count", at::kInt);
  require(evicted_logical_indices, "evicted_logical_indices", at::kInt);
  require(evicted_kv_offsets, "evicted_kv_offsets", at::kInt);
  require(hanging_token_count, "hanging_token_count", at::kInt);
  TORCH_CHECK(evicted_logical_indices.is_contiguous() && evicted_block_count.is_contiguous(),
              "count_block_evictions: evicted_logical_indices / evicted_block_count must be contiguous "
              "(written in place)");
  const Tensor offs = evicted_kv_offsets.contiguous(), hang = hanging_token_count.contiguous();
  c10::DeviceGuard guard(evicted_logical_indices.device());
  check(kvc_count_block_evictions(evicted_block_count.data_ptr<int32_t>(),
                                  evicted_logical_indices.data_ptr<int32_t>(), offs.data_ptr<int32_t>(),
                                  hang.data_ptr<int32_t>(), (int32_t)evicted_block_count.numel(),
                                  evicted_logical_indices.numel(), (int32_t)block_size, (int32_t)null_value,
                                  current_stream(evicted_logical_indices)));
  written(evicted_block_count);
  written(evicted_logical_indices);
}

// The move list schedule_t1_cache_moves made last on a (device, stream) brings its plan along
// (kvc_schedule_t1_cache_moves_ex / kvc_execute_cache_moves_planned, include/kvc_mi355x.h):
// execute_cache_moves of exactly those three tensors, untouched since, is then ONE launch.  "Exactly
// those, untouched": the same TensorImpl objects (held weakly: an address that was freed and handed
// out again does not pass) at the same version counters -- anything that wrote to them through
// torch has bumped those.  Whatever does not pass takes the self-contained op.
struct PlanRecord {
  Tensor plan;
  c10::weak_intrusive_ptr<c10::TensorImpl> who[3] = {
      c10::weak_intrusive_ptr<c10::TensorImpl>(c10::intrusive_ptr<c10::TensorImpl>()),
      c10::weak_intrusive_ptr<c10::TensorImpl>(c10::intrusive_ptr<c10::TensorImpl>()),
      c10::weak_intrusive_ptr<c10::TensorImpl>(c10::intrusive_ptr<c10::TensorImpl>())};
  int64_t version[3] = {-1, -1, -1};
  int32_t total_heads = 0, block_size = 0;
};
std::mutex g_plan_mu;
std::map<std::pair<int, void*>, PlanRecord> g_plans;
std::atomic<int64_t> g_planned_compactions{0};

bool version_of(const Tensor& t, int64_t& v) {
  try { v = (int64_t)t._version(); return true; } catch (const c10::Error&) { return false; }
}

void remember_plan(const Tensor& plan, const Tensor* const (&t)[3], int32_t total_heads, int32_t block_size) {
  PlanRecord r;
  r.plan = plan;
  for (int i = 0; i < 3; ++i) {
    if (!version_of(*t[i], r.version[i])) return;            // (no version counter: nothing to vouch with)
    r.who[i] = c10::weak_intrusive_ptr<c10::TensorImpl>(t[i]->getIntrusivePtr());
  }
  r.total_heads = total_heads; r.block_size = block_size;
  std::lock_guard<std::mutex> g(g_plan_mu);
  g_plans.insert_or_assign(std::make_pair((int)plan.get_device(), current_stream(plan)), std::move(r));
}

// the plan that vouches for (cmi, cmc, offs) on the current stream, or an undefined tensor
Tensor plan_of(const Tensor& like, const Tensor* const (&t)[3], int32_t total_heads, int32_t block_size) {
  std::lock_guard<std::mutex> g(g_plan_mu);
  auto it = g_plans.find(std::make_pair((int)like.get_device(), current_stream(like)));
  if (it == g_plans.end()) return Tensor();
  const PlanRecord& r = it->second;
  if (r.total_heads != total_heads || r.block_size != block_size) return Tensor();
  for (int i = 0; i < 3; ++i) {
    int64_t v = -1;
    if (!version_of(*t[i], v) || v != r.version[i]) return Tensor();
    auto alive = r.who[i].lock();
    if (!alive || alive.get() != t[i]->unsafeGetTensorImpl()) return Tensor();
  }
  return r.plan;
}
int64_t planned_compactions_op() { return g_planned_compactions.load(); }

// the bare op: rows that hold no move are left untouched (the Python wrapper of the reference
// zero-fills the workspace in front of it, vllm/_custom_ops.py:1168)
void schedule_t1_cache_moves(Tensor& cache_moves_idx, Tensor& cache_moves_count,
                             const Tensor& evicted_logical_indices, const Tensor& evicted_kv_count,
                             const Tensor& evicted_kv_offsets, const Tensor& block_tables,
                             const Tensor& context_lens, int64_t block_size) {
  for (auto p : {std::make_pair(&cache_moves_idx, "cache_moves_idx"), std::make_pair(&cache_moves_count, "cache_moves_count")})
    require(*p.first, p.second, at::kInt);
  for (auto p : {std::make_pair(&evicted_logical_indices, "evicted_logical_indices"),
                 std::make_pair(&evicted_kv_count, "evicted_kv_count"),
                 std::make_pair(&evicted_kv_offsets, "evicted_kv_offsets"),
                 std::make_pair(&block_tables, "block_tables"), std::make_pair(&context_lens, "context_lens")})
    require(*p.first, p.second, at::kInt);
  TORCH_CHECK(cache_moves_idx.is_contiguous() && cache_moves_count.is_contiguous(),
              "schedule_cache_moves: output tensors must be contiguous");
  TORCH_CHECK(evicted_kv_count.dim() == 3 && block_tables.dim() == 4, "schedule_cache_moves: bad shapes");
  const Tensor eli = evicted_logical_indices.contiguous(), ekc = evicted_kv_count.contiguous(),
               offs = evicted_kv_offsets.contiguous(), bt = block_tables.contiguous(),
               ctx = context_lens.contiguous();
  c10::DeviceGuard guard(cache_moves_idx.device());
  Tensor plan = workspace(cache_moves_idx, kvc_cache_moves_plan_bytes(), "cache_moves_plan");
  check(kvc_schedule_t1_cache_moves_ex(cache_moves_idx.data_ptr<int32_t>(), cache_moves_idx.size(0),
                                       cache_moves_count.data_ptr<int32_t>(), eli.data_ptr<int32_t>(),
                                       ekc.data_ptr<int32_t>(), offs.data_ptr<int32_t>(), bt.data_ptr<int32_t>(),
                                       ctx.data_ptr<int32_t>(), (int32_t)ekc.size(0), (int32_t)ekc.size(1),
                                       (int32_t)ekc.size(2), (int32_t)bt.size(3), (int32_t)block_size, 0, nullptr, 0,
                                       reinterpret_cast<int32_t*>(plan.data_ptr()), current_stream(cache_moves_idx)));
  written(cache_moves_idx);               // (before the plan remembers the versions it vouches for)
  written(cache_moves_count);
  const Tensor* const who[3] = {&cache_moves_idx, &cache_moves_count, &evicted_kv_offsets};
  remember_plan(plan, who, (int32_t)ekc.numel(), (int32_t)block_size);
}

void execute_cache_moves(Tensor& k_cache, Tensor& v_cache, Tensor& kv_metrics, Tensor& kv_position,
                         const Tensor& cache_moves_idx, const Tensor& cache_moves_count,
                         const Tensor& evicted_kv_offsets, int64_t /*blocks_per_head*/,
                         int64_t /*threads_per_head*/) {
  require(k_cache, "k_cache");
  require(v_cache, "v_cache");
  require(kv_metrics, "kv_metrics", at::kFloat);
  require(kv_position, "kv_position", at::kInt);
  require(cache_moves_idx, "cache_moves_indices", at::kInt);
  require(cache_moves_count, "cache_moves_count", at::kInt);
  require(evicted_kv_offsets, "evicted_kv_offsets", at::kInt);
  TORCH_CHECK(k_cache.dim() == 4 && v_cache.dim() == 3,
              "execute_cache_moves: k_cache must be [NB, hd/x, bs, x] and v_cache [NB, hd, bs]");
  TORCH_CHECK(k_cache.is_contiguous() && v_cache.is_contiguous() && kv_metrics.is_contiguous() &&
                  kv_position.is_contiguous(),
              "execute_cache_moves: caches, kv_metrics and kv_position must be contiguous (mutated in place)");
  const Tensor cmi = cache_moves_idx.contiguous(), cmc = cache_moves_count.contiguous(),
               offs = evicted_kv_offsets.contiguous();
  const int64_t num_blocks = v_cache.size(0), head_size = v_cache.size(1), block_size = v_cache.size(2);
  const int32_t total_heads = (int32_t)cmc.numel();
  c10::DeviceGuard guard(k_cache.device());
  const Tensor* const who[3] = {&cache_moves_idx, &cache_moves_count, &evicted_kv_offsets};
  const Tensor plan = plan_of(k_cache, who, total_heads, (int32_t)block_size);
  written(k_cache);
  written(v_cache);
  written(kv_metrics);
  written(kv_position);
  if (g_block_layout.load() == KVC_LAYOUT_SLOT_MAJOR) {
    // a slot is two contiguous runs: the moved bytes and nothing else; the list's own plan or one small launch
    Tensor ws = workspace(k_cache, kvc_cache_moves_plan_bytes(), "execute_cache_moves_slot_major");
    check(kvc_execute_cache_moves_slot_major(k_cache.data_ptr(), v_cache.data_ptr(), kv_metrics.data_ptr<float>(),
                                             kv_position.data_ptr<int32_t>(), cmi.data_ptr<int32_t>(),
                                             cmc.data_ptr<int32_t>(), offs.data_ptr<int32_t>(), total_heads, num_blocks,
                                             (int32_t)block_size, (int32_t)head_size, (int32_t)k_cache.element_size(),
                                             plan.defined() ? reinterpret_cast<const int32_t*>(plan.data_ptr()) : nullptr,
                                             ws.data_ptr(), (size_t)ws.numel(), current_stream(k_cache)));
    if (plan.defined()) g_planned_compactions.fetch_add(1);
    return;
  }
  if (plan.defined()) {
    check(kvc_execute_cache_moves_planned(k_cache.data_ptr(), v_cache.data_ptr(), kv_metrics.data_ptr<float>(),
                                          kv_position.data_ptr<int32_t>(), cmi.data_ptr<int32_t>(),
                                          cmc.data_ptr<int32_t>(), offs.data_ptr<int32_t>(), total_heads, num_blocks,
                                          (int32_t)block_size, (int32_t)head_size, (int32_t)k_cache.element_size(),
                                          (int32_t)k_cache.size(3), reinterpret_cast<const int32_t*>(plan.data_ptr()),
                                          current_stream(k_cache)));
    g_planned_compactions.fetch_add(1);
    return;
  }
  const size_t ws_bytes = kvc_execute_cache_moves_workspace_bytes(total_heads, num_blocks);
  Tensor ws = workspace(k_cache, ws_bytes, "execute_cache_moves");
  check(kvc_execute_cache_moves(k_cache.data_ptr(), v_cache.data_ptr(), kv_metrics.data_ptr<float>(),
                                kv_position.data_ptr<int32_t>(), cmi.data_ptr<int32_t>(),
                                cmc.data_ptr<int32_t>(), offs.data_ptr<int32_t>(), total_heads, num_blocks,
                                (int32_t)block_size, (int32_t)head_size, (int32_t)k_cache.element_size(),
                                (int32_t)k_cache.size(3), ws.data_ptr(), (size_t)ws.numel(),
                                current_stream(k_cache)));
}

// ------------------------------------------------------------------ _C_cache_ops
void kvcompress_reshape_and_cache(const Tensor& key, const Tensor& value, Tensor& key_cache,
                                  Tensor& value_cache, const Tensor& kv_metrics, const Tensor& slot_mapping,
                                  const Tensor& kv_metric_head_bias, std::string kv_cache_dtype,
                                  double k_scale, double v_scale) {
  require(key, "key");
  require(value, "value");
  require(key_cache, "key_cache");
  require(value_cache, "value_cache");
  require(kv_metrics, "kv_metrics", at::kFloat);
  require(slot_mapping, "slot_mapping", at::kLong);
  require(kv_metric_head_bias, "kv_metric_head_bias", at::kFloat);
  TORCH_CHECK(key.dim() == 3 && key.stride(2) == 1 && key.stride(1) == key.size(2) && value.stride(2) == 1 &&
                  value.stride(1) == value.size(2),
              "reshape_and_cache_kvc: key/value must be dense in their last two dims");
  const int64_t num_tokens = key.size(0), num_heads = key.size(1), head_size = key.size(2);
  const int64_t block_size = key_cache.size(2);
  const Tensor sm = slot_mapping.contiguous(), hb = kv_metric_head_bias.contiguous();
  c10::DeviceGuard guard(key.device());
  // (the fork's schema does not mark kv_metrics mutable, csrc/torch_bindings.cpp:353-360; the kernel sets
  // kv_metrics[slot] = bias all the same, csrc/kvcompress_cache_kernels.cu:27-89)
  float* met = const_cast<float*>(kv_metrics.data_ptr<float>());
  written(key_cache);
  written(value_cache);
  written(kv_metrics);
  if (kv_cache_dtype == "auto") {
    TORCH_CHECK(key.scalar_type() == key_cache.scalar_type() && value.scalar_type() == value_cache.scalar_type(),
                "reshape_and_cache_kvc: kv_cache_dtype 'auto' needs cache dtype == key/value dtype");
    check(kvc_reshape_and_cache_layout(key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
                                       met, sm.data_ptr<int64_t>(), hb.data_ptr<float>(), num_tokens,
                                       (int32_t)num_heads, (int32_t)head_size, (int32_t)block_size,
                                       (int32_t)key.element_size(), key.stride(0), value.stride(0),
                                       g_block_layout.load(), current_stream(key)));
    return;
  }
  int kind = 0;
  if (kv_cache_dtype == "fp8" || kv_cache_dtype == "fp8_e4m3") kind = 0;
  else if (kv_cache_dtype == "fp8_e5m2") kind = 1;
  else { TORCH_CHECK(false, "Unsupported data type of kv cache: ", kv_cache_dtype); }
  int src = 0;
  if (key.scalar_type() == at::kHalf) src = 0;
  else if (key.scalar_type() == at::kBFloat16) src = 1;
  else if (key.scalar_type() == at::kFloat) src = 2;
  else { TORCH_CHECK(false, "Unsupported input type of kv cache: ", key.scalar_type()); }
  TORCH_CHECK(value.scalar_type() == key.scalar_type(), "Unsupported input type of kv cache: ", value.scalar_type());
  TORCH_CHECK(key_cache.element_size() == 1 && value_cache.element_size() == 1,
              "reshape_and_cache_kvc: an fp8 kv cache must have 1-byte elements");
  check(kvc_reshape_and_cache_fp8_layout(key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
                                         met, sm.data_ptr<int64_t>(), hb.data_ptr<float>(), num_tokens,
                                         (int32_t)num_heads, (int32_t)head_size, (int32_t)block_size, src, kind,
                                         key.stride(0), value.stride(0), (float)k_scale, (float)v_scale,
                                         g_block_layout.load(), current_stream(key)));
}

// ------------------------------------------------------------------ _C (decode attention)
void paged_attention_kvc(Tensor& out, Tensor& kv_metric_out, const Tensor* exp_sums, const Tensor* max_logits,
                         const Tensor* tmp_out, const Tensor* tmp_kv_metric_out, const Tensor& query,
                         const Tensor& key_cache, const Tensor& value_cache, int64_t num_kv_heads, double scale,
                         const Tensor& block_tables, const Tensor& context_lens, const Tensor& kv_position,
                         const Tensor& last_position, const Tensor& kv_metric_buffer_len, int64_t block_size,
                         int64_t max_context_len, const std::optional<Tensor>& alibi_slopes,
                         const std::string& kv_cache_dtype, double k_scale, double v_scale,
                         bool record_kv_metrics) {
  require(out, "out");
  require(query, "query");
  require(key_cache, "key_cache");
  require(value_cache, "value_cache");
  require(kv_metric_out, "kv_metric_out", at::kFloat);
  for (auto p : {std::make_pair(&block_tables, "block_tables"), std::make_pair(&context_lens, "context_lens"),
                 std::make_pair(&kv_position, "kv_position"), std::make_pair(&last_position, "last_position"),
                 std::make_pair(&kv_metric_buffer_len, "kv_metric_buffer_len")})
    require(*p.first, p.second, at::kInt);
  int kvd = 0;
  if (kv_cache_dtype == "auto") kvd = 0;
  else if (kv_cache_dtype == "fp8" || kv_cache_dtype == "fp8_e4m3") kvd = 1;
  else if (kv_cache_dtype == "fp8_e5m2") kvd = 2;
  else { TORCH_CHECK(false, "Unsupported data type of kv cache: ", kv_cache_dtype); }
  int dt = 0;
  if (query.scalar_type() == at::kHalf) dt = 0;
  else if (query.scalar_type() == at::kBFloat16) dt = 1;
  else { TORCH_CHECK(false, "Unsupported data type: ", query.scalar_type()); }
  TORCH_CHECK(out.scalar_type() == query.scalar_type(), "paged_attention_kvc: query and output must share a dtype");
  if (kvd == 0) {
    TORCH_CHECK(key_cache.scalar_type() == query.scalar_type() && value_cache.scalar_type() == query.scalar_type(),
                "paged_attention_kvc: an \"auto\" cache has the query's dtype");
  } else {
    TORCH_CHECK(key_cache.element_size() == 1 && value_cache.element_size() == 1,
                "paged_attention_kvc: an fp8 kv cache must have 1-byte elements");
  }
  TORCH_CHECK(query.dim() == 3 && query.stride(1) == query.size(2) && query.stride(2) == 1,
              "paged_attention_kvc: query must be contiguous in (head, dim)");
  TORCH_CHECK(out.is_contiguous(), "paged_attention_kvc: out must be contiguous");
  const Tensor bt = block_tables.contiguous(), ctx = context_lens.contiguous(), pos = kv_position.contiguous(),
               last = last_position.contiguous(), buf = kv_metric_buffer_len.contiguous();
  Tensor alibi;
  if (alibi_slopes.has_value() && alibi_slopes->defined()) alibi = alibi_slopes->to(at::kFloat).contiguous();
  kvc_attention_params p;
  memset(&p, 0, sizeof(p));
  p.out = out.data_ptr();
  p.kv_metric_out = kv_metric_out.data_ptr<float>();
  p.query = query.data_ptr();
  p.key_cache = key_cache.data_ptr();
  p.value_cache = value_cache.data_ptr();
  p.block_tables = bt.data_ptr<int32_t>();
  p.context_lens = ctx.data_ptr<int32_t>();
  p.kv_position = pos.data_ptr<int32_t>();
  p.last_position = last.data_ptr<int32_t>();
  p.kv_metric_buffer_len = buf.data_ptr<int32_t>();
  p.alibi_slopes = alibi.defined() ? alibi.data_ptr<float>() : nullptr;
  p.q_stride = query.stride(0);
  p.kv_block_stride = key_cache.stride(0);
  p.scale = (float)scale; p.k_scale = (float)k_scale; p.v_scale = (float)v_scale;
  p.num_seqs = (int32_t)query.size(0); p.num_heads = (int32_t)query.size(1); p.num_kv_heads = (int32_t)num_kv_heads;
  p.head_size = (int32_t)query.size(2); p.block_size = (int32_t)block_size;
  p.max_num_blocks_per_seq = (int32_t)bt.size(-1);
  p.max_context_len = (int32_t)max_context_len;
  p.dtype = dt; p.kv_cache_dtype = kvd; p.record_kv_metrics = record_kv_metrics ? 1 : 0;
  p.schedule = g_attention_schedule.load();
  p.block_layout = g_block_layout.load();
  c10::DeviceGuard guard(query.device());
  Tensor scratch;                                   // v1: the small partition buffers the signature does not carry
  if (exp_sums != nullptr) {
    require(*exp_sums, "exp_sum", at::kFloat);
    require(*max_logits, "max_logits", at::kFloat);
    p.exp_sums = exp_sums->data_ptr<float>();
    p.max_logits = max_logits->data_ptr<float>();
    p.tmp_out = tmp_out->data_ptr();
    p.tmp_kv_metric_out = tmp_kv_metric_out->data_ptr<float>();
  } else if (kvc_paged_attention_decode_uses_partitions_in(p.num_seqs, p.num_heads, p.num_kv_heads, p.head_size,
                                                           p.max_context_len, p.schedule, p.block_layout)) {
    const int64_t parts = (max_context_len + 511) / 512;
    const int64_t n = (int64_t)p.num_seqs * p.num_heads * parts;
    const size_t nbytes = (size_t)n * 8 + (size_t)n * p.head_size * query.element_size() + 256;
    scratch = workspace(query, nbytes, "attn_v1");
    uint8_t* b = scratch.data_ptr<uint8_t>();
    p.exp_sums = reinterpret_cast<float*>(b);
    p.max_logits = reinterpret_cast<float*>(b + n * 4);
    p.tmp_out = b + n * 8;
    // the unnormalised per-partition weights are rescaled in place (same slot * qpk + q element)
    p.tmp_kv_metric_out = record_kv_metrics ? kv_metric_out.data_ptr<float>() : nullptr;
  }
  check(kvc_paged_attention_decode(&p, current_stream(query)));
  written(out);
  if (record_kv_metrics) written(kv_metric_out);
}

void kvcompress_paged_attention_v1(Tensor& out, Tensor& kv_metric_out, const Tensor& query, const Tensor& key_cache,
                                   const Tensor& value_cache, int64_t num_kv_heads, double scale,
                                   const Tensor& block_tables, const Tensor& context_lens, const Tensor& kv_position,
                                   const Tensor& last_position, const Tensor& kv_metric_buffer_len,
                                   int64_t block_size, int64_t max_context_len,
                                   const std::optional<Tensor>& alibi_slopes, std::string kv_cache_dtype,
                                   double k_scale, double v_scale, bool record_kv_metrics) {
  paged_attention_kvc(out, kv_metric_out, nullptr, nullptr, nullptr, nullptr, query, key_cache, value_cache,
                      num_kv_heads, scale, block_tables, context_lens, kv_position, last_position,
                      kv_metric_buffer_len, block_size, max_context_len, alibi_slopes, kv_cache_dtype, k_scale,
                      v_scale, record_kv_metrics);
}

void kvcompress_paged_attention_v2(Tensor& out, Tensor& kv_metric_out, const Tensor& exp_sums,
                                   const Tensor& max_logits, const Tensor& tmp_out, const Tensor& tmp_kv_metric_out,
                                   const Tensor& query, const Tensor& key_cache, const Tensor& value_cache,
                                   int64_t num_kv_heads, double scale, const Tensor& block_tables,
                                   const Tensor& context_lens, const Tensor& kv_position, const Tensor& last_position,
                                   const Tensor& kv_metric_buffer_len, int64_t block_size, int64_t max_context_len,
                                   const std::optional<Tensor>& alibi_slopes, std::string kv_cache_dtype,
                                   double k_scale, double v_scale, bool record_kv_metrics) {
  paged_attention_kvc(out, kv_metric_out, &exp_sums, &max_logits, &tmp_out, &tmp_kv_metric_out, query, key_cache,
                      value_cache, num_kv_heads, scale, block_tables, context_lens, kv_position, last_position,
                      kv_metric_buffer_len, block_size, max_context_len, alibi_slopes, kv_cache_dtype, k_scale,
                      v_scale, record_kv_metrics);
}

// The V1 scheduler pair (csrc/torch_bindings.cpp:374-394, csrc/kvcompress_eviction_kernels.cu
// schedule_cache_evictions / truncate_cache_evictions): dead in the reference behind `if False:`
// (vllm/kvcompress/scheduler.py:285) and mis-registered by its wrapper.  The
// schemas are here so that the op surface is complete; calling them says where the live path is.
constexpr const char* V1_DEAD =
    "schedule_cache_evictions (V1) is dead code in the reference; use "
    "vllm_kvcompress_amd.kvcompress.metrics.CompressionMetrics.schedule_evictions";

void schedule_cache_evictions(Tensor&, Tensor&, Tensor&, Tensor&, const Tensor&, const Tensor&, const Tensor&,
                              const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&,
                              const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t, bool,
                              const std::optional<Tensor>&, int64_t, int64_t, bool) {
  TORCH_CHECK(false, V1_DEAD);
}

void truncate_cache_evictions(Tensor&, Tensor&, Tensor&, const Tensor&, const Tensor&, int64_t, int64_t, int64_t) {
  TORCH_CHECK(false, V1_DEAD);
}

bool has_schema(const char* name) {
  return c10::Dispatcher::singleton().findSchema({name, ""}).has_value();
}

}  // namespace

// Schemas: the fork's own (csrc/torch_bindings.cpp:52-80, 353-362, 374-418).  FRAGMENTs, and only
// for names nobody defined yet, so the library can be loaded next to an extension that already
// carries the schemas; the kernels below are registered for the HIP (CUDA-key) backend either way.
TORCH_LIBRARY_FRAGMENT(_C_kvc_ops, m) {
  if (!has_schema("_C_kvc_ops::schedule_cache_evictions"))
    m.def("schedule_cache_evictions(Tensor! evicted_kv_indices, Tensor! evicted_logical_indices, "
          "Tensor! evicted_kv_count, Tensor! remaining_kv_count, Tensor evicted_kv_offsets, "
          "Tensor sorted_indices, Tensor seq_block_offsets, Tensor layer_by_block, Tensor head_by_block, "
          "Tensor virtual_block_num_by_block, Tensor evicted_blocks_per_seq, Tensor context_lens, "
          "Tensor hanging_token_count, Tensor kv_position, Tensor last_position, "
          "Tensor protected_window_size, int block_size, bool evict_evenly_per_layer, "
          "Tensor? control_layers, int max_evicted_kv, int null_eviction_index, bool truncate) -> ()");
  if (!has_schema("_C_kvc_ops::truncate_cache_evictions"))
    m.def("truncate_cache_evictions(Tensor! evicted_kv_indices, Tensor! evicted_logical_indices, "
          "Tensor! evicted_kv_count, Tensor evicted_kv_offsets, Tensor hanging_token_count, "
          "int block_size, int max_evicted_kv, int null_eviction_index) -> ()");
  if (!has_schema("_C_kvc_ops::count_block_evictions"))
    m.def("count_block_evictions(Tensor! evicted_block_count, Tensor! evicted_logical_indices, "
          "Tensor evicted_kv_offsets, Tensor hanging_token_count, int block_size, int null_value) -> ()");
  if (!has_schema("_C_kvc_ops::schedule_t1_cache_moves"))
    m.def("schedule_t1_cache_moves(Tensor! cache_moves_idx, Tensor! cache_moves_count, "
          "Tensor evicted_logical_indices, Tensor evicted_kv_count, Tensor evicted_kv_offsets, "
          "Tensor block_tables, Tensor context_lens, int block_size) -> ()");
  if (!has_schema("_C_kvc_ops::execute_cache_moves"))
    m.def("execute_cache_moves(Tensor! k_cache, Tensor! v_cache, Tensor! kv_metrics, Tensor! kv_position, "
          "Tensor cache_moves_idx, Tensor cache_moves_count, Tensor evicted_kv_offsets, "
          "int blocks_per_head, int threads_per_head) -> ()");
}
TORCH_LIBRARY_FRAGMENT(_C_cache_ops, m) {
  if (!has_schema("_C_cache_ops::kvcompress_reshape_and_cache"))
    m.def("kvcompress_reshape_and_cache(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache, "
          "Tensor kv_metrics, Tensor slot_mapping, Tensor kv_metric_head_bias, str kv_cache_dtype, "
          "float k_scale, float v_scale) -> ()");
}
TORCH_LIBRARY_FRAGMENT(_C, m) {
  if (!has_schema("_C::kvcompress_paged_attention_v1"))
    m.def("kvcompress_paged_attention_v1(Tensor! out, Tensor! kv_metric_out, Tensor query, Tensor key_cache, "
          "Tensor value_cache, int num_kv_heads, float scale, Tensor block_tables, Tensor context_lens, "
          "Tensor kv_position, Tensor last_position, Tensor kv_metric_buffer_len, int block_size, "
          "int max_context_len, Tensor? alibi_slopes, str kv_cache_dtype, float k_scale, float v_scale, "
          "bool record_kv_metrics) -> ()");
  if (!has_schema("_C::kvcompress_paged_attention_v2"))
    m.def("kvcompress_paged_attention_v2(Tensor! out, Tensor! kv_metric_out, Tensor exp_sums, Tensor max_logits, "
          "Tensor tmp_out, Tensor tmp_kv_metric_out, Tensor query, Tensor key_cache, Tensor value_cache, "
          "int num_kv_heads, float scale, Tensor block_tables, Tensor context_lens, Tensor kv_position, "
          "Tensor last_position, Tensor kv_metric_buffer_len, int block_size, int max_context_len, "
          "Tensor? alibi_slopes, str kv_cache_dtype, float k_scale, float v_scale, bool record_kv_metrics) -> ()");
}

// not in the fork: the start-up hooks of this binding
TORCH_LIBRARY_FRAGMENT(_kvc_mi355x, m) {
  m.def("reserve_workspace(Tensor like, int nbytes, str tag) -> ()", &reserve_workspace_op);
  m.def("set_attention_schedule(int schedule) -> ()", &set_attention_schedule_op);
  m.def("set_block_layout(int layout) -> ()", &set_block_layout_op);
  m.def("block_layout() -> int", &block_layout_op);
  // how many execute_cache_moves calls ran on the plan of the schedule_t1_cache_moves before them (tests)
  m.def("planned_compactions() -> int", &planned_compactions_op);
}

TORCH_LIBRARY_IMPL(_C_kvc_ops, CUDA, m) {
  m.impl("schedule_cache_evictions", &schedule_cache_evictions);
  m.impl("truncate_cache_evictions", &truncate_cache_evictions);
  m.impl("count_block_evictions", &count_block_evictions);
  m.impl("schedule_t1_cache_moves", &schedule_t1_cache_moves);
  m.impl("execute_cache_moves", &execute_cache_moves);
}
TORCH_LIBRARY_IMPL(_C_cache_ops, CUDA, m) {
  m.impl("kvcompress_reshape_and_cache", &kvcompress_reshape_and_cache);
}
TORCH_LIBRARY_IMPL(_C, CUDA, m) {
  m.impl("kvcompress_paged_attention_v1", &kvcompress_paged_attention_v1);
  m.impl("kvcompress_paged_attention_v2", &kvcompress_paged_attention_v2);
}
