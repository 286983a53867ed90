// kvc_harvest_layout.h -- the harvest buffer (kvc_schedule_params.harvest_buf, kvc_harvest_buffer_bytes()): what the
// small-eviction schedule's candidate lists look like to whoever fills them -- the schedule's own collecting pass,
// the harvesting aggregation pass (kvc_schedule_harvest.h) and the decode attention's fused-metric epilogue
// (kvc_attention_kernels.h, section "harvest in the epilogue").  No kernels here: shared by translation units.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace kvc {

constexpr int KREC = 256;            // record length of the small-eviction schedule (keys per head)
constexpr int CLAIM_SHARDS = 64;     // counters of claimed blocks, 128 B apart

// header (256 B, reserved) | pivot [B] u32 | claimed [CLAIM_SHARDS x 32] u32 | cnt [G] u32 | def [G] u32 |
// rec64 [G, KREC] u64 = (key << 32 | physical slot) | seen_ctx [G] i32 | seen_seq [2 B] i32;
// claimed | cnt | def are cleared by one fill per harvest.  seen_*: what lists made by the attention's epilogue
// were made WITH -- every head's context length, every sequence's position and protected window -- so that the
// schedule call can tell on the device whether they are the lists of ITS batch (kvc_schedule_params.harvest bit 3).
struct HvLayout { size_t pivot, claimed, cnt, def, rec64, seen_ctx, seen_seq, total; };
inline HvLayout hv_layout(int32_t G, int32_t B) {
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  HvLayout l;
  size_t o = 256;
  l.pivot = o;    o = up(o + (size_t)B * 4);
  l.claimed = o;  o = up(o + (size_t)CLAIM_SHARDS * 128);
  l.cnt = o;      o = up(o + (size_t)G * 4);
  l.def = o;      o = up(o + (size_t)G * 4);
  l.rec64 = o;    o = up(o + (size_t)G * KREC * 8);
  l.seen_ctx = o; o = up(o + (size_t)G * 4);
  l.seen_seq = o; o = up(o + (size_t)B * 8);
  l.total = o;
  return l;
}

}  // namespace kvc
