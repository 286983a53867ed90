// decode attention kernels, instantiated for fp16 queries and fp8 caches (see kvc_attention_kernels.h)
#include "kvc_attention_kernels.h"

namespace kvc {
#define KVC_X(HD, BS)                                                                 \
  template int launch_attention<_Float16, HD, BS, 1>(const AttnArgs&, int, hipStream_t); \
  template int launch_attention<_Float16, HD, BS, 2>(const AttnArgs&, int, hipStream_t); \
  template int launch_attention<_Float16, HD, BS, 3>(const AttnArgs&, int, hipStream_t); \
  template int launch_attention<_Float16, HD, BS, 4>(const AttnArgs&, int, hipStream_t);
KVC_ATT_F8_SHAPES(KVC_X)
#undef KVC_X
}  // namespace kvc
