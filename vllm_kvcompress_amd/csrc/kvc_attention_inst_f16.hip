// decode attention kernels, instantiated for fp16 queries and "auto" caches (see kvc_attention_kernels.h)
#include "kvc_attention_kernels.h"

namespace kvc {
#define KVC_X(HD, BS) template int launch_attention<_Float16, HD, BS, 0>(const AttnArgs&, int, hipStream_t);
KVC_ATT_SHAPES(KVC_X)
#undef KVC_X
}  // namespace kvc
