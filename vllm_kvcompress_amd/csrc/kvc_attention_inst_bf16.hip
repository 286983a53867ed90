// decode attention kernels, instantiated for bf16 queries and "auto" caches (see kvc_attention_kernels.h)
#include "kvc_attention_kernels.h"

namespace kvc {
#define KVC_X(HD, BS) template int launch_attention<__bf16, HD, BS, 0>(const AttnArgs&, int, hipStream_t);
KVC_ATT_SHAPES(KVC_X)
#undef KVC_X
}  // namespace kvc
