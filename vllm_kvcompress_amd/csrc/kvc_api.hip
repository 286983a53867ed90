// Error state + ABI version for libkvc_mi355x.so (see include/kvc_mi355x.h).
#include "kvc_common.h"
#include "../../include/kvc_mi355x.h"

namespace kvc {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace kvc

extern "C" int kvc_abi_version(void) { return KVC_ABI_VERSION; }
extern "C" const char* kvc_last_error(void) { return kvc::g_last_error.c_str(); }
