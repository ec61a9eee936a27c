// Error channel and version of libnfe_render.so (no device code here).
#include <cstdarg>
#include <cstdio>

#include "nfe_render.h"

namespace nfe {
static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
const char* last_error() { return g_err; }
}  // namespace nfe

extern "C" int nfe_abi_version(void) { return NFE_ABI_VERSION; }
extern "C" const char* nfe_last_error(void) { return nfe::last_error(); }
