// Error plumbing and version entry points of the C ABI (include/jenga_amd.h).
#include <cstdarg>
#include <cstdio>

#include "../../include/jenga_amd.h"

namespace jenga {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace jenga

extern "C" int jenga_abi_version(void) { return JENGA_ABI_VERSION; }
extern "C" const char* jenga_last_error(void) { return jenga::g_err; }
