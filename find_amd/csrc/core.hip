// Error plumbing and version queries of libfind_hip.so.
#include "common.h"

namespace find {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}
}  // namespace find

extern "C" int find_abi_version(void) { return FIND_ABI_VERSION; }
extern "C" const char* find_last_error(void) { return find::g_err; }
extern "C" const char* find_build_arch(void) { return "gfx950"; }
