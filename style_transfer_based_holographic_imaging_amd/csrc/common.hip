#include "common.h"
#include <cstring>

namespace holo {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace holo

extern "C" {
int holo_abi_version(void) { return HOLO_ABI_VERSION; }
const char* holo_last_error(void) { return holo::g_err; }
const char* holo_target_arch(void) { return "gfx950"; }
}
