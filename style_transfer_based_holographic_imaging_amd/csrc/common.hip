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
static int g_conv_tile = 0;
int conv_tile_override() { return g_conv_tile; }
}  // namespace holo

#ifndef HOLO_SOURCE_HASH
#define HOLO_SOURCE_HASH "unknown"
#endif
#ifndef HOLO_ROCM_VERSION
#define HOLO_ROCM_VERSION "unknown"
#endif

extern "C" {
// the tag lets the build recipe read the hash from the file without loading it (build.py: built_hash)
static const char kSourceHashTag[] = "@(#)holo_source_hash=" HOLO_SOURCE_HASH;
const char* holo_source_hash(void) { return kSourceHashTag + sizeof("@(#)holo_source_hash=") - 1; }
// the toolchain the object was compiled with: read by the build recipe only (a reason to REbuild, never a reason to refuse a load)
__attribute__((used)) static const char kRocmVersionTag[] = "@(#)holo_rocm_version=" HOLO_ROCM_VERSION;
int holo_debug_set_conv_tile(int tile) {
    HOLO_REQUIRE(tile >= 0 && tile <= 9, HOLO_ERR_ARG, "holo_debug_set_conv_tile: tile=%d outside [0,9]", tile);
    holo::g_conv_tile = tile;
    return HOLO_OK;
}
int holo_abi_version(void) { return HOLO_ABI_VERSION; }
const char* holo_last_error(void) { return holo::g_err; }
const char* holo_target_arch(void) { return "gfx950"; }
}
