// core.hip — error reporting and library identity for libide3d_hip.so.
#include "common.h"
#include <string.h>

namespace ide3d {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace ide3d

extern "C" const char* ide3d_last_error(void) { return ide3d::g_err; }
extern "C" int ide3d_abi_version(void) { return 6; }
extern "C" const char* ide3d_build_arch(void) { return "gfx950"; }
extern "C" const char* ide3d_build_flags(void) {
    static char buf[512] = "";
    static bool done = false;
    if (!done) {
        snprintf(buf, sizeof(buf), "%s%s%s", ide3d::modconv_build_flags(), ide3d::triplane_tile_build_flags(), ide3d::raymarch_build_flags());
        const size_t n = strlen(buf);
        if (n && buf[n - 1] == ' ') buf[n - 1] = 0;
        done = true;
    }
    return buf;
}
