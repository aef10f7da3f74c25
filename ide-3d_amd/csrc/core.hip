// core.hip — error reporting and library identity for libide3d_hip.so.
#include "common.h"
#include <string.h>
#include <stdio.h>
#include <map>
#include <mutex>
#include <tuple>

namespace ide3d {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static int g_excl_violations = 0;
static char g_excl_text[384] = "";
void note_exclusive_violation(const char* kernel, int blocks_per_cu) {
    ++g_excl_violations;
    snprintf(g_excl_text, sizeof(g_excl_text), "%d workgroups of %.200s fit one CU of this device: exclusive residency (DESIGN.md 4.2) does not hold", blocks_per_cu, kernel);
    fprintf(stderr, "[libide3d_hip] %s; the launch is refused (select IDE3D_CONV_ARITH=fp32)\n", g_excl_text);
}
int exclusive_violations() { return g_excl_violations; }
static thread_local bool g_refused = false;
void refuse_launch(const char* kernel) {
    g_refused = true;
    set_error("%.200s: more than one workgroup fits a CU of this device, exclusive residency (DESIGN.md 4.2) does not hold; launch refused "
              "(IDE3D_CONV_ARITH=fp32 selects the arithmetic that needs none)", kernel);
}
bool take_refused() { const bool r = g_refused; g_refused = false; return r; }

// One occupancy query per (kernel address, device, workgroup size, dynamic LDS): a kernel whose LDS size varies per launch (raymarch) is
// asked again for every size it is launched with.
bool exclusive_checked(const void* kernel, int threads, size_t dyn_lds, const char* name) {
    static std::mutex mu;
    static std::map<std::tuple<const void*, int, int, size_t>, int> table;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    int occ;
    {
        std::lock_guard<std::mutex> lock(mu);
        const auto key = std::make_tuple(kernel, dev, threads, dyn_lds);
        auto it = table.find(key);
        if (it == table.end()) {
            int n = 0;
            occ = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, dyn_lds) == hipSuccess && n > 0) ? n : -1;
            table.emplace(key, occ);
            if (occ > 1) note_exclusive_violation(name, occ);
        } else occ = it->second;
    }
    if (occ > 1) refuse_launch(name);
    return occ <= 1;
}

}  // namespace ide3d

extern "C" int ide3d_exclusive_violations(void) { return ide3d::g_excl_violations; }
extern "C" const char* ide3d_exclusive_violation_text(void) { return ide3d::g_excl_text; }
extern "C" const char* ide3d_last_error(void) { return ide3d::g_err; }
extern "C" int ide3d_abi_version(void) { return 8; }
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
