// dvm_api.cpp — ABI bookkeeping for libdvm_hip.so.
#include <stdarg.h>

#include "dvm_common.h"

namespace dvm {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace dvm

DVM_EXPORT int dvm_abi_version(void) { return DVM_ABI_VERSION; }
DVM_EXPORT const char *dvm_last_error(void) { return dvm::g_err; }
DVM_EXPORT int dvm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------- K1 launch timing
// Optional HIP-event bracket around every soft-correspondence kernel launch, recorded on the
// stream the kernel is launched on (bench.py's roofline leg).  Off by default; when off the
// launch path records nothing.
#include <vector>
namespace dvm {
static std::vector<hipEvent_t> g_ev;
static int g_ev_used = 0;
static bool g_prof_on = false;
void prof_begin(hipStream_t s) {
    if (g_prof_on && g_ev_used + 2 <= (int)g_ev.size()) (void)hipEventRecord(g_ev[g_ev_used], s);
}
void prof_end(hipStream_t s) {
    if (g_prof_on && g_ev_used + 2 <= (int)g_ev.size()) {
        (void)hipEventRecord(g_ev[g_ev_used + 1], s);
        g_ev_used += 2;
    }
}
}  // namespace dvm

DVM_EXPORT int dvm_profile_enable(int max_launches) {
    DVM_REQUIRE(max_launches >= 1 && max_launches <= (1 << 20), "dvm_profile_enable: bad max_launches %d", max_launches);
    while ((int)dvm::g_ev.size() < 2 * max_launches) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
            dvm::set_error("dvm_profile_enable: hipEventCreate failed");
            return DVM_ELAUNCH;
        }
        dvm::g_ev.push_back(e);
    }
    dvm::g_ev_used = 0;
    dvm::g_prof_on = true;
    return DVM_OK;
}

DVM_EXPORT int dvm_profile_read(double *total_ms, int *launches) {
    DVM_REQUIRE(total_ms && launches, "dvm_profile_read: null pointer");
    double tot = 0.0;
    int n = dvm::g_ev_used / 2;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(dvm::g_ev[2 * i + 1]) != hipSuccess ||
            hipEventElapsedTime(&ms, dvm::g_ev[2 * i], dvm::g_ev[2 * i + 1]) != hipSuccess) {
            dvm::set_error("dvm_profile_read: event %d not readable", i);
            return DVM_ELAUNCH;
        }
        tot += ms;
    }
    *total_ms = tot;
    *launches = n;
    dvm::g_ev_used = 0;
    return DVM_OK;
}

DVM_EXPORT int dvm_profile_disable(void) {
    dvm::g_prof_on = false;
    dvm::g_ev_used = 0;
    return DVM_OK;
}
