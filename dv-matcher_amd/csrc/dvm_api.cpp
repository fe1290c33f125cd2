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

// ---------------------------------------------------------------- per-device kernel attributes
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device's copy of a kernel: it has to be made
// once per (device, kernel), not once per process.  Thread-safe; no HIP allocation.
#include <mutex>
#include <set>
#include <utility>
namespace dvm {
static std::mutex g_attr_mu;
static std::set<std::pair<int, const void *>> g_attr_done;
void ensure_dyn_lds(const void *kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_attr_mu);
    if (g_attr_done.count({dev, kernel})) return;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) g_attr_done.insert({dev, kernel});
}
}  // namespace dvm

// ---------------------------------------------------------------- helper streams of dvm_pair_fwd_f32
// One context per (device, caller stream), created by dvm_pair_init — never inside a compute call, which therefore
// stays allocation-free and capturable; without a context dvm_pair_fwd_f32 runs everything on the caller's stream.
#include <map>
namespace dvm {
static std::mutex g_pair_mu;
static std::map<std::pair<int, hipStream_t>, PairCtx *> g_pair_ctx;
PairCtx *pair_ctx_find(hipStream_t caller) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_pair_mu);
    auto it = g_pair_ctx.find({dev, caller});
    return it == g_pair_ctx.end() ? nullptr : it->second;
}
static void pair_ctx_free(PairCtx *c) {
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->side2) (void)hipStreamDestroy(c->side2);
    for (hipEvent_t e : {c->ev_fork, c->ev_join, c->ev_join2})
        if (e) (void)hipEventDestroy(e);
    delete c;
}
}  // namespace dvm

DVM_EXPORT int dvm_pair_init(void *stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        dvm::set_error("dvm_pair_init: no current device");
        return DVM_ELAUNCH;
    }
    std::lock_guard<std::mutex> lock(dvm::g_pair_mu);
    const auto key = std::make_pair(dev, (hipStream_t)stream);
    if (dvm::g_pair_ctx.count(key)) return DVM_OK;
    dvm::PairCtx *c = new dvm::PairCtx();
    c->device = dev;
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming) != hipSuccess) {
        dvm::pair_ctx_free(c);
        dvm::set_error("dvm_pair_init: cannot create the helper streams / events on device %d", dev);
        return DVM_ELAUNCH;
    }
    dvm::g_pair_ctx[key] = c;
    return DVM_OK;
}

DVM_EXPORT int dvm_pair_destroy(void) {
    std::lock_guard<std::mutex> lock(dvm::g_pair_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &kv : dvm::g_pair_ctx) {
        (void)hipSetDevice(kv.first.first);
        dvm::pair_ctx_free(kv.second);
    }
    dvm::g_pair_ctx.clear();
    (void)hipSetDevice(cur);
    return DVM_OK;
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
