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

// ---------------------------------------------------------------- deterministic gradient sums
// LG-Net's backward combines partial sums of different workgroups with fp32 atomics in three places (row chunks of the weight
// gradient, the split inner loop of the SA backward, the order of a point's in-edges in the N2P gather): fastest, but the
// summation order — hence the low bits of the gradients — changes from run to run.  dvm_set_deterministic(1) (or
// DVM_DETERMINISTIC=1) fixes the order in all three: per-chunk partial tiles added in chunk order, no split, in-edge lists sorted.
#include <atomic>
#include <stdlib.h>
namespace dvm {
// ---------------------------------------------------------------- environment options: ONE place, read once (table: include/dvm.h)
const Options &options() {
    static const Options o = [] {
        Options r{0, -1, 0.0014f, 0.02f, -1, 1, 0};
        if (const char *e = getenv("DVM_DETERMINISTIC")) r.deterministic = atoi(e) != 0;
        if (const char *e = getenv("DVM_K1_ROUTE")) r.k1_route = atoi(e);
        if (const char *e = getenv("DVM_K1_ROUTE_P")) (void)sscanf(e, "%f,%f", &r.k1_p_coarse, &r.k1_p_lean);
        if (const char *e = getenv("DVM_LINEAR_CFG")) r.linear_cfg = atoi(e);
        if (const char *e = getenv("DVM_PAIR_OVERLAP")) r.pair_overlap = atoi(e) != 0;
        if (const char *e = getenv("DVM_DEBUG")) r.debug = atoi(e);
        return r;
    }();
    return o;
}
static std::atomic<int> g_deterministic{options().deterministic};
bool deterministic() { return g_deterministic.load(std::memory_order_relaxed) != 0; }
}  // namespace dvm
DVM_EXPORT int dvm_set_deterministic(int on) { return dvm::g_deterministic.exchange(on ? 1 : 0); }
DVM_EXPORT int dvm_get_deterministic(void) { return dvm::g_deterministic.load(std::memory_order_relaxed); }

// ---------------------------------------------------------------- per-device kernel attributes
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device's copy of a kernel: it has to be made
// per (device, kernel), not once per process, and it has to GROW when a later launch of the same kernel needs more
// (callers whose byte count depends on a runtime size: the CSR build, the GEMM tiles, the JBU tile): the registry keeps
// the largest value set so far and raises it when a bigger request arrives.  Thread-safe; no HIP allocation.
#include <map>
#include <mutex>
#include <utility>
namespace dvm {
static std::mutex g_attr_mu;
static std::map<std::pair<int, const void *>, int> g_attr_bytes;
void ensure_dyn_lds(const void *kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_attr_mu);
    auto it = g_attr_bytes.find({dev, kernel});
    if (it != g_attr_bytes.end() && it->second >= bytes) return;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) g_attr_bytes[{dev, kernel}] = bytes;
}
// compute units of the current device (cached per device): the grid of the persistent kernels
int device_cu_count() {
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    cus[dev] = n;
    return n;
}
}  // namespace dvm

// ---------------------------------------------------------------- helper streams of dvm_pair_fwd_f32
// One context per (device, caller stream), created by dvm_pair_init — never inside a compute call, which therefore
// stays allocation-free and capturable; without a context dvm_pair_fwd_f32 runs everything on the caller's stream.
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
    for (hipEvent_t e : {c->ev_fork, c->ev_join, c->ev_join2, c->ev_aux})
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
        hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_aux, hipEventDisableTiming) != hipSuccess) {
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

// ---------------------------------------------------------------- kernel launch timing
// Optional HIP-event brackets around the launches of the pair path's kernels, recorded on the stream the kernel is
// launched on (bench.py's roofline legs).  Slot 0 (DVM_PROF_K1_SWEEP) is the soft-correspondence sweep; the other
// slots (include/dvm.h, DVM_PROF_*) are the remaining kernels of one step.  Off by default; when off the launch path
// records nothing.  Only slots selected by dvm_profile_select() record (default: slot 0 only).
// The registry is shared by every host thread that launches through the library (the ABI allows one thread per
// stream): ONE mutex guards the event pool, the bracket list, the window flag and the slot names; a bracket's two
// events are reserved under the lock when it opens and belong to the opening thread until it closes (thread-local).
#include <string.h>

#include <vector>
namespace dvm {
struct ProfRec {
    int id;
    hipEvent_t e0, e1;
};
static std::mutex g_prof_mu;
static std::vector<hipEvent_t> g_ev;      // pool, 2 per bracket
static std::vector<ProfRec> g_rec;        // closed brackets of the current window
static int g_ev_used = 0;
static bool g_prof_on = false;
static unsigned g_prof_mask = 1u;
static unsigned g_prof_window = 0;        // bumped by enable / disable: a bracket opened in an earlier window is dropped
static char g_slot_name[DVM_PROF_COUNT][160] = {"softcorr_sweep2_kernel", "softcorr_refine_kernel", "mlp_f16x2_kernel", "grid_chamfer_kernel",
                                               "pool_kernel",            "grid_knn_self_kernel",   "fps_kernel",       "assemble_pooled_kernel"};
struct ProfOpen {
    int id = -1;
    unsigned window = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};
static thread_local ProfOpen g_open;      // (a bracket is opened and closed by the same host thread)
void prof_begin(hipStream_t s, int id) {
    hipEvent_t e0;
    {
        std::lock_guard<std::mutex> lock(g_prof_mu);
        if (!g_prof_on || !((g_prof_mask >> id) & 1u) || g_ev_used + 2 > (int)g_ev.size()) return;
        e0 = g_ev[g_ev_used];
        g_open.id = id, g_open.window = g_prof_window, g_open.e0 = e0, g_open.e1 = g_ev[g_ev_used + 1];
        g_ev_used += 2;
    }
    (void)hipEventRecord(e0, s);
}
void prof_end(hipStream_t s, int id) {
    if (g_open.id != id) return;
    g_open.id = -1;
    (void)hipEventRecord(g_open.e1, s);
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (g_prof_on && g_open.window == g_prof_window) g_rec.push_back(ProfRec{id, g_open.e0, g_open.e1});
}
// the kernel(s) a slot's bracket actually enclosed at its last launch (slot 0: whichever pass-A kernel the probe / alpha routed to)
void prof_note(int id, const char *name) {
    if (id < 0 || id >= DVM_PROF_COUNT) return;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (!g_prof_on) return;
    strncpy(g_slot_name[id], name, sizeof(g_slot_name[id]) - 1);
}
static int prof_read(int id, double *total_ms, int *launches) {
    std::vector<ProfRec> recs;
    {
        std::lock_guard<std::mutex> lock(g_prof_mu);
        for (const ProfRec &r : g_rec)
            if (r.id == id) recs.push_back(r);
    }
    double tot = 0.0;
    int n = 0;
    for (const ProfRec &r : recs) {   // (synchronises outside the lock: launches on other threads go on recording)
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) {
            set_error("dvm_profile_read: bracket %d of kernel slot %d not readable", n, id);
            return DVM_ELAUNCH;
        }
        tot += ms;
        ++n;
    }
    *total_ms = tot;
    *launches = n;
    return DVM_OK;
}
}  // namespace dvm

DVM_EXPORT int dvm_profile_enable(int max_launches) {
    DVM_REQUIRE(max_launches >= 1 && max_launches <= (1 << 20), "dvm_profile_enable: bad max_launches %d", max_launches);
    std::lock_guard<std::mutex> lock(dvm::g_prof_mu);
    while ((int)dvm::g_ev.size() < 2 * max_launches) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
            dvm::set_error("dvm_profile_enable: hipEventCreate failed");
            return DVM_ELAUNCH;
        }
        dvm::g_ev.push_back(e);
    }
    dvm::g_ev_used = 0;
    dvm::g_rec.clear();
    ++dvm::g_prof_window;
    dvm::g_prof_on = true;
    return DVM_OK;
}

DVM_EXPORT int dvm_profile_select(unsigned kernel_mask) {
    DVM_REQUIRE(kernel_mask != 0 && kernel_mask < (1u << DVM_PROF_COUNT), "dvm_profile_select: bad mask 0x%x", kernel_mask);
    std::lock_guard<std::mutex> lock(dvm::g_prof_mu);
    dvm::g_prof_mask = kernel_mask;
    return DVM_OK;
}

DVM_EXPORT int dvm_profile_read(double *total_ms, int *launches) {
    DVM_REQUIRE(total_ms && launches, "dvm_profile_read: null pointer");
    return dvm::prof_read(DVM_PROF_K1_SWEEP, total_ms, launches);
}

DVM_EXPORT int dvm_profile_read_kernel(int kernel, double *total_ms, int *launches) {
    DVM_REQUIRE(total_ms && launches, "dvm_profile_read_kernel: null pointer");
    DVM_REQUIRE(kernel >= 0 && kernel < DVM_PROF_COUNT, "dvm_profile_read_kernel: bad kernel slot %d", kernel);
    return dvm::prof_read(kernel, total_ms, launches);
}

DVM_EXPORT const char *dvm_profile_kernel_name(int kernel) {
    static thread_local char out[160];
    if (kernel < 0 || kernel >= DVM_PROF_COUNT) return "";
    std::lock_guard<std::mutex> lock(dvm::g_prof_mu);
    memcpy(out, dvm::g_slot_name[kernel], sizeof(out));
    return out;
}

DVM_EXPORT int dvm_profile_disable(void) {
    std::lock_guard<std::mutex> lock(dvm::g_prof_mu);
    dvm::g_prof_on = false;
    dvm::g_ev_used = 0;
    dvm::g_rec.clear();
    ++dvm::g_prof_window;
    dvm::g_prof_mask = 1u;
    return DVM_OK;
}
