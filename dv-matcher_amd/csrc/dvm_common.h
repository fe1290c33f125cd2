// dvm_common.h — shared host/device helpers for libdvm_hip.so (gfx950 only).
// Built with -ffp-contract=off: every fused multiply-add in this library is an explicit
// fmaf()/MFMA, every separate mul/add stays separate, because integer outputs (arg-min maps,
// top-k columns, kNN / FPS indices) are required to be bit-exact with the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dvm.h"

#define DVM_EXPORT extern "C" __attribute__((visibility("default")))

namespace dvm {

void set_error(const char *fmt, ...);
void prof_begin(hipStream_t s, int id = 0);  // dvm_api.cpp: optional event bracket around a launch (slot DVM_PROF_*)
void prof_end(hipStream_t s, int id = 0);
bool deterministic();   // dvm_api.cpp: dvm_set_deterministic / DVM_DETERMINISTIC — gradient sums in a fixed order (no float atomics between workgroups)
void prof_note(int id, const char *name);   // which kernel(s) the slot's bracket enclosed (reported by dvm_profile_kernel_name)
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (current device, kernel) — dvm_api.cpp
void ensure_dyn_lds(const void *kernel, int bytes);
int device_cu_count();   // compute units of the current device (cached)
// The library's environment options, read ONCE at load (dvm_api.cpp; the table is documented in include/dvm.h).
struct Options {
    int deterministic;        // DVM_DETERMINISTIC: initial value of the dvm_set_deterministic flag
    int k1_route;             // DVM_K1_ROUTE: -1 = the probe decides (default), else force K1_ROUTE_* for every launch at alpha >= 32
    float k1_p_coarse, k1_p_lean;   // DVM_K1_ROUTE_P="p_coarse,p_lean": the probe's thresholds
    int linear_cfg;           // DVM_LINEAR_CFG: tile configuration of dvm_linear_f32 (-1 = chosen per shape)
    int pair_overlap;         // DVM_PAIR_OVERLAP: initial value of dvm_pair_set_overlap (default 1)
    int debug;                // DVM_DEBUG: bit mask of synchronous diagnostics on stderr (DVM_DEBUG_*)
};
constexpr int DVM_DEBUG_K1_ROUTES = 1, DVM_DEBUG_K1_FLAGGED = 2, DVM_DEBUG_CHAMFER_STATS = 4, DVM_DEBUG_K1_STAMPS = 8, DVM_DEBUG_MLP_STAMPS = 16;
const Options &options();
// helper streams / events of dvm_pair_fwd_f32 for one (device, caller stream), made by dvm_pair_init — dvm_api.cpp
struct PairCtx {
    int device = 0;
    hipStream_t side = nullptr, side2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr, ev_aux = nullptr;
};
PairCtx *pair_ctx_find(hipStream_t caller);

#define DVM_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            dvm::set_error(__VA_ARGS__);  \
            return DVM_EINVAL;            \
        }                                 \
    } while (0)

#define DVM_CHECK_LAUNCH(name)                                                          \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            dvm::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
            return DVM_ELAUNCH;                                                         \
        }                                                                               \
    } while (0)

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// bump allocator over a caller-provided workspace
struct Arena {
    char *base;
    size_t cap, off;
    Arena(void *p, size_t n) : base((char *)p), cap(n), off(0) {}
    template <typename T>
    T *take(size_t count) {
        size_t bytes = align_up(count * sizeof(T));
        char *p = base ? base + off : nullptr;
        off += bytes;
        return (T *)p;
    }
    bool ok() const { return base != nullptr && off <= cap; }
};

// p + n that stays null for a null p: workspace layouts are also carved over a null base to compute their size, and
// offsetting a null pointer is undefined behaviour (found by the UBSan build, csrc/san/)
template <typename T>
static inline T *offset_ptr(T *p, size_t n) {
    return p ? p + n : nullptr;
}

constexpr int WAVE = 64;

// uniform-grid neighbour search (dvm_grid.hip)
struct GridBuf {
    float4 *pts;
    int32_t *ids;
    int32_t *start;
    float *params;
    int P, G;
};
size_t grid_bytes(int B, int P);
// view of shapes [b0, ...) of a batched grid
static inline GridBuf grid_slice(const GridBuf &g, int b0) {
    GridBuf r = g;
    const int G3 = g.G * g.G * g.G;
    r.pts = offset_ptr(g.pts, (size_t)b0 * g.P);
    r.ids = offset_ptr(g.ids, (size_t)b0 * g.P);
    r.start = offset_ptr(g.start, (size_t)b0 * (G3 + 1));
    r.params = offset_ptr(g.params, (size_t)b0 * 8);
    return r;
}
GridBuf grid_carve(Arena &ar, int B, int P);
void launch_grid_build(const float *xyz, int B, int Nsrc, const int32_t *sel, const GridBuf &gb, hipStream_t s);
void launch_grid_build_sets(const float *const *xyz, const int *Nsrc, const GridBuf *gb, int nsets, int B, hipStream_t s);   // up to 4 cloud sets, one launch
void launch_grid_knn_self(const GridBuf &gb, int B, int k, int32_t *idx, hipStream_t s);
void launch_grid_ring(const GridBuf &gnodes, int B, int32_t *ring, hipStream_t s);
void launch_grid_infl(const float *xyz, int B, int N, const GridBuf &gnodes, const GridBuf &gverts, int32_t *infl, float *dists,
                      double *nnd, hipStream_t s);
void launch_grid_chamfer(const GridBuf *gq, const GridBuf *gb, float *const *dout, int32_t *const *iout, int ngroups, int B,
                         hipStream_t s);

// ---------------------------------------------------------------- device helpers
// Sorted "k best" list in registers: keys ascending, ties keep the earlier (lower-index)
// entry first provided candidates arrive in ascending index order.
template <int K, typename KeyT>
struct KBest {
    KeyT key[K];
    int idx[K];
    __device__ __forceinline__ void init(KeyT inf) {
#pragma unroll
        for (int t = 0; t < K; ++t) {
            key[t] = inf;
            idx[t] = 0x7fffffff;
        }
    }
    __device__ __forceinline__ KeyT worst() const { return key[K - 1]; }
    __device__ __forceinline__ KeyT key_at(int q) const { return key[q]; }
    __device__ __forceinline__ int idx_at(int q) const { return idx[q]; }
    // insert (v, j) assuming v < key[K-1] was already tested by the caller (or test here)
    __device__ __forceinline__ void insert(KeyT v, int j) {
        if (!(v < key[K - 1])) return;
        key[K - 1] = v;
        idx[K - 1] = j;
#pragma unroll
        for (int p = K - 1; p > 0; --p) {
            bool sw = key[p] < key[p - 1];
            KeyT a = key[p - 1], b = key[p];
            int ia = idx[p - 1], ib = idx[p];
            key[p - 1] = sw ? b : a;
            key[p] = sw ? a : b;
            idx[p - 1] = sw ? ib : ia;
            idx[p] = sw ? ia : ib;
        }
    }
    // branch-free form of insert(): a candidate that does not beat the current worst leaves the list
    // unchanged (used inside wave-uniform loops where a divergent early-out costs register copies)
    __device__ __forceinline__ void insert_nb(KeyT v, int j) {
        const bool better = v < key[K - 1];
        key[K - 1] = better ? v : key[K - 1];
        idx[K - 1] = better ? j : idx[K - 1];
#pragma unroll
        for (int p = K - 1; p > 0; --p) {
            bool sw = key[p] < key[p - 1];
            KeyT a = key[p - 1], b = key[p];
            int ia = idx[p - 1], ib = idx[p];
            key[p - 1] = sw ? b : a;
            key[p] = sw ? a : b;
            idx[p - 1] = sw ? ib : ia;
            idx[p] = sw ? ia : ib;
        }
    }
    // insert honouring (key, idx) lexicographic order for candidates arriving in any order
    __device__ __forceinline__ void insert_lex(KeyT v, int j) {
        bool better = (v < key[K - 1]) || (v == key[K - 1] && j < idx[K - 1]);
        if (!better) return;
        key[K - 1] = v;
        idx[K - 1] = j;
#pragma unroll
        for (int p = K - 1; p > 0; --p) {
            bool sw = (key[p] < key[p - 1]) || (key[p] == key[p - 1] && idx[p] < idx[p - 1]);
            KeyT a = key[p - 1], b = key[p];
            int ia = idx[p - 1], ib = idx[p];
            key[p - 1] = sw ? b : a;
            key[p] = sw ? a : b;
            idx[p - 1] = sw ? ib : ia;
            idx[p] = sw ? ia : ib;
        }
    }
};

// The same list for NON-NEGATIVE float keys, each (key, index) pair packed into one double — high word the key's bit pattern,
// low word the index: such doubles order like (key, index) pairs, so a compare-swap of the insertion is v_min_f64 + v_max_f64
// instead of two compares, their combination and four selects (the idiom of the soft-correspondence sweeps).  insert_lex below
// costs 2 K instructions where KBest's costs 9 K: the xyz kNN executed 16 600 vector instructions per wave, most of them these.
// Keys must be >= +0 (a -0 is turned into +0), finite or +inf.  The key's bit pattern is biased by 2^20 in the high word, so that
// even a key of 0 packs into a NORMAL double (the order of the patterns is unchanged; +inf and NaN patterns stay finite doubles):
// the comparison does not depend on the fp64 denormal mode.
template <int K, int W = K - 1 /* the entry worst() reports: the K-best search certifies against it */>
struct KBestPacked {
    double e[K];
    static __device__ __forceinline__ double mn(double a, double b) {
        double r;
        asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ double mx(double a, double b) {
        double r;
        asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    __device__ __forceinline__ void init(float inf) {
#pragma unroll
        for (int t = 0; t < K; ++t) e[t] = __hiloint2double(__float_as_int(inf) + KBIAS, 0x7fffffff);
    }
    static constexpr int KBIAS = 1 << 20;
    __device__ __forceinline__ float key_at(int q) const { return __int_as_float(__double2hiint(e[q]) - KBIAS); }
    __device__ __forceinline__ int idx_at(int q) const { return __double2loint(e[q]); }
    __device__ __forceinline__ float worst() const { return key_at(W); }
    __device__ __forceinline__ void insert_lex(float v, int j) {
        const double x = __hiloint2double(__float_as_int(v + 0.f) + KBIAS, j);
        if (!(x < e[K - 1])) return;
        e[K - 1] = x;
#pragma unroll
        for (int p = K - 1; p > 0; --p) {
            const double lo = mn(e[p - 1], e[p]), hi = mx(e[p - 1], e[p]);
            e[p - 1] = lo;
            e[p] = hi;
        }
    }
};

// ATen cascade sum of squares of one contiguous row (see oracle/dvm_oracle.c dvo_aten_sum):
// 8-lane vectors, 4-way ILP, cascade levels of 16 vectors, lanes added last.
__device__ inline float aten_sumsq_row(const float *__restrict__ x, int K) {
    if (K < 8) {
        // scalar path: row viewed as (-1,4)
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        int si = K / 4;
        for (int i = 0; i < si; ++i)
            for (int k = 0; k < 4; ++k) {
                float v = x[i * 4 + k];
                p[k] = p[k] + v * v;
            }
        for (int i = si * 4; i < K; ++i) {
            float v = x[i];
            p[0] = p[0] + v * v;
        }
        for (int k = 1; k < 4; ++k) p[0] = p[0] + p[k];
        return p[0];
    }
    const int V = 8;
    int vec_size = K / V, size_ilp = vec_size / 4;
    int cl = 0;
    while ((1 << cl) < size_ilp) cl++;
    int level_power = cl / 4;
    if (level_power < 4) level_power = 4;
    int level_step = 1 << level_power, level_mask = level_step - 1;
    float fin_lane[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) fin_lane[l] = 0.f;
    // process lane by lane to keep register use small: each lane's sum is independent
    for (int l = 0; l < V; ++l) {
        float acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
        int i = 0;
        for (; i + level_step <= size_ilp;) {
            for (int j = 0; j < level_step; ++j, ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v = x[(i * 4 + k) * V + l];
                    acc[0][k] = acc[0][k] + v * v;
                }
#pragma unroll
            for (int j = 1; j < 4; ++j) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[j][k] = acc[j][k] + acc[j - 1][k];
                    acc[j - 1][k] = 0.f;
                }
                int mask = level_mask << (j * level_power);
                if ((i & mask) != 0) break;
            }
        }
        for (; i < size_ilp; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = x[(i * 4 + k) * V + l];
                acc[0][k] = acc[0][k] + v * v;
            }
#pragma unroll
        for (int j = 1; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[0][k] = acc[0][k] + acc[j][k];
        for (int ii = size_ilp * 4; ii < vec_size; ++ii) {
            float v = x[ii * V + l];
            acc[0][0] = acc[0][0] + v * v;
        }
#pragma unroll
        for (int k = 1; k < 4; ++k) acc[0][0] = acc[0][0] + acc[0][k];
        fin_lane[l] = acc[0][0];
    }
    float fin = 0.f;
    for (int k = vec_size * V; k < K; ++k) {
        float v = x[k];
        fin = fin + v * v;
    }
#pragma unroll
    for (int l = 0; l < V; ++l) fin = fin + fin_lane[l];
    return fin;
}

// Correctly rounded fp32 sqrt for x >= 0 (+0, subnormal, normal, +inf), written out explicitly:
// hipcc lowers some (SLP-vectorised) sqrtf calls to the bare 1-ulp v_sqrt_f32 even with
// -fhip-fp32-correctly-rounded-divide-sqrt, and a 1-ulp distance changes rankings.  Same
// sequence as LLVM's IEEE expansion: v_sqrt_f32, then pick among {s-1ulp, s, s+1ulp} by the sign
// of the fused residuals.
__device__ __forceinline__ float sqrt_rn(float x) {
    const bool tiny = x < 0x1p-96f;
    const float xs = tiny ? x * 0x1p+32f : x;
    float s = __builtin_amdgcn_sqrtf(xs);
    const int si = __float_as_int(s);
    const float s_dn = __int_as_float(si - 1), s_up = __int_as_float(si + 1);
    const float r_dn = fmaf(-s_dn, s, xs);
    const float r_up = fmaf(-s_up, s, xs);
    s = (r_dn <= 0.f) ? s_dn : s;
    s = (r_up > 0.f) ? s_up : s;
    s = tiny ? s * 0x1p-16f : s;
    return (xs == 0.f || xs == INFINITY) ? xs : s;
}

// |x|^2 of a 3-vector in ATen order: ((0 + x0^2) + x1^2) + x2^2
__device__ __forceinline__ float sumsq3(float x, float y, float z) {
    float a = x * x, b = y * y, c = z * z;
    return (a + b) + c;
}

// squared distance of 3-vectors, matmul form (torch.cdist default for >25 rows):
// fma chain over [-2a,|a|^2,1].[b,1,|b|^2], clamp at 0.  `a` is the ROW operand.
__device__ __forceinline__ float d2_mm3(float ax, float ay, float az, float na, float bx, float by, float bz,
                                        float nb) {
    float acc = fmaf(-2.f * ax, bx, 0.f);
    acc = fmaf(-2.f * ay, by, acc);
    acc = fmaf(-2.f * az, bz, acc);
    acc = acc + na;
    acc = acc + nb;
    return acc > 0.f ? acc : 0.f;
}

// squared distance of 3-vectors, difference form, no contraction: (dx^2 + dy^2) + dz^2
__device__ __forceinline__ float d2_diff3(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return (xx + yy) + zz;
}

// ---- XCD-aware block numbering.  Workgroups are dealt round-robin over the 8 XCDs (hardware block b runs on XCD b % 8),
// and every XCD has its own 4 MiB L2.  xcd_remap turns the hardware block number into a logical one such that each XCD
// works through ONE contiguous range of logical blocks, in order: blocks that gather from the same cloud (consecutive
// logical blocks) then share an L2 instead of pulling the cloud's rows into all eight.  Used by the soft-correspondence
// sweeps (a pair's key planes are re-read by its 8 query tiles).  For the gather kernels of the pair path (pass B of K1,
// pooling, Deformer rows, xyz kNN) it was measured neutral — their L2 misses are served by the Infinity Cache — and for
// the grid Chamfer kernel harmful (a contiguous range per XCD = whole query groups, whose costs differ: 2.5 -> 4.9 ms).
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    int q = nwg / 8, r = nwg % 8, xcd = orig % 8;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + orig / 8;
}
// ---- cross-lane exchanges without the LDS round trip of __shfl_xor (ds_bpermute): DPP within a 16-lane row,
// v_permlane16_swap / v_permlane32_swap (gfx950) across rows.  sum16 / sum32 add the SAME partners as the butterfly
// `for (o = 1; o < n; o <<= 1) v += __shfl_xor(v, o)` — after the xor-1 and xor-2 steps a quad holds one value, so the
// mirror within 8 / 16 lanes fetches the other quad's / half's value — and give the same bits.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_xor16(float v) {   // value of lane ^ 16
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // r[0]: odd rows <- the even row below, r[1]: even rows <- the odd row above
    return __uint_as_float((threadIdx.x & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor32(float v) {   // value of lane ^ 32
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float sum16(float v) {   // sum over the 16 lanes of a DPP row, in every lane of it
    v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_move<0x141>(v);   // row_half_mirror: the other quad of the 8
    v += dpp_move<0x140>(v);   // row_mirror: the other 8 of the 16
    return v;
}
__device__ __forceinline__ float sum32(float v) { v = sum16(v); return v + lane_xor16(v); }   // per 32-lane half

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace dvm
