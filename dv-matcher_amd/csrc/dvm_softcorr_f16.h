// dvm_softcorr_f16.h — what the two translation units of the fp16-split soft-correspondence sweep share
// (dvm_softcorr_f16.hip: splitting, first form of pass A, pass B, launchers; dvm_softcorr_coarse.hip: the one-plane screen).
#pragma once
#include "dvm_common.h"

namespace dvm {
namespace k1 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr float LOG2E = 1.4426950408889634f;
constexpr int HB_D = 128;
constexpr int HB_ROWB = 2 * HB_D * 2;   // 512 B per row: planes h | m, 128 fp16 each
constexpr int HB_KT = 64;               // keys per LDS tile (two 32-key sub-tiles)
constexpr int HB_WAVES = 8, HB_QB = 32 * HB_WAVES, HB_THREADS = 64 * HB_WAVES;
constexpr int HB_GLDS_PER_WAVE = HB_KT * HB_ROWB / 1024 / HB_WAVES;  // 4 LDS-DMA pieces (1 KiB = 2 rows) per wave per tile
constexpr int HB_STAGE = 16 * 64;       // floats per wave
constexpr int HB_KC = 12;               // candidates kept per row (top-10 + 2 of margin)
constexpr size_t HB_LDS_BYTES = (size_t)2 * HB_KT * HB_ROWB + 2 * HB_KT * sizeof(float) + (size_t)HB_WAVES * HB_STAGE * sizeof(float);
constexpr float HB_ERR = 2.5e-5f;  // |d2_passA - d2_chain| <= HB_ERR (|q|^2 + |k|^2): gamma_128 of both fp32 accumulations
                                   // (2 x 7.7e-6) + the dropped split terms (1.4e-6), with room to spare


// power of two s with max|x| * s in [2^11, 2^12): well inside fp16's range, the m-plane of every element within 2^-8 of
// the largest stays a normal fp16 number (smaller ones keep an absolute error of 2^-25 in scaled units: far below the
// error bound, which is relative to the largest norms), and |x s|^2 summed over 128 channels stays below 2^31 — the
// range of the three-piece fp16 representation of the norms in the second sweep form
__device__ __forceinline__ int scale_exp(int maxbits) {  // log2(s)
    const int e = ((maxbits >> 23) & 0xff) - 127;
    int k = 11 - e;
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    return maxbits == 0 ? 0 : k;
}
__device__ __forceinline__ float pow2i(int k) { return __int_as_float((127 + k) << 23); }

// Sorted candidate list with (key, column) packed into one double: high word = the float key's bits, low word =
// the column.  Doubles with the same sign order like their bit patterns, so a compare-swap of two entries is
// v_min_f64 + v_max_f64 (instead of a compare and four selects) and ties break on the column for free.
// Keys are squared distances: >= 0 up to rounding (a slightly negative key only reverses its own tie order).
// (plain v_min_f64 / v_max_f64: the C fmin/fmax add a canonicalising v_max_f64 x, x per operand; no NaNs here)
__device__ __forceinline__ double min64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double max64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int K>
struct PackedBest {
    double e[K];
    static __device__ __forceinline__ double pack(float key, int col) { return __hiloint2double(__float_as_int(key), col); }
    static __device__ __forceinline__ float key_of(double x) { return __int_as_float(__double2hiint(x)); }
    static __device__ __forceinline__ int col_of(double x) { return __double2loint(x); }
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int t = 0; t < K; ++t) e[t] = pack(INFINITY, 0x7fffffff);
    }
    __device__ __forceinline__ float key(int t) const { return key_of(e[t]); }
    // returns the entry that is outside the list afterwards (the evicted worst, or x itself)
    __device__ __forceinline__ double insert(double x) {
        const double out = max64(e[K - 1], x);
        e[K - 1] = min64(e[K - 1], x);
#pragma unroll
        for (int p = K - 1; p > 0; --p) {
            const double lo = min64(e[p - 1], e[p]), hi = max64(e[p - 1], e[p]);
            e[p - 1] = lo;
            e[p] = hi;
        }
        return out;
    }
};

// ---------------------------------------------------------------- pass A arguments
struct HBGroup {
    const char *qp, *kp;     // planes of the query / key side [B][rows][512]
    const int *qmax, *kmax;  // bit patterns of max|x| of either side (the split's scale)
    const float *nq, *nk;    // |.|^2 (ATen order); nk padded to whole tiles with +inf: [B][Mpad]
    int N, M, Mpad, tiles;
    int32_t *cidx;           // [B][N][HB_KC]
    float *cd2;              // [B][N][HB_KC] approximate squared distances, ascending
    float *lsum;             // [B][N][2] = (sum exp(s - cref), cref)
};
struct HBArgs {
    HBGroup g[2];
    int blocks0;
    float neg_alpha, cutw;
    const int *route;        // per (group, batch entry): which kernel sweeps it (K1_ROUTE_*), or nullptr = this launch takes all
    int nb;                  // batch entries per group
};
// Which pass-A kernel sweeps a (direction, pair): set per launch by k1_probe_kernel from the pair's own distance statistics
constexpr int K1_ROUTE_FULL = 0;     // first form, every softmax term (flat rows: nearly every column lies within the cut)
constexpr int K1_ROUTE_LEAN = 1;     // first form, lean
constexpr int K1_ROUTE_COARSE = 3;   // coarse screen (one fp16 plane; dvm_softcorr_coarse.hip): no softmax term outside the certified list

// ---------------------------------------------------------------- coarse screen (dvm_softcorr_coarse.hip)
constexpr int K1_KC_COARSE = 16;     // candidates per row handed to pass B (the other forms: HB_KC)
// |d2_coarse - d2_chain| <= HC_ERR (|q|^2 + |k|^2), d2_chain = the reference's fp32 chain (what pass B evaluates):
//   one-plane product: x s = h + e, |e| <= u |x s|, u = 2^-11 (fp16 round to nearest; elements below 2^-14 of the largest keep an
//     absolute error of 2^-25 in scaled units, as in the three-product forms): |q.k - qh.kh| <= (2 u + u^2) sum |q_i k_i|
//     <= (2^-10 + 2^-22) |q||k|, twice that in d2, and |q||k| <= (|q|^2 + |k|^2) / 2:          9.77e-4
//   the list entry keeps 19 bits of the accumulator (rounded down): 2^-14 (d2 + 2^-8 |q|^2), d2 <= 2 (|q|^2 + |k|^2):  1.23e-4
//   fp32 accumulation of the matrix instructions and of the chain (gamma_130 of sum |2 q_i k_i| + the norms, both):   1.6e-5
constexpr float HC_ERR = 1.12e-3f;
bool coarse_supports(int N, int M);
void launch_coarse(const HBArgs &a, const char *knf0, const char *knf1, const int *amaxc, int blocks, hipStream_t s);


// key-side norm fragments of the coarse screen [B][Mpad][32 B] (dvm_softcorr_coarse.hip)
// one launch for the norms of both sides (dvm_softcorr_coarse.hip::norm_prep_kernel): side sd has rows[sd] norms per batch element,
// padded to pad[sd]; nmax [B] (zeroed by the caller) always; npad [B][pad] / nfrag [B][pad][32 B] where not NULL; zero[]: counters to clear
struct NormPrep {
    const float *nrm[2];
    int rows[2], pad[2];
    float *nmax[2], *npad[2];
    char *nfrag[2];
    const int *amax;
    int32_t *zero[2];
};
void launch_norm_prep(const NormPrep &a, int B, hipStream_t s);

}  // namespace k1
}  // namespace dvm
