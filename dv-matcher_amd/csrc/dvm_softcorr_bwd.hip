// dvm_softcorr_bwd.hip — backward twin of the soft-correspondence kernel (SURVEY §8b "backward twins";
// reference: autograd through models/loss.py:110-114 `softmax(-alpha * cdist)` followed by the top-k
// keep of models/loss.py:1339-1347).
//
// Forward (dvm_softcorr.hip):  D_ij = |f1_i - f2_j|,  S = neg_alpha * D,  P_ij = exp(S_ij - smax_i) / l_i,
// outputs val_t = P_{i, idx_t} for the row's top-k.  With g_t = dL/dval_t, gp_t = g_t * val_t, G_i = sum_t gp_t:
//     dL/dS_ij = [j = idx_t] gp_t  -  P_ij * G_i
//     W_ij     = neg_alpha * dL/dS_ij / D_ij              (0 where D_ij = 0, like cdist's backward)
//     df1_i    = sum_j W_ij (f1_i - f2_j)   ,   df2_j = sum_i W_ij (f2_j - f1_i)
// The N x M matrices never exist in HBM: the dense term (-P*G) is recomputed tile by tile from row_smax /
// row_sum, flash-attention style, once row-major (df1) and once column-major (df2); the k-sparse term is a
// gather/scatter kernel.  Per 32x32 tile: 64 fp32 MFMAs (32x32x2) rebuild the distances, ~13 VALU per entry
// turn them into W in the accumulator layout, and 64 more MFMAs apply W to the staged rows — W's C-layout
// registers are fed straight back as the A operand (the contraction index is permuted consistently on the
// B side), so W never touches LDS.
#include <algorithm>

#include "dvm_common.h"

namespace dvm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void launch_rownorm2(const float *x, int rows, int K, float *out, hipStream_t s);

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int BW_D = 128;
constexpr int BW_KT = 64;            // inner rows per LDS tile (two 32-row sub-tiles)
constexpr int BW_LDK = BW_D + 4;     // padded row
constexpr int BW_WAVES = 4;
constexpr int BW_OB = 32 * BW_WAVES;  // outer rows per workgroup
constexpr int BW_THREADS = 64 * BW_WAVES;
constexpr int BW_LD_PER_THREAD = BW_KT * BW_D / 4 / BW_THREADS;  // 8 float4 per thread per tile
constexpr int BW_TILE_FLOATS = BW_KT * BW_LDK + 3 * BW_KT;       // rows + {norm, c2, coef}
constexpr size_t BW_LDS_BYTES = ((size_t)2 * BW_TILE_FLOATS + BW_OB) * sizeof(float);


// "outer" rows live in registers (32 per wave), "inner" rows stream through LDS.  The softmax row statistics
// (c2 = smax*log2e, coef = -neg_alpha*G/l) belong to f1's rows: they sit on the outer side in the df1 pass
// and on the inner side in the df2 pass; the other side's pointers are null (c2 = 0, coef = 1).
struct SBGroup {
    const float *fo, *fi, *no, *ni;
    const float *c2o, *coefo, *c2i, *coefi;
    float *dout;
    int No, Ni, tiles_o;
};
struct SBArgs {
    SBGroup g[2];
    int blocks0;  // B * g[0].tiles_o * split
    int split;    // the inner loop is cut into `split` pieces (small batches: fill the chip); outputs are atomics
    float a2;     // neg_alpha * log2(e)
};

__global__ __launch_bounds__(BW_THREADS, 2) void softcorr_bwd_mfma_kernel(const SBArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *const rsum = smem + 2 * BW_TILE_FLOATS;  // [BW_OB]

    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int grp = lid >= args.blocks0 ? 1 : 0;
    lid -= grp ? args.blocks0 : 0;
    const SBGroup &G = args.g[grp];
    const int No = G.No, Ni = G.Ni;
    const int sp = lid % args.split;
    lid /= args.split;
    const int ot = lid % G.tiles_o, b = lid / G.tiles_o;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    const float a2 = args.a2;

    const float *ibase = G.fi + (size_t)b * Ni * BW_D;
    const float *inb = G.ni + (size_t)b * Ni;
    const float *ic2 = G.c2i ? G.c2i + (size_t)b * Ni : nullptr;
    const float *icf = G.coefi ? G.coefi + (size_t)b * Ni : nullptr;

    const int orow = ot * BW_OB + wave * 32 + r32;
    const int orc = orow < No ? orow : No - 1;
    const float *op = G.fo + ((size_t)b * No + orc) * BW_D;
    float q[BW_D / 2];  // B operand of the distance GEMM: q[s] = -2 * fo[row][channel(s, h)]
#pragma unroll
    for (int c = 0; c < BW_D / 4; ++c) {
        f32x4 v = *(const f32x4 *)(op + 4 * c);
        q[2 * c] = -2.f * (h ? v.y : v.x);
        q[2 * c + 1] = -2.f * (h ? v.w : v.z);
    }
    const float nrm_o = G.no[(size_t)b * No + orc];
    const float c2_o = G.c2o ? G.c2o[(size_t)b * No + orc] : 0.f;
    const float coef_o = orow < No ? (G.coefo ? G.coefo[(size_t)b * No + orc] : 1.f) : 0.f;

    const int ntiles = (Ni + BW_KT - 1) / BW_KT;
    const int per = (ntiles + args.split - 1) / args.split;
    const int t0 = sp * per, t1 = min(ntiles, t0 + per);
    if (t0 >= t1) return;  // uniform over the workgroup

    f32x4 pre[BW_LD_PER_THREAD];
    float pres = 0.f;
    auto issue_loads = [&](int t) {
        const int j0 = t * BW_KT;
#pragma unroll
        for (int e = 0; e < BW_LD_PER_THREAD; ++e) {
            int id = tid + e * BW_THREADS;
            int r = id >> 5, c = id & 31;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j0 + r < Ni) v = *(const f32x4 *)(ibase + (size_t)(j0 + r) * BW_D + 4 * c);
            pre[e] = v;
        }
        if (tid < 3 * BW_KT) {  // threads 0..63: norm, 64..127: c2, 128..191: coef
            const int which = tid >> 6, j = j0 + (tid & 63);
            const bool ok = j < Ni;
            if (which == 0) pres = ok ? inb[j] : 0.f;
            else if (which == 1) pres = (ok && ic2) ? ic2[j] : 0.f;
            else pres = ok ? (icf ? icf[j] : 1.f) : 0.f;
        }
    };
    auto commit_loads = [&](int buf) {
        float *kt = smem + buf * BW_TILE_FLOATS;
#pragma unroll
        for (int e = 0; e < BW_LD_PER_THREAD; ++e) {
            int id = tid + e * BW_THREADS;
            int r = id >> 5, c = id & 31;
            // position p < 64 holds channel 2p, position 64 + p holds channel 2p + 1
            float2 ev = {pre[e].x, pre[e].z}, od = {pre[e].y, pre[e].w};
            *(float2 *)(kt + r * BW_LDK + 2 * c) = ev;
            *(float2 *)(kt + r * BW_LDK + 64 + 2 * c) = od;
        }
        if (tid < 3 * BW_KT) kt[BW_KT * BW_LDK + tid] = pres;
    };

    f32x16 acc2[4];  // [position block cb] : out[outer row (C layout)][position 4*r32 + cb]
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[cb][r] = 0.f;
    float rl = 0.f;  // this lane's share of sum_t W[t][o]

    issue_loads(t0);
    commit_loads(0);
    __syncthreads();

    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        const float *kt = smem + buf * BW_TILE_FLOATS;
        if (t + 1 < t1) issue_loads(t + 1);
#pragma unroll 1
        for (int sub = 0; sub < 2; ++sub) {
            const float *arow = kt + (sub * 32 + r32) * BW_LDK + h * 64;
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                f32x4 a = *(const f32x4 *)(arow + 4 * c);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, q[4 * c], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, q[4 * c + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, q[4 * c + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, q[4 * c + 3], acc, 0, 0, 0);
            }
            // this lane's 16 inner rows: local row = (r&3) + 8*(r>>2) + 4*h
            const float *sc = kt + BW_KT * BW_LDK + sub * 32 + 4 * h;
            float w[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 nb = *(const f32x4 *)(sc + 8 * g);
                const f32x4 cc = *(const f32x4 *)(sc + BW_KT + 8 * g);
                const f32x4 cf = *(const f32x4 *)(sc + 2 * BW_KT + 8 * g);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = 4 * g + u;
                    const float v = fmaxf((acc[r] + nrm_o) + nb[u], 0.f);
                    const float D = sqrt_rn(v);
                    const float e = __builtin_amdgcn_exp2f(fmaf(D, a2, -(c2_o + cc[u])));
                    const float wv = (coef_o * cf[u]) * e * __builtin_amdgcn_rcpf(D);
                    // padding rows of the last tile carry coef 0 but zero features: their "distance" |f_o| can be far
                    // below the row minimum, e overflows to +inf and 0 * inf would poison the whole output row
                    w[r] = (v > 0.f && cf[u] != 0.f) ? wv : 0.f;
                    rl += w[r];
                }
            }
            // apply: out[o][pos] += sum_t W[t][o] * X[t][pos]; step r contracts t = (r&3)+8*(r>>2)+4*h
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int trow = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const f32x4 x = *(const f32x4 *)(kt + trow * BW_LDK + 4 * r32);
                acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], x.x, acc2[0], 0, 0, 0);
                acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], x.y, acc2[1], 0, 0, 0);
                acc2[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], x.z, acc2[2], 0, 0, 0);
                acc2[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], x.w, acc2[3], 0, 0, 0);
            }
        }
        if (t + 1 < t1) commit_loads(buf ^ 1);
        __syncthreads();
    }

    // d_out[o] += (sum_t W[t][o]) * f_o - acc2[o]
    const float rtot = rl + __shfl_xor(rl, 32, 64);
    if (h == 0) rsum[wave * 32 + r32] = rtot;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int row = ot * BW_OB + wave * 32 + o;
        if (row >= No) continue;
        const float rr = rsum[wave * 32 + o];
        const float *fo = G.fo + ((size_t)b * No + row) * BW_D;
        float *dst = G.dout + ((size_t)b * No + row) * BW_D;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int pos = 4 * r32 + cb;
            const int ch = pos < 64 ? 2 * pos : 2 * (pos - 64) + 1;
            unsafeAtomicAdd(dst + ch, rr * fo[ch] - acc2[cb][r]);
        }
    }
}

// Any d (multiple of 4, <= 512): one wave per outer row, lanes own channels lane + 64u.  Reference-quality
// fallback and the cross-check for the MFMA kernel (variant 1).
struct SBScalarArgs {
    SBGroup g[2];
    long rows0;      // B * g[0].No
    long rows_total;  // rows0 + B * g[1].No
    int d;
    float a2;
};

__global__ __launch_bounds__(256) void softcorr_bwd_scalar_kernel(const SBScalarArgs args) {
    const int lane = threadIdx.x & 63;
    long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= args.rows_total) return;
    const int grp = gw >= args.rows0 ? 1 : 0;
    gw -= grp ? args.rows0 : 0;
    const SBGroup &G = args.g[grp];
    const int No = G.No, Ni = G.Ni, d = args.d;
    const int b = (int)(gw / No), row = (int)(gw % No);
    const float *fo = G.fo + ((size_t)b * No + row) * d;
    float ov[8], av[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int c = lane + 64 * u;
        ov[u] = c < d ? fo[c] : 0.f;
        av[u] = 0.f;
    }
    const float nrm_o = G.no[(size_t)b * No + row];
    const float c2_o = G.c2o ? G.c2o[(size_t)b * No + row] : 0.f;
    const float coef_o = G.coefo ? G.coefo[(size_t)b * No + row] : 1.f;
    float rsum = 0.f;
    for (int j = 0; j < Ni; ++j) {
        const float *fi = G.fi + ((size_t)b * Ni + j) * d;
        float xv[8], part = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = lane + 64 * u;
            xv[u] = c < d ? fi[c] : 0.f;
            part = fmaf(ov[u], xv[u], part);
        }
        const float dot = wave_sum(part);
        const float v = fmaxf((-2.f * dot + nrm_o) + G.ni[(size_t)b * Ni + j], 0.f);
        const float D = sqrt_rn(v);
        const float c2 = c2_o + (G.c2i ? G.c2i[(size_t)b * Ni + j] : 0.f);
        const float cf = coef_o * (G.coefi ? G.coefi[(size_t)b * Ni + j] : 1.f);
        const float w = v > 0.f ? cf * exp2f(fmaf(D, args.a2, -c2)) / D : 0.f;
        rsum += w;
#pragma unroll
        for (int u = 0; u < 8; ++u) av[u] = fmaf(w, xv[u], av[u]);
    }
    float *dst = G.dout + ((size_t)b * No + row) * d;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int c = lane + 64 * u;
        if (c < d) unsafeAtomicAdd(dst + c, rsum * ov[u] - av[u]);
    }
}

// Per f1 row: G = sum_t g_t*val_t, the row's dense-term coefficients, and the k-sparse term
// (gathered rows of f2; scatter-add into df2).  One wave per row, lanes own channels lane + 64u.
__global__ __launch_bounds__(256) void softcorr_bwd_prep_kernel(const float *__restrict__ f1, const float *__restrict__ f2,
                                                                const float *__restrict__ pi_val,
                                                                const int32_t *__restrict__ pi_idx,
                                                                const float *__restrict__ gval,
                                                                const float *__restrict__ row_smax,
                                                                const float *__restrict__ row_sum, int B, int N, int M, int d,
                                                                int topk, float neg_alpha, float *__restrict__ coef,
                                                                float *__restrict__ c2, float *__restrict__ df1,
                                                                float *__restrict__ df2) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * N) return;
    const int b = (int)(row / N);
    const float *a = f1 + (size_t)row * d;
    float av[8], own[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int c = lane + 64 * u;
        av[u] = c < d ? a[c] : 0.f;
        own[u] = 0.f;
    }
    float G = 0.f;
    for (int t = 0; t < topk; ++t) {
        const int j = pi_idx[(size_t)row * topk + t];
        const float gp = gval[(size_t)row * topk + t] * pi_val[(size_t)row * topk + t];
        G += gp;
        if (j < 0 || j >= M || gp == 0.f) continue;  // uniform over the wave
        const float *x = f2 + ((size_t)b * M + j) * d;
        float dx[8], part = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = lane + 64 * u;
            dx[u] = c < d ? av[u] - x[c] : 0.f;
            part = fmaf(dx[u], dx[u], part);
        }
        const float D = sqrt_rn(wave_sum(part));
        if (!(D > 0.f)) continue;
        const float w = neg_alpha * gp / D;
        float *dst = df2 + ((size_t)b * M + j) * d;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = lane + 64 * u;
            own[u] = fmaf(w, dx[u], own[u]);
            if (c < d) unsafeAtomicAdd(dst + c, -w * dx[u]);
        }
    }
    if (lane == 0) {
        coef[row] = -neg_alpha * G / row_sum[row];
        c2[row] = row_smax[row] * LOG2E;
    }
    float *dst = df1 + (size_t)row * d;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int c = lane + 64 * u;
        if (c < d) unsafeAtomicAdd(dst + c, own[u]);
    }
}

}  // namespace
}  // namespace dvm

using namespace dvm;

DVM_EXPORT size_t dvm_softcorr_bwd_workspace_bytes(int B, int N, int M, int d) {
    (void)d;
    return 3 * align_up((size_t)B * N * sizeof(float)) + align_up((size_t)B * M * sizeof(float));
}

DVM_EXPORT int dvm_softcorr_bwd_f32(const float *f1, const float *f2, int B, int N, int M, int d, float neg_alpha, int topk,
                                    const float *pi_val, const int32_t *pi_idx, const float *row_smax, const float *row_sum,
                                    const float *g_val, float *d_f1, float *d_f2, int variant, void *ws, size_t ws_bytes,
                                    void *stream) {
    DVM_REQUIRE(f1 && f2 && pi_val && pi_idx && row_smax && row_sum && g_val && d_f1 && d_f2, "dvm_softcorr_bwd_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1 && M >= 1, "dvm_softcorr_bwd_f32: empty input (B=%d N=%d M=%d)", B, N, M);
    DVM_REQUIRE(d >= 4 && d % 4 == 0 && d <= 512, "dvm_softcorr_bwd_f32: d=%d unsupported (need d%%4==0, 4<=d<=512)", d);
    DVM_REQUIRE(topk >= 1 && topk <= 16, "dvm_softcorr_bwd_f32: topk=%d unsupported (1..16)", topk);
    DVM_REQUIRE(neg_alpha < 0.f, "dvm_softcorr_bwd_f32: neg_alpha must be negative (got %g)", (double)neg_alpha);
    DVM_REQUIRE(variant >= 0 && variant <= 2, "dvm_softcorr_bwd_f32: bad variant %d", variant);
    DVM_REQUIRE(variant != 2 || d == BW_D, "dvm_softcorr_bwd_f32: MFMA variant needs d == 128");
    Arena ar(ws, ws_bytes);
    float *n1 = ar.take<float>((size_t)B * N);
    float *coef = ar.take<float>((size_t)B * N);
    float *c2 = ar.take<float>((size_t)B * N);
    float *n2 = ar.take<float>((size_t)B * M);
    if (!ar.ok()) {
        set_error("dvm_softcorr_bwd_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(d_f1, 0, (size_t)B * N * d * sizeof(float), s);
    (void)hipMemsetAsync(d_f2, 0, (size_t)B * M * d * sizeof(float), s);
    launch_rownorm2(f1, B * N, d, n1, s);
    launch_rownorm2(f2, B * M, d, n2, s);
    hipLaunchKernelGGL(softcorr_bwd_prep_kernel, dim3((unsigned)(((size_t)B * N + 3) / 4)), dim3(256), 0, s, f1, f2, pi_val, pi_idx,
                       g_val, row_smax, row_sum, B, N, M, d, topk, neg_alpha, coef, c2, d_f1, d_f2);
    const float a2 = neg_alpha * LOG2E;
    const bool mfma = (variant == 2) || (variant == 0 && d == BW_D);
    if (mfma) {
        SBArgs a;
        a.g[0] = SBGroup{f1, f2, n1, n2, c2, coef, nullptr, nullptr, d_f1, N, M, (N + BW_OB - 1) / BW_OB};
        a.g[1] = SBGroup{f2, f1, n2, n1, nullptr, nullptr, c2, coef, d_f2, M, N, (M + BW_OB - 1) / BW_OB};
        int split = 1;
        const int base = B * (a.g[0].tiles_o + a.g[1].tiles_o);
        const int min_tiles = (std::min(N, M) + BW_KT - 1) / BW_KT;
        while (base * split < 512 && split < 8 && min_tiles / (2 * split) >= 4) split *= 2;
        a.split = split;
        a.blocks0 = B * a.g[0].tiles_o * split;
        a.a2 = a2;
        ensure_dyn_lds((const void *)softcorr_bwd_mfma_kernel, (int)BW_LDS_BYTES);
        hipLaunchKernelGGL(softcorr_bwd_mfma_kernel, dim3(base * split), dim3(BW_THREADS), BW_LDS_BYTES, s, a);
    } else {
        SBScalarArgs a;
        a.g[0] = SBGroup{f1, f2, n1, n2, c2, coef, nullptr, nullptr, d_f1, N, M, 0};
        a.g[1] = SBGroup{f2, f1, n2, n1, nullptr, nullptr, c2, coef, d_f2, M, N, 0};
        a.rows0 = (long)B * N;
        a.rows_total = (long)B * N + (long)B * M;
        a.d = d;
        a.a2 = a2;
        const size_t rows = (size_t)B * N + (size_t)B * M;
        hipLaunchKernelGGL(softcorr_bwd_scalar_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a);
    }
    DVM_CHECK_LAUNCH("softcorr_bwd");
    return DVM_OK;
}
