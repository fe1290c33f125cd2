// dvm_softcorr.hip — K1: fused feature-distance + row-softmax + top-k soft correspondence.
//
// Replaces knnsearch_t_grad + topk_pi (reference models/loss.py:110-114, 1339-1347, 1404-1407):
// the N x M distance / softmax matrices are never written to HBM.  Squared distances are the
// matmul form of torch.cdist, [-2a,|a|^2,1].[b,1,|b|^2], evaluated as a k-ordered fp32 fma
// chain — on the matrix cores v_mfma_f32_32x32x2_f32 computes exactly that chain, so the
// distances equal the reference's CPU (MKL sgemm) values bit for bit and the top-k columns /
// arg-max map are bit-exact integers.
//
// Data layout: features row-major [B][N][d] fp32 in HBM; one workgroup owns 128 query rows
// (4 waves x 32) and sweeps all M keys through a double-buffered, k-deinterleaved LDS tile;
// the accumulator tile is [key][query] so that a query's candidates sit in one lane's
// registers (top-k insertion and online softmax need no cross-lane traffic until the end).
#include <stdlib.h>

#include "dvm_common.h"

namespace dvm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float LOG2E = 1.4426950408889634f;


// ------------------------------------------------------------------ row norms
__global__ void rownorm2_kernel(const float *__restrict__ x, int rows, int K, float *__restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    out[i] = aten_sumsq_row(x + (size_t)i * K, K);
}

// K = 128, coalesced: 32 lanes per row.  Same additions in the same order as aten_sumsq_row (K = 128:
// vec_size 16, size_ilp 4 -> lane l, ILP slot k accumulate x[(4i+k)*8 + l]^2 over i = 0..3, then
// ((k0 + k1) + k2) + k3, then lanes 0..7 added in order): element 32 i + 8 k + l sits in lane 8 k + l.
// All cross-lane traffic stays on the DPP network / v_permlane16_swap (17 dependent ds_bpermute round trips per wave
// otherwise: the kernel ran at 1.3 TB/s).
__device__ __forceinline__ float dpp_ror8(float v) {      // lane j of a 16-lane row <- lane (j + 8) % 16
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_shr1_zero(float v) {  // lane j <- lane j - 1 of its row; lane 0 of a row <- 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));
}
constexpr int RN_ROWS = 8;   // rows per 32-lane half: 4 KB of loads in flight per half before the first reduction
__global__ __launch_bounds__(256) void rownorm2_k128_kernel(const float *__restrict__ x, int rows, float *__restrict__ out,
                                                            int *__restrict__ absmax) {
    const int lane = threadIdx.x & 63, l32 = lane & 31;
    const long half = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;   // 32-lane half -> rows [half * 8, half * 8 + 8)
    float v[RN_ROWS][4];
#pragma unroll
    for (int q = 0; q < RN_ROWS; ++q) {
        const long row0 = half * RN_ROWS + q, row = row0 < rows ? row0 : rows - 1;
        const float *p = x + row * 128 + l32;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[q][i] = p[32 * i];
    }
    float am = 0.f;
#pragma unroll
    for (int q = 0; q < RN_ROWS; ++q) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s = s + v[q][i] * v[q][i];
            am = fmaxf(am, fabsf(v[q][i]));
        }
        // lanes 8k + l of a 32-lane half: k = 1 is 8 lanes up in the same 16-lane row, k = 2, 3 sit in the next row
        const unsigned su = __float_as_uint(s);
        const auto sw = __builtin_amdgcn_permlane16_swap(su, su, false, false);   // sw[1]: even rows <- the odd row above them
        const float up = __uint_as_float(sw[1]);
        const float r = ((s + dpp_ror8(s)) + up) + dpp_ror8(up);  // valid in lanes l32 < 8 (k = 0)
        // fin = (((r0 + r1) + r2) ... + r7): u_j <- u_{j-1} + r_j seven times leaves it in lane 7 of the half
        float u = r;
#pragma unroll
        for (int st = 0; st < 7; ++st) u = dpp_shr1_zero(u) + r;
        const long row0 = half * RN_ROWS + q;
        if (l32 == 7 && row0 < rows) out[row0] = u;
    }
    if (absmax) {  // optional: bit pattern of max |x| of the tensor (the fp16 split's scale, dvm_softcorr_f16.hip)
        int m = __float_as_int(am);   // non-negative floats order like their bit patterns
        m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x111, 0xf, 0xf, false));  // row_shr:1
        m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x112, 0xf, 0xf, false));  // row_shr:2
        m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x114, 0xf, 0xf, false));  // row_shr:4
        m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x118, 0xf, 0xf, false));  // row_shr:8
        m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x142, 0xa, 0xf, false));  // row_bcast:15
        m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x143, 0xc, 0xf, false));  // row_bcast:31 -> lane 63 = the wave's max
        // 256 slots (one address would serialise the waves); only waves that would raise a slot's maximum touch it
        int *slot = absmax + (blockIdx.x & 255);
        if (lane == 63 && m > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, m);
    }
}

// ------------------------------------------------------- per-row running state
template <int TOPK>
struct RowState {
    KBest<TOPK, float> kb;  // keyed on the post-sqrt distance, as torch.topk sees it
    float smax;             // running max of s = d * neg_alpha   (-inf initially)
    float l;                // sum exp(s - smax)
    __device__ __forceinline__ void init() {
        kb.init(INFINITY);
        smax = -INFINITY;
        l = 0.f;
    }
    __device__ __forceinline__ void rescale(float new_smax) {
        if (new_smax > smax) {
            l = l * exp2f((smax - new_smax) * LOG2E);  // smax=-inf, l=0 -> 0*0
            smax = new_smax;
        }
    }
    __device__ __forceinline__ void merge(const RowState &o) {
        float m = fmaxf(smax, o.smax);
        float a = (smax == -INFINITY) ? 0.f : l * exp2f((smax - m) * LOG2E);
        float b = (o.smax == -INFINITY) ? 0.f : o.l * exp2f((o.smax - m) * LOG2E);
        l = a + b;
        smax = m;
#pragma unroll
        for (int t = 0; t < TOPK; ++t) kb.insert_lex(o.kb.key[t], o.kb.idx[t]);
    }
};

template <int TOPK>
__device__ __forceinline__ void store_row(const RowState<TOPK> &st, int topk, int M, float neg_alpha, float *val,
                                          int32_t *idx, float *row_smax, float *row_sum) {
    float inv = 1.0f / st.l;
#pragma unroll
    for (int t = 0; t < TOPK; ++t) {
        if (t < topk) {
            bool live = t < M;
            float s = st.kb.key[t] * neg_alpha;
            val[t] = live ? exp2f((s - st.smax) * LOG2E) * inv : 0.f;
            idx[t] = live ? st.kb.idx[t] : 0;
        }
    }
    if (row_smax) *row_smax = st.smax;
    if (row_sum) *row_sum = st.l;
}

// ------------------------------------------------------------ scalar variant
// One thread per query row; keys staged through LDS in tiles of 32; the dot product is an
// explicit k-ordered fmaf chain.  Any d % 4 == 0.  Reference kernel for the MFMA variant and
// the fallback for d != 128.
constexpr int SC_KT = 32;   // keys per tile
constexpr int SC_DC = 32;   // feature chunk held in registers

template <int TOPK>
__global__ __launch_bounds__(128) void softcorr_scalar_kernel(const float *__restrict__ f1, const float *__restrict__ f2,
                                                              const float *__restrict__ n1, const float *__restrict__ n2,
                                                              int N, int M, int d, float neg_alpha, int topk,
                                                              float *__restrict__ pi_val, int32_t *__restrict__ pi_idx,
                                                              float *__restrict__ row_smax, float *__restrict__ row_sum) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [SC_KT][d] keys + [SC_KT] norms
    float *kt = smem;
    float *kn = smem + SC_KT * d;
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int ic = i < N ? i : N - 1;
    const float *q = f1 + ((size_t)b * N + ic) * d;
    const float na = n1[(size_t)b * N + ic];
    const float *kbase = f2 + (size_t)b * M * d;
    RowState<TOPK> st;
    st.init();
    for (int j0 = 0; j0 < M; j0 += SC_KT) {
        __syncthreads();
        for (int e = threadIdx.x; e < SC_KT * d / 4; e += blockDim.x) {
            int r = e / (d / 4), c = e % (d / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j0 + r < M) v = *(const f32x4 *)(kbase + (size_t)(j0 + r) * d + 4 * c);
            *(f32x4 *)(kt + r * d + 4 * c) = v;
        }
        if (threadIdx.x < SC_KT) kn[threadIdx.x] = (j0 + threadIdx.x < M) ? n2[(size_t)b * M + j0 + threadIdx.x] : INFINITY;
        __syncthreads();
        float acc[SC_KT];
#pragma unroll
        for (int j = 0; j < SC_KT; ++j) acc[j] = 0.f;
        for (int c0 = 0; c0 < d; c0 += SC_DC) {
            float qr[SC_DC];
            int cw = d - c0 < SC_DC ? d - c0 : SC_DC;
#pragma unroll
            for (int c = 0; c < SC_DC; c += 4) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (c < cw) v = *(const f32x4 *)(q + c0 + c);
                qr[c] = -2.f * v.x, qr[c + 1] = -2.f * v.y, qr[c + 2] = -2.f * v.z, qr[c + 3] = -2.f * v.w;
            }
#pragma unroll
            for (int j = 0; j < SC_KT; ++j) {
#pragma unroll
                for (int c = 0; c < SC_DC; c += 4) {
                    if (c < cw) {
                        f32x4 kv = *(const f32x4 *)(kt + j * d + c0 + c);
                        acc[j] = fmaf(qr[c], kv.x, acc[j]);
                        acc[j] = fmaf(qr[c + 1], kv.y, acc[j]);
                        acc[j] = fmaf(qr[c + 2], kv.z, acc[j]);
                        acc[j] = fmaf(qr[c + 3], kv.w, acc[j]);
                    }
                }
            }
        }
        // epilogue: distances, online softmax, top-k
        float dd[SC_KT];
        float tmin = INFINITY;
#pragma unroll
        for (int j = 0; j < SC_KT; ++j) {
            float d2 = (acc[j] + na) + kn[j];
            d2 = d2 > 0.f ? d2 : 0.f;
            dd[j] = sqrt_rn(d2);
            tmin = fminf(tmin, dd[j]);
        }
        st.rescale(tmin * neg_alpha);
#pragma unroll
        for (int j = 0; j < SC_KT; ++j) {
            float s = dd[j] * neg_alpha;
            st.l += exp2f((s - st.smax) * LOG2E);
            st.kb.insert(dd[j], j0 + j);
        }
    }
    if (i < N) {
        size_t row = (size_t)b * N + i;
        store_row<TOPK>(st, topk, M, neg_alpha, pi_val + row * topk, pi_idx + row * topk, row_smax ? row_smax + row : nullptr,
                        row_sum ? row_sum + row : nullptr);
    }
}

// ------------------------------------------------------------ dense Pi (API compatibility)
// knnsearch_t_grad as a dense (N x M) matrix for callers that really want it (reference
// models/loss.py:110-114, deform.py:241).  Recomputes the distances tile by tile and normalises
// with the row statistics of the fused kernel.
template <int DUMMY>
__global__ __launch_bounds__(128) void softcorr_dense_kernel(const float *__restrict__ f1, const float *__restrict__ f2,
                                                             const float *__restrict__ n1, const float *__restrict__ n2,
                                                             const float *__restrict__ row_smax, const float *__restrict__ row_sum,
                                                             int N, int M, int d, float neg_alpha, float *__restrict__ P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *kt = smem;
    float *kn = smem + SC_KT * d;
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int ic = i < N ? i : N - 1;
    const float *q = f1 + ((size_t)b * N + ic) * d;
    const float na = n1[(size_t)b * N + ic];
    const float smax = row_smax[(size_t)b * N + ic];
    const float inv = 1.0f / row_sum[(size_t)b * N + ic];
    const float *kbase = f2 + (size_t)b * M * d;
    for (int j0 = 0; j0 < M; j0 += SC_KT) {
        __syncthreads();
        for (int e = threadIdx.x; e < SC_KT * d / 4; e += blockDim.x) {
            int r = e / (d / 4), c = e % (d / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j0 + r < M) v = *(const f32x4 *)(kbase + (size_t)(j0 + r) * d + 4 * c);
            *(f32x4 *)(kt + r * d + 4 * c) = v;
        }
        if (threadIdx.x < SC_KT) kn[threadIdx.x] = (j0 + threadIdx.x < M) ? n2[(size_t)b * M + j0 + threadIdx.x] : INFINITY;
        __syncthreads();
        for (int j = 0; j < SC_KT && j0 + j < M; ++j) {
            float acc = 0.f;
            for (int c = 0; c < d; c += 4) {
                f32x4 qv = *(const f32x4 *)(q + c), kv = *(const f32x4 *)(kt + j * d + c);
                acc = fmaf(-2.f * qv.x, kv.x, acc);
                acc = fmaf(-2.f * qv.y, kv.y, acc);
                acc = fmaf(-2.f * qv.z, kv.z, acc);
                acc = fmaf(-2.f * qv.w, kv.w, acc);
            }
            float d2 = (acc + na) + kn[j];
            d2 = d2 > 0.f ? d2 : 0.f;
            float s = sqrt_rn(d2) * neg_alpha;
            if (i < N) P[((size_t)b * N + i) * M + j0 + j] = exp2f((s - smax) * LOG2E) * inv;
        }
    }
}

// -------------------------------------------------------------- MFMA variant
constexpr int MF_D = 128;
constexpr int MF_KT = 64;              // keys per LDS tile (two 32-key MFMA sub-tiles)
constexpr int MF_LDK = MF_D + 4;       // padded row (floats): 528 B, keeps ds_read_b128 conflict-free
constexpr int MF_QW = 32;              // queries per wave
constexpr int MF_WAVES = 8;             // waves 0-3 and 4-7 pair up on the 4 SIMDs (two per SIMD)
constexpr int MF_QB = MF_QW * MF_WAVES;  // 256 queries per workgroup
constexpr int MF_THREADS = 64 * MF_WAVES;
constexpr int MF_LD_PER_THREAD = MF_KT * MF_D / 4 / MF_THREADS;  // float4 loads per thread per tile = 8
constexpr int MF_STAGE = 16 * 64;  // floats per wave: this sub-tile's 16 squared distances of each lane, [r][lane]
constexpr size_t MF_LDS_BYTES = ((size_t)2 * (MF_KT * MF_LDK + MF_KT) + (size_t)MF_WAVES * MF_STAGE) * sizeof(float);

__device__ __forceinline__ float select16(const float (&v)[16], int b) {
    float a0 = (b & 1) ? v[1] : v[0], a1 = (b & 1) ? v[3] : v[2], a2 = (b & 1) ? v[5] : v[4], a3 = (b & 1) ? v[7] : v[6];
    float a4 = (b & 1) ? v[9] : v[8], a5 = (b & 1) ? v[11] : v[10], a6 = (b & 1) ? v[13] : v[12],
          a7 = (b & 1) ? v[15] : v[14];
    float b0 = (b & 2) ? a1 : a0, b1 = (b & 2) ? a3 : a2, b2 = (b & 2) ? a5 : a4, b3 = (b & 2) ? a7 : a6;
    float c0 = (b & 4) ? b1 : b0, c1 = (b & 4) ? b3 : b2;
    return (b & 8) ? c1 : c0;
}

// XCD-aware block remap (bijective for any grid): blocks that share `orig % 8` share an L2;
// give each XCD a contiguous range of logical ids so the row tiles of one pair reuse its keys
// from that L2.


// One launch covers up to two "groups" (the two directions of a pair batch: (f1 -> f2) and
// (f2 -> f1)), each with its own query/key tensors and outputs.
struct SCGroup {
    const float *q, *k, *nq, *nk;  // queries [B][N][128], keys [B][M][128], their |.|^2
    int N, M, tiles;               // tiles = ceil(N / MF_QB)
    float *val;
    int32_t *idx;
    float *smax, *sum;
};
struct SCArgs {
    SCGroup g[2];
    int blocks0;  // B * g[0].tiles : logical block ids below this belong to group 0
    float neg_alpha;
    float cutw;  // 20 / alpha: distance window above the row minimum whose softmax terms are kept (LEAN)
    int topk;
};

// Epilogue design: per candidate the fast path is {2 add, max, v_sqrt_f32, fma, v_exp_f32, add,
// 2 cmp} — the 1-ulp hardware sqrt feeds only softmax terms whose weight is < 3e-4.  Candidates
// that may enter the top-k (d2 <= threshold^2) or carry a significant weight are re-evaluated
// with the correctly rounded sqrt and the reference's rounding sequence (s = d*neg_alpha, s - c)
// inside a compacted, wave-uniform loop, so the ranking and the dominant terms stay exact.
template <int TOPK, bool LEAN>
__global__ __launch_bounds__(MF_THREADS, 2) void softcorr_mfma_kernel(const SCArgs args) {
    // __launch_bounds__(512, 2): two waves per SIMD, i.e. ONE 512-thread workgroup per CU
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *const ktile0 = smem;                       // [2][MF_KT][MF_LDK]
    float *const knorm0 = smem + 2 * MF_KT * MF_LDK;  // [2][MF_KT]
    float *const stage = knorm0 + 2 * MF_KT + (threadIdx.x >> 6) * MF_STAGE + (threadIdx.x & 63);  // this lane's column

    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int grp = lid >= args.blocks0 ? 1 : 0;
    lid -= grp ? args.blocks0 : 0;
    const SCGroup &G = args.g[grp];
    const int N = G.N, M = G.M;
    const int b = lid / G.tiles;
    const int qt = lid % G.tiles;
    const float neg_alpha = args.neg_alpha;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;

    const float *kbase = G.k + (size_t)b * M * MF_D;
    const float *knb = G.nk + (size_t)b * M;

    // this lane's query row and its B-operand fragment: q[s] = -2 * f1[row][2s + h]
    const int qrow = qt * MF_QB + wave * MF_QW + r32;
    const int qrc = qrow < N ? qrow : N - 1;
    const float *qp = G.q + ((size_t)b * N + qrc) * MF_D;
    float q[MF_D / 2];
#pragma unroll
    for (int c = 0; c < MF_D / 4; ++c) {
        f32x4 v = *(const f32x4 *)(qp + 4 * c);
        q[2 * c] = -2.f * (h ? v.y : v.x);
        q[2 * c + 1] = -2.f * (h ? v.w : v.z);
    }
    const float na = G.nq[(size_t)b * N + qrc];

    KBest<TOPK, float> kb;  // keyed on the correctly rounded distance
    kb.init(INFINITY);
    float cref = -INFINITY;  // running reference shift (~ max of s = d*neg_alpha), softmax is shift-invariant
    float l = 0.f;           // sum exp(s - cref)
    float thr2 = INFINITY;   // conservative squared-distance bound for "may enter the top-k"
    const float a2 = neg_alpha * LOG2E;

    const int ntiles = (M + MF_KT - 1) / MF_KT;
    f32x4 pre[MF_LD_PER_THREAD];
    float pren = 0.f;

    auto issue_loads = [&](int t) {
        int j0 = t * MF_KT;
#pragma unroll
        for (int e = 0; e < MF_LD_PER_THREAD; ++e) {
            int id = tid + e * MF_THREADS;
            int r = id >> 5, c = id & 31;  // 32 float4 per 128-float row
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j0 + r < M) v = *(const f32x4 *)(kbase + (size_t)(j0 + r) * MF_D + 4 * c);
            pre[e] = v;
        }
        if (tid < MF_KT) pren = (j0 + tid < M) ? knb[j0 + tid] : INFINITY;
    };
    auto commit_loads = [&](int buf) {
        float *kt = ktile0 + buf * (MF_KT * MF_LDK);
#pragma unroll
        for (int e = 0; e < MF_LD_PER_THREAD; ++e) {
            int id = tid + e * MF_THREADS;
            int r = id >> 5, c = id & 31;
            // k = 4c+{0,1,2,3} -> (h,s) = (0,2c) (1,2c) (0,2c+1) (1,2c+1)
            float2 ev = {pre[e].x, pre[e].z}, od = {pre[e].y, pre[e].w};
            *(float2 *)(kt + r * MF_LDK + 2 * c) = ev;
            *(float2 *)(kt + r * MF_LDK + 64 + 2 * c) = od;
        }
        if (tid < MF_KT) knorm0[buf * MF_KT + tid] = pren;
    };

    issue_loads(0);
    commit_loads(0);
    __syncthreads();

    // Phase structure.  Every wave alternates an MFMA phase M (64 dependent MFMAs, 4096 matrix-pipe
    // cycles) with a VALU phase V (the epilogue) of about the same length.  The two waves that
    // share a SIMD (w and w+4) would run them in lockstep — matrix pipe contended, then idle — so
    // waves 4-7 (role 1) defer the epilogue of each tile's second sub-tile across the barrier:
    //     role 0:   M0 V0 M1 V1 | barrier        role 1:   V1' M0 V0 M1 | barrier
    // After every barrier one wave of the SIMD starts in M and its partner in V.
    const int role = __builtin_amdgcn_readfirstlane(wave >> 2);

    auto mfma_chain = [&](const float *kt, int sub, f32x16 &acc, float (&nbv)[16], int buf) {
        const float *arow = kt + (sub * 32 + r32) * MF_LDK + h * 64;
        acc = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            f32x4 a = *(const f32x4 *)(arow + 4 * c);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, q[4 * c], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, q[4 * c + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, q[4 * c + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, q[4 * c + 3], acc, 0, 0, 0);
        }
        // |key|^2 of this lane's 16 keys: local key = (r&3) + 8*(r>>2) + 4*h
        const float *kn = knorm0 + buf * MF_KT + sub * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 nb = *(const f32x4 *)(kn + 8 * g);
            nbv[4 * g] = nb.x, nbv[4 * g + 1] = nb.y, nbv[4 * g + 2] = nb.z, nbv[4 * g + 3] = nb.w;
        }
    };

    // Epilogue.  fp32 MFMA executes on the same ALUs as VALU code (tools/probe_interleave.hip: every
    // VALU instruction adds its full issue time to the chain), so the epilogue is kept lean:
    //   LEAN (alpha >= 32): per candidate only {2 adds, 1 compare}: it is looked at again iff its squared
    //     distance is below lim2 = max(top-k bound, significance bound); everything else contributes
    //     < e^-20 to the softmax sum and is dropped.  Flagged candidates go through the exact path.
    //   !LEAN (flat softmax): every candidate adds a fast exp term; flagged ones are replaced exactly.
    // The 16 squared distances are parked in LDS ([r][lane], conflict-free) for the dynamic pick.
    float lim2 = INFINITY;
    const float cutw = args.cutw;
    auto epilogue = [&](const f32x16 &acc, const float (&nbv)[16], int jbase) {
        unsigned mask = 0;
        float c2 = 0.f;
        if (LEAN) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = (acc[r] + na) + nbv[r];
                stage[r * 64] = v;
                mask |= (v <= lim2) ? (1u << r) : 0u;
            }
        } else {
            float df[16];
            float tminf = INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = (acc[r] + na) + nbv[r];
                stage[r * 64] = v;
                mask |= (v <= lim2) ? (1u << r) : 0u;
                float f = __builtin_amdgcn_sqrtf(fabsf(v));
                df[r] = f;
                tminf = fminf(tminf, f);
            }
            const float cnew = tminf * neg_alpha;
            if (cnew > cref) {
                l = l * exp2f((cref - cnew) * LOG2E);  // cref = -inf, l = 0 -> 0 * 0
                cref = cnew;
            }
            c2 = cref * LOG2E;
            float lsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float arg = fmaf(df[r], a2, -c2);
                lsum += __builtin_amdgcn_exp2f(arg);
                mask |= (arg > -11.5f) ? (1u << r) : 0u;
            }
            l += lsum;
        }
        // Branch-free body (inactive lanes process +inf, a no-op): keeps the top-k registers in place
        // instead of copying them around a divergent region every iteration.
        // (counted, wave-uniform trip count: a `while (__any(mask))` loop makes the compiler copy the whole list — 60
        // v_mov — around a structurised exit on every iteration)
        const int iters = (int)__reduce_max_sync(~0ull, (unsigned)__popc(mask));
        for (int it = 0; it < iters; ++it) {
            const bool act = mask != 0;
            const int bpos = act ? (__ffs(mask) - 1) : 0;
            mask &= mask - 1;
            float v2 = stage[bpos * 64];
            v2 = act ? (v2 > 0.f ? v2 : 0.f) : INFINITY;
            const float de = sqrt_rn(v2);
            const float s = de * neg_alpha;  // -inf for inactive lanes
            if (LEAN) {
                const float cnew = fmaxf(cref, s);
                const float sc = (cnew == cref) ? 1.f : __builtin_amdgcn_exp2f((cref - cnew) * LOG2E);
                const float term = act ? __builtin_amdgcn_exp2f((s - cnew) * LOG2E) : 0.f;
                l = l * sc + term;
                cref = cnew;
            } else {
                const float dfast = __builtin_amdgcn_sqrtf(v2);
                // replace this term's fast value by the reference-rounded one (both 0 for +inf)
                l += __builtin_amdgcn_exp2f((s - cref) * LOG2E) - __builtin_amdgcn_exp2f(fmaf(dfast, a2, -c2));
            }
            kb.insert_nb(de, jbase + (bpos & 3) + 8 * (bpos >> 2));
        }
        // bounds for the next sub-tile.  Top-k: the row's k-th best is at most min(a_k, b_k, max(a_m, b_m))
        // with a, b the sorted lists of the two half-lanes and m = k/2 (2m elements lie below that max).
        {
            const float wk = kb.key[TOPK - 1], wm = kb.key[TOPK / 2 - 1], w0 = kb.key[0];
            const float pk = __shfl_xor(wk, 32, 64), pm = __shfl_xor(wm, 32, 64);
            const float w = fminf(fminf(wk, pk), fmaxf(wm, pm));
            thr2 = (w * w) * 1.0000004f;
            if (LEAN) {
                const float dmin = fminf(w0, __shfl_xor(w0, 32, 64));
                const float cut = dmin + cutw;  // beyond this the softmax term is < e^-20 of the largest
                lim2 = fmaxf(thr2, (cut * cut) * 1.000001f);
            } else {
                lim2 = thr2;
            }
        }
    };

    f32x16 acc;
    float nbv[16];
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) issue_loads(t + 1);
        const float *kt = ktile0 + buf * (MF_KT * MF_LDK);
        if (role == 1 && t > 0) epilogue(acc, nbv, (t - 1) * MF_KT + 32 + 4 * h);  // V1' of the previous tile
        mfma_chain(kt, 0, acc, nbv, buf);
        epilogue(acc, nbv, t * MF_KT + 4 * h);
        mfma_chain(kt, 1, acc, nbv, buf);
        if (role == 0) epilogue(acc, nbv, t * MF_KT + 32 + 4 * h);
        if (t + 1 < ntiles) commit_loads(buf ^ 1);
        __syncthreads();
    }
    if (role == 1) epilogue(acc, nbv, (ntiles - 1) * MF_KT + 32 + 4 * h);

    // merge the two half-lanes that share a query (lane, lane^32)
    {
        float co = __shfl_xor(cref, 32, 64), lo = __shfl_xor(l, 32, 64);
        float cm = fmaxf(cref, co);
        float a = (cref == -INFINITY) ? 0.f : l * exp2f((cref - cm) * LOG2E);
        float bb = (co == -INFINITY) ? 0.f : lo * exp2f((co - cm) * LOG2E);
        l = a + bb;
        cref = cm;
        float ok[TOPK];
        int oi[TOPK];
#pragma unroll
        for (int t = 0; t < TOPK; ++t) {
            ok[t] = __shfl_xor(kb.key[t], 32, 64);
            oi[t] = __shfl_xor(kb.idx[t], 32, 64);
        }
#pragma unroll
        for (int t = 0; t < TOPK; ++t) kb.insert_lex(ok[t], oi[t]);
    }
    if (h == 0 && qrow < N) {
        const size_t row = (size_t)b * N + qrow;
        const int topk = args.topk;
        const float smax = kb.key[0] * neg_alpha;                  // exact max of s
        const float lsm = l * exp2f((cref - smax) * LOG2E);        // sum exp(s - smax)
        const float inv = 1.0f / lsm;
        float *val = G.val + row * topk;
        int32_t *idx = G.idx + row * topk;
#pragma unroll
        for (int t = 0; t < TOPK; ++t) {
            if (t < topk) {
                bool live = t < M;
                float s = kb.key[t] * neg_alpha;
                val[t] = live ? exp2f((s - smax) * LOG2E) * inv : 0.f;
                idx[t] = live ? kb.idx[t] : 0;
            }
        }
        if (G.smax) G.smax[row] = smax;
        if (G.sum) G.sum[row] = lsm;
    }
}

constexpr float LEAN_MIN_ALPHA = 32.f;

template <int TOPK>
static void launch_softcorr_mfma(SCArgs &a, int blocks, hipStream_t s) {
    ensure_dyn_lds((const void *)softcorr_mfma_kernel<TOPK, true>, (int)MF_LDS_BYTES);
ensure_dyn_lds((const void *)softcorr_mfma_kernel<TOPK, false>, (int)MF_LDS_BYTES);
    const float alpha = -a.neg_alpha;
    a.cutw = 20.f / alpha;
    if (alpha >= LEAN_MIN_ALPHA)
        hipLaunchKernelGGL((softcorr_mfma_kernel<TOPK, true>), dim3(blocks), dim3(MF_THREADS), MF_LDS_BYTES, s, a);
    else
        hipLaunchKernelGGL((softcorr_mfma_kernel<TOPK, false>), dim3(blocks), dim3(MF_THREADS), MF_LDS_BYTES, s, a);
}

// both directions of B pairs in one launch; n1/n2 are the row norms of f1/f2 (computed once)
int launch_softcorr_both(const float *f1, const float *f2, const float *n1, const float *n2, int B, int N, int M,
                         float neg_alpha, float *val12, int32_t *idx12, float *val21, int32_t *idx21, hipStream_t s) {
    SCArgs a;
    a.g[0] = SCGroup{f1, f2, n1, n2, N, M, (N + MF_QB - 1) / MF_QB, val12, idx12, nullptr, nullptr};
    a.g[1] = SCGroup{f2, f1, n2, n1, M, N, (M + MF_QB - 1) / MF_QB, val21, idx21, nullptr, nullptr};
    a.blocks0 = B * a.g[0].tiles;
    a.neg_alpha = neg_alpha;
    a.topk = 10;
    prof_note(DVM_PROF_K1_SWEEP, "softcorr_mfma_kernel");
    prof_begin(s);
    launch_softcorr_mfma<10>(a, a.blocks0 + B * a.g[1].tiles, s);
    prof_end(s);
    return DVM_OK;
}

// dvm_softcorr_f16.hip
size_t softcorr_f16_ws_bytes(int B, int N, int M, bool both);
int launch_softcorr_f16(const float *f1, const float *f2, const float *n1, const float *n2, int B, int N, int M, float neg_alpha,
                         int topk, float *val12, int32_t *idx12, float *smax12, float *sum12, float *val21, int32_t *idx21,
                         float *smax21, float *sum21, const int *amax, void *ws, size_t ws_bytes, hipStream_t s, int *fuse_slots = nullptr);

// K == 128 only: also maxes the bit pattern of max |x| into the 256 slots of `absmax_slots` (zero them first);
// launch_absmax_finalize folds nt x 256 slots into nt values
void launch_rownorm2_absmax(const float *x, int rows, float *out, int *absmax_slots, hipStream_t s) {
    hipLaunchKernelGGL(rownorm2_k128_kernel, dim3((unsigned)((((long)rows + RN_ROWS - 1) / RN_ROWS * 32 + 255) / 256)), dim3(256), 0, s, x, rows, out,
                       absmax_slots);
}
__global__ void absmax_finalize_kernel(const int *__restrict__ slots, int *__restrict__ out) {
    int v = 0;
    for (int i = threadIdx.x; i < 256; i += 64) v = max(v, slots[blockIdx.x * 256 + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
void launch_absmax_finalize(const int *slots, int nt, int *out, hipStream_t s) {
    hipLaunchKernelGGL(absmax_finalize_kernel, dim3(nt), dim3(64), 0, s, slots, out);
}

void launch_rownorm2(const float *x, int rows, int K, float *out, hipStream_t s);

// norms of both sides + soft correspondence in both directions for the fused pair path (d = 128, top-10)
size_t softcorr_pair_ws_bytes(int B, int N, int M) { return align_up(514 * sizeof(int)) + softcorr_f16_ws_bytes(B, N, M, true); }
int launch_softcorr_pair(const float *f1, const float *f2, float *n1, float *n2, int B, int N, int M, float neg_alpha, float *val12,
                         int32_t *idx12, float *val21, int32_t *idx21, void *ws, size_t ws_bytes, hipStream_t s) {
    Arena ar(ws, ws_bytes);
    int *slots = ar.take<int>(2 * 256 + 2);
    int *amax = slots + 512;
    char *bws = ar.take<char>(0);
    if (!ar.ok()) {
        set_error("softcorr (pair): workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    // (row norms, absmax and the fp16 planes come out of ONE pass over the features inside launch_softcorr_f16: `slots`)
    return launch_softcorr_f16(f1, f2, n1, n2, B, N, M, neg_alpha, 10, val12, idx12, nullptr, nullptr, val21, idx21, nullptr, nullptr,
                                amax, bws, ws_bytes - ar.off, s, slots);
}

void launch_rownorm2(const float *x, int rows, int K, float *out, hipStream_t s) {
    if (K == 128)
        hipLaunchKernelGGL(rownorm2_k128_kernel, dim3((unsigned)((((long)rows + RN_ROWS - 1) / RN_ROWS * 32 + 255) / 256)), dim3(256), 0, s, x, rows, out,
                           (int *)nullptr);
    else
        hipLaunchKernelGGL(rownorm2_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, x, rows, K, out);
}

}  // namespace dvm

using namespace dvm;

DVM_EXPORT int dvm_rownorm2_f32(const float *x, int rows, int K, float *out, void *stream) {
    DVM_REQUIRE(x && out && rows >= 0 && K >= 1, "dvm_rownorm2_f32: bad arguments");
    if (rows == 0) return DVM_OK;
    launch_rownorm2(x, rows, K, out, (hipStream_t)stream);
    DVM_CHECK_LAUNCH("rownorm2");
    return DVM_OK;
}

DVM_EXPORT size_t dvm_softcorr_workspace_bytes(int B, int N, int M, int d) {
    return align_up((size_t)B * N * sizeof(float)) + align_up((size_t)B * M * sizeof(float)) +
           (d == MF_D ? align_up(514 * sizeof(int)) + softcorr_f16_ws_bytes(B, N, M, false) : 0);
}

DVM_EXPORT int dvm_softcorr_fwd_f32(const float *f1, const float *f2, int B, int N, int M, int d, float neg_alpha,
                                    int topk, float *pi_val, int32_t *pi_idx, float *row_smax, float *row_sum,
                                    int variant, void *ws, size_t ws_bytes, void *stream) {
    DVM_REQUIRE(f1 && f2 && pi_val && pi_idx, "dvm_softcorr_fwd_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1 && M >= 1, "dvm_softcorr_fwd_f32: empty input (B=%d N=%d M=%d)", B, N, M);
    DVM_REQUIRE(d >= 4 && d % 4 == 0 && d <= 512, "dvm_softcorr_fwd_f32: d=%d unsupported (need d%%4==0, 4<=d<=512)", d);
    DVM_REQUIRE(topk >= 1 && topk <= 16, "dvm_softcorr_fwd_f32: topk=%d unsupported (1..16)", topk);
    DVM_REQUIRE(neg_alpha < 0.f, "dvm_softcorr_fwd_f32: neg_alpha must be negative (got %g)", (double)neg_alpha);
    DVM_REQUIRE(variant >= 0 && variant <= 3, "dvm_softcorr_fwd_f32: bad variant %d", variant);
    DVM_REQUIRE(variant < 2 || d == MF_D, "dvm_softcorr_fwd_f32: the matrix-core variants need d == 128");
    DVM_REQUIRE(variant != 3 || topk <= 10, "dvm_softcorr_fwd_f32: the bf16 variant keeps 12 candidates (topk <= 10)");
    Arena ar(ws, ws_bytes);
    float *n1 = ar.take<float>((size_t)B * N);
    float *n2 = ar.take<float>((size_t)B * M);
    if (!ar.ok()) {
        set_error("dvm_softcorr_fwd_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    if (variant == 0 && d == MF_D && topk <= 10) variant = 3;   // auto: the fp16-split sweep (variants 1 / 2: the `variant` argument)
    if (variant == 3) {
        int *slots = ar.take<int>(2 * 256 + 2);
        int *amax = slots + 512;
        char *bws = ar.take<char>(0);
        if (!ar.ok()) {
            set_error("dvm_softcorr_fwd_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
            return DVM_ENOSPACE;
        }
        (void)hipMemsetAsync(slots, 0, 512 * sizeof(int), s);
        launch_rownorm2_absmax(f1, B * N, n1, slots, s);
        launch_rownorm2_absmax(f2, B * M, n2, slots + 256, s);
        launch_absmax_finalize(slots, 2, amax, s);
        int rc = launch_softcorr_f16(f1, f2, n1, n2, B, N, M, neg_alpha, topk, pi_val, pi_idx, row_smax, row_sum, nullptr, nullptr,
                                      nullptr, nullptr, amax, bws, ws_bytes - ar.off, s);
        if (rc != DVM_OK) return rc;
        DVM_CHECK_LAUNCH("softcorr (fp16)");
        return DVM_OK;
    }
    launch_rownorm2(f1, B * N, d, n1, s);
    launch_rownorm2(f2, B * M, d, n2, s);
    bool mfma = (variant == 2) || (variant == 0 && d == MF_D);
    prof_note(DVM_PROF_K1_SWEEP, mfma ? "softcorr_mfma_kernel" : "softcorr_scalar_kernel");
    prof_begin(s);
    if (mfma) {
        SCArgs a;
        a.g[0] = SCGroup{f1, f2, n1, n2, N, M, (N + MF_QB - 1) / MF_QB, pi_val, pi_idx, row_smax, row_sum};
        a.g[1] = a.g[0];
        a.blocks0 = B * a.g[0].tiles;
        a.neg_alpha = neg_alpha;
        a.topk = topk;
        if (topk <= 10)
            launch_softcorr_mfma<10>(a, a.blocks0, s);
        else
            launch_softcorr_mfma<16>(a, a.blocks0, s);
    } else {
        dim3 grid((N + 127) / 128, B);
        size_t lds = (size_t)(SC_KT * d + SC_KT) * sizeof(float);
        ensure_dyn_lds((const void *)softcorr_scalar_kernel<10>, 66 * 1024);
        ensure_dyn_lds((const void *)softcorr_scalar_kernel<16>, 66 * 1024);
        if (topk <= 10)
            hipLaunchKernelGGL(softcorr_scalar_kernel<10>, grid, dim3(128), lds, s, f1, f2, n1, n2, N, M, d, neg_alpha, topk,
                               pi_val, pi_idx, row_smax, row_sum);
        else
            hipLaunchKernelGGL(softcorr_scalar_kernel<16>, grid, dim3(128), lds, s, f1, f2, n1, n2, N, M, d, neg_alpha, topk,
                               pi_val, pi_idx, row_smax, row_sum);
    }
    prof_end(s);
    DVM_CHECK_LAUNCH("softcorr");
    return DVM_OK;
}

DVM_EXPORT size_t dvm_softcorr_dense_workspace_bytes(int B, int N, int M, int d) {
    return dvm_softcorr_workspace_bytes(B, N, M, d) + 2 * align_up((size_t)B * N * sizeof(float)) +
           align_up((size_t)B * N * sizeof(float)) + align_up((size_t)B * N * sizeof(int32_t));
}

DVM_EXPORT int dvm_softcorr_dense_f32(const float *f1, const float *f2, int B, int N, int M, int d, float neg_alpha, float *P,
                                      void *ws, size_t ws_bytes, void *stream) {
    DVM_REQUIRE(f1 && f2 && P, "dvm_softcorr_dense_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1 && M >= 1, "dvm_softcorr_dense_f32: empty input");
    DVM_REQUIRE(d >= 4 && d % 4 == 0 && d <= 512, "dvm_softcorr_dense_f32: d=%d unsupported", d);
    DVM_REQUIRE(neg_alpha < 0.f, "dvm_softcorr_dense_f32: neg_alpha must be negative");
    Arena ar(ws, ws_bytes);
    size_t scb = dvm_softcorr_workspace_bytes(B, N, M, d);
    char *scws = ar.take<char>(scb);
    float *smax = ar.take<float>((size_t)B * N), *ssum = ar.take<float>((size_t)B * N);
    float *v1 = ar.take<float>((size_t)B * N);
    int32_t *i1 = ar.take<int32_t>((size_t)B * N);
    if (!ar.ok()) {
        set_error("dvm_softcorr_dense_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    int rc = dvm_softcorr_fwd_f32(f1, f2, B, N, M, d, neg_alpha, 1, v1, i1, smax, ssum, 0, scws, scb, stream);
    if (rc != DVM_OK) return rc;
    // the norms are the first two carve-outs of the soft-correspondence workspace
    Arena a2(scws, scb);
    float *n1 = a2.take<float>((size_t)B * N);
    float *n2 = a2.take<float>((size_t)B * M);
    size_t lds = (size_t)(SC_KT * d + SC_KT) * sizeof(float);
    (void)hipFuncSetAttribute((const void *)softcorr_dense_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 66 * 1024);
    hipLaunchKernelGGL(softcorr_dense_kernel<0>, dim3((N + 127) / 128, B), dim3(128), lds, (hipStream_t)stream, f1, f2, n1, n2, smax,
                       ssum, N, M, d, neg_alpha, P);
    DVM_CHECK_LAUNCH("softcorr_dense");
    return DVM_OK;
}

