// dvm_sa_bwd.hip — backward twin of the SA_Layer attention core (reference models/model.py:113-121;
// SURVEY §8b "backward twins ... sa").
//
// Forward (dvm_backbone.hip, point-major):  E_ij = p_i . p_j (symmetric),  A_ij = exp(E_ij - m_i) / l_i,
//     c_j = 1e-9 + sum_i A_ij,  Ah_ij = A_ij / c_j,  xr_j = sum_i v_i Ah_ij.
// With G = dL/d xr:
//     dAh_ij = v_i . G_j                     dv_i = sum_j Ah_ij G_j
//     t_j    = sum_i dAh_ij Ah_ij = G_j . xr_j
//     dA_ij  = (dAh_ij - t_j) / c_j
//     u_i    = sum_j A_ij dA_ij              dE_ij = A_ij (dA_ij - u_i)
//     dp_i   = sum_j (dE_ij + dE_ji) p_j
// Nothing N x N is stored: two tile-recompute passes on the fp32 matrix cores.
//   pass R (rows):   per 32x32 tile  E (8 MFMA) + dAh (32) -> A, dA -> u_i += ..., dv_i += Ah G  (32 MFMA)
//   pass D (dp):     per tile  E (8) + dAh_ij (32) + dAh_ji (32) -> S = dE_ij + dE_ji -> dp_i += S p_j (16 MFMA)
// Both transposed entries of a tile are formed from the same E (symmetry), so dp needs no column pass and no
// atomics beyond the optional split of the inner loop.  The weight tiles go from the accumulator layout
// straight back into the A operand (the contraction index is permuted consistently on the B side).
#include <algorithm>

#include "dvm_common.h"

namespace dvm {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SB_P = 16, SB_C = 64, SB_LDP = 20, SB_LDC = 68;
constexpr int SB_ROWS = 128;  // outer rows per workgroup (4 waves x 32)

__device__ __forceinline__ f32x16 zero16() {
    return f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
}

// t_j = G_j . xr_j
__global__ void sa_bwd_prep_kernel(const float *__restrict__ g, const float *__restrict__ xr, long rows, float *__restrict__ t) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const f32x4 *a = (const f32x4 *)(g + i * SB_C), *b = (const f32x4 *)(xr + i * SB_C);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < SB_C / 4; ++c) {
        f32x4 x = a[c], y = b[c];
        s = fmaf(x.x, y.x, s);
        s = fmaf(x.y, y.y, s);
        s = fmaf(x.z, y.z, s);
        s = fmaf(x.w, y.w, s);
    }
    t[i] = s;
}

// stage rows [j0, j0+32) of a [N][W] matrix into LDS with the even/odd channel split the A operand wants:
// position q < W/2 holds channel 2q, position W/2 + q holds channel 2q + 1
template <int W, int LD>
__device__ __forceinline__ void stage_rows(const float *__restrict__ src, int j0, int N, float *__restrict__ dst, int tid) {
    constexpr int V = W / 4;  // float4 per row
    for (int e = tid; e < 32 * V; e += 256) {
        const int r = e / V, c = e % V;
        f32x4 q = {0.f, 0.f, 0.f, 0.f};
        if (j0 + r < N) q = *(const f32x4 *)(src + (size_t)(j0 + r) * W + 4 * c);
        float2 ev = {q.x, q.z}, od = {q.y, q.w};
        *(float2 *)(dst + r * LD + 2 * c) = ev;
        *(float2 *)(dst + r * LD + W / 2 + 2 * c) = od;
    }
}

// B-operand fragment of one outer row: frag[s] = row[2s + h]
template <int W>
__device__ __forceinline__ void load_frag(const float *__restrict__ row, int h, float (&frag)[W / 2]) {
#pragma unroll
    for (int c = 0; c < W / 4; ++c) {
        f32x4 q = *(const f32x4 *)(row + 4 * c);
        frag[2 * c] = h ? q.y : q.x;
        frag[2 * c + 1] = h ? q.w : q.z;
    }
}

// C[j][i] = sum_c tile[j][c] * frag_i[c]   (A from the staged tile, B from registers)
template <int W, int LD>
__device__ __forceinline__ f32x16 tile_dot(const float *__restrict__ tile, int r32, int h, const float (&frag)[W / 2]) {
    const float *jr = tile + r32 * LD + h * (W / 2);
    f32x16 acc = zero16();
#pragma unroll
    for (int c = 0; c < W / 8; ++c) {
        f32x4 a = *(const f32x4 *)(jr + 4 * c);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, frag[4 * c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, frag[4 * c + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, frag[4 * c + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, frag[4 * c + 3], acc, 0, 0, 0);
    }
    return acc;
}

struct SaBwdArgs {
    const float *p, *v, *g, *stats, *cinv, *t, *u;  // u: pass D only
    float *u_out, *dv, *dp;
    int N, split;
};

// pass R: u_i and dv_i.  Outer rows i in registers, inner rows j staged 32 at a time.
__global__ __launch_bounds__(256) void sa_bwd_rows_kernel(const SaBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float pt[32 * SB_LDP];
    __shared__ __attribute__((aligned(16))) float gt[32 * SB_LDC];
    __shared__ __attribute__((aligned(16))) float sc[3 * 32];  // t_j, cinv_j (0 past N), mask_j
    const int N = a.N;
    const int b = blockIdx.y;
    const int ot = blockIdx.x / a.split, sp = blockIdx.x % a.split;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const int irow = ot * SB_ROWS + wave * 32 + r32;
    const int irc = irow < N ? irow : N - 1;
    const float *pb = a.p + (size_t)b * N * SB_P, *vb = a.v + (size_t)b * N * SB_C, *gb = a.g + (size_t)b * N * SB_C;
    float pi[SB_P / 2];
    load_frag<SB_P>(pb + (size_t)irc * SB_P, h, pi);
    const float m_i = a.stats[((size_t)b * N + irc) * 2], il_i = irow < N ? a.stats[((size_t)b * N + irc) * 2 + 1] : 0.f;
    f32x16 dv0 = zero16(), dv1 = zero16();
    float ul = 0.f;
    const int ntiles = (N + 31) / 32, per = (ntiles + a.split - 1) / a.split;
    const int t0 = sp * per, t1 = (t0 + per < ntiles) ? t0 + per : ntiles;
    for (int tt = t0; tt < t1; ++tt) {
        const int j0 = tt * 32;
        __syncthreads();
        stage_rows<SB_P, SB_LDP>(pb, j0, N, pt, tid);
        stage_rows<SB_C, SB_LDC>(gb, j0, N, gt, tid);
        if (tid < 96) {
            const int which = tid >> 5, j = j0 + (tid & 31);
            const bool ok = j < N;
            sc[tid] = which == 0 ? (ok ? a.t[(size_t)b * N + j] : 0.f)
                                 : which == 1 ? (ok ? a.cinv[(size_t)b * N + j] : 0.f) : (ok ? 1.f : 0.f);
        }
        __syncthreads();
        // u_i = sum_j A_ij dA_ij with dA_ij = (G_j . v_i - t_j) / c_j, i.e. u_i = v_i . (sum_j Ah_ij G_j) - sum_j Ah_ij t_j
        // = v_i . dv_i - (Ah t)_i: the 32 matrix instructions of the G_j . v_i tile are not needed in this pass — the
        // first term is one dot product with the finished dv_i, the second a by-product of forming Ah.
        const f32x16 E = tile_dot<SB_P, SB_LDP>(pt, r32, h, pi);
        float ah[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 tj = *(const f32x4 *)(sc + 4 * h + 8 * g4);
            const f32x4 cj = *(const f32x4 *)(sc + 32 + 4 * h + 8 * g4);
            const f32x4 mk = *(const f32x4 *)(sc + 64 + 4 * h + 8 * g4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g4 + e;
                const float A = __expf(E[r] - m_i) * il_i * mk[e];
                ah[r] = A * cj[e];
                ul = fmaf(-ah[r], tj[e], ul);
            }
        }
        // dv_i += sum_j Ah_ij G_j : step r contracts j = (r&3) + 8*(r>>2) + 4*h; column n <-> LDS position 2n + cb
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float2 gg = *(const float2 *)(gt + j * SB_LDC + 2 * r32);
            dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ah[r], gg.x, dv0, 0, 0, 0);
            dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ah[r], gg.y, dv1, 0, 0, 0);
        }
    }
    ul += lane_xor32(ul);
    if (h == 0 && irow < N) unsafeAtomicAdd(a.u_out + (size_t)b * N + irow, ul);
    const int p0 = 2 * r32, p1 = 2 * r32 + 1;  // LDS positions -> channels
    const int c0 = p0 < 32 ? 2 * p0 : 2 * (p0 - 32) + 1, c1 = p1 < 32 ? 2 * p1 : 2 * (p1 - 32) + 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = ot * SB_ROWS + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;   // wave-half uniform
        const int rc = row < N ? row : N - 1;
        // v_row . dv_row (this split's part of dv): the 32 lanes of a half hold the row's 64 channels
        float d = dv0[r] * vb[(size_t)rc * SB_C + c0];
        d = fmaf(dv1[r], vb[(size_t)rc * SB_C + c1], d);
        d = sum32(d);
        if (row >= N) continue;
        if (r32 == 0) unsafeAtomicAdd(a.u_out + (size_t)b * N + row, d);
        float *dst = a.dv + ((size_t)b * N + row) * SB_C;
        unsafeAtomicAdd(dst + c0, dv0[r]);
        unsafeAtomicAdd(dst + c1, dv1[r]);
    }
}

// pass D: dp_i = sum_j (dE_ij + dE_ji) p_j
__global__ __launch_bounds__(256) void sa_bwd_dp_kernel(const SaBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float pt[32 * SB_LDP];
    __shared__ __attribute__((aligned(16))) float gt[32 * SB_LDC];
    __shared__ __attribute__((aligned(16))) float vt[32 * SB_LDC];
    __shared__ __attribute__((aligned(16))) float sc[6 * 32];  // m_j, il_j, t_j, cinv_j, u_j, mask_j
    const int N = a.N;
    const int b = blockIdx.y;
    const int ot = blockIdx.x / a.split, sp = blockIdx.x % a.split;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const int irow = ot * SB_ROWS + wave * 32 + r32;
    const int irc = irow < N ? irow : N - 1;
    const float *pb = a.p + (size_t)b * N * SB_P, *vb = a.v + (size_t)b * N * SB_C, *gb = a.g + (size_t)b * N * SB_C;
    float pi[SB_P / 2], vi[SB_C / 2], gi[SB_C / 2];
    load_frag<SB_P>(pb + (size_t)irc * SB_P, h, pi);
    load_frag<SB_C>(vb + (size_t)irc * SB_C, h, vi);
    load_frag<SB_C>(gb + (size_t)irc * SB_C, h, gi);
    const size_t ii = (size_t)b * N + irc;
    const float m_i = a.stats[ii * 2], il_i = a.stats[ii * 2 + 1], t_i = a.t[ii], c_i = a.cinv[ii], u_i = a.u[ii];
    f32x16 dp = zero16();
    const int ntiles = (N + 31) / 32, per = (ntiles + a.split - 1) / a.split;
    const int t0 = sp * per, t1 = (t0 + per < ntiles) ? t0 + per : ntiles;
    const int pcol = ((r32 & 15) & 1) * 8 + ((r32 & 15) >> 1);  // LDS position of channel (r32 & 15) in a p row
    for (int tt = t0; tt < t1; ++tt) {
        const int j0 = tt * 32;
        __syncthreads();
        stage_rows<SB_P, SB_LDP>(pb, j0, N, pt, tid);
        stage_rows<SB_C, SB_LDC>(gb, j0, N, gt, tid);
        stage_rows<SB_C, SB_LDC>(vb, j0, N, vt, tid);
        if (tid < 192) {
            const int which = tid >> 5, j = j0 + (tid & 31);
            const bool ok = j < N;
            const size_t jj = (size_t)b * N + (ok ? j : 0);
            float val;
            if (which == 0) val = ok ? a.stats[jj * 2] : 0.f;
            else if (which == 1) val = ok ? a.stats[jj * 2 + 1] : 0.f;
            else if (which == 2) val = ok ? a.t[jj] : 0.f;
            else if (which == 3) val = ok ? a.cinv[jj] : 0.f;
            else if (which == 4) val = ok ? a.u[jj] : 0.f;
            else val = ok ? 1.f : 0.f;
            sc[tid] = val;
        }
        __syncthreads();
        const f32x16 E = tile_dot<SB_P, SB_LDP>(pt, r32, h, pi);
        const f32x16 X = tile_dot<SB_C, SB_LDC>(gt, r32, h, vi);  // dAh_ij = G_j . v_i
        const f32x16 Y = tile_dot<SB_C, SB_LDC>(vt, r32, h, gi);  // dAh_ji = v_j . G_i
        float sv[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int o = 4 * h + 8 * g4;
            const f32x4 mj = *(const f32x4 *)(sc + o), lj = *(const f32x4 *)(sc + 32 + o), tj = *(const f32x4 *)(sc + 64 + o);
            const f32x4 cj = *(const f32x4 *)(sc + 96 + o), uj = *(const f32x4 *)(sc + 128 + o), mk = *(const f32x4 *)(sc + 160 + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g4 + e;
                const float Aij = __expf(E[r] - m_i) * il_i * mk[e];
                const float Aji = __expf(E[r] - mj[e]) * lj[e];  // il_j = 0 past N
                const float dEij = Aij * ((X[r] - tj[e]) * cj[e] - u_i);
                const float dEji = Aji * ((Y[r] - t_i) * c_i - uj[e]);
                sv[r] = dEij + dEji;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = (r & 3) + 8 * (r >> 2) + 4 * h;
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[r], pt[j * SB_LDP + pcol], dp, 0, 0, 0);
        }
    }
    if (r32 < SB_P) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = ot * SB_ROWS + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < N) unsafeAtomicAdd(a.dp + ((size_t)b * N + row) * SB_P + r32, dp[r]);
        }
    }
}

}  // namespace
}  // namespace dvm

using namespace dvm;

DVM_EXPORT size_t dvm_sa_attention_bwd_workspace_bytes(int B, int N) { return 2 * align_up((size_t)B * N * sizeof(float)); }

DVM_EXPORT int dvm_sa_attention_bwd_f32(const float *p, const float *v, const float *xr, const float *stats, const float *cinv,
                                        const float *g_xr, int B, int N, float *d_p, float *d_v, void *ws, size_t ws_bytes,
                                        void *stream) {
    DVM_REQUIRE(p && v && xr && stats && cinv && g_xr && d_p && d_v, "dvm_sa_attention_bwd_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1, "dvm_sa_attention_bwd_f32: empty input");
    Arena ar(ws, ws_bytes);
    float *t = ar.take<float>((size_t)B * N);
    float *u = ar.take<float>((size_t)B * N);
    if (!ar.ok()) {
        set_error("dvm_sa_attention_bwd_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(u, 0, (size_t)B * N * sizeof(float), s);
    (void)hipMemsetAsync(d_p, 0, (size_t)B * N * SB_P * sizeof(float), s);
    (void)hipMemsetAsync(d_v, 0, (size_t)B * N * SB_C * sizeof(float), s);
    const long rows = (long)B * N;
    hipLaunchKernelGGL(sa_bwd_prep_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, g_xr, xr, rows, t);
    SaBwdArgs a{p, v, g_xr, stats, cinv, t, u, u, d_v, d_p, N, 1};
    const int ot = (N + SB_ROWS - 1) / SB_ROWS, ntiles = (N + 31) / 32;
    int split = 1;
    if (!deterministic())   // (the split's partial sums meet in fp32 atomics: order not fixed)
        while (B * ot * split < 512 && split < 16 && ntiles / (2 * split) >= 4) split *= 2;
    a.split = split;
    dim3 grid(ot * split, B);
    hipLaunchKernelGGL(sa_bwd_rows_kernel, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(sa_bwd_dp_kernel, grid, dim3(256), 0, s, a);
    DVM_CHECK_LAUNCH("sa_attention_bwd");
    return DVM_OK;
}
