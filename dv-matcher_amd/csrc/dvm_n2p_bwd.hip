// dvm_n2p_bwd.hip — training twins of the N2P attention core (reference models/model.py:339-350, 375-386;
// SURVEY §8b "backward twins ... n2p").
//
// Per point i with feature-space neighbours j = idx[i, 0..K) and 4 heads of D = C/4 channels:
//     e_hj = q_h . (kp_j - kp_i)_h / sqrt(D),   a_h = softmax_j(e_h),   out_h = sum_j a_hj (vp_j - vp_i)_h
// The forward here also writes a [B,N,K,4] (10 MB at B=8, N=2048, K=40 — instead of the five (B,N,K,C)
// tensors the unfused formulation keeps for autograd).  Backward, with g = dL/d out:
//     da_hj = g_h . vp_j          (the - g_h . vp_i part is constant over j and cancels in the softmax backward)
//     de_hj = a_hj (da_hj - sum_j' a_hj' da_hj')
//     dq_h  = sum_j de_hj kp_j / sqrt(D)                      (sum_j de_hj = 0 removes kp_i)
//     dkp_j += de_hj q_h / sqrt(D) ,  dvp_j += a_hj g_h ,  dvp_i -= g_h
// One wave per point, a head = 16 consecutive lanes (C/64 channels per lane); neighbour rows are gathered
// with coalesced 256/512 B reads and scattered with hardware fp32 atomics.
#include "dvm_common.h"

namespace dvm {
namespace {

constexpr int NP_H = 4;
constexpr int NP_KMAX = 64;

template <int C>
__global__ __launch_bounds__(256) void n2p_core_fwd_kernel(const float *__restrict__ qkv, const int32_t *__restrict__ idx, int N,
                                                           int K, float *__restrict__ out, float *__restrict__ attn) {
    constexpr int CPL = C / 64, D = C / NP_H, LD = 3 * C;
    __shared__ float se[4][NP_KMAX * NP_H];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long pt0 = (long)blockIdx.x * 4 + wave;
    const bool valid = pt0 < N;
    const long pt = valid ? pt0 : N - 1;
    const size_t base = (size_t)blockIdx.y * N;
    const int hd = lane >> 4;
    const float scale = sqrtf((float)D);
    const float *self = qkv + (base + pt) * LD + lane * CPL;
    float qv[CPL], ki[CPL], vi[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) qv[c] = self[c], ki[c] = self[C + c], vi[c] = self[2 * C + c];
    const int32_t *nb = idx + (base + pt) * K;
    float m = -INFINITY, l = 0.f, acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
    for (int j = 0; j < K; ++j) {
        const float *nrow = qkv + (base + nb[j]) * LD + lane * CPL;
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) part = fmaf(qv[c], nrow[C + c] - ki[c], part);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) part += __shfl_xor(part, o, 64);
        const float e = part / scale;
        if ((lane & 15) == 0) se[wave][j * NP_H + hd] = e;
        const float mn = fmaxf(m, e);
        const float sc = __expf(m - mn);  // first neighbour: exp(-inf) = 0
        const float w = __expf(e - mn);
        l = l * sc + w;
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c] = fmaf(w, nrow[2 * C + c] - vi[c], acc[c] * sc);
        m = mn;
    }
    const float inv = 1.0f / l;
    if (valid) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) out[(base + pt) * C + lane * CPL + c] = acc[c] * inv;
    }
    __syncthreads();
    // a[j][h] = exp(e - m_h) / l_h; lane t handles entries t, t + 64, ... whose head is t % 4 = lane % 4
    const float mh = __shfl(m, (lane & 3) * 16, 64), ih = __shfl(inv, (lane & 3) * 16, 64);
    if (valid)
        for (int t = lane; t < K * NP_H; t += 64) attn[(base + pt) * K * NP_H + t] = __expf(se[wave][t] - mh) * ih;
}

template <int C>
__global__ __launch_bounds__(256) void n2p_core_bwd_kernel(const float *__restrict__ qkv, const int32_t *__restrict__ idx,
                                                           const float *__restrict__ attn, const float *__restrict__ gout, int N,
                                                           int K, float *__restrict__ dqkv) {
    constexpr int CPL = C / 64, D = C / NP_H, LD = 3 * C;
    __shared__ float sa[4][NP_KMAX * NP_H], sd[4][NP_KMAX * NP_H];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long pt0 = (long)blockIdx.x * 4 + wave;
    const bool valid = pt0 < N;
    const long pt = valid ? pt0 : N - 1;
    const size_t base = (size_t)blockIdx.y * N;
    const int hd = lane >> 4;
    const float inv_scale = 1.0f / sqrtf((float)D);
    const float *self = qkv + (base + pt) * LD + lane * CPL;
    float qv[CPL], gv[CPL], dq[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        qv[c] = self[c];
        gv[c] = valid ? gout[(base + pt) * C + lane * CPL + c] : 0.f;
        dq[c] = 0.f;
    }
    const int32_t *nb = idx + (base + pt) * K;
    for (int j = 0; j < K; ++j) {  // da_hj = g_h . vp_j
        const float *nrow = qkv + (base + nb[j]) * LD + 2 * C + lane * CPL;
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) part = fmaf(gv[c], nrow[c], part);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) part += __shfl_xor(part, o, 64);
        if ((lane & 15) == 0) sd[wave][j * NP_H + hd] = part;
    }
    __syncthreads();
    float dot = 0.f;  // sum_j a_hj da_hj for head lane % 4
    for (int t = lane; t < K * NP_H; t += 64) {
        const float a = attn[(base + pt) * K * NP_H + t];
        sa[wave][t] = a;
        dot = fmaf(a, sd[wave][t], dot);
    }
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) dot += __shfl_xor(dot, o, 64);
    for (int t = lane; t < K * NP_H; t += 64) sd[wave][t] = sa[wave][t] * (sd[wave][t] - dot) * inv_scale;  // de / sqrt(D)
    __syncthreads();
    if (!valid) return;
    for (int j = 0; j < K; ++j) {
        const size_t r = (base + nb[j]) * LD + lane * CPL;
        const float de = sd[wave][j * NP_H + hd], a = sa[wave][j * NP_H + hd];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            dq[c] = fmaf(de, qkv[r + C + c], dq[c]);
            unsafeAtomicAdd(dqkv + r + C + c, de * qv[c]);
            unsafeAtomicAdd(dqkv + r + 2 * C + c, a * gv[c]);
        }
    }
    float *dself = dqkv + (base + pt) * LD + lane * CPL;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        dself[c] = dq[c];  // nobody else writes the q part
        unsafeAtomicAdd(dself + 2 * C + c, -gv[c]);
    }
}

}  // namespace
}  // namespace dvm

using namespace dvm;

DVM_EXPORT int dvm_n2p_core_fwd_f32(const float *qkv, const int32_t *idx, int B, int N, int C, int K, int heads, float *out,
                                    float *attn, void *stream) {
    DVM_REQUIRE(qkv && idx && out && attn, "dvm_n2p_core_fwd_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1, "dvm_n2p_core_fwd_f32: empty input");
    DVM_REQUIRE((C == 64 || C == 128) && heads == NP_H, "dvm_n2p_core_fwd_f32: need C in {64,128}, heads == 4 (C=%d heads=%d)", C,
                heads);
    DVM_REQUIRE(K >= 1 && K <= NP_KMAX, "dvm_n2p_core_fwd_f32: K=%d unsupported (1..64)", K);
    dim3 grid((N + 3) / 4, B);
    hipStream_t s = (hipStream_t)stream;
    if (C == 64)
        hipLaunchKernelGGL(n2p_core_fwd_kernel<64>, grid, dim3(256), 0, s, qkv, idx, N, K, out, attn);
    else
        hipLaunchKernelGGL(n2p_core_fwd_kernel<128>, grid, dim3(256), 0, s, qkv, idx, N, K, out, attn);
    DVM_CHECK_LAUNCH("n2p_core_fwd");
    return DVM_OK;
}

DVM_EXPORT int dvm_n2p_core_bwd_f32(const float *qkv, const int32_t *idx, const float *attn, const float *g_out, int B, int N, int C,
                                    int K, int heads, float *d_qkv, void *stream) {
    DVM_REQUIRE(qkv && idx && attn && g_out && d_qkv, "dvm_n2p_core_bwd_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1, "dvm_n2p_core_bwd_f32: empty input");
    DVM_REQUIRE((C == 64 || C == 128) && heads == NP_H, "dvm_n2p_core_bwd_f32: need C in {64,128}, heads == 4 (C=%d heads=%d)", C,
                heads);
    DVM_REQUIRE(K >= 1 && K <= NP_KMAX, "dvm_n2p_core_bwd_f32: K=%d unsupported (1..64)", K);
    dim3 grid((N + 3) / 4, B);
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(d_qkv, 0, (size_t)B * N * 3 * C * sizeof(float), s);
    if (C == 64)
        hipLaunchKernelGGL(n2p_core_bwd_kernel<64>, grid, dim3(256), 0, s, qkv, idx, attn, g_out, N, K, d_qkv);
    else
        hipLaunchKernelGGL(n2p_core_bwd_kernel<128>, grid, dim3(256), 0, s, qkv, idx, attn, g_out, N, K, d_qkv);
    DVM_CHECK_LAUNCH("n2p_core_bwd");
    return DVM_OK;
}
