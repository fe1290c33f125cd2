// dvm_loss_bwd.hip — backward twins of the geometric loss kernels ("warp/arap, chamfer"):
//   rotation_6d_to_matrix (models/loss.py:39-45), DeformationGraph_geod.forward (lib/deformation_graph_point.py:
//   233-261: embedded-deformation warp + ARAP) and the Chamfer NN distances (models/loss.py:1216-1226, whose
//   upstream CUDA extension back-propagates through the arg-min indices held fixed).
// These are small, scatter-shaped problems (3 influence nodes per vertex, 9 ring nodes per node, one nearest
// neighbour per point): one thread per source element, fp32 atomics onto the few-KB gradient arrays.
#include "dvm_common.h"

namespace dvm {
namespace {

// d6 = (a1, a2) -> rows (b1, b2, b1 x b2); gR [9] -> g_d6 [6]
__device__ __forceinline__ void rot6d_bwd_dev(const float *__restrict__ d, const float *__restrict__ g, float *__restrict__ o) {
    const float a1[3] = {d[0], d[1], d[2]}, a2[3] = {d[3], d[4], d[5]};
    float n1 = sqrt_rn((a1[0] * a1[0] + a1[1] * a1[1]) + a1[2] * a1[2]);
    n1 = n1 > 1e-12f ? n1 : 1e-12f;
    const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float dot = (b1[0] * a2[0] + b1[1] * a2[1]) + b1[2] * a2[2];
    const float u[3] = {a2[0] - dot * b1[0], a2[1] - dot * b1[1], a2[2] - dot * b1[2]};
    float n2 = sqrt_rn((u[0] * u[0] + u[1] * u[1]) + u[2] * u[2]);
    n2 = n2 > 1e-12f ? n2 : 1e-12f;
    const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    float gb1[3] = {g[0], g[1], g[2]}, gb2[3] = {g[3], g[4], g[5]};
    const float gb3[3] = {g[6], g[7], g[8]};
    // b3 = b1 x b2 :  gb1 += b2 x gb3 ,  gb2 += gb3 x b1
    gb1[0] += b2[1] * gb3[2] - b2[2] * gb3[1];
    gb1[1] += b2[2] * gb3[0] - b2[0] * gb3[2];
    gb1[2] += b2[0] * gb3[1] - b2[1] * gb3[0];
    gb2[0] += gb3[1] * b1[2] - gb3[2] * b1[1];
    gb2[1] += gb3[2] * b1[0] - gb3[0] * b1[2];
    gb2[2] += gb3[0] * b1[1] - gb3[1] * b1[0];
    // b2 = u / |u|
    const float p2 = (gb2[0] * b2[0] + gb2[1] * b2[1]) + gb2[2] * b2[2];
    const float gu[3] = {(gb2[0] - p2 * b2[0]) / n2, (gb2[1] - p2 * b2[1]) / n2, (gb2[2] - p2 * b2[2]) / n2};
    // u = a2 - (b1.a2) b1
    const float pu = (gu[0] * b1[0] + gu[1] * b1[1]) + gu[2] * b1[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[3 + c] = gu[c] - pu * b1[c];
        gb1[c] += -dot * gu[c] - pu * a2[c];
    }
    // b1 = a1 / |a1|
    const float p1 = (gb1[0] * b1[0] + gb1[1] * b1[1]) + gb1[2] * b1[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (gb1[c] - p1 * b1[c]) / n1;
}
__global__ void rot6d_bwd_kernel(const float *__restrict__ d6, const float *__restrict__ gR, int total, float *__restrict__ gd6) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= total) return;
    rot6d_bwd_dev(d6 + (size_t)n * 6, gR + (size_t)n * 9, gd6 + (size_t)n * 6);
}
// The Deformer's output row def9 = [t (3) | d6 - (1,0,0,0,1,0) (6)] (models/loss.py:1258-1262): (dR [9], dT [3]) -> d def9 [9]
__global__ void def9_bwd_kernel(const float *__restrict__ def9, const float *__restrict__ dR, const float *__restrict__ dT, int total,
                                float *__restrict__ ddef9) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= total) return;
    const float *d = def9 + (size_t)n * 9;
    const float d6[6] = {d[3] + 1.f, d[4] + 0.f, d[5] + 0.f, d[6] + 0.f, d[7] + 1.f, d[8] + 0.f};
    float *o = ddef9 + (size_t)n * 9;
    float g6[6];
    rot6d_bwd_dev(d6, dR + (size_t)n * 9, g6);
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = dT[(size_t)n * 3 + c];
#pragma unroll
    for (int c = 0; c < 6; ++c) o[3 + c] = g6[c];
}

// warped_i = sum_s w_s (R_s (v_i - g_s) + g_s + t_s):  dT_s += w_s gw_i ,  dR_s += w_s gw_i (v_i - g_s)^T
__global__ __launch_bounds__(256) void dg_warp_bwd_kernel(const float *__restrict__ xyz, int N, int Nn,
                                                          const int32_t *__restrict__ nodes_idx, const int32_t *__restrict__ infl,
                                                          const float *__restrict__ weights, const float *__restrict__ gw,
                                                          float *__restrict__ dR, float *__restrict__ dT) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float *p = xyz + (size_t)b * N * 3;
    const size_t row = (size_t)b * N + i;
    const float v[3] = {p[3 * i], p[3 * i + 1], p[3 * i + 2]};
    const float g[3] = {gw[row * 3], gw[row * 3 + 1], gw[row * 3 + 2]};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int nb = infl[row * 3 + s];
        const int gv = nodes_idx[(size_t)b * Nn + nb];
        const float w = weights[row * 3 + s];
        const float d[3] = {v[0] - p[3 * gv], v[1] - p[3 * gv + 1], v[2] - p[3 * gv + 2]};
        float *r = dR + ((size_t)b * Nn + nb) * 9, *t = dT + ((size_t)b * Nn + nb) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float wg = w * g[c];
            unsafeAtomicAdd(t + c, wg);
#pragma unroll
            for (int e = 0; e < 3; ++e) unsafeAtomicAdd(r + 3 * c + e, wg * d[e]);
        }
    }
}

// arap_b = sum_{a, q} |e_aq|^2 / Nn,  e = (g_a + t_a) - (g_b + t_b) - R_a (g_a - g_b),  b = ring[a][q]
__global__ __launch_bounds__(256) void dg_arap_bwd_kernel(const float *__restrict__ xyz, int N, int Nn,
                                                          const int32_t *__restrict__ nodes_idx, const int32_t *__restrict__ ring,
                                                          const float *__restrict__ R, const float *__restrict__ T,
                                                          const float *__restrict__ garap, int garap_stride, float *__restrict__ dR,
                                                          float *__restrict__ dT) {
    const int b = blockIdx.y;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= Nn) return;
    const float *p = xyz + (size_t)b * N * 3;
    const size_t na = (size_t)b * Nn + a;
    const int va = nodes_idx[na];
    const float *ra = R + na * 9, *ta = T + na * 3;
    const float ga[3] = {p[3 * va], p[3 * va + 1], p[3 * va + 2]};
    const float k = 2.f * garap[(size_t)b * garap_stride] / (float)Nn;
    float accT[3] = {0.f, 0.f, 0.f}, accR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < 9; ++q) {
        const int nb = ring[na * 9 + q];
        const size_t nbg = (size_t)b * Nn + nb;
        const int vb = nodes_idx[nbg];
        const float *tb = T + nbg * 3;
        const float gb[3] = {p[3 * vb], p[3 * vb + 1], p[3 * vb + 2]};
        const float d[3] = {ga[0] - gb[0], ga[1] - gb[1], ga[2] - gb[2]};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float rv = (ra[3 * c] * d[0] + ra[3 * c + 1] * d[1]) + ra[3 * c + 2] * d[2];
            const float ge = k * (((ga[c] + ta[c]) - (gb[c] + tb[c])) - rv);
            accT[c] += ge;
            unsafeAtomicAdd(dT + nbg * 3 + c, -ge);
#pragma unroll
            for (int e = 0; e < 3; ++e) accR[3 * c + e] -= ge * d[e];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) unsafeAtomicAdd(dT + na * 3 + c, accT[c]);
#pragma unroll
    for (int c = 0; c < 9; ++c) unsafeAtomicAdd(dR + na * 9 + c, accR[c]);
}

// d1_i = |a_i - b_{i1(i)}|^2, d2_j = |b_j - a_{i2(j)}|^2 with the indices held fixed
__global__ __launch_bounds__(256) void chamfer_bwd_kernel(const float *__restrict__ a, const float *__restrict__ bp,
                                                          const int32_t *__restrict__ i1, const int32_t *__restrict__ i2,
                                                          const float *__restrict__ g1, const float *__restrict__ g2, int N, int M,
                                                          float *__restrict__ da, float *__restrict__ db) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N + M) return;
    const bool first = t < N;
    const int i = first ? t : t - N;
    const float *src = first ? a + ((size_t)b * N + i) * 3 : bp + ((size_t)b * M + i) * 3;
    const int j = first ? i1[(size_t)b * N + i] : i2[(size_t)b * M + i];
    const float *oth = first ? bp + ((size_t)b * M + j) * 3 : a + ((size_t)b * N + j) * 3;
    const float g = 2.f * (first ? g1[(size_t)b * N + i] : g2[(size_t)b * M + i]);
    float *dsrc = first ? da + ((size_t)b * N + i) * 3 : db + ((size_t)b * M + i) * 3;
    float *doth = first ? db + ((size_t)b * M + j) * 3 : da + ((size_t)b * N + j) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = g * (src[c] - oth[c]);
        unsafeAtomicAdd(dsrc + c, v);
        unsafeAtomicAdd(doth + c, -v);
    }
}

// The source-side gradient of the two Chamfer side MEANS of a pair (terms d1.mean(), d2.mean() with gradients gt[p][off], gt[p][off+1]):
// d_a only (the target cloud is an input of the criterion); d_a zeroed by the caller.  blockIdx.z selects one of two problems.
struct ChBwdSrc {
    const float *a[2], *b[2];
    const int32_t *i1[2], *i2[2];
    float *da[2];
    int off[2];
};
__global__ __launch_bounds__(256) void chamfer_bwd_src_kernel(const ChBwdSrc q, const float *__restrict__ gt, int gstride, int N, int M) {
    const int z = blockIdx.z, b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N + M) return;
    const float *a = q.a[z], *bp = q.b[z];
    float *da = q.da[z];
    const bool first = t < N;
    const int i = first ? t : t - N;
    const int j = first ? q.i1[z][(size_t)b * N + i] : q.i2[z][(size_t)b * M + i];
    const float *pa = a + ((size_t)b * N + (first ? i : j)) * 3, *pb = bp + ((size_t)b * M + (first ? j : i)) * 3;
    const float g = 2.f * (first ? gt[(size_t)b * gstride + q.off[z]] / (float)N : gt[(size_t)b * gstride + q.off[z] + 1] / (float)M);
    float *d = da + ((size_t)b * N + (first ? i : j)) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) unsafeAtomicAdd(d + c, g * (pa[c] - pb[c]));
}

}  // namespace

// g_arap read at g_arap[b * garap_stride]
void launch_dg_warp_arap_bwd(const float *xyz, int B, int N, const int32_t *nodes_idx, const int32_t *ring, const int32_t *infl_idx,
                             const float *weights, const float *R, const float *T, const float *g_warped, const float *g_arap, int garap_stride,
                             float *d_R, float *d_T, hipStream_t s) {
    const int Nn = N / 2;
    (void)hipMemsetAsync(d_R, 0, (size_t)B * Nn * 9 * sizeof(float), s);
    (void)hipMemsetAsync(d_T, 0, (size_t)B * Nn * 3 * sizeof(float), s);
    hipLaunchKernelGGL(dg_warp_bwd_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, xyz, N, Nn, nodes_idx, infl_idx, weights,
                       g_warped, d_R, d_T);
    hipLaunchKernelGGL(dg_arap_bwd_kernel, dim3((Nn + 255) / 256, B), dim3(256), 0, s, xyz, N, Nn, nodes_idx, ring, R, T, g_arap, garap_stride,
                       d_R, d_T);
}
void launch_def9_bwd(const float *def9, const float *dR, const float *dT, int rows, float *ddef9, hipStream_t s) {
    hipLaunchKernelGGL(def9_bwd_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, def9, dR, dT, rows, ddef9);
}
// da0 / da1 [B][N][3] (zeroed here): gradients of a0 / a1 through the side means at gt[b][off0], gt[b][off0+1] / gt[b][off1], gt[b][off1+1]
void launch_chamfer_bwd_src2(const float *a0, const float *a1, const float *b0, const float *b1, const int32_t *i1a, const int32_t *i2a,
                             const int32_t *i1b, const int32_t *i2b, const float *gt, int gstride, int off0, int off1, int B, int N, int M,
                             float *da0, float *da1, hipStream_t s) {
    ChBwdSrc q;
    q.a[0] = a0, q.a[1] = a1, q.b[0] = b0, q.b[1] = b1, q.i1[0] = i1a, q.i1[1] = i1b, q.i2[0] = i2a, q.i2[1] = i2b;
    q.da[0] = da0, q.da[1] = da1, q.off[0] = off0, q.off[1] = off1;
    (void)hipMemsetAsync(da0, 0, (size_t)B * N * 3 * sizeof(float), s);
    (void)hipMemsetAsync(da1, 0, (size_t)B * N * 3 * sizeof(float), s);
    hipLaunchKernelGGL(chamfer_bwd_src_kernel, dim3((N + M + 255) / 256, B, 2), dim3(256), 0, s, q, gt, gstride, N, M);
}
}  // namespace dvm

using namespace dvm;

DVM_EXPORT int dvm_rot6d_bwd_f32(const float *d6, const float *g_R, int rows, float *g_d6, void *stream) {
    DVM_REQUIRE(d6 && g_R && g_d6 && rows >= 1, "dvm_rot6d_bwd_f32: bad arguments");
    hipLaunchKernelGGL(rot6d_bwd_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, d6, g_R, rows, g_d6);
    DVM_CHECK_LAUNCH("rot6d_bwd");
    return DVM_OK;
}

DVM_EXPORT int dvm_dg_warp_arap_bwd_f32(const float *xyz, int B, int N, const int32_t *nodes_idx, const int32_t *ring,
                                        const int32_t *infl_idx, const float *weights, const float *R, const float *T,
                                        const float *g_warped, const float *g_arap, float *d_R, float *d_T, void *stream) {
    DVM_REQUIRE(xyz && nodes_idx && ring && infl_idx && weights && R && T && g_warped && g_arap && d_R && d_T,
                "dvm_dg_warp_arap_bwd_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 2, "dvm_dg_warp_arap_bwd_f32: bad sizes (B=%d N=%d)", B, N);
    launch_dg_warp_arap_bwd(xyz, B, N, nodes_idx, ring, infl_idx, weights, R, T, g_warped, g_arap, 1, d_R, d_T, (hipStream_t)stream);
    DVM_CHECK_LAUNCH("dg_warp_arap_bwd");
    return DVM_OK;
}

DVM_EXPORT int dvm_chamfer_bwd_f32(const float *a, const float *b, const int32_t *idx1, const int32_t *idx2, const float *g_d1,
                                   const float *g_d2, int B, int N, int M, float *d_a, float *d_b, void *stream) {
    DVM_REQUIRE(a && b && idx1 && idx2 && g_d1 && g_d2 && d_a && d_b, "dvm_chamfer_bwd_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1 && M >= 1, "dvm_chamfer_bwd_f32: empty input");
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(d_a, 0, (size_t)B * N * 3 * sizeof(float), s);
    (void)hipMemsetAsync(d_b, 0, (size_t)B * M * 3 * sizeof(float), s);
    hipLaunchKernelGGL(chamfer_bwd_kernel, dim3((N + M + 255) / 256, B), dim3(256), 0, s, a, b, idx1, idx2, g_d1, g_d2, N, M, d_a,
                       d_b);
    DVM_CHECK_LAUNCH("chamfer_bwd");
    return DVM_OK;
}
