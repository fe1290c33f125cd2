// dvm_bn.hip — training-mode BatchNorm1d over (B,C,N) fused with the residual add in front of it and the
// (Leaky)ReLU behind it, forward and backward (reference: nn.BatchNorm1d + the adds / activations around it in
// models/model.py:97-123, 325-395, 506-529).  Two launches each way instead of the 3-5 elementwise / reduction
// launches of the unfused form, each tensor read at most twice and written once.
//   z = x (+ res);  mean_c, var_c over (b, n);  y = act(gamma_c (z - mean_c) / sqrt(var_c + eps) + beta_c),
//   act(t) = t > 0 ? t : slope * t   (slope = 1: no activation, 0: ReLU, 0.2: LeakyReLU)
// Statistics are accumulated as (sum, sum of squares) in fp64 (the kernels are memory-bound; the few fp64 adds are free).
#include "dvm_common.h"

namespace dvm {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double block_sum(double v, double *sm /* [4] */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[wave] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// grid (C, S): split s of channel c covers elements e = b * N + n in [s * chunk, (s + 1) * chunk)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float *__restrict__ x, const float *__restrict__ res, int B, int C,
                                                       int N, int chunk, double *__restrict__ partial) {
    __shared__ double sm[4];
    const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const long total = (long)B * N;
    const long lo = (long)s * chunk, hi = lo + chunk < total ? lo + chunk : total;
    double da = 0.0, dq = 0.0;
    for (long e = lo + threadIdx.x; e < hi; e += 256) {
        const long b = e / N, n = e - b * N;
        const size_t off = ((size_t)b * C + c) * N + n;
        float v = x[off];
        if (res) v += res[off];
        da += (double)v;
        dq = fma((double)v, (double)v, dq);
    }
    const double ta = block_sum(da, sm), tq = block_sum(dq, sm);
    if (threadIdx.x == 0) {
        partial[((size_t)c * S + s) * 2] = ta;
        partial[((size_t)c * S + s) * 2 + 1] = tq;
    }
}

// grid (ceil(B*N / 1024), C)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                       const double *__restrict__ partial, int S, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, int B, int C, int N, float eps, float slope,
                                                       float momentum, float *__restrict__ y, float *__restrict__ mean_out,
                                                       float *__restrict__ invstd_out, float *__restrict__ running_mean,
                                                       float *__restrict__ running_var) {
    const int c = blockIdx.y;
    const long total = (long)B * N;
    double ta = 0.0, tq = 0.0;
    for (int s = 0; s < S; ++s) ta += partial[((size_t)c * S + s) * 2], tq += partial[((size_t)c * S + s) * 2 + 1];
    const double mean = ta / (double)total;
    double var = tq / (double)total - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float mf = (float)mean;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        mean_out[c] = mf;
        invstd_out[c] = invstd;
        if (running_mean) {  // PyTorch: running = (1 - m) running + m * batch; the variance unbiased
            const double unb = total > 1 ? var * (double)total / (double)(total - 1) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mf;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    const long e0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e0 >= total) return;
    if ((N & 3) == 0) {  // 4 consecutive n of one (b, c) row
        const long b = e0 / N, n = e0 - b * N;
        const size_t off = ((size_t)b * C + c) * N + n;
        f32x4 v = *(const f32x4 *)(x + off);
        if (res) v += *(const f32x4 *)(res + off);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = (v[k] - mf) * sc + bt;
            o[k] = t > 0.f ? t : t * slope;
        }
        *(f32x4 *)(y + off) = o;
    } else {
        for (long e = e0; e < e0 + 4 && e < total; ++e) {
            const long b = e / N, n = e - b * N;
            const size_t off = ((size_t)b * C + c) * N + n;
            float v = x[off];
            if (res) v += res[off];
            const float t = (v - mf) * sc + bt;
            y[off] = t > 0.f ? t : t * slope;
        }
    }
}

// backward pass 1: per (c, s) partial sums of dz and dz * xhat, dz = dy * act'(.) taken from the sign of y
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                            const float *__restrict__ x, const float *__restrict__ res,
                                                            const float *__restrict__ mean, const float *__restrict__ invstd, int B,
                                                            int C, int N, int chunk, float slope, double *__restrict__ partial) {
    __shared__ double sm[4];
    const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const long total = (long)B * N;
    const long lo = (long)s * chunk, hi = lo + chunk < total ? lo + chunk : total;
    const float mf = mean[c], is = invstd[c];
    double da = 0.0, dq = 0.0;
    for (long e = lo + threadIdx.x; e < hi; e += 256) {
        const long b = e / N, n = e - b * N;
        const size_t off = ((size_t)b * C + c) * N + n;
        float v = x[off];
        if (res) v += res[off];
        const float dz = dy[off] * (y[off] > 0.f ? 1.f : slope);
        da += (double)dz;
        dq = fma((double)dz, (double)((v - mf) * is), dq);
    }
    const double ta = block_sum(da, sm), tq = block_sum(dq, sm);
    if (threadIdx.x == 0) {
        partial[((size_t)c * S + s) * 2] = ta;
        partial[((size_t)c * S + s) * 2 + 1] = tq;
    }
}

// backward pass 2: dx = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat));  dgamma = sum dz xhat, dbeta = sum dz
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                           const float *__restrict__ x, const float *__restrict__ res,
                                                           const float *__restrict__ mean, const float *__restrict__ invstd,
                                                           const float *__restrict__ gamma, const double *__restrict__ partial, int S,
                                                           int B, int C, int N, float slope, float *__restrict__ dx,
                                                           float *__restrict__ dgamma, float *__restrict__ dbeta) {
    const int c = blockIdx.y;
    const long total = (long)B * N;
    double ta = 0.0, tq = 0.0;
    for (int s = 0; s < S; ++s) ta += partial[((size_t)c * S + s) * 2], tq += partial[((size_t)c * S + s) * 2 + 1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (dgamma) dgamma[c] = (float)tq;
        if (dbeta) dbeta[c] = (float)ta;
    }
    const float mf = mean[c], is = invstd[c];
    const float g = gamma ? gamma[c] : 1.f;
    const float k0 = g * is, m1 = (float)(ta / (double)total), m2 = (float)(tq / (double)total);
    const long e0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e0 >= total) return;
    if ((N & 3) == 0) {
        const long b = e0 / N, n = e0 - b * N;
        const size_t off = ((size_t)b * C + c) * N + n;
        f32x4 v = *(const f32x4 *)(x + off);
        if (res) v += *(const f32x4 *)(res + off);
        const f32x4 g4 = *(const f32x4 *)(dy + off), y4 = *(const f32x4 *)(y + off);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dz = g4[k] * (y4[k] > 0.f ? 1.f : slope);
            o[k] = k0 * ((dz - m1) - ((v[k] - mf) * is) * m2);
        }
        *(f32x4 *)(dx + off) = o;
    } else {
        for (long e = e0; e < e0 + 4 && e < total; ++e) {
            const long b = e / N, n = e - b * N;
            const size_t off = ((size_t)b * C + c) * N + n;
            float v = x[off];
            if (res) v += res[off];
            const float dz = dy[off] * (y[off] > 0.f ? 1.f : slope);
            dx[off] = k0 * ((dz - m1) - ((v - mf) * is) * m2);
        }
    }
}

int splits_for(int B, int C, int N) {  // enough workgroups to fill the chip, at least ~1k elements each
    const long total = (long)B * N;
    int S = 1;
    while ((long)C * S < 1024 && total / (S * 2) >= 1024 && S < 64) S *= 2;
    return S;
}

}  // namespace
}  // namespace dvm

using namespace dvm;

DVM_EXPORT size_t dvm_bn_workspace_bytes(int B, int C, int N) { return align_up((size_t)C * splits_for(B, C, N) * 2 * sizeof(double)); }

DVM_EXPORT int dvm_bn_act_train_fwd_f32(const float *x, const float *res, const float *gamma, const float *beta, int B, int C, int N,
                                        float eps, float slope, float momentum, float *y, float *save_mean, float *save_invstd,
                                        float *running_mean, float *running_var, void *ws, size_t ws_bytes, void *stream) {
    DVM_REQUIRE(x && y && save_mean && save_invstd, "dvm_bn_act_train_fwd_f32: null pointer");
    DVM_REQUIRE(B >= 1 && C >= 1 && N >= 1, "dvm_bn_act_train_fwd_f32: empty input (B=%d C=%d N=%d)", B, C, N);
    DVM_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "dvm_bn_act_train_fwd_f32: running_mean/var go together");
    const int S = splits_for(B, C, N);
    Arena ar(ws, ws_bytes);
    double *partial = ar.take<double>((size_t)C * S * 2);
    if (!ar.ok()) {
        set_error("dvm_bn_act_train_fwd_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * N;
    const int chunk = (int)((total + S - 1) / S);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, S), dim3(256), 0, s, x, res, B, C, N, chunk, partial);
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((total + 1023) / 1024), C), dim3(256), 0, s, x, res, partial, S, gamma, beta, B, C,
                       N, eps, slope, momentum, y, save_mean, save_invstd, running_mean, running_var);
    DVM_CHECK_LAUNCH("bn_act_train_fwd");
    return DVM_OK;
}

DVM_EXPORT int dvm_bn_act_train_bwd_f32(const float *dy, const float *y, const float *x, const float *res, const float *gamma,
                                        const float *save_mean, const float *save_invstd, int B, int C, int N, float slope, float *dx,
                                        float *dgamma, float *dbeta, void *ws, size_t ws_bytes, void *stream) {
    DVM_REQUIRE(dy && y && x && save_mean && save_invstd && dx, "dvm_bn_act_train_bwd_f32: null pointer");
    DVM_REQUIRE(B >= 1 && C >= 1 && N >= 1, "dvm_bn_act_train_bwd_f32: empty input (B=%d C=%d N=%d)", B, C, N);
    const int S = splits_for(B, C, N);
    Arena ar(ws, ws_bytes);
    double *partial = ar.take<double>((size_t)C * S * 2);
    if (!ar.ok()) {
        set_error("dvm_bn_act_train_bwd_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * N;
    const int chunk = (int)((total + S - 1) / S);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, S), dim3(256), 0, s, dy, y, x, res, save_mean, save_invstd, B, C, N, chunk, slope,
                       partial);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((total + 1023) / 1024), C), dim3(256), 0, s, dy, y, x, res, save_mean,
                       save_invstd, gamma, partial, S, B, C, N, slope, dx, dgamma, dbeta);
    DVM_CHECK_LAUNCH("bn_act_train_bwd");
    return DVM_OK;
}
