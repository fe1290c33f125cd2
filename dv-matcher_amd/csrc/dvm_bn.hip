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

// ---------------------------------------------------------------- point-major (rows x C) variants
// The training path keeps activations point-major [B*N][C] (the layout of the attention / kNN kernels and of
// dvm_linear_f32's inference mode), so that no transposes sit between the operators.  Same arithmetic as above: fp64
// (sum, sum of squares) per channel in a fixed order (row chunk by row chunk), one elementwise pass.
constexpr int PM_CG = 8, PM_SL = 32;   // finalize kernels: 8 channels per workgroup, 32 lanes over the partials of each

// grid (S, C / CG): rows [s * chunk, (s+1) * chunk) of one group of CG channels (pm_group_for: 128, 64 or all of them).
// A thread owns 4 consecutive channels (one 16-byte load per operand and row) of every RPP-th row, RPP = 256 / (CG/4)
// rows per pass; partial[(s * C + c) * 2 + {0,1}].
template <bool BWD>
__global__ __launch_bounds__(256) void bn_pm_reduce_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                           const float *__restrict__ dy, const float *__restrict__ y,
                                                           const float *__restrict__ mean, const float *__restrict__ invstd /* BWD */, long R,
                                                           int C, int CG, long chunk, float slope, double *__restrict__ partial) {
    __shared__ double sa[1024], sq[1024];     // [rl][c], RPP * CG <= 1024
    {   // group blockIdx.z of a merged call: rows [z R, (z + 1) R), its own statistics and partials
        const size_t go = (size_t)blockIdx.z * R * C;
        x += go;
        if (res) res += go;
        if (BWD) dy += go, y += go, mean += (size_t)blockIdx.z * C, invstd += (size_t)blockIdx.z * C;
        partial += (size_t)blockIdx.z * gridDim.x * C * 2;
    }
    const int C4 = CG >> 2, RPP = 256 / C4;
    const int c4 = threadIdx.x % C4, rl = threadIdx.x / C4, cl = c4 * 4, c = blockIdx.y * CG + cl;
    const long lo = (long)blockIdx.x * chunk, hi = lo + chunk < R ? lo + chunk : R;
    double da[4] = {0.0, 0.0, 0.0, 0.0}, dq[4] = {0.0, 0.0, 0.0, 0.0};
    if (rl < RPP) {
        f32x4 mf = {0.f, 0.f, 0.f, 0.f}, is = mf;
        if (BWD) mf = *(const f32x4 *)(mean + c), is = *(const f32x4 *)(invstd + c);
        auto body = [&](long r) {
            const size_t off = (size_t)r * C + c;
            f32x4 v = *(const f32x4 *)(x + off);
            if (res) v += *(const f32x4 *)(res + off);
            if (BWD) {
                const f32x4 g4 = *(const f32x4 *)(dy + off), y4 = *(const f32x4 *)(y + off);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dz = g4[k] * (y4[k] > 0.f ? 1.f : slope);
                    da[k] += (double)dz;
                    dq[k] = fma((double)dz, (double)((v[k] - mf[k]) * is[k]), dq[k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    da[k] += (double)v[k];
                    dq[k] = fma((double)v[k], (double)v[k], dq[k]);
                }
            }
        };
        long r = lo + rl;
        for (; r + 3L * RPP < hi; r += 4L * RPP) {   // 4 rows in flight
            body(r);
            body(r + RPP);
            body(r + 2L * RPP);
            body(r + 3L * RPP);
        }
        for (; r < hi; r += RPP) body(r);
#pragma unroll
        for (int k = 0; k < 4; ++k) sa[rl * CG + cl + k] = da[k], sq[rl * CG + cl + k] = dq[k];
    }
    __syncthreads();
    for (int cc = threadIdx.x; cc < CG; cc += 256) {
        double ta = 0.0, tq = 0.0;
        for (int q = 0; q < RPP; ++q) ta += sa[q * CG + cc], tq += sq[q * CG + cc];
        const size_t o = ((size_t)blockIdx.x * C + blockIdx.y * CG + cc) * 2;
        partial[o] = ta;
        partial[o + 1] = tq;
    }
}

// sum of the S partials of 8 channels by one workgroup (32 lanes per channel, fixed order) -> (ta, tq) in lanes sl == 0
__device__ inline bool pm_sum_partials(const double *__restrict__ partial, int S, int C, int c, int cl, int sl, double &ta, double &tq) {
    __shared__ double pa[PM_SL][PM_CG], pq[PM_SL][PM_CG];
    double a = 0.0, q = 0.0;
    if (c < C)
        for (int s = sl; s < S; s += PM_SL) {
            const double2 v = *(const double2 *)(partial + ((size_t)s * C + c) * 2);
            a += v.x, q += v.y;
        }
    pa[sl][cl] = a, pq[sl][cl] = q;
    __syncthreads();
    if (sl != 0 || c >= C) return false;
    ta = 0.0, tq = 0.0;
#pragma unroll
    for (int k = 0; k < PM_SL; ++k) ta += pa[k][cl], tq += pq[k][cl];
    return true;
}

// grid (ceil(C / 8)) x 256 threads: partials -> fin[g][c] = {mean, invstd, gamma * invstd, beta} (forward; + running
// statistics) or {mean(dz), mean(dz xhat), gamma * invstd, -} (backward; + dgamma, dbeta), for every group of a merged call
__global__ __launch_bounds__(256) void bn_pm_finalize_fwd_kernel(const double *__restrict__ partial, int S, long R, int C,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                                 float momentum, float *__restrict__ fin, float *__restrict__ mean_out,
                                                                 float *__restrict__ invstd_out, float *__restrict__ running_mean,
                                                                 float *__restrict__ running_var, float *__restrict__ unb_out, int groups,
                                                                 const double *__restrict__ glob) {
    // glob (cross-rank statistics): [groups][C][2] totals + [groups] row counts, summed over the ranks — then `partial` is not read
    const int cl = threadIdx.x & (PM_CG - 1), sl = threadIdx.x / PM_CG, c = blockIdx.x * PM_CG + cl;
    for (int g = 0; g < groups; ++g) {   // group after group: the running statistics take the groups' updates in order
        double ta, tq;
        bool lead;
        double Rd = (double)R;
        if (glob) {
            lead = sl == 0 && c < C;
            if (lead) ta = glob[((size_t)g * C + c) * 2], tq = glob[((size_t)g * C + c) * 2 + 1], Rd = glob[(size_t)groups * C * 2 + g];
        } else {
            lead = pm_sum_partials(partial + (size_t)g * S * C * 2, S, C, c, cl, sl, ta, tq);
        }
        if (lead) {
            const double mean = ta / Rd;
            double var = tq / Rd - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps)), mf = (float)mean;
            mean_out[(size_t)g * C + c] = mf;
            invstd_out[(size_t)g * C + c] = invstd;
            const double unb = Rd > 1.0 ? var * Rd / (Rd - 1.0) : var;
            if (unb_out) unb_out[(size_t)g * C + c] = (float)unb;   // for a deferred running-statistics update: the same float
            if (running_mean) {
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mf;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
            }
            float *f = fin + ((size_t)g * C + c) * 4;
            f[0] = mf;
            f[1] = invstd;
            f[2] = (gamma ? gamma[c] : 1.f) * invstd;
            f[3] = beta ? beta[c] : 0.f;
        }
        if (!glob) __syncthreads();   // (pm_sum_partials' shared arrays are reused by the next group)
    }
}

// partials -> per-(group, channel) totals in pm_sum_partials' order + the groups' row counts: what the ranks of a data-parallel
// step combine (tot [groups][C][2] + [groups], doubles)
__global__ __launch_bounds__(256) void bn_pm_totals_kernel(const double *__restrict__ partial, int S, long R, int C, int groups,
                                                           double *__restrict__ tot) {
    const int cl = threadIdx.x & (PM_CG - 1), sl = threadIdx.x / PM_CG, c = blockIdx.x * PM_CG + cl;
    for (int g = 0; g < groups; ++g) {
        double ta, tq;
        if (pm_sum_partials(partial + (size_t)g * S * C * 2, S, C, c, cl, sl, ta, tq)) {
            tot[((size_t)g * C + c) * 2] = ta;
            tot[((size_t)g * C + c) * 2 + 1] = tq;
            if (c == 0) tot[(size_t)groups * C * 2 + g] = (double)R;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void bn_pm_finalize_bwd_kernel(const double *__restrict__ partial, int S, long R, int C,
                                                                 const float *__restrict__ gamma, const float *__restrict__ mean,
                                                                 const float *__restrict__ invstd, float *__restrict__ fin,
                                                                 float *__restrict__ dgamma, float *__restrict__ dbeta, int accumulate, int groups,
                                                                 const double *__restrict__ glob) {
    // glob (cross-rank statistics, as in the forward): the means of dz and dz xhat are taken over ALL ranks' rows; the parameter
    // gradients stay this rank's own sums (the gradient all-reduce adds the ranks')
    const int cl = threadIdx.x & (PM_CG - 1), sl = threadIdx.x / PM_CG, c = blockIdx.x * PM_CG + cl;
    float sg = 0.f, sb = 0.f;   // the groups' parameter gradients, added in group order
    bool any = false;
    for (int g = 0; g < groups; ++g) {
        double ta, tq;
        const bool lead = pm_sum_partials(partial + (size_t)g * S * C * 2, S, C, c, cl, sl, ta, tq);
        if (lead) {
            any = true;
            sg += (float)tq, sb += (float)ta;
            float *f = fin + ((size_t)g * C + c) * 4;
            double Rd = (double)R;
            if (glob) ta = glob[((size_t)g * C + c) * 2], tq = glob[((size_t)g * C + c) * 2 + 1], Rd = glob[(size_t)groups * C * 2 + g];
            f[0] = (float)(ta / Rd);
            f[1] = (float)(tq / Rd);
            f[2] = (gamma ? gamma[c] : 1.f) * invstd[(size_t)g * C + c];
            f[3] = 0.f;
        }
        __syncthreads();
    }
    if (!any) return;
    // accumulate: an atomic add — two network calls whose backward passes run side by side on two streams add into the same buffers
    // (a zeroed buffer plus two addends gives the same bits in either order)
    if (dgamma) {
        if (accumulate) atomicAdd(dgamma + c, sg);
        else dgamma[c] = sg;
    }
    if (dbeta) {
        if (accumulate) atomicAdd(dbeta + c, sb);
        else dbeta[c] = sb;
    }
}

// elementwise pass, 4 consecutive channels per thread (C % 4 == 0)
template <bool BWD>
__global__ __launch_bounds__(256) void bn_pm_apply_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                          const float *__restrict__ dy, const float *__restrict__ y_in,
                                                          const float *__restrict__ fin, const float *__restrict__ mean,
                                                          const float *__restrict__ invstd, long R, int C, float slope,
                                                          float *__restrict__ out, long Rg) {
    const long q0 = (long)blockIdx.x * 256 + threadIdx.x;   // float4 index
    const long e0 = q0 * 4;
    if (e0 >= R * C) return;
    if (Rg < R) {   // merged call: the row's group selects the statistics (R = all rows, Rg = rows per group)
        const long g = (e0 / C) / Rg;
        fin += (size_t)g * C * 4;
        if (BWD) mean += (size_t)g * C, invstd += (size_t)g * C;
    }
    const int c = 4 * (q0 < (1L << 32) ? (int)((unsigned)q0 % (unsigned)(C >> 2)) : (int)(q0 % (C >> 2)));
    f32x4 v = *(const f32x4 *)(x + e0);
    if (res) v += *(const f32x4 *)(res + e0);
    f32x4 o;
    if (BWD) {
        const f32x4 g4 = *(const f32x4 *)(dy + e0), y4 = *(const f32x4 *)(y_in + e0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dz = g4[k] * (y4[k] > 0.f ? 1.f : slope);
            o[k] = fin[(c + k) * 4 + 2] * ((dz - fin[(c + k) * 4]) - ((v[k] - mean[c + k]) * invstd[c + k]) * fin[(c + k) * 4 + 1]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = (v[k] - fin[(c + k) * 4]) * fin[(c + k) * 4 + 2] + fin[(c + k) * 4 + 3];
            o[k] = t > 0.f ? t : t * slope;
        }
    }
    *(f32x4 *)(out + e0) = o;
}

// channel group of one reduction workgroup: 512- or 256-byte row segments, else the whole row (C <= 1024)
int pm_group_for(int C) { return C % 128 == 0 ? 128 : C % 64 == 0 ? 64 : C; }
// row chunks of the point-major reduction: ~256 workgroups in all, every thread row (256 / (CG/4) per workgroup) >= 4 rows
long pm_chunk_for(long R, int C) {
    const int CG = pm_group_for(C), groups = C / CG;
    const long rpp = 256 / (CG >> 2);
    long chunk = 4 * rpp;
    while (((R + chunk - 1) / chunk) * groups > 256) chunk *= 2;
    return chunk;
}
int pm_splits_for(long R, int C) {
    const long chunk = pm_chunk_for(R, C);
    return (int)((R + chunk - 1) / chunk);
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

DVM_EXPORT size_t dvm_bn_pm_workspace_bytes(long R, int C) {
    if (R < 1 || C < 4 || C % 4 != 0 || C > 1024) return 0;
    return align_up((size_t)C * pm_splits_for(R, C) * 2 * sizeof(double)) + align_up((size_t)C * 4 * sizeof(float));
}

DVM_EXPORT int dvm_bn_act_train_fwd_pm_f32(const float *x, const float *res, const float *gamma, const float *beta, long R, int C, float eps,
                                           float slope, float momentum, float *y, float *save_mean, float *save_invstd,
                                           float *running_mean, float *running_var, void *ws, size_t ws_bytes, void *stream) {
    return dvm_bn_act_train_fwd_pm_var_f32(x, res, gamma, beta, R, C, 1, eps, slope, momentum, y, save_mean, save_invstd, nullptr, running_mean,
                                           running_var, ws, ws_bytes, stream);
}

DVM_EXPORT size_t dvm_bn_pm_groups_workspace_bytes(long R, int C, int groups) {
    if (R < 1 || C < 4 || C % 4 != 0 || C > 1024 || groups < 1) return 0;
    return align_up((size_t)groups * C * pm_splits_for(R, C) * 2 * sizeof(double)) + align_up((size_t)groups * C * 4 * sizeof(float));
}

DVM_EXPORT size_t dvm_bn_pm_sync_bytes(int C, int groups) { return align_up(((size_t)groups * C * 2 + groups) * sizeof(double)); }

DVM_EXPORT int dvm_bn_act_train_fwd_pm_var_f32(const float *x, const float *res, const float *gamma, const float *beta, long R, int C, int groups,
                                               float eps, float slope, float momentum, float *y, float *save_mean, float *save_invstd,
                                               float *save_var_unbiased, float *running_mean, float *running_var, void *ws, size_t ws_bytes,
                                               void *stream) {
    return dvm_bn_act_train_fwd_pm_sync_f32(x, res, gamma, beta, R, C, groups, eps, slope, momentum, y, save_mean, save_invstd, save_var_unbiased,
                                            running_mean, running_var, ws, ws_bytes, nullptr, nullptr, stream);
}

DVM_EXPORT int dvm_bn_act_train_fwd_pm_sync_f32(const float *x, const float *res, const float *gamma, const float *beta, long R, int C, int groups,
                                                float eps, float slope, float momentum, float *y, float *save_mean, float *save_invstd,
                                                float *save_var_unbiased, float *running_mean, float *running_var, void *ws, size_t ws_bytes,
                                                const dvm_collective *coll, void *sync_buf, void *stream) {
    DVM_REQUIRE(!coll || (coll->allreduce && sync_buf), "dvm_bn_act_train_fwd_pm_sync_f32: a collective needs its function and the statistics buffer");
    DVM_REQUIRE(groups >= 1 && groups <= 64, "dvm_bn_act_train_fwd_pm_var_f32: bad group count %d", groups);
    DVM_REQUIRE(x && y && save_mean && save_invstd, "dvm_bn_act_train_fwd_pm_f32: null pointer");
    DVM_REQUIRE(R >= 1 && C >= 4 && C % 4 == 0 && C <= 1024, "dvm_bn_act_train_fwd_pm_f32: need R >= 1 and C a multiple of 4, at most 1024 (R=%ld C=%d)", R, C);
    DVM_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "dvm_bn_act_train_fwd_pm_f32: running_mean/var go together");
    const int S = pm_splits_for(R, C);
    Arena ar(ws, ws_bytes);
    double *partial = ar.take<double>((size_t)groups * C * S * 2);
    float *fin = ar.take<float>((size_t)groups * C * 4);
    if (!ar.ok()) {
        set_error("dvm_bn_act_train_fwd_pm_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int CG = pm_group_for(C);
    const long Rall = R * groups;   // R: rows PER GROUP
    hipLaunchKernelGGL(bn_pm_reduce_kernel<false>, dim3(S, C / CG, groups), dim3(256), 0, s, x, res, nullptr, nullptr, nullptr, nullptr, R, C, CG,
                       pm_chunk_for(R, C), slope, partial);
    const double *glob = nullptr;
    if (coll) {   // per-(group, channel) totals + row counts -> summed over the ranks by the caller's collective, on this stream
        double *tot = (double *)sync_buf;
        hipLaunchKernelGGL(bn_pm_totals_kernel, dim3((C + PM_CG - 1) / PM_CG), dim3(256), 0, s, partial, S, R, C, groups, tot);
        const int rc = coll->allreduce(coll->user, tot, (size_t)groups * C * 2 + groups, 1, 0, stream);
        DVM_REQUIRE(rc == 0, "dvm_bn_act_train_fwd_pm_sync_f32: the caller's all-reduce failed (%d)", rc);
        glob = tot;
    }
    hipLaunchKernelGGL(bn_pm_finalize_fwd_kernel, dim3((C + PM_CG - 1) / PM_CG), dim3(256), 0, s, partial, S, R, C, gamma, beta, eps, momentum, fin,
                       save_mean, save_invstd, running_mean, running_var, save_var_unbiased, groups, glob);
    hipLaunchKernelGGL(bn_pm_apply_kernel<false>, dim3((unsigned)((Rall * C / 4 + 255) / 256)), dim3(256), 0, s, x, res, nullptr, nullptr, fin,
                       nullptr, nullptr, Rall, C, slope, y, R);
    DVM_CHECK_LAUNCH("bn_act_train_fwd_pm");
    return DVM_OK;
}

DVM_EXPORT int dvm_bn_act_train_bwd_pm_f32(const float *dy, const float *y, const float *x, const float *res, const float *gamma,
                                           const float *save_mean, const float *save_invstd, long R, int C, float slope, float *dx,
                                           float *dgamma, float *dbeta, int accumulate, void *ws, size_t ws_bytes, void *stream) {
    return dvm_bn_act_train_bwd_pm_groups_f32(dy, y, x, res, gamma, save_mean, save_invstd, R, C, 1, slope, dx, dgamma, dbeta, accumulate, ws, ws_bytes,
                                              stream);
}

DVM_EXPORT int dvm_bn_act_train_bwd_pm_groups_f32(const float *dy, const float *y, const float *x, const float *res, const float *gamma,
                                                  const float *save_mean, const float *save_invstd, long R, int C, int groups, float slope,
                                                  float *dx, float *dgamma, float *dbeta, int accumulate, void *ws, size_t ws_bytes, void *stream) {
    return dvm_bn_act_train_bwd_pm_sync_f32(dy, y, x, res, gamma, save_mean, save_invstd, R, C, groups, slope, dx, dgamma, dbeta, accumulate, ws, ws_bytes,
                                            nullptr, nullptr, stream);
}

DVM_EXPORT int dvm_bn_act_train_bwd_pm_sync_f32(const float *dy, const float *y, const float *x, const float *res, const float *gamma,
                                                const float *save_mean, const float *save_invstd, long R, int C, int groups, float slope,
                                                float *dx, float *dgamma, float *dbeta, int accumulate, void *ws, size_t ws_bytes,
                                                const dvm_collective *coll, void *sync_buf, void *stream) {
    DVM_REQUIRE(!coll || (coll->allreduce && sync_buf), "dvm_bn_act_train_bwd_pm_sync_f32: a collective needs its function and the statistics buffer");
    DVM_REQUIRE(dy && y && x && save_mean && save_invstd && dx, "dvm_bn_act_train_bwd_pm_f32: null pointer");
    DVM_REQUIRE(groups >= 1 && groups <= 64, "dvm_bn_act_train_bwd_pm_groups_f32: bad group count %d", groups);
    DVM_REQUIRE(R >= 1 && C >= 4 && C % 4 == 0 && C <= 1024, "dvm_bn_act_train_bwd_pm_f32: need R >= 1 and C a multiple of 4, at most 1024 (R=%ld C=%d)", R, C);
    const int S = pm_splits_for(R, C);
    Arena ar(ws, ws_bytes);
    double *partial = ar.take<double>((size_t)groups * C * S * 2);
    float *fin = ar.take<float>((size_t)groups * C * 4);
    if (!ar.ok()) {
        set_error("dvm_bn_act_train_bwd_pm_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int CG = pm_group_for(C);
    const long Rall = R * groups;   // R: rows PER GROUP
    hipLaunchKernelGGL(bn_pm_reduce_kernel<true>, dim3(S, C / CG, groups), dim3(256), 0, s, x, res, dy, y, save_mean, save_invstd, R, C, CG,
                       pm_chunk_for(R, C), slope, partial);
    const double *glob = nullptr;
    if (coll) {
        double *tot = (double *)sync_buf;
        hipLaunchKernelGGL(bn_pm_totals_kernel, dim3((C + PM_CG - 1) / PM_CG), dim3(256), 0, s, partial, S, R, C, groups, tot);
        const int rc = coll->allreduce(coll->user, tot, (size_t)groups * C * 2 + groups, 1, 0, stream);
        DVM_REQUIRE(rc == 0, "dvm_bn_act_train_bwd_pm_sync_f32: the caller's all-reduce failed (%d)", rc);
        glob = tot;
    }
    hipLaunchKernelGGL(bn_pm_finalize_bwd_kernel, dim3((C + PM_CG - 1) / PM_CG), dim3(256), 0, s, partial, S, R, C, gamma, save_mean, save_invstd, fin,
                       dgamma, dbeta, accumulate, groups, glob);
    hipLaunchKernelGGL(bn_pm_apply_kernel<true>, dim3((unsigned)((Rall * C / 4 + 255) / 256)), dim3(256), 0, s, x, res, dy, y, fin, save_mean,
                       save_invstd, Rall, C, slope, dx, R);
    DVM_CHECK_LAUNCH("bn_act_train_bwd_pm");
    return DVM_OK;
}

// Deferred running-statistics update of up to 32 BatchNorms in one launch: running = (1 - momentum) running + momentum batch, the
// expression (and roundings) of the fused forward kernel — for callers that ran the forward with running_mean == NULL (two network
// calls side by side on two streams share every BatchNorm: their updates are applied afterwards, in the reference's call order).
namespace dvm {
namespace {
struct RunUpd {
    float *rm[32], *rv[32];
    const float *mean[32], *var[32];
    int C[32], count;
};
__global__ __launch_bounds__(256) void bn_running_update_kernel(const RunUpd u, float momentum) {
    const int k = blockIdx.y;
    if (k >= u.count) return;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < u.C[k]; c += gridDim.x * blockDim.x) {
        u.rm[k][c] = (1.f - momentum) * u.rm[k][c] + momentum * u.mean[k][c];
        u.rv[k][c] = (1.f - momentum) * u.rv[k][c] + momentum * u.var[k][c];
    }
}
}  // namespace
int launch_bn_running_update(float *const *rm, float *const *rv, const float *const *mean, const float *const *var, const int *C, int count,
                             float momentum, hipStream_t s) {
    for (int k0 = 0; k0 < count; k0 += 32) {
        RunUpd u;
        u.count = count - k0 < 32 ? count - k0 : 32;
        for (int k = 0; k < u.count; ++k) u.rm[k] = rm[k0 + k], u.rv[k] = rv[k0 + k], u.mean[k] = mean[k0 + k], u.var[k] = var[k0 + k], u.C[k] = C[k0 + k];
        hipLaunchKernelGGL(bn_running_update_kernel, dim3(2, u.count), dim3(256), 0, s, u, momentum);
    }
    return DVM_OK;
}
}  // namespace dvm
