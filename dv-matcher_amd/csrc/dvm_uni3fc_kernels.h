// dvm_uni3fc_kernels.h — the small element-wise kernels LG-Net's native orchestration (dvm_uni3fc.hip: eval forward,
// dvm_uni3fc_train.hip: training forward / backward) runs between the library's own layer launches: what the Python
// paths of dv-matcher_amd/models/model.py leave to torch (the position encoding added transposed, x - x_r, channel
// concatenations, the max over the points, ...).  Included by both translation units (internal linkage).
#pragma once
#include "dvm_common.h"

namespace dvm {
typedef float f32x4 __attribute__((ext_vector_type(4)));
namespace {

// out[b][n][c] += pe[b][c][n]   (f is point-major, the position encoding channel-major: 32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void add_transposed_kernel(float *__restrict__ f, const float *__restrict__ pe, int N, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 8 rows per pass
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, n = n0 + tx;
        tile[r][tx] = (c < C && n < N) ? pe[((size_t)b * C + c) * N + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, c = c0 + tx;
        if (n < N && c < C) {
            const size_t o = ((size_t)b * N + n) * C + c;
            f[o] = f[o] + tile[tx][r];
        }
    }
}

// out = a - b
__global__ __launch_bounds__(256) void sub_kernel(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b, long n4, f32x4 *__restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) out[i] = a[i] - b[i];
}

// out[r][c] = t[c] + (x[r][c] + a[r][c]) * s[c], product and sum rounded separately   (the eval-mode BatchNorm of the attention
// residual as models/model.py::_N2P.infer_pm evaluates it with torch.addcmul: the two paths agree bit for bit; C % 4 == 0)
__global__ __launch_bounds__(256) void add_affine_kernel(const f32x4 *__restrict__ x, const f32x4 *__restrict__ a, const f32x4 *__restrict__ s,
                                                         const f32x4 *__restrict__ t, long n4, int c4, f32x4 *__restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4);
        const f32x4 v = x[i] + a[i], sv = s[c], tv = t[c];
        f32x4 o;
        o.x = __fadd_rn(tv.x, __fmul_rn(v.x, sv.x)), o.y = __fadd_rn(tv.y, __fmul_rn(v.y, sv.y)), o.z = __fadd_rn(tv.z, __fmul_rn(v.z, sv.z)), o.w = __fadd_rn(tv.w, __fmul_rn(v.w, sv.w));
        out[i] = o;
    }
}

// out[b][c] = max over n of x[b][n][c]: rows split over blockIdx.z, partial maxima combined with an ordered-integer atomic
// (out pre-set to -inf); the maximum does not depend on the order
__device__ __forceinline__ void atomic_max_float(float *addr, float v) {
    if (!(__float_as_uint(v) >> 31))   // by the SIGN BIT: -0.0f belongs to the negative branch (as an int it is INT_MIN and would never win)
        atomicMax((int *)addr, __float_as_int(v));
    else
        atomicMin((unsigned *)addr, __float_as_uint(v));
}
__global__ __launch_bounds__(256) void colmax_kernel(const float *__restrict__ x, int N, int C, int rows_per, float *__restrict__ out) {
    __shared__ float part[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int n0 = blockIdx.z * rows_per, n1 = min(N, n0 + rows_per);
    float m = -INFINITY;
    if (c < C)
        for (int n = n0 + g; n < n1; n += 4) m = fmaxf(m, x[((size_t)b * N + n) * C + c]);
    part[g][threadIdx.x & 63] = m;
    __syncthreads();
    if (g == 0 && c < C) {
        m = fmaxf(fmaxf(part[0][threadIdx.x], part[1][threadIdx.x]), fmaxf(part[2][threadIdx.x], part[3][threadIdx.x]));
        atomic_max_float(out + (size_t)b * C + c, m);
    }
}
__global__ void fill_kernel(float *__restrict__ p, long n, float v) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// out[r] = [s0[r] | s1[r] | s2[r] | s3[r]]   (ns sources of C floats each, C % 4 == 0)
struct CatArgs {
    const f32x4 *src[4];
    int ns, c4;
    long rows;
    f32x4 *out;
};
__global__ __launch_bounds__(256) void concat_kernel(const CatArgs a) {
    const long total = a.rows * a.ns * a.c4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = a.ns * a.c4, col = (int)(i % w);
        const long r = i / w;
        a.out[i] = a.src[col / a.c4][r * a.c4 + col % a.c4];
    }
}

// ---------------------------------------------------------------- training-only helpers (dvm_uni3fc_train.hip)
// out = a + b  (+ c when c != nullptr)
__global__ __launch_bounds__(256) void add3_kernel(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b, const f32x4 *__restrict__ c, long n4,
                                                   f32x4 *__restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 v = a[i] + b[i];
        if (c) v = v + c[i];
        out[i] = v;
    }
}
// s = g + d ; neg = -d   (SA_Layer backward: the gradient of x - x_r reaches x with +, x_r with -)
__global__ __launch_bounds__(256) void add_neg_kernel(const f32x4 *__restrict__ g, const f32x4 *__restrict__ d, long n4, f32x4 *__restrict__ s,
                                                      f32x4 *__restrict__ neg) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 dv = d[i];
        s[i] = g[i] + dv;
        neg[i] = -dv;
    }
}
// g *= (y > 0 ? 1 : slope)   (ATen's leaky_relu_backward with the activation's OUTPUT as `self`: slope > 0)
__global__ __launch_bounds__(256) void act_bwd_kernel(f32x4 *__restrict__ g, const f32x4 *__restrict__ y, long n4, float slope) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 v = g[i];
        const f32x4 yv = y[i];
        v.x = yv.x > 0.f ? v.x : v.x * slope, v.y = yv.y > 0.f ? v.y : v.y * slope, v.z = yv.z > 0.f ? v.z : v.z * slope, v.w = yv.w > 0.f ? v.w : v.w * slope;
        g[i] = v;
    }
}
// dst[r][0..C) = src[r][off .. off+C) (+ add[r][0..C) when add != nullptr): a column slice of a wider gradient (the backward of a
// channel concatenation), optionally summed with the gradient that reached the same tensor along its other path
__global__ __launch_bounds__(256) void slice_add_kernel(const float *__restrict__ src, int ld, int off, const f32x4 *__restrict__ add, long rows, int c4,
                                                        f32x4 *__restrict__ dst) {
    const long total = rows * c4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c4;
        const int c = (int)(i % c4);
        f32x4 v = *(const f32x4 *)(src + r * ld + off + 4 * c);
        if (add) v = v + add[i];
        dst[i] = v;
    }
}
// Max over the points WITH its position (training: torch.max(dim=1) routes the gradient to the arg-max row): partial maxima
// are combined with a 64-bit atomic on (order-preserving bits of the value << 32 | ~row), so equal maxima resolve to the
// lowest row whatever the order; packed must be zeroed first.
__device__ __forceinline__ unsigned ordered_bits(float v) {
    const unsigned b = __float_as_uint(v);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__global__ __launch_bounds__(256) void colargmax_kernel(const float *__restrict__ x, int N, int C, int rows_per, unsigned long long *__restrict__ packed) {
    __shared__ unsigned long long part[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int n0 = blockIdx.z * rows_per, n1 = min(N, n0 + rows_per);
    unsigned long long best = 0ull;
    if (c < C)
        for (int n = n0 + g; n < n1; n += 4) {
            const unsigned long long key = ((unsigned long long)ordered_bits(x[((size_t)b * N + n) * C + c]) << 32) | (unsigned)(~n);
            best = key > best ? key : best;
        }
    part[g][threadIdx.x & 63] = best;
    __syncthreads();
    if (g == 0 && c < C) {
#pragma unroll
        for (int q = 1; q < 4; ++q) best = part[q][threadIdx.x] > best ? part[q][threadIdx.x] : best;
        atomicMax(packed + (size_t)b * C + c, best);
    }
}
// rows [mx[b] (Cg) | x[b][n] (Cx)] -> cat [R][Cg + Cx]; mx / arg are decoded from `packed` by the block that owns row 0 of a shape
__global__ __launch_bounds__(256) void cat_prefix_kernel(const unsigned long long *__restrict__ packed, const float *__restrict__ x, int N, int Cg, int Cx,
                                                         float *__restrict__ mx, int32_t *__restrict__ arg, float *__restrict__ cat) {
    const int W = Cg + Cx, w4 = W / 4;
    const int b = blockIdx.y;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)N * w4; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / w4), c = 4 * (int)(i % w4);
        f32x4 v;
        if (c < Cg) {
            float e[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned long long k = packed[(size_t)b * Cg + c + q];
                const unsigned ob = (unsigned)(k >> 32);
                e[q] = __uint_as_float(ob ^ ((ob >> 31) ? 0x80000000u : 0xffffffffu));
                if (n == 0) mx[(size_t)b * Cg + c + q] = e[q], arg[(size_t)b * Cg + c + q] = (int32_t)(~(unsigned)k);
            }
            v.x = e[0], v.y = e[1], v.z = e[2], v.w = e[3];
        } else {
            v = *(const f32x4 *)(x + ((size_t)b * N + n) * Cx + (c - Cg));
        }
        *(f32x4 *)(cat + ((size_t)b * N + n) * W + c) = v;
    }
}
// Backward of the two steps above for the Cg prefix columns of dcat [R][ld]: dmax[b][c] = sum_n dcat[b][n][c] (the broadcast),
// then dwide[b][n][c] = (n == arg[b][c]) ? dmax[b][c] : 0 (the max).  The column sums are formed in a fixed order: S row
// slices per shape (partial [S][B][Cg]; inside a slice rows strided by 4 per thread, the four partials combined in order),
// and the S slices are added in order by the one thread of max_bwd_kernel that owns the arg-max row.
__global__ __launch_bounds__(256) void prefix_colsum_kernel(const float *__restrict__ dcat, int N, int ld, int Cg, int rows_per, float *__restrict__ partial) {
    __shared__ float part[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int n0 = blockIdx.z * rows_per, n1 = min(N, n0 + rows_per);
    float s = 0.f;
    if (c < Cg)
        for (int n = n0 + g; n < n1; n += 4) s += dcat[((size_t)b * N + n) * ld + c];
    part[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < Cg)
        partial[((size_t)blockIdx.z * gridDim.y + b) * Cg + c] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void max_bwd_kernel(const float *__restrict__ partial, int S, const int32_t *__restrict__ arg, int N, int C,
                                                      float *__restrict__ dwide) {
    const int b = blockIdx.y, B = gridDim.y, c4n = C / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)N * c4n; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / c4n), c = 4 * (int)(i % c4n);
        const int32_t *ag = arg + (size_t)b * C + c;
        float e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            e[q] = 0.f;
            if (ag[q] == n)
                for (int s = 0; s < S; ++s) e[q] += partial[((size_t)s * B + b) * C + c + q];
        }
        f32x4 v;
        v.x = e[0], v.y = e[1], v.z = e[2], v.w = e[3];
        *(f32x4 *)(dwide + ((size_t)b * N + n) * C + c) = v;
    }
}
// out[c] += sum_r g[r][c]  (bias gradients) in a fixed order, two launches: column sums per row chunk (rows strided over the
// 256 / C thread groups, group partials combined in order), then ONE block adds the chunks in order.  (A one-launch form —
// the block that finishes last adds the chunks, found through a device-scope counter — was measured at 39 - 46 us per call:
// the device-scope release in front of the counter writes back the L2 on this eight-XCD part.)  C in {32, 64, 128}.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ g, long R, int C, long rows_per, float *__restrict__ partial) {
    __shared__ float part[8][128];
    const int groups = 256 / C, c = threadIdx.x % C, h = threadIdx.x / C;
    const long r0 = (long)blockIdx.x * rows_per, r1 = r0 + rows_per < R ? r0 + rows_per : R;
    float s = 0.f;
    for (long r = r0 + h; r < r1; r += groups) s += g[r * C + c];
    part[h][c] = s;
    __syncthreads();
    if (h == 0) {
        float t = part[0][c];
        for (int k = 1; k < groups; ++k) t += part[k][c];
        partial[(size_t)blockIdx.x * C + c] = t;
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float *__restrict__ partial, int chunks, int C, float *__restrict__ out) {
    __shared__ float part[8][128];
    const int groups = 256 / C, c = threadIdx.x % C, h = threadIdx.x / C;
    float t = 0.f;
    for (int k = h; k < chunks; k += groups) t += partial[(size_t)k * C + c];
    part[h][c] = t;
    __syncthreads();
    if (h == 0) {
        float tot = part[0][c];
        for (int k = 1; k < groups; ++k) tot += part[k][c];
        atomicAdd(out + c, tot);   // (atomic: see bn_pm_finalize_bwd_kernel — two calls' backward passes may run side by side)
    }
}
// The three C x C projection weights of the N2P blocks stacked as [q | k | v] (3C x C per block): packed copies for the forward /
// input-gradient GEMMs, and the reverse step for the weight gradient — grads[j] += stacked[j].
struct StackArgs {
    const float *src[21];
    float *dst[21];
    int n[21];        // floats per matrix (C*C)
    int count;
};
__global__ __launch_bounds__(256) void stack_copy_kernel(const StackArgs a, int accumulate) {
    const int m = blockIdx.y;
    if (m >= a.count) return;
    const f32x4 *s = (const f32x4 *)a.src[m];
    f32x4 *d = (f32x4 *)a.dst[m];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n[m] / 4; i += gridDim.x * blockDim.x) {
        if (accumulate) {
            const f32x4 v = s[i];
            float *o = (float *)(d + i);
            atomicAdd(o, v.x), atomicAdd(o + 1, v.y), atomicAdd(o + 2, v.z), atomicAdd(o + 3, v.w);
        } else {
            d[i] = s[i];
        }
    }
}

inline unsigned blocks_for(long n, int cap = 4096) { return (unsigned)((n + 255) / 256 < cap ? (n + 255) / 256 : cap); }

}  // namespace
}  // namespace dvm
