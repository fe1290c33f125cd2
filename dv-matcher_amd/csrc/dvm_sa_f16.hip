// dvm_sa_f16.hip — SA_Layer attention core (models/model.py:113-121) with both contractions on the 16-bit matrix cores.
// The fp32-MFMA kernels (dvm_backbone.hip) spend 40 matrix instructions of 64 cycles per 32x32 tile; here every
// operand is split into two fp16 planes (x = h + m, |x - h - m| <= 2^-22 |x|) and each contraction runs as three
// products hh + hm + mh of the 32x32x16 fp16 instruction, accumulated in fp32:
//   E tile      : p has 16 channels = exactly one k-step            -> 3 instructions (was 8 of the fp32 kind)
//   x_r += V^T w: 32 keys = two k-steps, 64 channels = two tiles    -> 12 instructions (was 32)
// at 32 instead of 64 cycles each.  Operands are pre-split once per call (sa_split_kernel) into layouts a lane can
// load as its MFMA fragment with ONE 16-byte load straight from L2 — no LDS staging, no barriers in the key loop:
//   pp [B][N][plane][16]                    : row fragment = 8 channels 8*hh..8*hh+7
//   vp [B][N/32][plane][64 channels][32 pos]: the softmax weights come out of the first MFMA in its accumulator layout
//                                             (register r of lane-half h = key (r&3) + 8 (r>>2) + 4h); they are fed
//                                             back as the B operand with k-slot (step s, half hh, j) = register 8s + j,
//                                             and V^T is stored with its keys permuted to the same slots
//                                             (pos = 16 s + 8 hh + j), so nothing moves between the two contractions.
#include "dvm_common.h"

namespace dvm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int SF_P = 16, SF_C = 64;

__device__ __forceinline__ void split2(float x, _Float16 &h, _Float16 &m) {
    h = (_Float16)x;
    m = (_Float16)(x - (float)h);
}

// thread per (b, point): p row -> pp, v row -> its slot of the permuted V^T tile
__global__ __launch_bounds__(256) void sa_split_kernel(const float *__restrict__ p, const float *__restrict__ v, int B, int N, int T,
                                                       _Float16 *__restrict__ pp, _Float16 *__restrict__ vp) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long rows = (long)B * T * 32;  // padded to whole 32-key tiles
    if (g >= rows) return;
    const int b = (int)(g / ((long)T * 32)), i = (int)(g % ((long)T * 32));
    const bool live = i < N;
    const int kappa = i & 31, tile = i >> 5;
    const int hh = (kappa >> 2) & 1, r = (kappa & 3) + 4 * (kappa >> 3);
    const int pos = 16 * (r >> 3) + 8 * hh + (r & 7);
    _Float16 *vt = vp + (((size_t)b * T + tile) * 2) * SF_C * 32;
    const float *vr = v + ((size_t)b * N + (live ? i : 0)) * SF_C;
    for (int c = 0; c < SF_C; ++c) {
        _Float16 h, m;
        split2(live ? vr[c] : 0.f, h, m);
        vt[(size_t)c * 32 + pos] = h;
        vt[(size_t)SF_C * 32 + (size_t)c * 32 + pos] = m;
    }
    if (live) {
        const float *pr = p + ((size_t)b * N + i) * SF_P;
        _Float16 *o = pp + ((size_t)b * N + i) * 2 * SF_P;
#pragma unroll
        for (int c = 0; c < SF_P; ++c) {
            _Float16 h, m;
            split2(pr[c], h, m);
            o[c] = h;
            o[SF_P + c] = m;
        }
    }
}

__device__ __forceinline__ f32x16 energy_tile(const f16x8 ah, const f16x8 am, const f16x8 bh, const f16x8 bm) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bh, acc, 0, 0, 0);  // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    return acc;
}

// this lane's fragment of point `i` (clamped): 8 channels of either plane
__device__ __forceinline__ void load_p(const _Float16 *__restrict__ pp, size_t row, int hh, f16x8 &h, f16x8 &m) {
    const _Float16 *q = pp + row * 2 * SF_P + 8 * hh;
    h = *(const f16x8 *)q;
    m = *(const f16x8 *)(q + SF_P);
}

// pass 1: (m_i, l_i) of softmax_j(E_ij).  One workgroup = 32 rows i; its 4 waves take every 4th 32-key tile.
__global__ __launch_bounds__(256) void sa_rowstats_f16_kernel(const _Float16 *__restrict__ pp, int N, int kchunk,
                                                              float *__restrict__ stats) {
    __shared__ float red[4][32][2];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r32 = lane & 31, h = lane >> 5;
    const int irow = blockIdx.x * 32 + r32;
    const size_t base = (size_t)b * N;
    f16x8 bh, bm;
    load_p(pp, base + (irow < N ? irow : N - 1), h, bh, bm);
    const int zbeg = blockIdx.z * kchunk, zend = zbeg + kchunk < N ? zbeg + kchunk : N;
    const bool partial = gridDim.z > 1;
    float m = -INFINITY, l = 0.f;
    for (int j0 = zbeg + wave * 32; j0 < zend; j0 += 128) {
        const int jr = j0 + r32;
        f16x8 ah, am;
        load_p(pp, base + (jr < N ? jr : N - 1), h, ah, am);
        f32x16 acc = energy_tile(ah, am, bh, bm);
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float e = j < N ? acc[r] : -INFINITY;
            acc[r] = e;
            tmax = fmaxf(tmax, e);
        }
        if (tmax > m) {
            l = l * __expf(m - tmax);
            m = tmax;
        }
        if (tmax != -INFINITY) {
#pragma unroll
            for (int r = 0; r < 16; ++r) l += __expf(acc[r] - m);
        }
    }
    const float mo = __shfl_xor(m, 32, 64), lo = __shfl_xor(l, 32, 64);
    const float mm = fmaxf(m, mo);
    const float ll = (m == -INFINITY ? 0.f : l * __expf(m - mm)) + (mo == -INFINITY ? 0.f : lo * __expf(mo - mm));
    if (h == 0) {
        red[wave][r32][0] = mm;
        red[wave][r32][1] = ll;
    }
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid < 32 && blockIdx.x * 32 + tid < N) {
        const float gm = fmaxf(fmaxf(red[0][tid][0], red[1][tid][0]), fmaxf(red[2][tid][0], red[3][tid][0]));
        float gl = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float wm = red[w][tid][0];
            if (wm != -INFINITY) gl += red[w][tid][1] * __expf(wm - gm);
        }
        const size_t o = (((size_t)blockIdx.z * gridDim.y + b) * N + blockIdx.x * 32 + tid) * 2;
        stats[o] = gm;
        stats[o + 1] = partial ? gl : 1.0f / gl;
    }
}

// pass 2: x_r[j,:] = sum_i v_i w_ij / (1e-9 + sum_i w_ij),  w_ij = exp(E_ij - m_i) / l_i.  Same work split as pass 1.
__global__ __launch_bounds__(256) void sa_apply_f16_kernel(const _Float16 *__restrict__ pp, const _Float16 *__restrict__ vp,
                                                           const float *__restrict__ stats, int N, int T, int kchunk,
                                                           float *__restrict__ xr, float *__restrict__ cinv_out) {
    __shared__ float red[4 * 32 * 64];
    __shared__ float csum[4][32];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r32 = lane & 31, h = lane >> 5;
    const int jcol = blockIdx.x * 32 + r32;
    const size_t base = (size_t)b * N;
    f16x8 bh, bm;
    load_p(pp, base + (jcol < N ? jcol : N - 1), h, bh, bm);
    const int zbeg = blockIdx.z * kchunk, zend = zbeg + kchunk < N ? zbeg + kchunk : N;
    const bool partial = gridDim.z > 1;
    f32x16 o0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, o1 = o0;
    float colsum = 0.f;
    int e_col = 100;  // the column's current weight scale 2^e_col (any first weight lowers it)
    for (int i0 = zbeg + wave * 32; i0 < zend; i0 += 128) {
        const int ir = i0 + r32;
        f16x8 ah, am;
        load_p(pp, base + (ir < N ? ir : N - 1), h, ah, am);
        const f32x16 acc = energy_tile(ah, am, bh, bm);
        // softmax weights of this lane's 16 keys (4 runs of 4 consecutive keys)
        float w[16];
        float cmax = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int k0 = i0 + 8 * g + 4 * h;  // keys k0 .. k0+3 <-> registers 4g .. 4g+3
            f32x4 s01 = {0.f, 0.f, 0.f, 0.f}, s23 = s01;  // (m, 1/l) pairs; padding keys: 1/l = 0
            const float *sp = stats + (base + k0) * 2;
            if (k0 + 3 < N) {
                s01 = *(const f32x4 *)sp;
                s23 = *(const f32x4 *)(sp + 4);
            } else {
                if (k0 < N) s01.x = sp[0], s01.y = sp[1];
                if (k0 + 1 < N) s01.z = sp[2], s01.w = sp[3];
                if (k0 + 2 < N) s23.x = sp[4], s23.y = sp[5];
            }
            const float mk[4] = {s01.x, s01.z, s23.x, s23.z}, il[4] = {s01.y, s01.w, s23.y, s23.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = 4 * g + u;
                w[r] = il[u] != 0.f ? __expf(acc[r] - mk[u]) * il[u] : 0.f;
                colsum += w[r];
                cmax = fmaxf(cmax, w[r]);
            }
        }
        // fp16 spans only 2^-14 .. 2^15 at full precision, the weights of a column many decades: the column (= this
        // lane and its partner in the other half) carries a power-of-two scale 2^e that keeps its largest weight so far
        // below 2^15 — weights more than 2^-24 below the column's largest are irrelevant to its sum — and the
        // accumulators are rescaled (exactly) whenever a larger weight lowers e.
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
        if (cmax > 0.f) {
            const int need = 14 - ((int)((__float_as_uint(cmax) >> 23) & 0xffu) - 127);
            if (need < e_col) {
                const float f = __uint_as_float((unsigned)(need - e_col + 127) << 23);  // 2^(need - e_col), exact
#pragma unroll
                for (int r = 0; r < 16; ++r) o0[r] *= f, o1[r] *= f;
                e_col = need;
            }
        }
        const float sc = __uint_as_float((unsigned)(e_col + 127) << 23);
        f16x8 wh[2], wm[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            _Float16 hv, mv;
            split2(w[r] * sc, hv, mv);
            wh[r >> 3][r & 7] = hv;
            wm[r >> 3][r & 7] = mv;
        }
        // x_r += V^T w : A = permuted V^T fragments straight from L2 (this lane: channel ct*32 + r32, slots 16 s + 8 h ..+7)
        const _Float16 *vt = vp + (((size_t)b * T + (i0 >> 5)) * 2) * SF_C * 32 + (size_t)r32 * 32 + 8 * h;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f16x8 v0h = *(const f16x8 *)(vt + 16 * s), v0m = *(const f16x8 *)(vt + SF_C * 32 + 16 * s);
            const f16x8 v1h = *(const f16x8 *)(vt + 32 * 32 + 16 * s), v1m = *(const f16x8 *)(vt + SF_C * 32 + 32 * 32 + 16 * s);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0m, wh[s], o0, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h, wm[s], o0, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h, wh[s], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1m, wh[s], o1, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h, wm[s], o1, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h, wh[s], o1, 0, 0, 0);
        }
    }
    colsum += __shfl_xor(colsum, 32, 64);
    if (h == 0) csum[wave][r32] = colsum;
    float *rw = red + wave * (32 * 64);
    const float unscale = __uint_as_float((unsigned)(127 - e_col) << 23);  // 2^-e_col (e_col in [-112, 100])
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        rw[r * 64 + lane] = o0[r] * unscale;
        rw[(16 + r) * 64 + lane] = o1[r] * unscale;
    }
    __syncthreads();
    const float cs = (csum[0][r32] + csum[1][r32]) + (csum[2][r32] + csum[3][r32]);
    const float inv = partial ? 1.0f : 1.0f / (1e-9f + cs);
    const size_t orow = ((size_t)blockIdx.z * gridDim.y + b) * N + jcol;
    if (cinv_out && wave == 0 && h == 0 && jcol < N) cinv_out[orow] = partial ? cs : inv;
    if (jcol < N) {
        float *o = xr + orow * SF_C;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = wave * 8 + q;  // merged register index: 0..15 -> o0, 16..31 -> o1
            const float *src = red + r * 64 + lane;
            const float sum = (src[0] + src[32 * 64]) + (src[2 * 32 * 64] + src[3 * 32 * 64]);
            const int rr = r & 15;
            o[(r >> 4) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h] = sum * inv;
        }
    }
}

}  // namespace

size_t sa_f16_ws_bytes(int B, int N) {
    const size_t T = (N + 31) / 32;
    return align_up((size_t)B * N * 2 * SF_P * sizeof(_Float16)) + align_up((size_t)B * T * 2 * SF_C * 32 * sizeof(_Float16));
}

void sa_f16_carve(void *ws, int B, int N, _Float16 *&pp, _Float16 *&vp) {
    pp = (_Float16 *)ws;
    vp = (_Float16 *)((char *)ws + align_up((size_t)B * N * 2 * SF_P * sizeof(_Float16)));
}

void launch_sa_split_f16(const float *p, const float *v, int B, int N, _Float16 *pp, _Float16 *vp, hipStream_t s) {
    const int T = (N + 31) / 32;
    const long rows = (long)B * T * 32;
    hipLaunchKernelGGL(sa_split_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, p, v, B, N, T, pp, vp);
}

void launch_sa_rowstats_f16(const _Float16 *pp, int B, int N, int kchunk, int Z, float *stats, hipStream_t s) {
    hipLaunchKernelGGL(sa_rowstats_f16_kernel, dim3((N + 31) / 32, B, Z), dim3(256), 0, s, pp, N, kchunk, stats);
}

void launch_sa_apply_f16(const _Float16 *pp, const _Float16 *vp, const float *stats, int B, int N, int kchunk, int Z, float *xr,
                         float *cinv, hipStream_t s) {
    hipLaunchKernelGGL(sa_apply_f16_kernel, dim3((N + 31) / 32, B, Z), dim3(256), 0, s, pp, vp, stats, N, (N + 31) / 32, kchunk, xr,
                       cinv);
}

}  // namespace dvm
