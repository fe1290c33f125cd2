// dvm_proj.hip — visual-feature injection around the (external) image backbone:
//   dvm_proj2img_f32 : point cloud -> 224x224 colour-mapped depth rendering   (reference models/model.py:584-650, 563-581)
//   dvm_i2p_f32      : image features -> per-point features, bicubic resize + gather (+ L2 normalise) fused
//                                                                            (reference models/model.py:653-678, 701-708)
// The ViT/upsampler between the two is a torch module (library GEMMs) and not part of this file.
#include "dvm_common.h"

namespace dvm {
namespace {

constexpr int IMG = 224, NOFF = 25;

__device__ const unsigned PIYG_LUT[256 * 3] = {
#include "piyg_lut.inc"
};

// params[b][8]: min_x, min_y, grid, off_x, off_y, e (the fixed-point exponent of the depth sums)

// ---- per-shape extent: min/max of x, y, max|z| -> pc_min, grid_size, centring offsets, fixed-point exponent
__global__ __launch_bounds__(256) void proj_range_kernel(const float *__restrict__ pts, int N, float *__restrict__ pc_min,
                                                         float *__restrict__ grid_size, float *__restrict__ offsets,
                                                         float *__restrict__ params) {
    __shared__ float red[5][256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *p = pts + (size_t)b * N * 3;
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY, az = 0.f;
    for (int i = tid; i < N; i += 256) {
        float x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
        mnx = fminf(mnx, x);
        mny = fminf(mny, y);
        mxx = fmaxf(mxx, x);
        mxy = fmaxf(mxy, y);
        az = fmaxf(az, fabsf(z));
    }
    red[0][tid] = mnx;
    red[1][tid] = mny;
    red[2][tid] = mxx;
    red[3][tid] = mxy;
    red[4][tid] = az;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            red[0][tid] = fminf(red[0][tid], red[0][tid + o]);
            red[1][tid] = fminf(red[1][tid], red[1][tid + o]);
            red[2][tid] = fmaxf(red[2][tid], red[2][tid + o]);
            red[3][tid] = fmaxf(red[3][tid], red[3][tid + o]);
            red[4][tid] = fmaxf(red[4][tid], red[4][tid + o]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        mnx = red[0][0], mny = red[1][0], mxx = red[2][0], mxy = red[3][0], az = red[4][0];
        const float rx = mxx - mnx, ry = mxy - mny;
        const float grid = __fdiv_rn(fmaxf(rx, ry), (float)(IMG - 3));
        // idx = floor((p - min) / grid) is monotone in p: its maximum is taken at the coordinate maximum, its minimum is 0.
        // dense = idx + {-2..2} + 1  =>  max = idx_max + 3, min = -1;  centre = floor((max + min) / 2)
        const float ix = floorf(__fdiv_rn(mxx - mnx, grid)), iy = floorf(__fdiv_rn(mxy - mny, grid));
        const float cx = floorf(((ix + 3.f) + (-1.f)) / 2.f), cy = floorf(((iy + 3.f) + (-1.f)) / 2.f);
        const float ox = (float)(IMG / 2) - (float)(int)cx - 1.f, oy = (float)(IMG / 2) - (float)(int)cy - 1.f;
        // depth sums are accumulated in fixed point (order-independent, hence reproducible): z * 2^e with
        // 25 * N * max|z| * 2^e < 2^62
        const double bound = (double)az * 25.0 * (double)N;
        int e = 40;
        if (bound > 0.0) {
            int lg = ilogb(bound) + 1;
            e = 61 - lg;
            if (e > 40) e = 40;
        }
        pc_min[2 * b] = mnx;
        pc_min[2 * b + 1] = mny;
        grid_size[b] = grid;
        offsets[2 * b] = ox;
        offsets[2 * b + 1] = oy;
        float *q = params + 8 * b;
        q[0] = mnx, q[1] = mny, q[2] = grid, q[3] = ox, q[4] = oy, q[5] = (float)e;
    }
}

// ---- splat: every point adds its depth to the 5x5 pixels around its cell
__global__ __launch_bounds__(256) void proj_splat_kernel(const float *__restrict__ pts, int N, const float *__restrict__ params,
                                                         unsigned long long *__restrict__ acc) {
    const int b = blockIdx.y;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)N * NOFF) return;
    const int i = (int)(g / NOFF), o = (int)(g % NOFF);
    const float *q = params + 8 * b;
    const float *p = pts + ((size_t)b * N + i) * 3;
    const float ix = floorf(__fdiv_rn(p[0] - q[0], q[2])), iy = floorf(__fdiv_rn(p[1] - q[1], q[2]));
    float px = (ix + (float)(o / 5 - 2)) + 1.f + q[3], py = (iy + (float)(o % 5 - 2)) + 1.f + q[4];
    const float sx = (px < 0.f ? 1.f : 0.f) - (px > (float)(IMG - 1) ? 1.f : 0.f);   // the reference's single-pixel shift
    const float sy = (py < 0.f ? 1.f : 0.f) - (py > (float)(IMG - 1) ? 1.f : 0.f);
    px += sx;
    py += sy;
    int r = (int)px, c = (int)py;
    r = r < 0 ? 0 : (r > IMG - 1 ? IMG - 1 : r);
    c = c < 0 ? 0 : (c > IMG - 1 ? IMG - 1 : c);
    const long long term = __double2ll_rn(ldexp((double)p[2], (int)q[5]));
    const size_t pix = (size_t)b * IMG * IMG + (size_t)r * IMG + c;
    atomicAdd(acc + pix, (unsigned long long)term);
}

__device__ __forceinline__ float depth_value(unsigned long long a, int e) {
    const float s = (float)ldexp((double)(long long)a, -e);       // the summed depth
    const float sg = 1.f / (1.f + expf(-s));                       // nn.Sigmoid
    return __fdiv_rn(sg - 0.485f, 0.229f);                         // channel 0 of (img - mean) / std
}

// ---- per image min / max of the normalised depth (over ALL pixels, empty ones included)
__global__ __launch_bounds__(256) void proj_minmax_kernel(const unsigned long long *__restrict__ acc, const float *__restrict__ params,
                                                          float *__restrict__ mm) {
    __shared__ float smn[256], smx[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int e = (int)params[8 * b + 5];
    float mn = INFINITY, mx = -INFINITY;
    for (int i = tid; i < IMG * IMG; i += 256) {
        float v = depth_value(acc[(size_t)b * IMG * IMG + i], e);
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    smn[tid] = mn;
    smx[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            smn[tid] = fminf(smn[tid], smn[tid + o]);
            smx[tid] = fmaxf(smx[tid], smx[tid + o]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        mm[2 * b] = smn[0];
        mm[2 * b + 1] = smx[0];
    }
}

// ---- colour: PiYG[(v - min) / (max - min)], pixels whose depth sum is exactly zero become -1
__global__ __launch_bounds__(256) void proj_colour_kernel(const unsigned long long *__restrict__ acc, const float *__restrict__ params,
                                                          const float *__restrict__ mm, float *__restrict__ img) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= IMG * IMG) return;
    const int e = (int)params[8 * b + 5];
    const unsigned long long a = acc[(size_t)b * IMG * IMG + i];
    const float s = (float)ldexp((double)(long long)a, -e);
    float rgb[3] = {-1.f, -1.f, -1.f};
    if (s != 0.f) {
        const float v = depth_value(a, e);
        const float d = __fdiv_rn(v - mm[2 * b], mm[2 * b + 1] - mm[2 * b]);
        if (d == d) {                                            // NaN (flat image) maps to matplotlib's "bad" colour: 0
            float x = d * 256.f;
            int k = x >= 256.f ? 255 : (x < 0.f ? 0 : (int)x);   // to_rgba: int(x * N), x == 1 -> N - 1, clip=True
            if (k > 255) k = 255;
            rgb[0] = __uint_as_float(PIYG_LUT[3 * k]);
            rgb[1] = __uint_as_float(PIYG_LUT[3 * k + 1]);
            rgb[2] = __uint_as_float(PIYG_LUT[3 * k + 2]);
        } else {
            rgb[0] = rgb[1] = rgb[2] = 0.f;
        }
    }
    float *o = img + (size_t)b * 3 * IMG * IMG + i;
    o[0] = rgb[0];
    o[IMG * IMG] = rgb[1];
    o[2 * IMG * IMG] = rgb[2];
}

// ---- I2P: bicubic (A = -0.75, align_corners = False) sample of f at each point's pixel, all channels, optional L2 norm
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
    const float A = -0.75f;
    float x = t + 1.f;
    c[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
    x = t;
    c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    x = 1.f - t;
    c[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    x = 2.f - t;
    c[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}

// one wave per point, lanes over channels
__global__ __launch_bounds__(256) void i2p_kernel(const float *__restrict__ pts, const float *__restrict__ f,
                                                  const float *__restrict__ pc_min, const float *__restrict__ grid_size,
                                                  const float *__restrict__ offsets, int N, int C, int H, int W, int normalize,
                                                  float *__restrict__ out, int ldo) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const float *p = pts + ((size_t)b * N + i) * 3;
    const float g = grid_size[b];
    float px = (floorf(__fdiv_rn(p[0] - pc_min[2 * b], g)) + 1.f) + offsets[2 * b];
    float py = (floorf(__fdiv_rn(p[1] - pc_min[2 * b + 1], g)) + 1.f) + offsets[2 * b + 1];
    int r = (int)px, c = (int)py;
    r = r < 0 ? 0 : (r > IMG - 1 ? IMG - 1 : r);
    c = c < 0 ? 0 : (c > IMG - 1 ? IMG - 1 : c);
    float cy[4], cx[4];
    int ys[4], xs[4];
    if (H == IMG && W == IMG) {                                  // F.interpolate to the same size is the identity
        cy[0] = cy[2] = cy[3] = 0.f, cy[1] = 1.f;
        cx[0] = cx[2] = cx[3] = 0.f, cx[1] = 1.f;
        for (int k = 0; k < 4; ++k) ys[k] = r, xs[k] = c;
    } else {
        const float sh = (float)H / (float)IMG, sw = (float)W / (float)IMG;
        const float fy = sh * ((float)r + 0.5f) - 0.5f, fx = sw * ((float)c + 0.5f) - 0.5f;
        const float y0 = floorf(fy), x0 = floorf(fx);
        cubic_coeffs(fy - y0, cy);
        cubic_coeffs(fx - x0, cx);
        for (int k = 0; k < 4; ++k) {
            int yy = (int)y0 - 1 + k, xx = (int)x0 - 1 + k;
            ys[k] = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
            xs[k] = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
        }
    }
    const float *fb = f + (size_t)b * C * H * W;
    float *o = out + ((size_t)b * N + i) * ldo;
    float ss = 0.f;
    // away from the left/right border the four taps of a row are adjacent: one 16-byte load per row (4-byte aligned
    // is all a global dwordx4 load needs) instead of four scattered dwords -- the whole wave takes the same branch
    const bool adjacent = xs[1] == xs[0] + 1 && xs[2] == xs[0] + 2 && xs[3] == xs[0] + 3;
    struct __attribute__((packed, aligned(4))) Taps { float v[4]; };
    for (int ch = lane; ch < C; ch += 64) {
        const float *pl = fb + (size_t)ch * H * W;
        float rows[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float *row = pl + (size_t)ys[k] * W;
            float t0, t1, t2, t3;
            if (adjacent) {
                const Taps t = *(const Taps *)(row + xs[0]);
                t0 = t.v[0], t1 = t.v[1], t2 = t.v[2], t3 = t.v[3];
            } else {
                t0 = row[xs[0]], t1 = row[xs[1]], t2 = row[xs[2]], t3 = row[xs[3]];
            }
            rows[k] = ((t0 * cx[0] + t1 * cx[1]) + t2 * cx[2]) + t3 * cx[3];
        }
        const float v = ((rows[0] * cy[0] + rows[1] * cy[1]) + rows[2] * cy[2]) + rows[3] * cy[3];
        ss += v * v;
        o[ch] = v;
    }
    if (normalize) {
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) ss += __shfl_xor(ss, s, 64);
        const float den = fmaxf(sqrtf(ss), 1e-12f);              // F.normalize: x / max(|x|, eps)
        for (int ch = lane; ch < C; ch += 64) o[ch] = __fdiv_rn(o[ch], den);
    }
}

}  // namespace
}  // namespace dvm

using namespace dvm;

DVM_EXPORT size_t dvm_proj2img_workspace_bytes(int B) {
    return align_up((size_t)B * IMG * IMG * sizeof(unsigned long long)) + align_up((size_t)B * 8 * sizeof(float)) +
           align_up((size_t)B * 2 * sizeof(float));
}

DVM_EXPORT int dvm_proj2img_f32(const float *pts, int B, int N, float *img, float *pc_min, float *grid_size, float *offsets,
                                void *ws, size_t ws_bytes, void *stream) {
    DVM_REQUIRE(pts && img && pc_min && grid_size && offsets, "dvm_proj2img_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1, "dvm_proj2img_f32: empty input (B=%d N=%d)", B, N);
    Arena ar(ws, ws_bytes);
    unsigned long long *acc = ar.take<unsigned long long>((size_t)B * IMG * IMG);
    float *params = ar.take<float>((size_t)B * 8);
    float *mm = ar.take<float>((size_t)B * 2);
    if (!ar.ok()) {
        set_error("dvm_proj2img_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(acc, 0, (size_t)B * IMG * IMG * sizeof(unsigned long long), s);
    hipLaunchKernelGGL(proj_range_kernel, dim3(B), dim3(256), 0, s, pts, N, pc_min, grid_size, offsets, params);
    hipLaunchKernelGGL(proj_splat_kernel, dim3((unsigned)(((long)N * NOFF + 255) / 256), B), dim3(256), 0, s, pts, N, params, acc);
    hipLaunchKernelGGL(proj_minmax_kernel, dim3(B), dim3(256), 0, s, acc, params, mm);
    hipLaunchKernelGGL(proj_colour_kernel, dim3((IMG * IMG + 255) / 256, B), dim3(256), 0, s, acc, params, mm, img);
    DVM_CHECK_LAUNCH("proj2img");
    return DVM_OK;
}

DVM_EXPORT int dvm_i2p_f32(const float *pts, const float *f, const float *pc_min, const float *grid_size, const float *offsets,
                           int B, int N, int C, int H, int W, int normalize, float *out, int ldo, void *stream) {
    DVM_REQUIRE(pts && f && pc_min && grid_size && offsets && out, "dvm_i2p_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1 && C >= 1 && H >= 1 && W >= 1 && ldo >= C, "dvm_i2p_f32: bad sizes (B=%d N=%d C=%d H=%d W=%d ldo=%d)",
                B, N, C, H, W, ldo);
    hipLaunchKernelGGL(i2p_kernel, dim3((N + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, pts, f, pc_min, grid_size, offsets, N, C,
                       H, W, normalize, out, ldo);
    DVM_CHECK_LAUNCH("i2p");
    return DVM_OK;
}


// ---------------------------------------------------------------- adaptive convolution of the JBU upsampler
// out[b,c,h,w] = sum_{i,j < d} in[b,c,h+i,w+j] * kern[b,h,w,i,j]   — the per-pixel d x d filtering step of FeatUp's joint
// bilateral upsampler (its `AdaptiveConv` CUDA op; here a gfx950 kernel).  `in` is the reflect-padded bicubic upsample
// [B,C,H+d-1,W+d-1], `kern` the combined spatial x range kernel [B,H,W,d,d].
// One workgroup = one row segment of 64 pixels x 4 channel lanes: the segment's d*d kernel weights (the same for every
// channel) are staged once in LDS as [tap][pixel]; every thread then walks its channels, reading input rows that its
// 63 neighbours read at adjacent addresses (coalesced, L1/L2-resident across the d vertical taps).
namespace dvm {
namespace {
constexpr int AC_PIX = 64, AC_CL = 4;
// kern_cm: the kernel tensor is [B,d*d,H,W] (tap-major, what dvm_jbu_kernel_f32 writes) instead of [B,H,W,d,d]
__global__ __launch_bounds__(AC_PIX * AC_CL) void adaptive_conv_kernel(const float *__restrict__ in, const float *__restrict__ kern, int C,
                                                                        int H, int W, int d, int kern_cm, float *__restrict__ out) {
    extern __shared__ float kw[];   // [d*d][AC_PIX]
    const int px = threadIdx.x & (AC_PIX - 1), cl = threadIdx.x / AC_PIX;
    const int w0 = blockIdx.x * AC_PIX, h = blockIdx.y, b = blockIdx.z;
    const int taps = d * d, Wp = W + d - 1, Hp = H + d - 1;
    if (kern_cm) {
        for (int e = threadIdx.x; e < taps * AC_PIX; e += AC_PIX * AC_CL) {
            const int t = e / AC_PIX, p = e % AC_PIX;   // consecutive threads read consecutive pixels of a tap (contiguous)
            kw[e] = (w0 + p < W) ? kern[(((size_t)b * taps + t) * H + h) * W + w0 + p] : 0.f;
        }
    } else {
        for (int e = threadIdx.x; e < taps * AC_PIX; e += AC_PIX * AC_CL) {
            const int p = e / taps, t = e % taps;   // consecutive threads read consecutive taps of a pixel (contiguous)
            kw[t * AC_PIX + p] = (w0 + p < W) ? kern[(((size_t)b * H + h) * W + w0 + p) * taps + t] : 0.f;
        }
    }
    __syncthreads();
    const int w = w0 + px;
    if (w >= W) return;
    for (int c = cl; c < C; c += AC_CL) {
        const float *ip = in + (((size_t)b * C + c) * Hp + h) * Wp + w;
        float acc = 0.f;
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) acc = fmaf(ip[(size_t)i * Wp + j], kw[(i * d + j) * AC_PIX + px], acc);
        out[(((size_t)b * C + c) * H + h) * W + w] = acc;
    }
}
}  // namespace
}  // namespace dvm

// Bicubic resize (A = -0.75, align_corners = False, border-clamped taps: torch's upsample_bicubic2d) written straight into
// its reflect-padded frame: out[b,c,h',w'] = up(in)[reflect(h' - pad), reflect(w' - pad)], h' < Ho + 2 pad.  One pass, one
// thread per output element (coalesced along w); the low-resolution source is small and cache-resident.  Replaces
// F.interpolate(mode='bicubic') + F.pad(mode='reflect') in front of the adaptive convolution (the generic ATen resize kernel
// took 16 ms for 24 x 384 x 256^2; this is bound by the 2.5 GB it writes).
namespace dvm {
namespace {
__device__ __forceinline__ void cubic_w(float t, float w[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}
// one output element, any scale: ATen's upsample_bicubic2d (align_corners = False, A = -0.75, source index clamped)
__device__ __forceinline__ float bicubic_at(const float *__restrict__ src, int Hi, int Wi, float sh, float sw, int h, int w) {
    const float fy = sh * (h + 0.5f) - 0.5f, fx = sw * (w + 0.5f) - 0.5f;
    const int iy = (int)floorf(fy), ix = (int)floorf(fx);
    float wy[4], wx[4];
    cubic_w(fy - iy, wy);
    cubic_w(fx - ix, wx);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), Hi - 1);
        float row = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) row = fmaf(wx[b], src[(size_t)yy * Wi + min(max(ix - 1 + b, 0), Wi - 1)], row);
        acc = fmaf(wy[a], row, acc);
    }
    return acc;
}
__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__global__ __launch_bounds__(256) void bicubic_pad_kernel(const float *__restrict__ in, int Hi, int Wi, int Ho, int Wo, int pad,
                                                          float *__restrict__ out) {
    const int Wp = Wo + 2 * pad, Hp = Ho + 2 * pad;
    const int wq = blockIdx.x * 256 + threadIdx.x, hq = blockIdx.y;
    const size_t bc = blockIdx.z;
    if (wq >= Wp) return;
    out[(bc * Hp + hq) * Wp + wq] = bicubic_at(in + bc * Hi * Wi, Hi, Wi, (float)Hi / Ho, (float)Wi / Wo, reflect_idx(hq - pad, Ho),
                                               reflect_idx(wq - pad, Wo));
}

// Exactly x2 with an odd pad (every JBU stage: 16 -> 32 -> ... -> 256, pad 3): the padded outputs (2a, 2a+1) of either
// axis are the unpadded (odd, even) pair 2a - pad, 2a + 1 - pad, which interpolate the SAME four source rows / columns
// with the two constant weight sets t = 0.25 and t = 0.75.  A thread owns a 2 x 2 block: 16 loads for 4 outputs instead
// of 64, weights formed once; same fma chains per output as bicubic_at, bit for bit.  Blocks that touch the reflected
// frame take the per-element path.
__global__ __launch_bounds__(256) void bicubic_pad_x2_kernel(const float *__restrict__ in, int Hi, int Wi, int pad,
                                                             float *__restrict__ out) {
    const int Ho = 2 * Hi, Wo = 2 * Wi, Wp = Wo + 2 * pad, Hp = Ho + 2 * pad;
    const int nx = (Wp + 1) >> 1, ny = (Hp + 1) >> 1;            // blocks per row / column of one plane
    const int t = blockIdx.x * 256 + threadIdx.x;               // packed: no idle lanes when nx is not a multiple of 64
    if (t >= nx * ny) return;
    const int ay = t / nx, ax = t - ay * nx;
    const size_t bc = blockIdx.y;
    const int wq = 2 * ax, hq = 2 * ay;
    const float *src = in + bc * Hi * Wi;
    float *o = out + (bc * Hp + hq) * Wp + wq;
    const int h0 = hq - pad, w0 = wq - pad;   // odd; h0 + 1, w0 + 1 even
    const bool interior = h0 >= 0 && h0 + 1 < Ho && w0 >= 0 && w0 + 1 < Wo && hq + 1 < Hp && wq + 1 < Wp;
    if (!interior) {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
                if (hq + dy < Hp && wq + dx < Wp)
                    o[(size_t)dy * Wp + dx] = bicubic_at(src, Hi, Wi, 0.5f, 0.5f, reflect_idx(h0 + dy, Ho), reflect_idx(w0 + dx, Wo));
        return;
    }
    // h0 = 2k + 1: fy = k + 0.25, rows k-1 .. k+2;  h0 + 1 = 2(k+1): fy = k + 0.75, the same rows
    const int ky = (h0 - 1) >> 1, kx = (w0 - 1) >> 1;
    float wlo[4], whi[4];
    cubic_w(0.25f, wlo);
    cubic_w(0.75f, whi);
    float v[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(ky - 1 + a, 0), Hi - 1);
#pragma unroll
        for (int b = 0; b < 4; ++b) v[a][b] = src[(size_t)yy * Wi + min(max(kx - 1 + b, 0), Wi - 1)];
    }
    float r[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float row0 = 0.f, row1 = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) row0 = fmaf(wlo[b], v[a][b], row0), row1 = fmaf(whi[b], v[a][b], row1);
        r[0][0] = fmaf(wlo[a], row0, r[0][0]);
        r[0][1] = fmaf(wlo[a], row1, r[0][1]);
        r[1][0] = fmaf(whi[a], row0, r[1][0]);
        r[1][1] = fmaf(whi[a], row1, r[1][1]);
    }
    *(float2 *)o = make_float2(r[0][0], r[0][1]);
    *(float2 *)(o + Wp) = make_float2(r[1][0], r[1][1]);
}
}  // namespace
}  // namespace dvm

DVM_EXPORT int dvm_bicubic_resize_pad_f32(const float *in, int BC, int Hi, int Wi, int Ho, int Wo, int pad, float *out, void *stream) {
    DVM_REQUIRE(in && out, "dvm_bicubic_resize_pad_f32: null pointer");
    DVM_REQUIRE(BC >= 1 && Hi >= 1 && Wi >= 1 && Ho >= 1 && Wo >= 1 && pad >= 0 && pad < Ho && pad < Wo, "dvm_bicubic_resize_pad_f32: bad sizes");
    DVM_REQUIRE(BC <= 65535 * 1 && Ho + 2 * pad <= 65535, "dvm_bicubic_resize_pad_f32: B*C=%d or height exceeds the grid limit", BC);
    if (Ho == 2 * Hi && Wo == 2 * Wi && (pad & 1) == 1)
        hipLaunchKernelGGL(dvm::bicubic_pad_x2_kernel, dim3((((Wo + 2 * pad + 1) / 2) * ((Ho + 2 * pad + 1) / 2) + 255) / 256, BC), dim3(256), 0,
                           (hipStream_t)stream, in, Hi, Wi, pad, out);
    else
        hipLaunchKernelGGL(dvm::bicubic_pad_kernel, dim3((Wo + 2 * pad + 255) / 256, Ho + 2 * pad, BC), dim3(256), 0, (hipStream_t)stream, in, Hi,
                           Wi, Ho, Wo, pad, out);
    DVM_CHECK_LAUNCH("bicubic_resize_pad");
    return DVM_OK;
}

// The combined kernel of one JBU stage before its learned correction: for every pixel p and tap t (offset (i,j) in the
// d x d window, reflect padding), softmax_t(temp * <proj[p], proj[p + t]>) * exp(-|t|^2 / (2 sigma^2)), renormalised over
// the taps.  A workgroup owns a 32 x 8 pixel tile: the key vectors of the tile and its 3-pixel halo (32 x 14 x 38 floats,
// 68 KB) are staged in LDS once — straight from L2 the 1568 loads per pixel were the whole cost (9.9 ms per 24 x 256^2
// images) — then every thread holds its own key vector and the d*d logits in registers.  Written tap-major so that the
// 1x1 "fixup" convolutions and the adaptive convolution read it coalesced.  Replaces an unfold of key_dim*d*d floats per
// pixel (400 MB per 256^2 image).
namespace dvm {
namespace {
constexpr int JT_W = 32, JT_H = 8;
template <int KD, int D>
__global__ __launch_bounds__(256) void jbu_kernel_kernel(const float *__restrict__ proj, const float *__restrict__ temp_p,
                                                         const float *__restrict__ sigma_p, int H, int W, float *__restrict__ out) {
    constexpr int R = D / 2, TAPS = D * D, LW = JT_W + 2 * R, LH = JT_H + 2 * R;
    extern __shared__ float tile[];   // [KD][LH][LW]
    const int tx = threadIdx.x & (JT_W - 1), ty = threadIdx.x / JT_W;
    const int w0 = blockIdx.x * JT_W, h0 = blockIdx.y * JT_H, b = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const float *pb = proj + (size_t)b * KD * plane;
    for (int e = threadIdx.x; e < KD * LH * LW; e += 256) {
        const int c = e / (LH * LW), r = e % (LH * LW), y = r / LW, x = r % LW;
        int hh = h0 + y - R, ww = w0 + x - R;
        hh = hh < 0 ? -hh : (hh >= H ? 2 * H - 2 - hh : hh);       // reflect (no edge repeat), like F.pad(mode='reflect')
        ww = ww < 0 ? -ww : (ww >= W ? 2 * W - 2 - ww : ww);
        hh = min(max(hh, 0), H - 1);                               // (rows / columns of a partial tile beyond the reflection)
        ww = min(max(ww, 0), W - 1);
        tile[e] = pb[c * plane + (size_t)hh * W + ww];
    }
    __syncthreads();
    const int w = w0 + tx, h = h0 + ty;
    if (w >= W || h >= H) return;
    float q[KD];
#pragma unroll
    for (int c = 0; c < KD; ++c) q[c] = tile[(c * LH + ty + R) * LW + tx + R];
    float temp = expf(*temp_p);
    temp = fminf(fmaxf(temp, 1e-4f), 1e4f);
    const float sig = *sigma_p, inv2s2 = 1.f / (2.f * sig * sig);
    float logit[TAPS];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const float *nb = tile + (ty + i) * LW + tx + j;
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < KD; ++c) dot = fmaf(q[c], nb[c * LH * LW], dot);
            const float l = temp * dot;
            logit[i * D + j] = l;
            mx = fmaxf(mx, l);
        }
    float se = 0.f;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        logit[t] = expf(logit[t] - mx);
        se += logit[t];
    }
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            // spatial kernel on linspace(-1, 1, D) coordinates
            const float y = -1.f + 2.f * i / (D - 1), x = -1.f + 2.f * j / (D - 1);
            const float v = (logit[i * D + j] / se) * expf(-(x * x + y * y) * inv2s2);
            logit[i * D + j] = v;
            tot += v;
        }
    const float inv = 1.f / fmaxf(tot, 1e-7f);
    float *ob = out + (size_t)b * TAPS * plane + (size_t)h * W + w;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) ob[t * plane] = logit[t] * inv;
}

// Adaptive convolution, d = 7: 32 x 8 pixel tiles; a thread keeps its pixel's 49 weights in registers for all channels and
// reads the inputs of 8 channels at a time from an LDS tile (the first version read weights from LDS and inputs from
// L1 / L2: 12.5 ms per 24 x 384 x 256^2; the inputs' 49-fold reuse now stays inside the CU).
// A thread owns TWO vertically adjacent pixels (tile 32 x 16): the 8 input rows they cover are read once and feed both
// weight sets — 56 LDS reads per 98 fmas instead of 49 per 49 (the kernel is bound by its LDS reads).
constexpr int AT_CC = 4, AT_TH = 16;
__global__ __launch_bounds__(256) void adaptive_conv7_kernel(const float *__restrict__ in, const float *__restrict__ kern, int C, int H, int W,
                                                             int kern_cm, int csplit, float *__restrict__ out) {
    constexpr int D = 7, R = 3, LW = JT_W + 2 * R, LH = AT_TH + 2 * R, PER = AT_CC * LH * LW, SLOTS = (PER + 255) / 256;
    __shared__ float tile[PER];
    const int tx = threadIdx.x & (JT_W - 1), ty = threadIdx.x / JT_W;
    const int w0 = blockIdx.x * JT_W, h0 = blockIdx.y * AT_TH;
    const int b = blockIdx.z / csplit, cs = blockIdx.z % csplit;
    const int cper = (C + csplit - 1) / csplit, c_beg = cs * cper, c_end = min(C, c_beg + cper);
    const int w = w0 + tx, ha = h0 + 2 * ty, Wp = W + 2 * R, Hp = H + 2 * R;
    const bool live0 = w < W && ha < H, live1 = w < W && ha + 1 < H;
    float k0[D * D], k1[D * D];
#pragma unroll
    for (int t = 0; t < D * D; ++t) {
        const int hq = live0 ? ha : 0, hr = live1 ? ha + 1 : 0, wq = live0 ? w : 0;
        k0[t] = kern_cm ? kern[(((size_t)b * D * D + t) * H + hq) * W + wq] : kern[(((size_t)b * H + hq) * W + wq) * D * D + t];
        k1[t] = kern_cm ? kern[(((size_t)b * D * D + t) * H + hr) * W + wq] : kern[(((size_t)b * H + hr) * W + wq) * D * D + t];
    }
    // the tile elements this thread stages are the same for every channel chunk: their offsets (relative to the chunk's
    // first channel plane) are formed once, and the next chunk is fetched into registers while the current one is consumed
    const size_t plane = (size_t)Hp * Wp;
    int off[SLOTS];
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
        const int e = threadIdx.x + 256 * q;
        const int cc = e / (LH * LW), r = e % (LH * LW), y = r / LW, x = r % LW;
        const int hh = h0 + y, ww = w0 + x;   // padded coordinates
        off[q] = (e < PER && hh < Hp && ww < Wp) ? (int)(cc * plane + (size_t)hh * Wp + ww) : -1;
    }
    const float *bin = in + (size_t)b * C * plane;
    float nxt[SLOTS];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int q = 0; q < SLOTS; ++q)
            nxt[q] = (off[q] >= 0 && c0 + (threadIdx.x + 256 * q) / (LH * LW) < c_end) ? bin[(size_t)c0 * plane + off[q]] : 0.f;
    };
    fetch(c_beg);
    for (int c0 = c_beg; c0 < c_end; c0 += AT_CC) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SLOTS; ++q)
            if (threadIdx.x + 256 * q < PER) tile[threadIdx.x + 256 * q] = nxt[q];
        __syncthreads();
        if (c0 + AT_CC < c_end) fetch(c0 + AT_CC);
#pragma unroll
        for (int cc = 0; cc < AT_CC; ++cc) {
            if (c0 + cc >= c_end) break;
            const float *tp = tile + (cc * LH + 2 * ty) * LW + tx;
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int i = 0; i <= D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const float v = tp[i * LW + j];
                    if (i < D) a0 = fmaf(v, k0[i * D + j], a0);
                    if (i > 0) a1 = fmaf(v, k1[(i - 1) * D + j], a1);
                }
            float *o = out + (((size_t)b * C + c0 + cc) * H + ha) * W + w;
            if (live0) o[0] = a0;
            if (live1) o[W] = a1;
        }
    }
}
}  // namespace
}  // namespace dvm

DVM_EXPORT int dvm_jbu_kernel_f32(const float *proj, const float *range_temp, const float *sigma_spatial, int B, int key_dim, int H, int W, int d,
                                  float *out, void *stream) {
    DVM_REQUIRE(proj && range_temp && sigma_spatial && out, "dvm_jbu_kernel_f32: null pointer");
    DVM_REQUIRE(B >= 1 && H >= 4 && W >= 4 && B <= 65535 && H <= 65535, "dvm_jbu_kernel_f32: bad sizes (B=%d H=%d W=%d)", B, H, W);
    DVM_REQUIRE(key_dim == 32 && d == 7, "dvm_jbu_kernel_f32: only FeatUp's key_dim = 32, radius = 3 (got key_dim=%d d=%d)", key_dim, d);
    const size_t lds = (size_t)32 * (dvm::JT_H + 6) * (dvm::JT_W + 6) * sizeof(float);
    dvm::ensure_dyn_lds((const void *)dvm::jbu_kernel_kernel<32, 7>, (int)lds);
    hipLaunchKernelGGL((dvm::jbu_kernel_kernel<32, 7>), dim3((W + dvm::JT_W - 1) / dvm::JT_W, (H + dvm::JT_H - 1) / dvm::JT_H, B), dim3(256), lds,
                       (hipStream_t)stream, proj, range_temp, sigma_spatial, H, W, out);
    DVM_CHECK_LAUNCH("jbu_kernel");
    return DVM_OK;
}

DVM_EXPORT int dvm_adaptive_conv_f32(const float *in, const float *kern, int B, int C, int H, int W, int d, int kern_tap_major, float *out,
                                     void *stream) {
    DVM_REQUIRE(in && kern && out, "dvm_adaptive_conv_f32: null pointer");
    DVM_REQUIRE(B >= 1 && C >= 1 && H >= 1 && W >= 1 && d >= 1 && d <= 15 && (d & 1), "dvm_adaptive_conv_f32: bad sizes (B=%d C=%d H=%d W=%d d=%d)", B,
                C, H, W, d);
    DVM_REQUIRE(B <= 65535 && H <= 65535, "dvm_adaptive_conv_f32: B or H exceeds the grid limit");
    if (d == 7) {
        const int tiles = ((W + dvm::JT_W - 1) / dvm::JT_W) * ((H + dvm::AT_TH - 1) / dvm::AT_TH) * B;
        int csplit = 1;   // small maps: split the channels over more workgroups until the chip is covered
        while (tiles * csplit < 1024 && csplit * 2 * dvm::AT_CC <= C) csplit *= 2;
        DVM_REQUIRE((long)B * csplit <= 65535, "dvm_adaptive_conv_f32: B exceeds the grid limit");
        DVM_REQUIRE((long)dvm::AT_CC * (H + 6) * (W + 6) < (1L << 31), "dvm_adaptive_conv_f32: image too large (H=%d W=%d)", H, W);
        hipLaunchKernelGGL(dvm::adaptive_conv7_kernel, dim3((W + dvm::JT_W - 1) / dvm::JT_W, (H + dvm::AT_TH - 1) / dvm::AT_TH, B * csplit),
                           dim3(256), 0, (hipStream_t)stream, in, kern, C, H, W, kern_tap_major, csplit, out);
        DVM_CHECK_LAUNCH("adaptive_conv7");
        return DVM_OK;
    }
    const size_t lds = (size_t)d * d * dvm::AC_PIX * sizeof(float);
    hipLaunchKernelGGL(dvm::adaptive_conv_kernel, dim3((W + dvm::AC_PIX - 1) / dvm::AC_PIX, H, B), dim3(dvm::AC_PIX * dvm::AC_CL), lds,
                       (hipStream_t)stream, in, kern, C, H, W, d, kern_tap_major, out);
    DVM_CHECK_LAUNCH("adaptive_conv");
    return DVM_OK;
}
