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
