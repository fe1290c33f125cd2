// dvm_mlp_bf16.hip — the Deformer's decoder MLP (262 -> 512 -> 256 -> 128 -> 9, ELU) on the bf16
// matrix cores with fp32-level accuracy.
//
// gfx950's fp32 MFMA runs at the vector rate on the vector ALUs; the bf16 MFMA is 16x
// faster and is a separate pipe.  Every fp32 operand x is split exactly into three bf16 planes
//     x = h + m + l (+ r, |r| <= 2^-24 |x|),   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)
// and a product is evaluated as the six partial products  hh + hm + mh + hl + lh + mm  with fp32
// accumulation (the dropped terms are <= 2^-24 relative).  That is 6/16 of the fp32-MFMA time for the
// same accuracy class as an fp32 chain (reference models/model.py:433-452, 476-477; floats only, no
// integer output depends on it).  Activations live pre-split in LDS (three bf16 planes per node),
// weights are pre-split and packed in B-fragment order, one 16-B load per lane, plane and k-step.
#include "dvm_common.h"

namespace dvm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int MB_NODES = 32, MB_WAVES = 8, MB_THREADS = 64 * MB_WAVES;
constexpr int MB_ZS = 264;  // z row stride in HBM (floats)
// K (padded to a multiple of 16) of each layer and the LDS row strides in bytes (three planes of K bf16 + pad
// so that 32 rows hit distinct 16-B slots: stride/4 = 4*odd mod 64)
constexpr int MB_K0 = 272, MB_K1 = 512, MB_K2 = 256, MB_K3 = 128;
constexpr int MB_SA = 3 * MB_K0 * 2 + 16;  // 1648 B: holds z (K0) and h1 (K2)
constexpr int MB_SB = 3 * MB_K1 * 2 + 16;  // 3088 B: holds h0 (K1) and h2 (K3)
constexpr size_t MB_LDS_BYTES = (size_t)MB_NODES * (MB_SA + MB_SB);

__device__ __forceinline__ void split3(float x, __bf16 &h, __bf16 &m, __bf16 &l) {
    h = (__bf16)x;
    float r = x - (float)h;
    m = (__bf16)r;
    float r2 = r - (float)m;
    l = (__bf16)r2;
}

// Packed weights: Wp[otile][step][plane][lane][8] bf16 with element j of lane (o = lane&31, hh = lane>>5)
// = plane(W[otile*32 + o][16*step + 8*hh + j])  (0 outside the matrix).
// the four layers in ONE launch (blockIdx.y = layer), gated like the kernel they serve: as the range-safe fallback of the fp16x2
// MLP (gate != nullptr) nothing is packed unless the flag is set
struct PackLayersBf16 {
    const float *W[4];
    int O[4], I[4], otiles[4], steps[4];
    __bf16 *Wp[4];
};
__device__ __forceinline__ void pack_weights_bf16_body(const float *__restrict__ W, int O, int I, int otiles, int steps, __bf16 *__restrict__ Wp);
__global__ void pack_layers_bf16_kernel(const PackLayersBf16 a, const int *__restrict__ gate) {
    if (gate && *gate == 0) return;
    const int q = blockIdx.y;
    pack_weights_bf16_body(a.W[q], a.O[q], a.I[q], a.otiles[q], a.steps[q], a.Wp[q]);
}
__device__ __forceinline__ void pack_weights_bf16_body(const float *__restrict__ W, int O, int I, int otiles, int steps, __bf16 *__restrict__ Wp) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)otiles * steps * 64 * 8;
    if (g >= total) return;
    int j = (int)(g & 7);
    int lane = (int)((g >> 3) & 63);
    int step = (int)((g >> 9) % steps);
    int ot = (int)(g / (512L * steps));
    int o = ot * 32 + (lane & 31), c = 16 * step + 8 * (lane >> 5) + j;
    float w = (o < O && c < I) ? W[(size_t)o * I + c] : 0.f;
    __bf16 h, m, l;
    split3(w, h, m, l);
    size_t base = (((size_t)ot * steps + step) * 3) * 512 + (size_t)lane * 8 + j;
    Wp[base] = h;
    Wp[base + 512] = m;
    Wp[base + 1024] = l;
}

// one 32x32 output tile: acc[node][out] = sum_k act[node][k] W[out][k]  over `steps` k-steps of 16
__device__ __forceinline__ f32x16 mlp_tile_bf16(const char *__restrict__ act_row /* node row + 16*hh */, int plane_bytes,
                                                const __bf16 *__restrict__ wp /* tile base + lane*8 */, int steps) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // weight fragments come straight from L2 (every workgroup streams all 1.8 MB): keep two k-steps in flight
    bf16x8 bh = *(const bf16x8 *)(wp), bm = *(const bf16x8 *)(wp + 512), bl = *(const bf16x8 *)(wp + 1024);
    const __bf16 *w1 = wp + (size_t)(1 < steps ? 1 : 0) * 1536;
    bf16x8 ch = *(const bf16x8 *)(w1), cm = *(const bf16x8 *)(w1 + 512), cl = *(const bf16x8 *)(w1 + 1024);
    for (int s = 0; s < steps; ++s) {
        const bf16x8 ah = *(const bf16x8 *)(act_row + 32 * s);
        const bf16x8 am = *(const bf16x8 *)(act_row + plane_bytes + 32 * s);
        const bf16x8 al = *(const bf16x8 *)(act_row + 2 * plane_bytes + 32 * s);
        const __bf16 *wn = wp + (size_t)(s + 2 < steps ? s + 2 : steps - 1) * 1536;
        const bf16x8 nh = *(const bf16x8 *)(wn), nm = *(const bf16x8 *)(wn + 512), nl = *(const bf16x8 *)(wn + 1024);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);  // small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        bh = ch, bm = cm, bl = cl;
        ch = nh, cm = nm, cl = nl;
    }
    return acc;
}

__device__ __forceinline__ float elu_fast(float x) {
    return x > 0.f ? x : __builtin_amdgcn_exp2f(x * 1.4426950408889634f) - 1.f;
}

// bias + ELU, split, store the three planes of this lane's 16 (node, out) values
__device__ __forceinline__ void mlp_store_bf16(const f32x16 &acc, const float *__restrict__ bias, int o, int O, char *dst,
                                               int stride, int plane_bytes, int hh) {
    if (o >= O) return;
    const float bv = bias[o];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int node = (r & 3) + 8 * (r >> 2) + 4 * hh;
        __bf16 h, m, l;
        split3(elu_fast(acc[r] + bv), h, m, l);
        char *p = dst + node * stride + 2 * o;
        *(__bf16 *)(p) = h;
        *(__bf16 *)(p + plane_bytes) = m;
        *(__bf16 *)(p + 2 * plane_bytes) = l;
    }
}

__global__ __launch_bounds__(MB_THREADS) void mlp_bf16x3_kernel(const float *__restrict__ z, int rows,
                                                                const __bf16 *__restrict__ Wp0, const float *__restrict__ b0,
                                                                const __bf16 *__restrict__ Wp1, const float *__restrict__ b1,
                                                                const __bf16 *__restrict__ Wp2, const float *__restrict__ b2,
                                                                const __bf16 *__restrict__ Wp3, const float *__restrict__ b3,
                                                                float *__restrict__ out, const int *__restrict__ gate) {
    if (gate && *gate == 0) return;  // fallback launch behind the fp16x2 kernel: runs only if that one left fp16's range
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *bufA = smem;                        // [32][MB_SA]
    char *bufB = smem + MB_NODES * MB_SA;     // [32][MB_SB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, hh = lane >> 5;
    const int row0 = blockIdx.x * MB_NODES;

    // stage z: split every element into its three planes (columns 262..271 are zero)
    for (int e = tid; e < MB_NODES * (MB_K0 / 4); e += MB_THREADS) {
        const int r = e / (MB_K0 / 4), c = e % (MB_K0 / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row0 + r < rows && 4 * c < MB_ZS) v = *(const f32x4 *)(z + (size_t)(row0 + r) * MB_ZS + 4 * c);
        char *p = bufA + r * MB_SA + 8 * c;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __bf16 h, m, l;
            split3(v[q], h, m, l);
            *(__bf16 *)(p + 2 * q) = h;
            *(__bf16 *)(p + 2 * MB_K0 + 2 * q) = m;
            *(__bf16 *)(p + 4 * MB_K0 + 2 * q) = l;
        }
    }
    __syncthreads();
    // layer 0: K0 -> 512 : 16 output tiles, 2 per wave
    for (int q = 0; q < 2; ++q) {
        const int ot = wave * 2 + q;
        f32x16 acc = mlp_tile_bf16(bufA + r32 * MB_SA + 16 * hh, 2 * MB_K0, Wp0 + (size_t)ot * (MB_K0 / 16) * 1536 + lane * 8,
                                   MB_K0 / 16);
        mlp_store_bf16(acc, b0, ot * 32 + r32, 512, bufB, MB_SB, 2 * MB_K1, hh);
    }
    __syncthreads();
    // layer 1: 512 -> 256 : 8 tiles, 1 per wave
    {
        const int ot = wave;
        f32x16 acc = mlp_tile_bf16(bufB + r32 * MB_SB + 16 * hh, 2 * MB_K1, Wp1 + (size_t)ot * (MB_K1 / 16) * 1536 + lane * 8,
                                   MB_K1 / 16);
        mlp_store_bf16(acc, b1, ot * 32 + r32, 256, bufA, MB_SA, 2 * MB_K2, hh);
    }
    __syncthreads();
    // layer 2: 256 -> 128 : 4 tiles
    if (wave < 4) {
        const int ot = wave;
        f32x16 acc = mlp_tile_bf16(bufA + r32 * MB_SA + 16 * hh, 2 * MB_K2, Wp2 + (size_t)ot * (MB_K2 / 16) * 1536 + lane * 8,
                                   MB_K2 / 16);
        mlp_store_bf16(acc, b2, ot * 32 + r32, 128, bufB, MB_SB, 2 * MB_K3, hh);
    }
    __syncthreads();
    // layer 3: 128 -> 9 : one tile, straight to HBM
    if (wave == 0) {
        f32x16 acc = mlp_tile_bf16(bufB + r32 * MB_SB + 16 * hh, 2 * MB_K3, Wp3 + lane * 8, MB_K3 / 16);
        const int o = r32;
        const float bv = o < 9 ? b3[o] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int node = (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (o < 9 && row0 + node < rows) out[(size_t)(row0 + node) * 9 + o] = acc[r] + bv;
        }
    }
}

size_t mlp_bf16_pack_bytes() {
    return align_up(((size_t)16 * (MB_K0 / 16) + (size_t)8 * (MB_K1 / 16) + (size_t)4 * (MB_K2 / 16) + (size_t)1 * (MB_K3 / 16)) * 1536 *
                    sizeof(__bf16));
}

// z [rows][264] fp32 -> out [rows][9]; scratch = mlp_bf16_pack_bytes() bytes
void launch_mlp_rows_bf16(const float *z, int rows, const float *W0, const float *b0, const float *W1, const float *b1,
                          const float *W2, const float *b2, const float *W3, const float *b3, void *scratch, float *out,
                          hipStream_t s, const int *gate) {
    __bf16 *Wp0 = (__bf16 *)scratch;
    __bf16 *Wp1 = Wp0 + (size_t)16 * (MB_K0 / 16) * 1536;
    __bf16 *Wp2 = Wp1 + (size_t)8 * (MB_K1 / 16) * 1536;
    __bf16 *Wp3 = Wp2 + (size_t)4 * (MB_K2 / 16) * 1536;
    {
        PackLayersBf16 a;
        long maxth = 0;
        auto layer = [&](int q, const float *W, int O, int I, int otiles, int steps, __bf16 *Wp) {
            a.W[q] = W, a.O[q] = O, a.I[q] = I, a.otiles[q] = otiles, a.steps[q] = steps, a.Wp[q] = Wp;
            const long th = (long)otiles * steps * 512;
            maxth = th > maxth ? th : maxth;
        };
        layer(0, W0, 512, 262, 16, MB_K0 / 16, Wp0);
        layer(1, W1, 256, 512, 8, MB_K1 / 16, Wp1);
        layer(2, W2, 128, 256, 4, MB_K2 / 16, Wp2);
        layer(3, W3, 9, 128, 1, MB_K3 / 16, Wp3);
        hipLaunchKernelGGL(pack_layers_bf16_kernel, dim3((unsigned)((maxth + 255) / 256), 4), dim3(256), 0, s, a, gate);
    }
    ensure_dyn_lds((const void *)mlp_bf16x3_kernel, (int)MB_LDS_BYTES);
    hipLaunchKernelGGL(mlp_bf16x3_kernel, dim3((rows + MB_NODES - 1) / MB_NODES), dim3(MB_THREADS), MB_LDS_BYTES, s, z, rows, Wp0,
                       b0, Wp1, b1, Wp2, b2, Wp3, b3, out, gate);
}

}  // namespace dvm
