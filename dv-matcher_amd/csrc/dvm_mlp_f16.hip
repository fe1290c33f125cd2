// dvm_mlp_f16.hip — the Deformer's decoder MLP (262 -> 512 -> 256 -> 128 -> 9, ELU) on the 16-bit matrix cores
// with fp32-level accuracy, 64 nodes per workgroup.
//
// The bf16x3 kernel (dvm_mlp_bf16.hip) is bound by streaming its pre-split weights from L2: every 32-node
// workgroup reads all 1.8 MB (13 TB/s of L2 -> CU traffic at 256 pairs).  Here
//   * operands are split 2-way into fp16 planes, x*s = h + m (+ r, |r| <= 2^-22 |x*s|), with fixed power-of-two
//     scales (activations 2^5, weights 2^8): three partial products hh + hm + mh instead of six, 2 planes of
//     weights instead of 3;
//   * a workgroup carries 64 nodes (two 32-row MFMA tiles), so a weight fragment is loaded once per two tiles;
//     the 512-wide hidden layer is produced in two halves that layer 1 consumes immediately (split-K, its
//     accumulators stay in registers), which is what lets 64 nodes fit: LDS = z/h1 (70 KB) + h0-half/h2 (66 KB).
//   -> 0.6 MB of weights per 32 nodes instead of 1.8 MB, half the matrix work.
// fp16 has a narrow range: an activation or weight beyond +-60000/scale raises a flag and the caller re-runs the
// launch with the bf16x3 kernel (gated on that flag, so it costs one empty launch otherwise).  Values below
// 2^-3/scale lose relative — not absolute — precision in the m plane (<= 2^-24 of the scale).
// (reference models/model.py:433-452, 476-477; floats only, no integer output depends on it)
#include <stdlib.h>

#include "dvm_common.h"

namespace dvm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int MH_NODES = 64, MH_WAVES = 8, MH_THREADS = 64 * MH_WAVES;
constexpr int MH_ZS = 264;                                     // z row stride in HBM (floats)
constexpr int MH_K0 = 272, MH_K1 = 512, MH_K2 = 256, MH_K3 = 128;  // K padded to multiples of 16
constexpr int MH_SZ = 2 * MH_K0 * 2 + 16;   // 1104 B: z (2 planes of 272) and later h1 (2 x 256)
constexpr int MH_SH = 2 * 256 * 2 + 16;     // 1040 B: one half of h0 (2 x 256) and later h2 (2 x 128)
constexpr size_t MH_LDS_BYTES = (size_t)MH_NODES * (MH_SZ + MH_SH);
constexpr float MH_SA = 32.f, MH_SW = 256.f, MH_INV = 1.f / (32.f * 256.f);  // activation / weight scales
constexpr float MH_LIMIT = 60000.f;

__device__ __forceinline__ void split2(float xs, _Float16 &h, _Float16 &m) {
    h = (_Float16)xs;
    m = (_Float16)(xs - (float)h);
}

// Packed weights: Wp[otile][step][plane][lane][8] fp16, element j of lane (o = lane&31, hh = lane>>5)
// = plane(S_w * W[otile*32 + o][16*step + 8*hh + j])  (0 outside the matrix)
// the four layers of the MLP in ONE launch (blockIdx.y = layer; four launches before: four gaps on the main stream in front of the MLP)
struct PackLayersF16 {
    const float *W[4];
    int O[4], I[4], otiles[4], steps[4];
    _Float16 *Wp[4];
};
__device__ __forceinline__ void pack_weights_f16_body(const float *__restrict__ W, int O, int I, int otiles, int steps, _Float16 *__restrict__ Wp,
                                                      int *__restrict__ flag);
__global__ void pack_layers_f16_kernel(const PackLayersF16 a, int *__restrict__ flag) {
    const int q = blockIdx.y;
    pack_weights_f16_body(a.W[q], a.O[q], a.I[q], a.otiles[q], a.steps[q], a.Wp[q], flag);
}
__device__ __forceinline__ void pack_weights_f16_body(const float *__restrict__ W, int O, int I, int otiles, int steps, _Float16 *__restrict__ Wp,
                                                      int *__restrict__ flag) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)otiles * steps * 64 * 8;
    if (g >= total) return;
    int j = (int)(g & 7);
    int lane = (int)((g >> 3) & 63);
    int step = (int)((g >> 9) % steps);
    int ot = (int)(g / (512L * steps));
    int o = ot * 32 + (lane & 31), c = 16 * step + 8 * (lane >> 5) + j;
    float w = (o < O && c < I) ? W[(size_t)o * I + c] * MH_SW : 0.f;
    if (!(fabsf(w) <= MH_LIMIT)) atomicOr(flag, 1);
    _Float16 h, m;
    split2(w, h, m);
    size_t base = (((size_t)ot * steps + step) * 2) * 512 + (size_t)lane * 8 + j;
    Wp[base] = h;
    Wp[base + 512] = m;
}

// acc[t][node][out] += sum_k act_t[node][k] W[out][k] over `steps` k-steps, for NT 32-node tiles sharing the weight
// fragments.  a0: this lane's row of tile 0 (+ 16*hh); tile t is 32 rows further.  The weight fragments come straight
// from L2 (every workgroup streams the same 1.2 MB), MH_AHEAD k-steps ahead of their matrix instructions.
// the first MH_AHEAD k-steps of a weight stream, requested AHEAD of the phase that uses them (before the activation stores and
// the barrier of the previous phase: each of the six phases of a workgroup otherwise starts with an exposed L2 round trip)
template <int MH_AHEAD>
struct WFrag {
    f16x8 h[MH_AHEAD], m[MH_AHEAD];
    __device__ __forceinline__ void request(const _Float16 *__restrict__ wp, int steps) {
#pragma unroll
        for (int u = 0; u < MH_AHEAD; ++u) {
            const _Float16 *w = wp + (size_t)(u < steps ? u : steps - 1) * 1024;
            h[u] = *(const f16x8 *)(w), m[u] = *(const f16x8 *)(w + 512);
        }
    }
};

// One k-loop of a phase, software-pipelined BY HAND: at step s the weight fragments of step s + MH_AHEAD and the activation
// fragments of step s + 1 are requested, then step s's matrix instructions run - with scheduling barriers in between, because
// left alone the compiler sinks every load to just above its first use to save registers (ISA: `global_load` / `s_waitcnt
// vmcnt(0)` / `v_mfma` - a whole L2 round trip exposed per k-step - and `ds_read` x 2 / `s_waitcnt lgkmcnt(1)` / `v_mfma`).
template <int NT, int MH_AHEAD, int STEPS, int ABL = 0>
__device__ __forceinline__ void mma_tiles(const char *__restrict__ a0, int row_stride, int plane_bytes,
                                          const _Float16 *__restrict__ wp /* (otile, first step) base + lane*8 */,
                                          f32x16 (&acc)[NT], const WFrag<MH_AHEAD> &first) {
    f16x8 wh[MH_AHEAD], wm[MH_AHEAD];
#pragma unroll
    for (int u = 0; u < MH_AHEAD; ++u) wh[u] = first.h[u], wm[u] = first.m[u];
    f16x8 ah[2][NT], am[2][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const char *ar = a0 + t * 32 * row_stride;
        ah[0][t] = *(const f16x8 *)(ar), am[0][t] = *(const f16x8 *)(ar + plane_bytes);
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int u = s % MH_AHEAD, cur = s & 1;
        const f16x8 bh = wh[u], bm = wm[u];
        if (!(ABL & 1) && s + MH_AHEAD < STEPS) {
            const _Float16 *wn = wp + (size_t)(s + MH_AHEAD) * 1024;
            wh[u] = *(const f16x8 *)(wn), wm[u] = *(const f16x8 *)(wn + 512);
        }
        if (s + 1 < STEPS) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const char *ar = a0 + t * 32 * row_stride + ((ABL & 2) ? 0 : 32 * (s + 1));
                ah[cur ^ 1][t] = *(const f16x8 *)(ar), am[cur ^ 1][t] = *(const f16x8 *)(ar + plane_bytes);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // weights as the A operand, activations as B: the accumulator tile is [out][node] - a lane owns ONE node (its
            // r32) and 4 x 4 consecutive outputs, so the activation stores below are 8-byte ones (the fragments of the
            // two operands have the same register layout: swapping them transposes the tile and nothing else)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, am[cur][t], acc[t], 0, 0, 0);  // small terms first
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bm, ah[cur][t], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah[cur][t], acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ float elu_fast(float x) {
    return x > 0.f ? x : __builtin_amdgcn_exp2f(x * 1.4426950408889634f) - 1.f;
}

// Two values -> their packed fp16 planes: h = rn16(a), m = rn16(a - h) (a - h is exact in fp32), three instructions per pair:
// the packed round-to-nearest conversion (gfx950) and one v_fma_mix per value, which reads the fp16 h directly and writes its
// half of m.  (The compiler's form of split2 is convert, convert back, subtract, convert, pack: 4.5 per value.)
__device__ __forceinline__ void split2x2(float a0, float a1, unsigned &h, unsigned &m) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a0), "v"(a1));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(m) : "v"(a0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(m) : "v"(a1), "v"(h));
}

// bias + ELU + scale + split of this lane's 16 (out, node) values, then four 8-byte stores per plane: register r = 4 g + e holds
// output ocol0 + 8 g + e of the lane's node.  Everything runs in the SCALED domain (xs = S_a x: the scales are powers of two,
// so every rounding is the one of the unscaled formula): xs = fma(acc, S_a / (S_a S_w), S_a b); S_a elu(x) = xs > 0 ? xs :
// fma(exp2(xs log2e / S_a), S_a, -S_a).  The range guard is a running max of |a| (one instruction; compared once per kernel).
// These epilogues are vector work that no matrix instruction overlaps (all eight waves reach them together): at 14.5
// instructions per value they were 6.5 of a workgroup's 31 us.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct Bias16 {
    f32x4 v[4];
    __device__ __forceinline__ void request(const float *__restrict__ b, int ocol0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = *(const f32x4 *)(b + ocol0 + 8 * g);
    }
};
__device__ __forceinline__ void store_act(const f32x16 &acc, const Bias16 &bv, int ocol0, char *dst /* this lane's node row */, int plane_bytes,
                                          float &amax) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xs = fmaf(acc[4 * g + e], MH_INV * MH_SA, bv.v[g][e] * MH_SA);
            const float ex = fmaf(__builtin_amdgcn_exp2f(xs * (1.4426950408889634f / MH_SA)), MH_SA, -MH_SA);
            a[e] = xs > 0.f ? xs : ex;
            amax = fmaxf(amax, fabsf(a[e]));
        }
        unsigned h0, m0, h1, m1;
        split2x2(a[0], a[1], h0, m0);
        split2x2(a[2], a[3], h1, m1);
        const u32x2 h = {h0, h1}, m = {m0, m1};
        char *p = dst + 2 * (ocol0 + 8 * g);
        *(u32x2 *)(p) = h;
        *(u32x2 *)(p + plane_bytes) = m;
    }
}

template <int AHEAD, int ABL = 0>
__global__ __launch_bounds__(MH_THREADS) void mlp_f16x2_kernel(const float *__restrict__ z, int rows,
                                                               const _Float16 *__restrict__ Wp0, const float *__restrict__ b0,
                                                               const _Float16 *__restrict__ Wp1, const float *__restrict__ b1,
                                                               const _Float16 *__restrict__ Wp2, const float *__restrict__ b2,
                                                               const _Float16 *__restrict__ Wp3, const float *__restrict__ b3,
                                                               float *__restrict__ out, int *__restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *bufZ = smem;                       // [64][MH_SZ]
    char *bufH = smem + MH_NODES * MH_SZ;    // [64][MH_SH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, hh = lane >> 5;
    const int row0 = blockIdx.x * MH_NODES;
    float amax = 0.f;   // largest |activation| of this lane (scaled): beyond fp16's range -> the flag (inf included; a NaN is a NaN in the result either way)
    // this wave's five biases and the first weight fragments of layer 0: requested before anything else, in flight while the z
    // rows are staged (each bias load used to sit between a phase's last matrix instruction and its activation stores)
    // (the biases of a phase — 16 per lane — are requested at the head of the phase: they land under its matrix instructions)
    WFrag<AHEAD> wfirst;
    wfirst.request(Wp0 + (size_t)wave * (MH_K0 / 16) * 1024 + lane * 8, MH_K0 / 16);

    // stage z: scale, split into the two planes (columns 262..271 are zero)
    // (all of a thread's 9 loads are requested before the first is used: row and column are clamped and the value selected —
    // a predicated load is a branch around a load with a full wait in front of it, nine dependent round trips per workgroup)
    constexpr int ZIT = (MH_NODES * (MH_K0 / 4) + MH_THREADS - 1) / MH_THREADS;
    f32x4 zv[ZIT];
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
        const int e = tid + it * MH_THREADS, ec = e < MH_NODES * (MH_K0 / 4) ? e : 0;
        const int r = ec / (MH_K0 / 4), c = ec % (MH_K0 / 4);
        const int rr = row0 + r < rows ? row0 + r : rows - 1, cc = 4 * c < MH_ZS ? 4 * c : MH_ZS - 4;
        zv[it] = *(const f32x4 *)(z + (size_t)rr * MH_ZS + cc);
    }
#pragma unroll
    for (int it = 0; it < ZIT; ++it) {
        const int e = tid + it * MH_THREADS;
        if (e >= MH_NODES * (MH_K0 / 4)) break;
        const int r = e / (MH_K0 / 4), c = e % (MH_K0 / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row0 + r < rows && 4 * c < MH_ZS) v = zv[it];
        char *p = bufZ + r * MH_SZ + 8 * c;
        float a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = (4 * c + q < 262 ? v[q] : 0.f) * MH_SA;
            amax = fmaxf(amax, fabsf(a[q]));
        }
        unsigned h0, m0, h1, m1;
        split2x2(a[0], a[1], h0, m0);
        split2x2(a[2], a[3], h1, m1);
        const u32x2 h = {h0, h1}, m = {m0, m1};
        *(u32x2 *)(p) = h;
        *(u32x2 *)(p + 2 * MH_K0) = m;
    }
    __syncthreads();

    // layer 0 in two halves of 256 outputs; layer 1 consumes each half at once (split-K, accumulators in registers)
    const _Float16 *const w0p[2] = {Wp0 + (size_t)wave * (MH_K0 / 16) * 1024 + lane * 8, Wp0 + (size_t)(8 + wave) * (MH_K0 / 16) * 1024 + lane * 8};
    const _Float16 *const w1p[2] = {Wp1 + ((size_t)wave * (MH_K1 / 16)) * 1024 + lane * 8, Wp1 + ((size_t)wave * (MH_K1 / 16) + 16) * 1024 + lane * 8};
    const _Float16 *const w2p = Wp2 + (size_t)(wave & 3) * (MH_K2 / 16) * 1024 + lane * 8;
    const _Float16 *const w3p = Wp3 + lane * 8;
    f32x16 acc1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;
    WFrag<AHEAD> wa = wfirst, wb;   // (wfirst: layer 0, first half — requested before the z rows were staged)
    Bias16 bias1;
    for (int hlf = 0; hlf < 2; ++hlf) {
        {
            const int ot = 8 * hlf + wave;  // of 16 output tiles
            f32x16 acc0[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc0[t][r] = 0.f;
            Bias16 bv;
            bv.request(b0, ot * 32 + 4 * hh);
            mma_tiles<2, AHEAD, MH_K0 / 16, ABL>(bufZ + r32 * MH_SZ + 16 * hh, MH_SZ, 2 * MH_K0, w0p[hlf], acc0, wa);
            wb.request(w1p[hlf], 16);   // layer 1's first fragments travel while this half's activations are stored
#pragma unroll
            for (int t = 0; t < 2; ++t) store_act(acc0[t], bv, wave * 32 + 4 * hh, bufH + (t * 32 + r32) * MH_SH, 2 * 256, amax);
        }
        __syncthreads();
        // layer 1, K-half hlf: out tile = wave (8 tiles = 256 outputs), k-steps 16*hlf .. 16*hlf+15
        if (hlf == 1) bias1.request(b1, wave * 32 + 4 * hh);
        mma_tiles<2, AHEAD, 16, ABL>(bufH + r32 * MH_SH + 16 * hh, MH_SH, 2 * 256, w1p[hlf], acc1, wb);
        if (hlf == 0) wa.request(w0p[1], MH_K0 / 16); else wa.request(w2p, MH_K2 / 16);   // the next phase's, across the barrier
        __syncthreads();
    }
    {   // h1 -> bufZ (z is dead)
#pragma unroll
        for (int t = 0; t < 2; ++t) store_act(acc1[t], bias1, wave * 32 + 4 * hh, bufZ + (t * 32 + r32) * MH_SZ, 2 * MH_K2, amax);
    }
    __syncthreads();
    // layer 2: 256 -> 128 : 4 output tiles x 2 node tiles, one pair per wave
    {
        const int ot = wave & 3, nt = wave >> 2;
        f32x16 acc2[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[0][r] = 0.f;
        Bias16 bias2;
        bias2.request(b2, ot * 32 + 4 * hh);
        mma_tiles<1, AHEAD, MH_K2 / 16, ABL>(bufZ + (nt * 32 + r32) * MH_SZ + 16 * hh, MH_SZ, 2 * MH_K2, w2p, acc2, wa);
        if (wave < 2) wb.request(w3p, MH_K3 / 16);
        store_act(acc2[0], bias2, ot * 32 + 4 * hh, bufH + (nt * 32 + r32) * MH_SH, 2 * MH_K3, amax);
    }
    __syncthreads();
    // layer 3: 128 -> 9 : one output tile per node tile, straight to HBM
    if (wave < 2) {
        f32x16 acc3[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[0][r] = 0.f;
        mma_tiles<1, AHEAD, MH_K3 / 16, ABL>(bufH + (wave * 32 + r32) * MH_SH + 16 * hh, MH_SH, 2 * MH_K3, w3p, acc3, wb);
        const int node = wave * 32 + r32;   // (transposed tile: the lane's node, outputs (r & 3) + 8 (r >> 2) + 4 hh — r < 5 reaches 0..8)
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int o = (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (o < 9 && row0 + node < rows) out[(size_t)(row0 + node) * 9 + o] = acc3[0][r] * MH_INV + b3[o];
        }
    }
    if (__any(!(amax <= MH_LIMIT)) && lane == 0) atomicOr(flag, 1);
}

}  // namespace

size_t mlp_f16_pack_bytes() {
    return align_up(((size_t)16 * (MH_K0 / 16) + (size_t)8 * (MH_K1 / 16) + (size_t)4 * (MH_K2 / 16) + (size_t)1 * (MH_K3 / 16)) * 1024 *
                    sizeof(_Float16)) +
           align_up(sizeof(int));
}

// z [rows][264] fp32 -> out [rows][9]; scratch = mlp_f16_pack_bytes() bytes.  Returns the device flag that is non-zero
// when a value left fp16's range (the results are then invalid and the bf16x3 kernel must overwrite them).
int *launch_mlp_rows_f16(const float *z, int rows, const float *W0, const float *b0, const float *W1, const float *b1,
                         const float *W2, const float *b2, const float *W3, const float *b3, void *scratch, float *out,
                         hipStream_t s) {
    _Float16 *Wp0 = (_Float16 *)scratch;
    _Float16 *Wp1 = Wp0 + (size_t)16 * (MH_K0 / 16) * 1024;
    _Float16 *Wp2 = Wp1 + (size_t)8 * (MH_K1 / 16) * 1024;
    _Float16 *Wp3 = Wp2 + (size_t)4 * (MH_K2 / 16) * 1024;
    int *flag = (int *)((char *)scratch + mlp_f16_pack_bytes() - align_up(sizeof(int)));
    (void)hipMemsetAsync(flag, 0, sizeof(int), s);
    {
        PackLayersF16 a;
        long maxth = 0;
        auto layer = [&](int q, const float *W, int O, int I, int otiles, int steps, _Float16 *Wp) {
            a.W[q] = W, a.O[q] = O, a.I[q] = I, a.otiles[q] = otiles, a.steps[q] = steps, a.Wp[q] = Wp;
            const long th = (long)otiles * steps * 512;
            maxth = th > maxth ? th : maxth;
        };
        layer(0, W0, 512, 262, 16, MH_K0 / 16, Wp0);
        layer(1, W1, 256, 512, 8, MH_K1 / 16, Wp1);
        layer(2, W2, 128, 256, 4, MH_K2 / 16, Wp2);
        layer(3, W3, 9, 128, 1, MH_K3 / 16, Wp3);
        hipLaunchKernelGGL(pack_layers_f16_kernel, dim3((unsigned)((maxth + 255) / 256), 4), dim3(256), 0, s, a, flag);
    }
    // weight k-steps requested ahead of their matrix instructions: 2 / 4 / 8 measured alike (2.74 / 2.83 / 2.82 ms per launch
    // at 512 pairs) — the waves' 59 % parked cycles (SQ_WAIT_ANY) are not the L2 latency of the weights (DVM_MLP_AHEAD = A/B)
    static const int ahead = [] {
        const char *e = getenv("DVM_MLP_AHEAD");
        return e ? atoi(e) : 4;
    }();
    prof_begin(s, DVM_PROF_MLP);
    const dim3 grid((rows + MH_NODES - 1) / MH_NODES), block(MH_THREADS);
    // (ablation, WRONG results: 1 = the weight fragments are not re-loaded in the k-loops, 2 = the activation fragments are read
    // from one LDS address, 3 = both: what is left is the matrix instructions, the activation stores and the barriers)
    // compiled in only with `make ABLATE=1` (-DDVM_ABLATE): a stray environment variable must not be able to corrupt a production run
#ifdef DVM_ABLATE
    static const int abl = [] { const char *e = getenv("DVM_MLP_ABLATE"); return e ? atoi(e) : 0; }();
    if (abl == 1 || abl == 2 || abl == 3) {
        static bool warned = false;
        if (!warned) warned = true, fprintf(stderr, "libdvm_hip: DVM_MLP_ABLATE=%d: the Deformer MLP returns WRONG results (timing experiment)\n", abl);
        auto k = abl == 1 ? mlp_f16x2_kernel<2, 1> : abl == 2 ? mlp_f16x2_kernel<2, 2> : mlp_f16x2_kernel<2, 3>;
        ensure_dyn_lds((const void *)k, (int)MH_LDS_BYTES);
        hipLaunchKernelGGL(k, grid, block, MH_LDS_BYTES, s, z, rows, Wp0, b0, Wp1, b1, Wp2, b2, Wp3, b3, out, flag);
    } else
#endif
    if (ahead <= 2) {
        ensure_dyn_lds((const void *)mlp_f16x2_kernel<2>, (int)MH_LDS_BYTES);
        hipLaunchKernelGGL(mlp_f16x2_kernel<2>, grid, block, MH_LDS_BYTES, s, z, rows, Wp0, b0, Wp1, b1, Wp2, b2, Wp3, b3, out, flag);
    } else if (ahead <= 4) {
        ensure_dyn_lds((const void *)mlp_f16x2_kernel<4>, (int)MH_LDS_BYTES);
        hipLaunchKernelGGL(mlp_f16x2_kernel<4>, grid, block, MH_LDS_BYTES, s, z, rows, Wp0, b0, Wp1, b1, Wp2, b2, Wp3, b3, out, flag);
    } else {
        ensure_dyn_lds((const void *)mlp_f16x2_kernel<8>, (int)MH_LDS_BYTES);
        hipLaunchKernelGGL(mlp_f16x2_kernel<8>, grid, block, MH_LDS_BYTES, s, z, rows, Wp0, b0, Wp1, b1, Wp2, b2, Wp3, b3, out, flag);
    }
    prof_end(s, DVM_PROF_MLP);
    return flag;
}

}  // namespace dvm
