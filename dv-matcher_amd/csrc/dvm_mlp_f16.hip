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
#include "dvm_mlp_f16.h"

namespace dvm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int MH_WAVES = 8, MH_THREADS = 64 * MH_WAVES;    // (MH_NODES, MH_K0, MH_SZ, the scales: dvm_mlp_f16.h)
constexpr int MH_K1 = 512, MH_K2 = 256, MH_K3 = 128;  // K padded to multiples of 16 (MH_K0 = 272)
constexpr int MH_SH = 2 * 256 * 2 + 16;     // 1040 B: one half of h0 (2 x 256) and later h2 (2 x 128) / h1 (persistent form)
constexpr size_t MH_LDS_BYTES = (size_t)MH_NODES * (MH_SZ + MH_SH);
constexpr float MH_INV = 1.f / (MH_SA * MH_SW);
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
    int plane_form;   // 1: for the persistent kernel — layer 0's columns in the plane order of z, layer 3 in the accumulator order of layer 2
};
__device__ __forceinline__ void pack_weights_f16_body(const float *__restrict__ W, int O, int I, int otiles, int steps, _Float16 *__restrict__ Wp,
                                                      int *__restrict__ flag, bool zplane_cols);
// Layer 3 for the persistent kernel, which multiplies it onto layer 2's activations straight from that layer's accumulator
// registers: W3x[ot][s][plane][lane][8], element j of lane (o = lane & 31, hh = lane >> 5) = plane(S_w W3[o][k]) with
// k = 32 ot + 4 hh + 8 (2 s + (j >> 2)) + (j & 3) — the output index held by register 4 (2 s + (j >> 2)) + (j & 3) of a lane with that hh
__device__ __forceinline__ void pack_w3x_body(const float *__restrict__ W, _Float16 *__restrict__ Wp, int *__restrict__ flag) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 4 * 2 * 64 * 8) return;
    const int j = g & 7, lane = (g >> 3) & 63, s2 = (g >> 9) & 1, ot = g >> 10;
    const int o = lane & 31, k = 32 * ot + 4 * (lane >> 5) + 8 * (2 * s2 + (j >> 2)) + (j & 3);
    const float w = o < 9 ? W[(size_t)o * MH_K3 + k] * MH_SW : 0.f;
    if (!(fabsf(w) <= MH_LIMIT)) atomicOr(flag, 1);
    _Float16 h, m;
    split2(w, h, m);
    const size_t base = ((size_t)(ot * 2 + s2) * 2) * 512 + (size_t)lane * 8 + j;
    Wp[base] = h;
    Wp[base + 512] = m;
}
__global__ void pack_layers_f16_kernel(const PackLayersF16 a, int *__restrict__ flag) {
    const int q = blockIdx.y;
    if (a.plane_form && q == 3) pack_w3x_body(a.W[3], a.Wp[3], flag);
    else pack_weights_f16_body(a.W[q], a.O[q], a.I[q], a.otiles[q], a.steps[q], a.Wp[q], flag, a.plane_form && q == 0);
}
__device__ __forceinline__ void pack_weights_f16_body(const float *__restrict__ W, int O, int I, int otiles, int steps, _Float16 *__restrict__ Wp,
                                                      int *__restrict__ flag, bool zplane_cols) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)otiles * steps * 64 * 8;
    if (g >= total) return;
    int j = (int)(g & 7);
    int lane = (int)((g >> 3) & 63);
    int step = (int)((g >> 9) % steps);
    int ot = (int)(g / (512L * steps));
    int o = ot * 32 + (lane & 31), c = 16 * step + 8 * (lane >> 5) + j;
    if (zplane_cols) c = mh_zcol_of_plane_col(c) < 0 ? I : mh_zcol_of_plane_col(c);   // (layer 0 against the plane form of z: dvm_mlp_f16.h)
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
template <bool GUARD = true>
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
            if (GUARD) amax = fmaxf(amax, fabsf(a[e]));
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

// (The one-workgroup-per-block kernel of round 3, mlp_f16x2_kernel — the same arithmetic, z rows staged through registers, 1.87 ms
// against 1.76 ms — and its timing ablations were removed in round 5: profiles/notes_r4.md.)


// ---------------------------------------------------------------- the persistent form (round 4)
// Cycle stamps of the kernel above (DVM_MLP_STAMPS, profiles/r4_mlp_stamps.txt): its matrix loops run AT the pipe's floor for two
// waves per SIMD (layer 0: 6 577 cycles per half for 2 x 102 instructions of 32 cycles), but they are 57 % of a workgroup's
// 52 200 cycles: staging the z rows (global -> registers -> scale, split -> LDS, behind one exposed HBM round trip) takes 6 700,
// the activation stores and their barriers 11 700, the barriers behind layer 1 3 200, layer 3 (two of eight waves) 2 200.  A CU
// holds ONE workgroup (137 KB of LDS), so nothing runs beside any of that.  This form:
//  * z arrives as fp16 PLANES in the LDS row layout (written so by the kernel that assembles the rows: dvm_mlp_f16.h), and a
//    workgroup WALKS 64-row blocks: the next block's 69 KiB go global -> LDS by LDS-DMA — no registers, no vector
//    instructions — as soon as the second half of layer 0 has read the current block (h1 goes to the OTHER buffer, which h0's
//    second half has left by then), i.e. under layer 1's second half, layer 2 and the output phase;
//  * layer 3 is multiplied onto layer 2's activations straight from the accumulator registers: a lane's 16 outputs of layer 2
//    ARE the B operand of two k-steps once the weights are packed in that order (pack_w3x_body): six matrix instructions per
//    wave, partial sums of the four output tiles through 12 KB of LDS, added in a fixed order.  No h2 buffer, no phase in which
//    six of eight waves idle, one barrier less;
//  * the range guard is the OUTPUT: a value beyond fp16's range becomes (inf, -inf) planes, whose products sum to NaN in every
//    output of the next layer — any overflow anywhere reaches the node's nine outputs as NaN.  One instruction per activation less.
constexpr int MP_PS = 12;                                                  // floats per (output tile, node) partial row (9 used)
constexpr size_t MP_LDS_BYTES = MH_LDS_BYTES + (size_t)4 * MH_NODES * MP_PS * sizeof(float);
constexpr int MP_ZBLOCK = MH_NODES * MH_SZ;                                // 70 656 B = 69 KiB per 64-row block
static_assert(MP_ZBLOCK % 1024 == 0, "a z block is a whole number of LDS-DMA pieces");

// fp32 z rows [rows][stride] (reference column order) -> the plane form; rows beyond `rows` up to `rows_padded` are zero
__global__ void split_rows_kernel(const float *__restrict__ z, int rows, int rows_padded, int stride, char *__restrict__ zp) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)rows_padded * (MH_K0 / 4)) return;
    const int r = (int)(g / (MH_K0 / 4)), c = 4 * (int)(g % (MH_K0 / 4));
    float a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int zc = mh_zcol_of_plane_col(c + q);
        a[q] = (r < rows && zc >= 0) ? z[(size_t)r * stride + zc] * MH_SA : 0.f;
    }
    unsigned h0, m0, h1, m1;
    split2x2(a[0], a[1], h0, m0);
    split2x2(a[2], a[3], h1, m1);
    const u32x2 h = {h0, h1}, m = {m0, m1};
    char *p = zp + (size_t)r * MH_SZ + 2 * c;
    *(u32x2 *)(p) = h;
    *(u32x2 *)(p + 2 * MH_K0) = m;
}

// STAMP slots: 1 / 4 layer-0 halves, 2 / 5 their stores + barrier, 12 layer-1 halves (both), 3 / 6 the barriers behind them
// (5 includes the DMA issue), 7 h1 stores + barrier, 8 layer 2, 9 its epilogue + layer 3 + partials + DMA wait + barrier,
// 10 output phase, 0 before the first block, 11 whole kernel; per BLOCK of 64 rows (the totals are divided by the blocks walked).
template <int AHEAD, bool STAMP = false>
__global__ __launch_bounds__(MH_THREADS) void mlp_f16x2p_kernel(const char *__restrict__ zp, int rows, int nblocks, int bpw,
                                                                const _Float16 *__restrict__ Wp0, const float *__restrict__ b0,
                                                                const _Float16 *__restrict__ Wp1, const float *__restrict__ b1,
                                                                const _Float16 *__restrict__ Wp2, const float *__restrict__ b2,
                                                                const _Float16 *__restrict__ W3x, const float *__restrict__ b3,
                                                                float *__restrict__ out, int *__restrict__ flag,
                                                                unsigned long long *__restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long T[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0, tstart = 0, rstart = 0;
    if (STAMP) tstart = tlast = __builtin_amdgcn_s_memtime(), rstart = __builtin_amdgcn_s_memrealtime();
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if (STAMP) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            T[slot] += now - tlast;
            tlast = now;
        }
    };
    auto stamp_acc = [&](int slot, const f32x16 &a) __attribute__((always_inline)) {
        if (STAMP) {
            const int x = __builtin_amdgcn_readfirstlane(__float_as_int(a[0]));
            asm volatile("" ::"s"(x));
            stamp(slot);
        }
    };
    char *const bufZ = smem;                                   // [64][MH_SZ]: the z planes of the current block
    char *const bufH = smem + MH_NODES * MH_SZ;                // [64][MH_SH]: a half of h0, then h1
    float *const bufP = (float *)(smem + MH_LDS_BYTES);        // [4][64][MP_PS]: layer 3's partial outputs per output tile of layer 2
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31;

    // a block's planes: 69 pieces of 1 KiB, nine per wave (the tail pieces are brought twice: same bytes)
    auto stage_z = [&](int blk) __attribute__((always_inline)) {
        const char *src = zp + (size_t)blk * MP_ZBLOCK + lane * 16;
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            const int p = min(wave + 8 * e, MP_ZBLOCK / 1024 - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)p * 1024),
                                             (__attribute__((address_space(3))) void *)(bufZ + p * 1024), 16, 0, 0);
        }
    };
    // barrier behind LDS-DMA: every wave first waits for its own pieces (see dvm_softcorr_sweep2.hip)
    auto dma_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    int blk = blockIdx.x * bpw;                                   // this workgroup walks blocks [blk, blk_end)
    const int blk_end = min(blk + bpw, nblocks);
    if (blk >= blk_end) return;
    stage_z(blk);
    // the output phase's indices: thread t handles flat output t (and t + 512 < 576)
    const int on0 = tid / 9, oo0 = tid - 9 * on0, on1 = (tid + MH_THREADS) / 9, oo1 = (tid + MH_THREADS) - 9 * on1;
    const float b3a = b3[oo0], b3b = b3[oo1];
    bool bad = false;
    int lane8 = lane * 8;
    WFrag<AHEAD> wa, wb;
    wa.request(Wp0 + (size_t)wave * (MH_K0 / 16) * 1024 + lane8, MH_K0 / 16);
    dma_barrier();
    stamp(0);
    float unused_amax = 0.f;
    for (; blk < blk_end; ++blk) {
        // (laundered once per block: the weight streams are block-invariant, and the compiler otherwise hoists whatever loads and
        // addresses it can out of this loop — into registers the matrix loops need)
        asm volatile("" : "+v"(lane8));
        const int hh = lane8 >> 8;   // (from the laundered value: the bias loads stay inside the loop as well)
        const int row0 = blk * MH_NODES;
        const _Float16 *const w0p[2] = {Wp0 + (size_t)wave * (MH_K0 / 16) * 1024 + lane8, Wp0 + (size_t)(8 + wave) * (MH_K0 / 16) * 1024 + lane8};
        const _Float16 *const w1p[2] = {Wp1 + ((size_t)wave * (MH_K1 / 16)) * 1024 + lane8, Wp1 + ((size_t)wave * (MH_K1 / 16) + 16) * 1024 + lane8};
        const _Float16 *const w2p = Wp2 + (size_t)(wave & 3) * (MH_K2 / 16) * 1024 + lane8;
        const _Float16 *const w3p = W3x + (size_t)(wave & 3) * 2 * 1024 + lane8;
        f32x16 acc1[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;
        Bias16 bias1;
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            {
                const int ot = 8 * hlf + wave;
                f32x16 acc0[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc0[t][r] = 0.f;
                Bias16 bv;
                bv.request(b0, ot * 32 + 4 * hh);
                mma_tiles<2, AHEAD, MH_K0 / 16>(bufZ + r32 * MH_SZ + 16 * hh, MH_SZ, 2 * MH_K0, w0p[hlf], acc0, wa);
                stamp_acc(hlf ? 4 : 1, acc0[1]);
                wb.request(w1p[hlf], 16);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    store_act<false>(acc0[t], bv, wave * 32 + 4 * hh, bufH + (t * 32 + r32) * MH_SH, 2 * 256, unused_amax);
            }
            __syncthreads();
            if (hlf == 1) {   // every wave is through layer 0: the next block's planes may land on this block's
                if (blk + 1 < blk_end) stage_z(blk + 1);
                bias1.request(b1, wave * 32 + 4 * hh);
            }
            stamp(hlf ? 5 : 2);
            mma_tiles<2, AHEAD, 16>(bufH + r32 * MH_SH + 16 * hh, MH_SH, 2 * 256, w1p[hlf], acc1, wb);
            if (hlf == 0) wa.request(w0p[1], MH_K0 / 16); else wa.request(w2p, MH_K2 / 16);
            stamp_acc(12, acc1[1]);
            __syncthreads();
            stamp(hlf ? 6 : 3);
        }
        // h1 -> bufH (h0's second half is dead)
#pragma unroll
        for (int t = 0; t < 2; ++t) store_act<false>(acc1[t], bias1, wave * 32 + 4 * hh, bufH + (t * 32 + r32) * MH_SH, 2 * MH_K2, unused_amax);
        __syncthreads();
        stamp(7);
        // layer 2: 4 output tiles x 2 node tiles, one pair per wave; layer 3 on its accumulators
        {
            const int ot = wave & 3, nt = wave >> 2;
            f32x16 acc2[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[0][r] = 0.f;
            Bias16 bias2;
            bias2.request(b2, ot * 32 + 4 * hh);
            f16x8 w3h[2], w3m[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) w3h[u] = *(const f16x8 *)(w3p + (size_t)u * 1024), w3m[u] = *(const f16x8 *)(w3p + (size_t)u * 1024 + 512);
            mma_tiles<1, AHEAD, MH_K2 / 16>(bufH + (nt * 32 + r32) * MH_SH + 16 * hh, MH_SH, 2 * MH_K2, w2p, acc2, wa);
            stamp_acc(8, acc2[0]);
            // bias + ELU + split as store_act, but the planes stay in registers: registers 8 u .. 8 u + 7 of the lane are the eight
            // k-slots of this lane in k-step u of layer 3 (the packing of W3x)
            unsigned hf[8], mf[8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float a[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xs = fmaf(acc2[0][4 * g + e], MH_INV * MH_SA, bias2.v[g][e] * MH_SA);
                    const float ex = fmaf(__builtin_amdgcn_exp2f(xs * (1.4426950408889634f / MH_SA)), MH_SA, -MH_SA);
                    a[e] = xs > 0.f ? xs : ex;
                }
                split2x2(a[0], a[1], hf[2 * g], mf[2 * g]);
                split2x2(a[2], a[3], hf[2 * g + 1], mf[2 * g + 1]);
            }
            f32x16 acc3 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 hq = {hf[4 * u], hf[4 * u + 1], hf[4 * u + 2], hf[4 * u + 3]}, mq = {mf[4 * u], mf[4 * u + 1], mf[4 * u + 2], mf[4 * u + 3]};
                const f16x8 ah = __builtin_bit_cast(f16x8, hq), am = __builtin_bit_cast(f16x8, mq);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3h[u], am, acc3, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3m[u], ah, acc3, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3h[u], ah, acc3, 0, 0, 0);
            }
            // register r = output (r & 3) + 8 (r >> 2) + 4 hh of the lane's node: outputs 0..8 are r = 0..3 of either half and r = 4 of half 0
            float *pr = bufP + ((size_t)ot * MH_NODES + nt * 32 + r32) * MP_PS;
            const f32x4 q = {acc3[0], acc3[1], acc3[2], acc3[3]};
            *(f32x4 *)(pr + 4 * hh) = q;
            if (hh == 0) pr[8] = acc3[4];
        }
        dma_barrier();   // the partial outputs are visible — and the next block's planes have landed
        stamp(9);
        if (blk + 1 < blk_end) wa.request(w0p[0], MH_K0 / 16);   // layer 0's first fragments travel under the output phase
        {
            const float *p0 = bufP + (size_t)on0 * MP_PS + oo0;
            const float v0 = ((p0[0] + p0[MH_NODES * MP_PS]) + (p0[2 * MH_NODES * MP_PS] + p0[3 * MH_NODES * MP_PS])) * MH_INV + b3a;
            if (row0 + on0 < rows) {
                out[(size_t)row0 * 9 + tid] = v0;
                bad = bad || !(fabsf(v0) <= 3.0e38f);
            }
            if (tid < MH_NODES * 9 - MH_THREADS) {
                const float *p1 = bufP + (size_t)on1 * MP_PS + oo1;
                const float v1 = ((p1[0] + p1[MH_NODES * MP_PS]) + (p1[2 * MH_NODES * MP_PS] + p1[3 * MH_NODES * MP_PS])) * MH_INV + b3b;
                if (row0 + on1 < rows) {
                    out[(size_t)row0 * 9 + tid + MH_THREADS] = v1;
                    bad = bad || !(fabsf(v1) <= 3.0e38f);
                }
            }
        }
        stamp(10);
    }
    if (__any(bad) && lane == 0) atomicOr(flag, 1);
    if (STAMP) {
        T[11] = __builtin_amdgcn_s_memtime() - tstart;
        T[13] = __builtin_amdgcn_s_memrealtime() - rstart;   // the constant 100 MHz counter: T[11] / T[13] = the shader clock in units of 100 MHz
        if (lane == 0 && stamps)
            for (int i = 0; i < 14; ++i) stamps[((size_t)blockIdx.x * MH_WAVES + wave) * 14 + i] = T[i];
    }
}

}  // namespace

static size_t mlp_f16_w3x_offset() {
    return align_up(((size_t)16 * (MH_K0 / 16) + (size_t)8 * (MH_K1 / 16) + (size_t)4 * (MH_K2 / 16) + (size_t)1 * (MH_K3 / 16)) * 1024 * sizeof(_Float16));
}
size_t mlp_f16_pack_bytes() {   // [Wp0 | Wp1 | Wp2 | Wp3 | W3x (persistent form) | flag]
    return mlp_f16_w3x_offset() + align_up((size_t)4 * 2 * 1024 * sizeof(_Float16)) + align_up(sizeof(int));
}
size_t mlp_zplane_bytes(int rows) { return (size_t)((rows + MH_NODES - 1) / MH_NODES) * MP_ZBLOCK; }
size_t mlp_zplane_row_bytes() { return MH_SZ; }

void launch_split_rows(const float *z, int rows, int stride, void *zp, hipStream_t s) {
    const int rp = (rows + MH_NODES - 1) / MH_NODES * MH_NODES;
    const long th = (long)rp * (MH_K0 / 4);
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, z, rows, rp, stride, (char *)zp);
}

// The persistent form: zp = the plane form of the rows (mlp_zplane_bytes(rows) bytes, dvm_mlp_f16.h) -> out [rows][9].  Returns the
// device flag that is non-zero when a weight left fp16's range or an output is not finite (an activation beyond the range
// surfaces there): the results are then invalid and the bf16x3 kernel must overwrite them from the fp32 rows.
int *launch_mlp_planes_f16(const void *zp, int rows, const float *W0, const float *b0, const float *W1, const float *b1, const float *W2,
                           const float *b2, const float *W3, const float *b3, void *scratch, float *out, hipStream_t s) {
    _Float16 *Wp0 = (_Float16 *)scratch;
    _Float16 *Wp1 = Wp0 + (size_t)16 * (MH_K0 / 16) * 1024;
    _Float16 *Wp2 = Wp1 + (size_t)8 * (MH_K1 / 16) * 1024;
    _Float16 *W3x = (_Float16 *)((char *)scratch + mlp_f16_w3x_offset());
    int *flag = (int *)((char *)scratch + mlp_f16_pack_bytes() - align_up(sizeof(int)));
    (void)hipMemsetAsync(flag, 0, sizeof(int), s);
    {
        PackLayersF16 a;
        a.plane_form = 1;
        long maxth = 0;
        auto layer = [&](int q, const float *W, int O, int I, int otiles, int steps, _Float16 *Wp) {
            a.W[q] = W, a.O[q] = O, a.I[q] = I, a.otiles[q] = otiles, a.steps[q] = steps, a.Wp[q] = Wp;
            const long th = (long)otiles * steps * 512;
            maxth = th > maxth ? th : maxth;
        };
        layer(0, W0, 512, 262, 16, MH_K0 / 16, Wp0);
        layer(1, W1, 256, 512, 8, MH_K1 / 16, Wp1);
        layer(2, W2, 128, 256, 4, MH_K2 / 16, Wp2);
        layer(3, W3, 9, 128, 1, MH_K3 / 16, W3x);
        hipLaunchKernelGGL(pack_layers_f16_kernel, dim3((unsigned)((maxth + 255) / 256), 4), dim3(256), 0, s, a, flag);
    }
    const int nblocks = (rows + MH_NODES - 1) / MH_NODES;
    // blocks per workgroup: the next block's planes travel under the current block, so the more the better for THIS kernel — but a
    // workgroup holds its compute unit (146 KB of LDS) until it is through, and whatever else is queued on the device (the next
    // call's coordinate chain on its helper stream) gets a compute unit only when one retires: 4 is the measured optimum (one
    // workgroup per compute unit walking its whole share: fewest cycles, slowest step — 12.0 vs 11.7 ms)
    constexpr int bpw_env = 4;
    const int cus = device_cu_count();
    int bpw = bpw_env > 0 ? bpw_env : (nblocks + cus - 1) / cus;
    if (bpw < 1) bpw = 1;
    const dim3 grid((nblocks + bpw - 1) / bpw), block(MH_THREADS);
    if (options().debug & DVM_DEBUG_MLP_STAMPS) {   // diagnostic: synchronous, allocates — never taken in production
        unsigned long long *dbuf = nullptr;
        const size_t n = (size_t)grid.x * MH_WAVES * 14;
        if (hipMalloc(&dbuf, n * sizeof(unsigned long long)) != hipSuccess) return flag;
        ensure_dyn_lds((const void *)mlp_f16x2p_kernel<4, true>, (int)MP_LDS_BYTES);
        hipLaunchKernelGGL((mlp_f16x2p_kernel<4, true>), grid, block, MP_LDS_BYTES, s, (const char *)zp, rows, nblocks, bpw, Wp0, b0, Wp1, b1, Wp2, b2, W3x, b3, out,
                           flag, dbuf);
        (void)hipStreamSynchronize(s);
        unsigned long long *hbuf = (unsigned long long *)malloc(n * sizeof(unsigned long long));
        (void)hipMemcpy(hbuf, dbuf, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double tot[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t w = 0; w < (size_t)grid.x * MH_WAVES; ++w)
            for (int i = 0; i < 14; ++i) tot[i] += (double)hbuf[w * 14 + i];
        const double nw = (double)nblocks * MH_WAVES;   // per wave and 64-row block
        fprintf(stderr, "MLP stamps, persistent form (%d blocks on %u workgroups), cycles per wave and block: start %.0f | L0a mfma %.0f store+bar %.0f | "
                        "L0b mfma %.0f store+bar+dma %.0f | L1 mfma (both halves) %.0f bars %.0f + %.0f | h1 store+bar %.0f | L2 mfma %.0f "
                        "epilogue+L3+wait+bar %.0f | output %.0f | whole %.0f | shader clock %.0f MHz\n", nblocks, grid.x, tot[0] / nw, tot[1] / nw, tot[2] / nw, tot[4] / nw, tot[5] / nw,
                tot[12] / nw, tot[3] / nw, tot[6] / nw, tot[7] / nw, tot[8] / nw, tot[9] / nw, tot[10] / nw, tot[11] / nw, 100.0 * tot[11] / tot[13]);
        free(hbuf);
        (void)hipFree(dbuf);
        return flag;
    }
    prof_note(DVM_PROF_MLP, "mlp_f16x2p_kernel");
    prof_begin(s, DVM_PROF_MLP);
    ensure_dyn_lds((const void *)mlp_f16x2p_kernel<4>, (int)MP_LDS_BYTES);
    hipLaunchKernelGGL((mlp_f16x2p_kernel<4>), grid, block, MP_LDS_BYTES, s, (const char *)zp, rows, nblocks, bpw, Wp0, b0, Wp1, b1, Wp2, b2, W3x, b3, out, flag,
                       (unsigned long long *)nullptr);
    prof_end(s, DVM_PROF_MLP);
    return flag;
}

}  // namespace dvm
