// dvm_deformer.hip — Deformer.forward on the GPU without the (B,N,k,128) gathers and without the
// dense Pi.   Reference: models/model.py:464-478 (Deformer), 433-452 (MLP); the caller-side
// gathers it replaces are models/loss.py:1254-1257.
//
//   pool   g[i,:]  = sum_s conv_w[s] * feat[idx[i,s],:] + conv_b        (Conv2d k->1, 1x1)
//   xfer   g2'[v,:] = sum_t P[v,t] * g2[pidx[v,t],:]                    (Pi~ @ g2, sparse)
//   z[n]   = [verts1_v, g1_v, verts12_v, g2'_v],  v = fps1[n]           (262 floats)
//   MLP    262 -> 512 -> 256 -> 128 -> 9, ELU between layers.
//
// The MLP is 0.6 GFLOP per direction at N = 2048 — the second largest contraction of the path —
// and runs on the fp32 matrix cores: 32 nodes per workgroup, activations resident in LDS
// (k-deinterleaved so an MFMA A-fragment is one ds_read_b128 per 4 k-steps), weights pre-packed
// into the MFMA B-fragment order so each k-step is one coalesced 256-B load per wave.  The
// accumulation is the same k-ordered fma chain as the oracle's (and torch's CPU addmm).
#include "dvm_common.h"
#include "dvm_mlp_f16.h"

namespace dvm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DF_C = 128;

// pooled[r,:] = sum_s w[s]*feat[idx[row(r),s],:] + bias ; row(r) = rowmap ? rowmap[r] : r.
// order (optional, with rowmap == nullptr): a permutation of the points; thread group t handles point order[t] (and
// writes its row) — in grid-cell order consecutive groups are spatial neighbours, their neighbour lists overlap and the
// gathered rows hit in L1/L2.
__global__ __launch_bounds__(256) void pool_kernel(const float *__restrict__ feat, const int32_t *__restrict__ idx,
                                                   const int32_t *__restrict__ rowmap, int P, int nrows, int k,
                                                   const float *__restrict__ cw, const float *__restrict__ cb,
                                                   float *__restrict__ out, int out_stride, int out_off,
                                                   const int32_t *__restrict__ order) {
    const int b = blockIdx.y;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)nrows * (DF_C / 4)) return;
    int r = (int)(g / (DF_C / 4));
    const int c4 = (int)(g % (DF_C / 4));
    if (order) r = order[(size_t)b * nrows + r];
    const int v = rowmap ? rowmap[(size_t)b * nrows + r] : r;
    const float *fb = feat + (size_t)b * P * DF_C;
    const int32_t *ix = idx + ((size_t)b * P + v) * k;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (k == 10) {   // the Deformer's k: all ten neighbour rows requested before the first is used (same fma order)
        int nb[10];
#pragma unroll
        for (int s = 0; s < 10; ++s) nb[s] = ix[s];
        f32x4 f[10];
#pragma unroll
        for (int s = 0; s < 10; ++s) f[s] = *(const f32x4 *)(fb + (size_t)nb[s] * DF_C + 4 * c4);
#pragma unroll
        for (int s = 0; s < 10; ++s) {
            const float w = cw[s];
            acc.x = fmaf(w, f[s].x, acc.x);
            acc.y = fmaf(w, f[s].y, acc.y);
            acc.z = fmaf(w, f[s].z, acc.z);
            acc.w = fmaf(w, f[s].w, acc.w);
        }
    } else {
        for (int s = 0; s < k; ++s) {
            f32x4 f = *(const f32x4 *)(fb + (size_t)ix[s] * DF_C + 4 * c4);
            float w = cw[s];
            acc.x = fmaf(w, f.x, acc.x);
            acc.y = fmaf(w, f.y, acc.y);
            acc.z = fmaf(w, f.z, acc.z);
            acc.w = fmaf(w, f.w, acc.w);
        }
    }
    const float bias = cb[0];
    float *o = out + ((size_t)b * nrows + r) * out_stride + out_off + 4 * c4;
    if (((out_stride | out_off) & 3) == 0) {   // (kernel-uniform) one 16-byte store per lane: the row goes out as whole lines
        const f32x4 v = {acc.x + bias, acc.y + bias, acc.z + bias, acc.w + bias};
        *(f32x4 *)o = v;
    } else {   // the Deformer's z rows (offset 3): four 4-byte stores at a 16-byte stride
        o[0] = acc.x + bias, o[1] = acc.y + bias, o[2] = acc.z + bias, o[3] = acc.w + bias;
    }
}

// z[n, 0:3] = verts1[v]; z[n,131:134] = verts12[v]; z[n,134:262] = sum_t P[v,t] g2[pidx[v,t],:]
// (ascending column order); z[n,262:ZS] = 0.
constexpr int DF_IN = 262;
constexpr int DF_ZS = 264;  // z row stride (padded to a multiple of 4)

template <int TOPK>
__global__ __launch_bounds__(256) void assemble_kernel(const float *__restrict__ verts1, const float *__restrict__ verts12,
                                                       const float *__restrict__ g2, const float *__restrict__ pi_val,
                                                       const int32_t *__restrict__ pi_idx, const int32_t *__restrict__ fps1,
                                                       int N, int M, int Nn, int topk, float *__restrict__ z) {
    const int b = blockIdx.y;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)Nn * (DF_C / 4)) return;
    const int n = (int)(g / (DF_C / 4)), c4 = (int)(g % (DF_C / 4));
    const int v = fps1[(size_t)b * Nn + n];
    const size_t row = (size_t)b * N + v;
    float pv[TOPK];
    int pc[TOPK];
#pragma unroll
    for (int t = 0; t < TOPK; ++t) {
        const bool live = t < topk;
        const size_t o = row * topk + (live ? t : 0);   // (clamped, then selected: a predicated load is a branch and a wait)
        const float pvt = pi_val[o];
        const int pct = pi_idx[o];
        pv[t] = live ? pvt : 0.f;
        pc[t] = live ? pct : 0x7fffffff;
    }
    // insertion sort by column
#pragma unroll
    for (int a = 1; a < TOPK; ++a) {
#pragma unroll
        for (int p = a; p > 0; --p) {
            bool sw = pc[p] < pc[p - 1];
            int c0 = pc[p - 1], c1 = pc[p];
            float v0 = pv[p - 1], v1 = pv[p];
            pc[p - 1] = sw ? c1 : c0;
            pc[p] = sw ? c0 : c1;
            pv[p - 1] = sw ? v1 : v0;
            pv[p] = sw ? v0 : v1;
        }
    }
    const float *g2b = g2 + (size_t)b * M * DF_C;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (topk == TOPK) {   // (uniform) the usual case: all TOPK rows requested before the first is used (no predicated loads)
        f32x4 f[TOPK];
#pragma unroll
        for (int t = 0; t < TOPK; ++t) f[t] = *(const f32x4 *)(g2b + (size_t)pc[t] * DF_C + 4 * c4);
#pragma unroll
        for (int t = 0; t < TOPK; ++t) {
            acc.x = fmaf(pv[t], f[t].x, acc.x);
            acc.y = fmaf(pv[t], f[t].y, acc.y);
            acc.z = fmaf(pv[t], f[t].z, acc.z);
            acc.w = fmaf(pv[t], f[t].w, acc.w);
        }
    } else {
#pragma unroll
        for (int t = 0; t < TOPK; ++t) {
            if (t < topk) {
                f32x4 f = *(const f32x4 *)(g2b + (size_t)pc[t] * DF_C + 4 * c4);
                acc.x = fmaf(pv[t], f.x, acc.x);
                acc.y = fmaf(pv[t], f.y, acc.y);
                acc.z = fmaf(pv[t], f.z, acc.z);
                acc.w = fmaf(pv[t], f.w, acc.w);
            }
        }
    }
    float *zr = z + ((size_t)b * Nn + n) * DF_ZS;
    zr[134 + 4 * c4] = acc.x, zr[135 + 4 * c4] = acc.y, zr[136 + 4 * c4] = acc.z, zr[137 + 4 * c4] = acc.w;
    if (c4 == 0) {
        for (int c = 0; c < 3; ++c) {
            zr[c] = verts1[row * 3 + c];
            zr[131 + c] = verts12[row * 3 + c];
        }
        zr[262] = 0.f, zr[263] = 0.f;
    }
}

// z row from POOLED features of both clouds (used by the two-direction path, where g(feat, idx) is
// computed once per cloud):  z[n] = [vsrc_v, gsrc[v,:], vcorr_v, sum_t P[v,t] gtgt[pidx[v,t],:]], v = fps[n]
// PLANES: the row goes out in the plane form the persistent MLP kernel stages by LDS-DMA (dvm_mlp_f16.h: scaled by 32, split into
// two fp16 planes, the two 128-wide blocks first) instead of as 264 floats; gate (fp32 form only): the launch does nothing unless
// *gate != 0 — the fp32 rows are needed only by the range fallback of that kernel.
// (round 6: blockIdx.z = direction — the pair path's two directions are one launch)
struct AssembleSide {
    const float *vsrc, *vcorr, *gsrc, *gtgt, *pi_val;
    const int32_t *pi_idx, *fps;
    int N, M, Nn;
    float *z;
};
struct AssembleArgs {
    AssembleSide d[2];
    int topk;
    const int *gate;
};
template <int TOPK, bool PLANES = false>
__global__ __launch_bounds__(256) void assemble_pooled_kernel(const AssembleArgs args) {
    const int *__restrict__ gate = args.gate;
    if (gate && *gate == 0) return;
    const AssembleSide &A = args.d[blockIdx.z];
    const float *__restrict__ vsrc = A.vsrc, *__restrict__ vcorr = A.vcorr, *__restrict__ gsrc = A.gsrc, *__restrict__ gtgt = A.gtgt;
    const float *__restrict__ pi_val = A.pi_val;
    const int32_t *__restrict__ pi_idx = A.pi_idx, *__restrict__ fps = A.fps;
    const int N = A.N, M = A.M, Nn = A.Nn, topk = args.topk;
    float *__restrict__ z = A.z;
    const int b = blockIdx.y;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)Nn * (DF_C / 4)) return;
    const int n = (int)(g / (DF_C / 4)), c4 = (int)(g % (DF_C / 4));
    const int v = fps[(size_t)b * Nn + n];
    const size_t row = (size_t)b * N + v;
    // The correspondences of a node in ascending column order (the reference's sparse sum runs over the columns).  The 32 lanes of
    // a node used to sort the ten (column, weight) pairs each for itself - 45 compare-swaps of five instructions, 225 of the
    // kernel's ~300 vector instructions per lane.  Now lane t holds pair t, ranks it against the others (ten shuffles and
    // compares), sends it to the lane of its rank (ds_permute), and every lane reads the sorted pairs from there.
    const int lane = threadIdx.x & 63, base = lane & 32, l32 = lane & 31;
    const bool live = l32 < topk;
    const size_t o = row * topk + (live ? l32 : 0);   // (clamped, then selected: a predicated load is a branch and a wait)
    const float pvt = pi_val[o];
    const int pct = pi_idx[o];
    const float my_pv = live ? pvt : 0.f;
    const int my_pc = live ? pct : 0x7fffffff;
    int rank = 0;
#pragma unroll
    for (int u = 0; u < TOPK; ++u) {
        const int pcu = __shfl(my_pc, base + u, 64);
        rank += (pcu < my_pc || (pcu == my_pc && u < l32)) ? 1 : 0;   // (unused slots share one key: their order is their position)
    }
    const int dst = l32 < TOPK ? base + rank : lane;   // (ranks of lanes 0 .. TOPK-1 are a permutation of 0 .. TOPK-1)
    const int spc = __builtin_amdgcn_ds_permute(dst << 2, my_pc);
    const float spv = __int_as_float(__builtin_amdgcn_ds_permute(dst << 2, __float_as_int(my_pv)));
    float pv[TOPK];
    int pc[TOPK];
#pragma unroll
    for (int t = 0; t < TOPK; ++t) {
        pc[t] = __shfl(spc, base + t, 64);
        pv[t] = __shfl(spv, base + t, 64);
    }
    const float *gt = gtgt + (size_t)b * M * DF_C;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (topk == TOPK) {   // (uniform) the usual case: all TOPK rows requested before the first is used (no predicated loads)
        f32x4 f[TOPK];
#pragma unroll
        for (int t = 0; t < TOPK; ++t) f[t] = *(const f32x4 *)(gt + (size_t)pc[t] * DF_C + 4 * c4);
#pragma unroll
        for (int t = 0; t < TOPK; ++t) {
            acc.x = fmaf(pv[t], f[t].x, acc.x);
            acc.y = fmaf(pv[t], f[t].y, acc.y);
            acc.z = fmaf(pv[t], f[t].z, acc.z);
            acc.w = fmaf(pv[t], f[t].w, acc.w);
        }
    } else {
#pragma unroll
        for (int t = 0; t < TOPK; ++t) {
            if (t < topk) {
                f32x4 f = *(const f32x4 *)(gt + (size_t)pc[t] * DF_C + 4 * c4);
                acc.x = fmaf(pv[t], f.x, acc.x);
                acc.y = fmaf(pv[t], f.y, acc.y);
                acc.z = fmaf(pv[t], f.z, acc.z);
                acc.w = fmaf(pv[t], f.w, acc.w);
            }
        }
    }
    f32x4 gs = *(const f32x4 *)(gsrc + row * DF_C + 4 * c4);
    if (PLANES) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        char *zr = (char *)z + ((size_t)b * Nn + n) * MH_SZ;
        unsigned h0, m0, h1, m1;
        split2x2(gs.x * MH_SA, gs.y * MH_SA, h0, m0);
        split2x2(gs.z * MH_SA, gs.w * MH_SA, h1, m1);
        *(u32x2 *)(zr + 8 * c4) = u32x2{h0, h1};
        *(u32x2 *)(zr + 2 * MH_K0 + 8 * c4) = u32x2{m0, m1};
        split2x2(acc.x * MH_SA, acc.y * MH_SA, h0, m0);
        split2x2(acc.z * MH_SA, acc.w * MH_SA, h1, m1);
        *(u32x2 *)(zr + 256 + 8 * c4) = u32x2{h0, h1};
        *(u32x2 *)(zr + 2 * MH_K0 + 256 + 8 * c4) = u32x2{m0, m1};
        if (c4 == 0) {   // plane columns 256..261 = the two points, 262..271 = 0
            unsigned h2, m2;
            split2x2(vsrc[row * 3] * MH_SA, vsrc[row * 3 + 1] * MH_SA, h0, m0);
            split2x2(vsrc[row * 3 + 2] * MH_SA, vcorr[row * 3] * MH_SA, h1, m1);
            split2x2(vcorr[row * 3 + 1] * MH_SA, vcorr[row * 3 + 2] * MH_SA, h2, m2);
            *(u32x4 *)(zr + 512) = u32x4{h0, h1, h2, 0u};
            *(u32x4 *)(zr + 528) = u32x4{0u, 0u, 0u, 0u};
            *(u32x4 *)(zr + 2 * MH_K0 + 512) = u32x4{m0, m1, m2, 0u};
            *(u32x4 *)(zr + 2 * MH_K0 + 528) = u32x4{0u, 0u, 0u, 0u};
        }
        return;
    }
    float *zr = z + ((size_t)b * Nn + n) * DF_ZS;
    zr[3 + 4 * c4] = gs.x, zr[4 + 4 * c4] = gs.y, zr[5 + 4 * c4] = gs.z, zr[6 + 4 * c4] = gs.w;
    zr[134 + 4 * c4] = acc.x, zr[135 + 4 * c4] = acc.y, zr[136 + 4 * c4] = acc.z, zr[137 + 4 * c4] = acc.w;
    if (c4 == 0) {
        for (int c = 0; c < 3; ++c) {
            zr[c] = vsrc[row * 3 + c];
            zr[131 + c] = vcorr[row * 3 + c];
        }
        zr[262] = 0.f, zr[263] = 0.f;
    }
}

// ELU as torch evaluates it on fp32: exp(x) - 1 (not expm1); v_exp_f32 keeps the epilogue off the
// fp32 ALU that the matrix instructions of this kernel also run on
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : __builtin_amdgcn_exp2f(x * 1.4426950408889634f) - 1.f; }

// ---------------------------------------------------------------- scalar MLP layer (check variant)
__global__ void mlp_layer_scalar_kernel(const float *__restrict__ in, int in_stride, const float *__restrict__ W,
                                        const float *__restrict__ bias, int rows, int I, int O, int act,
                                        float *__restrict__ out, int out_stride) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)rows * O) return;
    int r = (int)(g / O), o = (int)(g % O);
    const float *x = in + (size_t)r * in_stride, *w = W + (size_t)o * I;
    float acc = 0.f;
    for (int c = 0; c < I; ++c) acc = fmaf(x[c], w[c], acc);
    acc = acc + bias[o];
    out[(size_t)r * out_stride + o] = act ? elu1(acc) : acc;
}

// ---------------------------------------------------------------- MFMA MLP
// Packed weights: Wp[otile][step][lane] = W[otile*32 + (lane&31)][2*step + (lane>>5)] (0 outside).
__global__ void pack_weights_kernel(const float *__restrict__ W, int O, int I, int otiles, int steps, float *__restrict__ Wp) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)otiles * steps * 64;
    if (g >= total) return;
    int lane = (int)(g % 64);
    int step = (int)((g / 64) % steps);
    int ot = (int)(g / (64L * steps));
    int o = ot * 32 + (lane & 31), c = 2 * step + (lane >> 5);
    Wp[g] = (o < O && c < I) ? W[(size_t)o * I + c] : 0.f;
}

constexpr int ML_NODES = 32;
constexpr int ML_WAVES = 8;
constexpr int ML_THREADS = 64 * ML_WAVES;
// activation buffers, k-deinterleaved: element (node, c) at node*stride + (c&1)*half + (c>>1)
constexpr int ML_SA = 268;  // holds z (264 -> halves of 132) and h1 (256 -> halves of 128)
constexpr int ML_SB = 516;  // holds h0 (512 -> halves of 256) and h2 (128 -> halves of 64)
constexpr size_t ML_LDS_BYTES = (size_t)ML_NODES * (ML_SA + ML_SB) * sizeof(float);

// one 32x32 output tile: acc[node][out] = sum_k act[node][k] * W[out][k], K = 2*steps (steps % 4 == 0)
__device__ __forceinline__ f32x16 mlp_tile(const float *__restrict__ act_row /* lane's node row + h*half */,
                                           const float *__restrict__ wp /* + lane */, int steps) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // The chain is dependent (one MFMA per 64 cycles), so the operands of the NEXT 8 k-steps are
    // fetched (weights: 8 coalesced 256-B loads from L2; activations: 2 ds_read_b128) while the
    // current 8 MFMAs run.  steps % 4 == 0; a trailing group of 4 is handled after the loop.
    float b[8], bn[8];
    f32x4 a0, a1, an0, an1;
    const int full = steps & ~7;
#pragma unroll
    for (int u = 0; u < 8; ++u) b[u] = (u < steps) ? wp[(size_t)u * 64] : 0.f;
    a0 = *(const f32x4 *)(act_row);
    a1 = (steps > 4) ? *(const f32x4 *)(act_row + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < full; s += 8) {
        const int sn = s + 8;
#pragma unroll
        for (int u = 0; u < 8; ++u) bn[u] = (sn + u < steps) ? wp[(size_t)(sn + u) * 64] : 0.f;
        an0 = (sn < steps) ? *(const f32x4 *)(act_row + sn) : f32x4{0.f, 0.f, 0.f, 0.f};
        an1 = (sn + 4 < steps) ? *(const f32x4 *)(act_row + sn + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b[3], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b[4], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b[5], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b[6], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b[7], acc, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = bn[u];
        a0 = an0;
        a1 = an1;
    }
    if (steps & 4) {  // trailing 4 steps (operands already loaded)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b[3], acc, 0, 0, 0);
    }
    return acc;
}

// write a finished tile (bias + optional ELU) into the next activation buffer
__device__ __forceinline__ void mlp_store(const f32x16 &acc, const float *__restrict__ bias, int o, int O, bool act,
                                          float *__restrict__ dst, int stride, int half, int h) {
    const float bv = o < O ? bias[o] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int node = (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[r] + bv;
        v = act ? elu1(v) : v;
        if (o < O) dst[node * stride + (o & 1) * half + (o >> 1)] = v;
    }
}

__global__ __launch_bounds__(ML_THREADS) void mlp_mfma_kernel(const float *__restrict__ z, int rows,
                                                              const float *__restrict__ Wp0, const float *__restrict__ b0,
                                                              const float *__restrict__ Wp1, const float *__restrict__ b1,
                                                              const float *__restrict__ Wp2, const float *__restrict__ b2,
                                                              const float *__restrict__ Wp3, const float *__restrict__ b3,
                                                              float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *bufA = smem;                      // [32][ML_SA]
    float *bufB = smem + ML_NODES * ML_SA;   // [32][ML_SB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * ML_NODES;

    // stage z (de-interleave k): 32 rows x 264 floats
    for (int e = tid; e < ML_NODES * (DF_ZS / 4); e += ML_THREADS) {
        int r = e / (DF_ZS / 4), c = e % (DF_ZS / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row0 + r < rows) v = *(const f32x4 *)(z + (size_t)(row0 + r) * DF_ZS + 4 * c);
        float2 ev = {v.x, v.z}, od = {v.y, v.w};
        *(float2 *)(bufA + r * ML_SA + 2 * c) = ev;
        *(float2 *)(bufA + r * ML_SA + 132 + 2 * c) = od;
    }
    __syncthreads();
    // layer 0: 264(262) -> 512 : 16 output tiles, 2 per wave; steps = 132
    for (int q = 0; q < 2; ++q) {
        int ot = wave * 2 + q;
        f32x16 acc = mlp_tile(bufA + r32 * ML_SA + h * 132, Wp0 + (size_t)ot * 132 * 64 + lane, 132);
        mlp_store(acc, b0, ot * 32 + r32, 512, true, bufB, ML_SB, 256, h);
    }
    __syncthreads();
    // layer 1: 512 -> 256 : 8 tiles, 1 per wave; steps = 256
    {
        int ot = wave;
        f32x16 acc = mlp_tile(bufB + r32 * ML_SB + h * 256, Wp1 + (size_t)ot * 256 * 64 + lane, 256);
        mlp_store(acc, b1, ot * 32 + r32, 256, true, bufA, ML_SA, 128, h);
    }
    __syncthreads();
    // layer 2: 256 -> 128 : 4 tiles; steps = 128
    if (wave < 4) {
        int ot = wave;
        f32x16 acc = mlp_tile(bufA + r32 * ML_SA + h * 128, Wp2 + (size_t)ot * 128 * 64 + lane, 128);
        mlp_store(acc, b2, ot * 32 + r32, 128, true, bufB, ML_SB, 64, h);
    }
    __syncthreads();
    // layer 3: 128 -> 9 : one tile; steps = 64; straight to HBM
    if (wave == 0) {
        f32x16 acc = mlp_tile(bufB + r32 * ML_SB + h * 64, Wp3 + lane, 64);
        const int o = r32;
        const float bv = o < 9 ? b3[o] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int node = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (o < 9 && row0 + node < rows) out[(size_t)(row0 + node) * 9 + o] = acc[r] + bv;
        }
    }
}

struct DeformerWs {
    float *g2, *z, *Wp0, *Wp1, *Wp2, *Wp3, *h0, *h1, *h2;
    char *zp;   // the rows in the plane form (input of the persistent MLP kernel)
};

size_t mlp_pack_floats();
void launch_mlp_rows(const float *z, int rows, const float *W0, const float *b0, const float *W1, const float *b1, const float *W2,
                     const float *b2, const float *W3, const float *b3, float *wp, float *out, hipStream_t s, int variant, void *zp);

static size_t carve(Arena &ar, int B, int M, int Nn, DeformerWs &w) {
    w.g2 = ar.take<float>((size_t)B * M * DF_C);
    w.z = ar.take<float>((size_t)B * Nn * DF_ZS);
    w.Wp0 = ar.take<float>(mlp_pack_floats());  // packed weights of all layers (either kernel's format)
    w.Wp1 = offset_ptr(w.Wp0, (size_t)16 * 132 * 64);
    w.Wp2 = offset_ptr(w.Wp1, (size_t)8 * 256 * 64);
    w.Wp3 = offset_ptr(w.Wp2, (size_t)4 * 128 * 64);
    w.h0 = ar.take<float>((size_t)B * Nn * 512);
    w.h1 = ar.take<float>((size_t)B * Nn * 256);
    w.h2 = ar.take<float>((size_t)B * Nn * 128);
    w.zp = ar.take<char>(mlp_zplane_bytes(B * Nn));
    return ar.off;
}

int launch_deformer(const float *feat1, const float *feat2, const float *verts1, const float *verts12, const int32_t *idx11,
                    const int32_t *idx22, const float *pi_val, const int32_t *pi_idx, const int32_t *fps1, int B, int N,
                    int M, int Nn, int k, int topk, const float *conv_w, const float *conv_b, const float *W0,
                    const float *b0, const float *W1, const float *b1, const float *W2, const float *b2, const float *W3,
                    const float *b3, float *out, int variant, void *ws, size_t ws_bytes, hipStream_t s) {
    Arena ar(ws, ws_bytes);
    DeformerWs w;
    carve(ar, B, M, Nn, w);
    if (!ar.ok()) {
        set_error("dvm_deformer_fwd_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    const int rows = B * Nn;
    // g2 for every target point; g1 only at the graph nodes (written straight into z[:,3:131])
    hipLaunchKernelGGL(pool_kernel, dim3((unsigned)(((long)M * 32 + 255) / 256), B), dim3(256), 0, s, feat2, idx22,
                       (const int32_t *)nullptr, M, M, k, conv_w, conv_b, w.g2, DF_C, 0, (const int32_t *)nullptr);
    hipLaunchKernelGGL(pool_kernel, dim3((unsigned)(((long)Nn * 32 + 255) / 256), B), dim3(256), 0, s, feat1, idx11, fps1, N,
                       Nn, k, conv_w, conv_b, w.z, DF_ZS, 3, (const int32_t *)nullptr);
    if (topk <= 10)
        hipLaunchKernelGGL(assemble_kernel<10>, dim3((unsigned)(((long)Nn * 32 + 255) / 256), B), dim3(256), 0, s, verts1,
                           verts12, w.g2, pi_val, pi_idx, fps1, N, M, Nn, topk, w.z);
    else
        hipLaunchKernelGGL(assemble_kernel<16>, dim3((unsigned)(((long)Nn * 32 + 255) / 256), B), dim3(256), 0, s, verts1,
                           verts12, w.g2, pi_val, pi_idx, fps1, N, M, Nn, topk, w.z);
    if (variant == 1) {
        auto layer = [&](const float *in, int is, const float *W, const float *bb, int I, int O, int act, float *o, int os) {
            long th = (long)rows * O;
            hipLaunchKernelGGL(mlp_layer_scalar_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, in, is, W, bb, rows,
                               I, O, act, o, os);
        };
        layer(w.z, DF_ZS, W0, b0, DF_IN, 512, 1, w.h0, 512);
        layer(w.h0, 512, W1, b1, 512, 256, 1, w.h1, 256);
        layer(w.h1, 256, W2, b2, 256, 128, 1, w.h2, 128);
        layer(w.h2, 128, W3, b3, 128, 9, 0, out, 9);
    } else {
        launch_mlp_rows(w.z, rows, W0, b0, W1, b1, W2, b2, W3, b3, w.Wp0, out, s, variant, w.zp);
    }
    return DVM_OK;
}

__global__ void pad_rows_kernel(const float *__restrict__ in, int rows, int I, int stride, float *__restrict__ out) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)rows * stride) return;
    int r = (int)(g / stride), c = (int)(g % stride);
    out[g] = c < I ? in[(size_t)r * I + c] : 0.f;
}

void launch_pool_all(const float *feat, const int32_t *idx, int B, int P, int k, const float *cw, const float *cb, float *out,
                     hipStream_t s, const int32_t *order) {
    prof_begin(s, DVM_PROF_POOL);
    hipLaunchKernelGGL(pool_kernel, dim3((unsigned)(((long)P * 32 + 255) / 256), B), dim3(256), 0, s, feat, idx,
                       (const int32_t *)nullptr, P, P, k, cw, cb, out, DF_C, 0, order);
    prof_end(s, DVM_PROF_POOL);
}
void launch_assemble_pooled(const float *vsrc, const float *vcorr, const float *gsrc, const float *gtgt, const float *pi_val,
                            const int32_t *pi_idx, const int32_t *fps, int B, int N, int M, int Nn, float *z, hipStream_t s, const int *gate) {
    AssembleArgs a;
    a.d[0] = a.d[1] = AssembleSide{vsrc, vcorr, gsrc, gtgt, pi_val, pi_idx, fps, N, M, Nn, z};
    a.topk = 10, a.gate = gate;
    if (!gate) prof_begin(s, DVM_PROF_ASSEMBLE);
    hipLaunchKernelGGL(assemble_pooled_kernel<10>, dim3((unsigned)(((long)Nn * 32 + 255) / 256), B, 1), dim3(256), 0, s, a);
    if (!gate) prof_end(s, DVM_PROF_ASSEMBLE);
}
// the rows in the plane form (zp: row (b, n) at ((b Nn + n) MH_SZ) bytes)
void launch_assemble_pooled_planes(const float *vsrc, const float *vcorr, const float *gsrc, const float *gtgt, const float *pi_val,
                                   const int32_t *pi_idx, const int32_t *fps, int B, int N, int M, int Nn, void *zp, hipStream_t s) {
    AssembleArgs a;
    a.d[0] = a.d[1] = AssembleSide{vsrc, vcorr, gsrc, gtgt, pi_val, pi_idx, fps, N, M, Nn, (float *)zp};
    a.topk = 10, a.gate = nullptr;
    prof_begin(s, DVM_PROF_ASSEMBLE);
    hipLaunchKernelGGL((assemble_pooled_kernel<10, true>), dim3((unsigned)(((long)Nn * 32 + 255) / 256), B, 1), dim3(256), 0, s, a);
    prof_end(s, DVM_PROF_ASSEMBLE);
}
// both directions of the pair path in one launch: side 0 = (clouds 1 -> 2), side 1 = (2 -> 1); planes: the plane form (zp*), else the
// fp32 rows behind `gate`
void launch_assemble_pooled_pair(const float *verts1, const float *verts2, const float *verts12, const float *verts21, const float *g1,
                                 const float *g2, const float *val12, const int32_t *idx12, const float *val21, const int32_t *idx21,
                                 const int32_t *nodes1, const int32_t *nodes2, int B, int N, int M, void *z12, void *z21, bool planes,
                                 const int *gate, hipStream_t s) {
    AssembleArgs a;
    a.d[0] = AssembleSide{verts1, verts12, g1, g2, val12, idx12, nodes1, N, M, N / 2, (float *)z12};
    a.d[1] = AssembleSide{verts2, verts21, g2, g1, val21, idx21, nodes2, M, N, M / 2, (float *)z21};
    a.topk = 10, a.gate = gate;
    const int nn = N / 2 > M / 2 ? N / 2 : M / 2;
    const dim3 grid((unsigned)(((long)nn * 32 + 255) / 256), B, 2);
    if (!gate) prof_begin(s, DVM_PROF_ASSEMBLE);
    if (planes) hipLaunchKernelGGL((assemble_pooled_kernel<10, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(assemble_pooled_kernel<10>, grid, dim3(256), 0, s, a);
    if (!gate) prof_end(s, DVM_PROF_ASSEMBLE);
}
size_t mlp_bf16_pack_bytes();
void launch_mlp_rows_bf16(const float *z, int rows, const float *W0, const float *b0, const float *W1, const float *b1,
                          const float *W2, const float *b2, const float *W3, const float *b3, void *scratch, float *out,
                          hipStream_t s, const int *gate);
size_t mlp_f16_pack_bytes();
void launch_split_rows(const float *z, int rows, int stride, void *zp, hipStream_t s);
int *launch_mlp_planes_f16(const void *zp, int rows, const float *W0, const float *b0, const float *W1, const float *b1, const float *W2,
                           const float *b2, const float *W3, const float *b3, void *scratch, float *out, hipStream_t s);
// variant 0 on rows that are in the plane form already; returns the range flag for launch_mlp_fallback
const int *launch_mlp_planes(const void *zp, int rows, const float *W0, const float *b0, const float *W1, const float *b1, const float *W2,
                             const float *b2, const float *W3, const float *b3, float *wp, float *out, hipStream_t s) {
    return launch_mlp_planes_f16(zp, rows, W0, b0, W1, b1, W2, b2, W3, b3, wp, out, s);
}
// the bf16x3 kernel on the fp32 rows, gated on the flag
void launch_mlp_fallback(const float *z, int rows, const float *W0, const float *b0, const float *W1, const float *b1, const float *W2,
                         const float *b2, const float *W3, const float *b3, float *wp, float *out, hipStream_t s, const int *flag) {
    launch_mlp_rows_bf16(z, rows, W0, b0, W1, b1, W2, b2, W3, b3, (char *)wp + mlp_f16_pack_bytes(), out, s, flag);
}
size_t mlp_pack_floats() {
    size_t f32 = (size_t)16 * 132 * 64 + (size_t)8 * 256 * 64 + (size_t)4 * 128 * 64 + (size_t)64 * 64;
    size_t h16 = (mlp_f16_pack_bytes() + mlp_bf16_pack_bytes() + 3) / 4;  // variant 0 keeps both packings
    return f32 > h16 ? f32 : h16;
}
// z [rows][264] -> out [rows][9]; wp = scratch of mlp_pack_floats() floats.
// variant 0: fp16x2-split matrix-core kernel, 64 nodes per workgroup (dvm_mlp_f16.hip), followed by the bf16x3 kernel
//            gated on its out-of-range flag; 3: bf16x3-split kernel (dvm_mlp_bf16.hip); 2: fp32-MFMA kernel
void launch_mlp_rows(const float *z, int rows, const float *W0, const float *b0, const float *W1, const float *b1, const float *W2,
                     const float *b2, const float *W3, const float *b3, float *wp, float *out, hipStream_t s, int variant, void *zp) {
    if (variant == 0 && zp) {
        launch_split_rows(z, rows, DF_ZS, zp, s);
        const int *flag = launch_mlp_planes_f16(zp, rows, W0, b0, W1, b1, W2, b2, W3, b3, wp, out, s);
        launch_mlp_rows_bf16(z, rows, W0, b0, W1, b1, W2, b2, W3, b3, (char *)wp + mlp_f16_pack_bytes(), out, s, flag);
        return;
    }
    if (variant == 0) variant = 3;   // (no plane buffer: the range-safe bf16x3 kernel)
    if (variant == 3) {
        launch_mlp_rows_bf16(z, rows, W0, b0, W1, b1, W2, b2, W3, b3, wp, out, s, nullptr);
        return;
    }
    float *Wp0 = wp, *Wp1 = Wp0 + (size_t)16 * 132 * 64, *Wp2 = Wp1 + (size_t)8 * 256 * 64, *Wp3 = Wp2 + (size_t)4 * 128 * 64;
    auto pack = [&](const float *W, int O, int I, int otiles, int steps, float *Wp) {
        long th = (long)otiles * steps * 64;
        hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, W, O, I, otiles, steps, Wp);
    };
    pack(W0, 512, DF_IN, 16, 132, Wp0);
    pack(W1, 256, 512, 8, 256, Wp1);
    pack(W2, 128, 256, 4, 128, Wp2);
    pack(W3, 9, 128, 1, 64, Wp3);
    ensure_dyn_lds((const void *)mlp_mfma_kernel, (int)ML_LDS_BYTES);
    hipLaunchKernelGGL(mlp_mfma_kernel, dim3((rows + ML_NODES - 1) / ML_NODES), dim3(ML_THREADS), ML_LDS_BYTES, s, z, rows, Wp0, b0,
                       Wp1, b1, Wp2, b2, Wp3, b3, out);
}

size_t deformer_ws_bytes(int B, int M, int Nn) {
    Arena ar(nullptr, 0);
    DeformerWs w;
    return carve(ar, B, M, Nn, w);
}

}  // namespace dvm

using namespace dvm;

DVM_EXPORT size_t dvm_deformer_workspace_bytes(int B, int N, int M, int Nn) {
    (void)N;
    return deformer_ws_bytes(B, M, Nn);
}

DVM_EXPORT int dvm_deformer_fwd_f32(const float *feat1, const float *feat2, const float *verts1, const float *verts12,
                                    const int32_t *idx11, const int32_t *idx22, const float *pi_val, const int32_t *pi_idx,
                                    const int32_t *fps1, int B, int N, int M, int Nn, int k, int topk, const float *conv_w,
                                    const float *conv_b, const float *W0, const float *b0, const float *W1, const float *b1,
                                    const float *W2, const float *b2, const float *W3, const float *b3, float *out,
                                    int variant, void *ws, size_t ws_bytes, void *stream) {
    DVM_REQUIRE(feat1 && feat2 && verts1 && verts12 && idx11 && idx22 && pi_val && pi_idx && fps1 && out,
                "dvm_deformer_fwd_f32: null tensor pointer");
    DVM_REQUIRE(conv_w && conv_b && W0 && b0 && W1 && b1 && W2 && b2 && W3 && b3, "dvm_deformer_fwd_f32: null weight pointer");
    DVM_REQUIRE(B >= 1 && N >= 1 && M >= 1 && Nn >= 1, "dvm_deformer_fwd_f32: empty input");
    DVM_REQUIRE(k >= 1 && k <= 64 && topk >= 1 && topk <= 16, "dvm_deformer_fwd_f32: k=%d topk=%d out of range", k, topk);
    DVM_REQUIRE(variant >= 0 && variant <= 3, "dvm_deformer_fwd_f32: bad variant %d", variant);
    int rc = launch_deformer(feat1, feat2, verts1, verts12, idx11, idx22, pi_val, pi_idx, fps1, B, N, M, Nn, k, topk, conv_w,
                             conv_b, W0, b0, W1, b1, W2, b2, W3, b3, out, variant, ws, ws_bytes, (hipStream_t)stream);
    if (rc != DVM_OK) return rc;
    DVM_CHECK_LAUNCH("deformer");
    return DVM_OK;
}

/* Deformer's decoder alone (reference MLP, models/model.py:433-452, as used at 476-477):
 * z [rows,262] -> out [rows,9]. */
DVM_EXPORT size_t dvm_deformer_mlp_workspace_bytes(int rows) { return deformer_ws_bytes(1, 1, rows); }

DVM_EXPORT int dvm_deformer_mlp_fwd_f32(const float *z, int rows, const float *W0, const float *b0, const float *W1,
                                        const float *b1, const float *W2, const float *b2, const float *W3, const float *b3,
                                        float *out, void *ws, size_t ws_bytes, void *stream) {
    DVM_REQUIRE(z && out && rows >= 1, "dvm_deformer_mlp_fwd_f32: bad arguments");
    DVM_REQUIRE(W0 && b0 && W1 && b1 && W2 && b2 && W3 && b3, "dvm_deformer_mlp_fwd_f32: null weight pointer");
    Arena ar(ws, ws_bytes);
    DeformerWs w;
    carve(ar, 1, 1, rows, w);
    if (!ar.ok()) {
        set_error("dvm_deformer_mlp_fwd_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    long th = (long)rows * DF_ZS;
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, z, rows, DF_IN, DF_ZS, w.z);
    launch_mlp_rows(w.z, rows, W0, b0, W1, b1, W2, b2, W3, b3, w.Wp0, out, s, 0, w.zp);
    DVM_CHECK_LAUNCH("deformer_mlp");
    return DVM_OK;
}
