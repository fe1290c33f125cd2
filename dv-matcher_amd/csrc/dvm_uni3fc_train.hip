// dvm_uni3fc_train.hip — LG-Net's TRAINING forward and backward (`Uni3FC.forward` under autograd in train mode, reference
// models/model.py:680-761 with the conv blocks 506-529, N2PAttention[_DIM] 325-395 and SA_Layer 97-123; `loss.backward()` of
// train.py:110) behind two C-ABI calls: dvm_uni3fc_train_fwd_f32 / dvm_uni3fc_train_bwd_f32.
//
// Nothing new is computed here: every layer is one of the library's own launches — dvm_linear_f32 (the reference's fp32 chain)
// forward and as dX = dY W, dvm_linear_wgrad_f32, the fused training BatchNorm pair dvm_bn_act_train_{fwd,bwd}_pm_f32, the
// kNN / N2P / SA cores and their backward twins — in the order of dv-matcher_amd/models/model.py::Uni3FC._forward_train_pm and of
// the autograd graph that method records.  That Python path enqueues ~450 forward and ~450 backward launches through ~1500
// Python / autograd hops per call and is host-bound (21.8 ms of host time per 22.3-ms step at B = 8, N = 2048); here the same
// launches are enqueued natively.  Activations the backward needs are kept in a caller-provided arena (laid out by carve());
// parameter gradients are ADDED into caller-provided buffers (the flat gradient bucket: autograd's `p.grad += g` without
// a launch per parameter).  The local (kNN attention) and global (self-attention) chains run on two streams in both passes
// when the caller's stream has a context from dvm_pair_init.
//
// Differences from the autograd path that do not change the mathematics: SA_Layer's q/k and v projections are two GEMMs
// instead of one over stacked weights (same per-column chains: bit-identical forward), and their input gradients are summed
// through the GEMM's residual operand instead of one contraction over 80 columns (fp32 rounding order only).
#include "dvm_uni3fc_kernels.h"

namespace dvm {

void launch_linear(const float *x, const float *w, int B, int N, int K, int Co, int channel_major, const float *bias,
                   const float *res, const float *alpha, const float *beta, float slope, float *y, hipStream_t s, const float *xg,
                   int Cg, const float *post_res, float post_scale);   // dvm_gemm.hip

// dvm_bn.hip
int launch_bn_running_update(float *const *rm, float *const *rv, const float *const *mean, const float *const *var, const int *C, int count,
                             float momentum, hipStream_t s);

namespace {

// ---- parameter table (include/dvm.h, dvm_uni3fc_train_fwd_f32): raw trainable tensors + BatchNorm running statistics
enum { TC_W = 0, TC_G, TC_B, TC_RM, TC_RV, TC_N };                                                       // conv block
enum { TS_WK = 0, TS_WV, TS_BV, TS_WT, TS_BT, TS_G, TS_B, TS_RM, TS_RV, TS_N };                          // SA_Layer
enum { TN_WQ = 0, TN_WK, TN_WV, TN_G1, TN_B1, TN_RM1, TN_RV1, TN_FF0, TN_FF2, TN_G2, TN_B2, TN_RM2, TN_RV2, TN_N };   // N2P block
constexpr int T_CONV0 = 0;
constexpr int T_SA0 = T_CONV0 + 8 * TC_N;
constexpr int T_NP0 = T_SA0 + 4 * TS_N;
constexpr int T_TOTAL = T_NP0 + 7 * TN_N;
static_assert(T_TOTAL == DVM_U3_TRAIN_NPARAMS, "parameter table layout and include/dvm.h disagree");

constexpr int CONV_K[8] = {1152, 384, 256, 256, 768, 768, 256, 512};
constexpr int CONV_CO[8] = {384, 64, 512, 512, 128, 128, 128, 128};
constexpr int NP_C[7] = {64, 64, 64, 64, 128, 128, 128};

constexpr int MAXG = 2;   // groups (merged network calls) per native call
struct BnSave {
    float *mean, *invstd, *var;   // var: unbiased batch variance (for a deferred running-statistics update)
};
struct ConvSave {   // y = act(bn(z)), z = x W^T
    float *z, *y;
    BnSave bn;
};
struct NpSave {
    int32_t *idx;
    float *qkv, *att, *attn, *x1, *h, *ffo, *out;
    BnSave bn1, bn2;
};
struct SaSave {
    float *p, *v, *xr, *stats, *cinv, *d, *t, *y, *out;
    BnSave bn;
};
struct ChainScratch {   // per stream: library workspaces + gradient ping-pong buffers of the backward
    void *bnws, *knnws, *saws, *sabws, *npbws, *wgws;
    size_t bn_bytes, knn_bytes, sa_bytes, sab_bytes, npb_bytes, wg_bytes;
    void *syncbuf;         // cross-rank BatchNorm totals of one layer (dvm_bn_pm_sync_bytes(512, MAXG)): the caller's collective works in place
    float *colpart;        // colsum_accum partials
    unsigned *counter;
    float *g768, *g512a, *g512b, *g256a, *g256b, *ga, *gb, *gc, *gd, *ge, *dqkv, *dh, *dp;   // gradient scratch (backward)
    float *dmax;
};
struct TrainWs {
    // saved by the forward
    float *pe, *pews, *fpe, *pesync;
    ConvSave cv[8];     // cv[1].y is the caller's `tmp`, cv[7].y the caller's `feat`
    NpSave np[7];
    SaSave sa[4];
    float *loc, *catL, *glo, *catG, *mxL, *mxG, *ycat2, *ycat4;
    int32_t *argL, *argG;
    unsigned long long *packL, *packG;
    float *wqkv[7];     // stacked q|k|v weights (when the three parameters are not already contiguous)
    float *dwqkv[7];    // stacked q|k|v weight gradients (same condition), zeroed at the start of a backward
    size_t dwqkv_off, dwqkv_bytes;
    ChainScratch cs[2]; // [0] caller's stream (local chain, trunk), [1] helper stream (global chain)
};

void carve_chain(Arena &ar, int B, int N, int K, ChainScratch &c, bool local) {
    const size_t R = (size_t)B * N;
    c.bn_bytes = 0;
    for (int C : {64, 128, 384, 512})
        for (int G : {1, MAXG}) {   // (a group of a merged call has its own, shorter reduction)
            const size_t b = dvm_bn_pm_groups_workspace_bytes((long)R / G > 0 ? (long)R / G : 1, C, G);
            c.bn_bytes = b > c.bn_bytes ? b : c.bn_bytes;
        }
    c.bnws = ar.take<char>(c.bn_bytes);
    c.syncbuf = ar.take<char>(dvm_bn_pm_sync_bytes(512, MAXG));
    c.knn_bytes = local ? dvm_knn_neg_workspace_bytes(B, N, N, 128, K) : 0;
    c.knnws = local ? (void *)ar.take<char>(c.knn_bytes) : nullptr;
    c.sa_bytes = local ? 0 : dvm_sa_attention_train_fwd_workspace_bytes(B, N);
    c.saws = c.sa_bytes ? (void *)ar.take<char>(c.sa_bytes) : nullptr;
    c.sab_bytes = local ? 0 : dvm_sa_attention_bwd_workspace_bytes(B, N);
    c.sabws = c.sab_bytes ? (void *)ar.take<char>(c.sab_bytes) : nullptr;
    c.npb_bytes = local ? dvm_n2p_core_bwd_workspace_bytes(B, N, K) : 0;
    c.npbws = local ? (void *)ar.take<char>(c.npb_bytes) : nullptr;
    c.wg_bytes = 0;   // per-chunk partial tiles of the weight gradient (deterministic mode): the largest layer of the chain
    for (int i = 0; i < 8; ++i) {
        const size_t nb = dvm_linear_wgrad_workspace_bytes((long)R, CONV_CO[i], CONV_K[i]);
        c.wg_bytes = nb > c.wg_bytes ? nb : c.wg_bytes;
    }
    for (int C : {64, 128})
        for (size_t nb : {dvm_linear_wgrad_workspace_bytes((long)R, 3 * C, C), dvm_linear_wgrad_workspace_bytes((long)R, 4 * C, C),
                          dvm_linear_wgrad_workspace_bytes((long)R, C, 4 * C)})
            c.wg_bytes = nb > c.wg_bytes ? nb : c.wg_bytes;
    c.wgws = ar.take<char>(c.wg_bytes);
    c.colpart = ar.take<float>(1024 * 128);
    c.counter = ar.take<unsigned>(64);
    c.g768 = ar.take<float>(R * 768);
    c.g512a = ar.take<float>(R * 512);
    c.g512b = ar.take<float>(R * 512);
    c.g256a = ar.take<float>(R * 256);
    c.g256b = ar.take<float>(R * 256);
    c.ga = ar.take<float>(R * 128);
    c.gb = ar.take<float>(R * 128);
    c.gc = ar.take<float>(R * 128);
    c.gd = ar.take<float>(R * 128);
    c.ge = ar.take<float>(R * 128);
    c.dqkv = local ? ar.take<float>(R * 384) : nullptr;
    c.dh = local ? ar.take<float>(R * 512) : nullptr;
    c.dp = local ? nullptr : ar.take<float>(R * 16);
    c.dmax = ar.take<float>((size_t)16 * B * 512);   // per-slice column sums of the max prefix
}

void carve(Arena &ar, int B, int N, int K, TrainWs &w) {
    const size_t R = (size_t)B * N;
    w.pe = ar.take<float>(R * 384);
    w.pews = ar.take<float>(dvm_pos_encoding_workspace_bytes() / sizeof(float));
    w.pesync = ar.take<float>(2 * MAXG);
    w.fpe = ar.take<float>(R * 384);
    for (int i = 0; i < 8; ++i) {
        w.cv[i].z = ar.take<float>(R * CONV_CO[i]);
        w.cv[i].y = (i == 1 || i == 7) ? nullptr : ar.take<float>(R * CONV_CO[i]);   // tmp / feat live in the caller's tensors
        w.cv[i].bn.mean = ar.take<float>(MAXG * CONV_CO[i]);
        w.cv[i].bn.invstd = ar.take<float>(MAXG * CONV_CO[i]);
        w.cv[i].bn.var = ar.take<float>(MAXG * CONV_CO[i]);
    }
    for (int l = 0; l < 7; ++l) {
        const int C = NP_C[l];
        NpSave &n = w.np[l];
        n.idx = ar.take<int32_t>(R * K);
        n.qkv = ar.take<float>(R * 3 * C);
        n.att = ar.take<float>(R * C);
        n.attn = ar.take<float>(R * K * 4);
        n.x1 = ar.take<float>(R * C);
        n.h = ar.take<float>(R * 4 * C);
        n.ffo = ar.take<float>(R * C);
        n.out = ar.take<float>(R * C);
        n.bn1.mean = ar.take<float>(MAXG * C), n.bn1.invstd = ar.take<float>(MAXG * C), n.bn1.var = ar.take<float>(MAXG * C);
        n.bn2.mean = ar.take<float>(MAXG * C), n.bn2.invstd = ar.take<float>(MAXG * C), n.bn2.var = ar.take<float>(MAXG * C);
        w.wqkv[l] = ar.take<float>((size_t)3 * C * C);
    }
    w.dwqkv_off = ar.off;
    for (int l = 0; l < 7; ++l) w.dwqkv[l] = ar.take<float>((size_t)3 * NP_C[l] * NP_C[l]);
    w.dwqkv_bytes = ar.off - w.dwqkv_off;
    for (int l = 0; l < 4; ++l) {
        SaSave &a = w.sa[l];
        a.p = ar.take<float>(R * 16);
        a.v = ar.take<float>(R * 64);
        a.xr = ar.take<float>(R * 64);
        a.stats = ar.take<float>(R * 2);
        a.cinv = ar.take<float>(R);
        a.d = ar.take<float>(R * 64);
        a.t = ar.take<float>(R * 64);
        a.y = ar.take<float>(R * 64);
        a.out = ar.take<float>(R * 64);
        a.bn.mean = ar.take<float>(MAXG * 64), a.bn.invstd = ar.take<float>(MAXG * 64), a.bn.var = ar.take<float>(MAXG * 64);
    }
    w.loc = ar.take<float>(R * 256);
    w.catL = ar.take<float>(R * 768);
    w.glo = ar.take<float>(R * 256);
    w.catG = ar.take<float>(R * 768);
    w.mxL = ar.take<float>((size_t)B * 512), w.mxG = ar.take<float>((size_t)B * 512);
    w.argL = ar.take<int32_t>((size_t)B * 512), w.argG = ar.take<int32_t>((size_t)B * 512);
    w.packL = ar.take<unsigned long long>((size_t)B * 512), w.packG = ar.take<unsigned long long>((size_t)B * 512);
    w.ycat2 = ar.take<float>(R * 256);
    w.ycat4 = ar.take<float>(R * 512);
    carve_chain(ar, B, N, K, w.cs[0], true);
    carve_chain(ar, B, N, K, w.cs[1], false);
}

#define T_TRY(call)                    \
    do {                               \
        const int rc_ = (call);        \
        if (rc_ != DVM_OK) return rc_; \
    } while (0)

struct Net {
    const float *const *P;   // parameters
    float *const *GR;        // gradients (backward; entries of running statistics unused)
    const int32_t *const *knn_forced = nullptr;   // per N2P block: neighbour sets to use instead of the library's own (tests)
    int32_t *const *knn_log = nullptr;            // per N2P block: receives the library's own sets (tests)
    int B, N, K;   // B: ALL shapes of the call (groups x shapes per group)
    int G = 1;     // groups: network calls of the reference merged into this one (each with its own BatchNorm statistics and
                   // position-encoding range, exactly as if it had been a call of its own); rows of group g: [g Rg, (g + 1) Rg)
    long R, Rg;
    float eps, momentum;
    bool defer_stats = false;   // leave the running statistics alone (dvm_uni3fc_train_running_stats_f32 applies the updates later)
    const dvm_collective *coll = nullptr;   // data-parallel step: BatchNorm statistics and the position-encoding range over ALL ranks
    TrainWs w;
};

// ---------------------------------------------------------------- forward pieces
int bn_fwd(const Net &n, const float *x, const float *res, const float *g, const float *b, float *rm, float *rv, int C, float slope, float *y,
           const BnSave &sv, const ChainScratch &c, hipStream_t s) {
    // per group its own batch statistics (one launch set for all groups); the running statistics take the groups' updates in order
    return dvm_bn_act_train_fwd_pm_sync_f32(x, res, g, b, n.Rg, C, n.G, n.eps, slope, n.momentum, y, sv.mean, sv.invstd, sv.var,
                                            n.defer_stats ? nullptr : rm, n.defer_stats ? nullptr : rv, c.bnws, c.bn_bytes, n.coll, c.syncbuf, s);
}

// conv block i: y = leaky_0.2(bn(x W^T))
int conv_fwd(const Net &n, int i, const float *x, float *y, const ChainScratch &c, hipStream_t s) {
    const float *const *p = n.P + T_CONV0 + i * TC_N;
    const ConvSave &cv = n.w.cv[i];
    T_TRY(dvm_linear_f32(x, p[TC_W], n.B, n.N, CONV_K[i], CONV_CO[i], 0, nullptr, nullptr, nullptr, nullptr, 1.f, cv.z, s));
    return bn_fwd(n, cv.z, nullptr, p[TC_G], p[TC_B], (float *)p[TC_RM], (float *)p[TC_RV], CONV_CO[i], 0.2f, y, cv.bn, c, s);
}

bool stacked(const float *q, const float *k, const float *v, int C) { return k == q + (size_t)C * C && v == k + (size_t)C * C; }

int n2p_fwd(const Net &n, int l, const float *xin, const ChainScratch &c, hipStream_t s) {
    const float *const *p = n.P + T_NP0 + l * TN_N;
    const NpSave &sv = n.w.np[l];
    const int C = NP_C[l];
    const float *wqkv = stacked(p[TN_WQ], p[TN_WK], p[TN_WV], C) ? p[TN_WQ] : n.w.wqkv[l];
    T_TRY(dvm_knn_neg_f32(xin, xin, n.B, n.N, n.N, C, n.K, sv.idx, c.knnws, c.knn_bytes, s));
    const size_t idx_bytes = (size_t)n.R * n.K * sizeof(int32_t);
    if (n.knn_log && n.knn_log[l]) (void)hipMemcpyAsync(n.knn_log[l], sv.idx, idx_bytes, hipMemcpyDeviceToDevice, s);
    if (n.knn_forced && n.knn_forced[l]) (void)hipMemcpyAsync(sv.idx, n.knn_forced[l], idx_bytes, hipMemcpyDeviceToDevice, s);
    T_TRY(dvm_linear_f32(xin, wqkv, n.B, n.N, C, 3 * C, 0, nullptr, nullptr, nullptr, nullptr, 1.f, sv.qkv, s));
    T_TRY(dvm_n2p_core_fwd_f32(sv.qkv, sv.idx, n.B, n.N, C, n.K, 4, sv.att, sv.attn, s));
    T_TRY(bn_fwd(n, xin, sv.att, p[TN_G1], p[TN_B1], (float *)p[TN_RM1], (float *)p[TN_RV1], C, 1.f, sv.x1, sv.bn1, c, s));
    T_TRY(dvm_linear_f32(sv.x1, p[TN_FF0], n.B, n.N, C, 4 * C, 0, nullptr, nullptr, nullptr, nullptr, 0.2f, sv.h, s));
    T_TRY(dvm_linear_f32(sv.h, p[TN_FF2], n.B, n.N, 4 * C, C, 0, nullptr, nullptr, nullptr, nullptr, 1.f, sv.ffo, s));
    return bn_fwd(n, sv.x1, sv.ffo, p[TN_G2], p[TN_B2], (float *)p[TN_RM2], (float *)p[TN_RV2], C, 1.f, sv.out, sv.bn2, c, s);
}

int sa_fwd(const Net &n, int l, const float *xin, const ChainScratch &c, hipStream_t s) {
    const float *const *p = n.P + T_SA0 + l * TS_N;
    const SaSave &sv = n.w.sa[l];
    const long n4 = n.R * 16;
    T_TRY(dvm_linear_f32(xin, p[TS_WK], n.B, n.N, 64, 16, 0, nullptr, nullptr, nullptr, nullptr, 1.f, sv.p, s));
    T_TRY(dvm_linear_f32(xin, p[TS_WV], n.B, n.N, 64, 64, 0, p[TS_BV], nullptr, nullptr, nullptr, 1.f, sv.v, s));
    T_TRY(dvm_sa_attention_train_fwd_f32(sv.p, sv.v, n.B, n.N, sv.xr, sv.stats, sv.cinv, c.saws, c.sa_bytes, s));
    hipLaunchKernelGGL(sub_kernel, dim3(blocks_for(n4)), dim3(256), 0, s, (const f32x4 *)xin, (const f32x4 *)sv.xr, n4, (f32x4 *)sv.d);
    T_TRY(dvm_linear_f32(sv.d, p[TS_WT], n.B, n.N, 64, 64, 0, p[TS_BT], nullptr, nullptr, nullptr, 1.f, sv.t, s));
    T_TRY(bn_fwd(n, sv.t, nullptr, p[TS_G], p[TS_B], (float *)p[TS_RM], (float *)p[TS_RV], 64, 0.f, sv.y, sv.bn, c, s));
    hipLaunchKernelGGL(add3_kernel, dim3(blocks_for(n4)), dim3(256), 0, s, (const f32x4 *)xin, (const f32x4 *)sv.y, (const f32x4 *)nullptr, n4,
                       (f32x4 *)sv.out);
    return DVM_OK;
}

void concat4(const float *s0, const float *s1, const float *s2, const float *s3, int ns, int C, long rows, float *out, hipStream_t s) {
    CatArgs a;
    a.src[0] = (const f32x4 *)s0, a.src[1] = (const f32x4 *)s1, a.src[2] = (const f32x4 *)s2, a.src[3] = (const f32x4 *)s3;
    a.ns = ns, a.c4 = C / 4, a.rows = rows, a.out = (f32x4 *)out;
    hipLaunchKernelGGL(concat_kernel, dim3(blocks_for(rows * ns * C / 4)), dim3(256), 0, s, a);
}

// [max over the points of wide (B,N,512) | x (B,N,256)] -> cat (B,N,768), with the arg-max rows kept for the backward
void max_prefix(const Net &n, const float *wide, const float *x, unsigned long long *pack, float *mx, int32_t *arg, float *cat, hipStream_t s) {
    (void)hipMemsetAsync(pack, 0, (size_t)n.B * 512 * sizeof(unsigned long long), s);
    const int splits = n.N >= 512 ? 16 : 1, rows_per = (n.N + splits - 1) / splits;
    hipLaunchKernelGGL(colargmax_kernel, dim3(512 / 64, n.B, splits), dim3(256), 0, s, wide, n.N, 512, rows_per, pack);
    hipLaunchKernelGGL(cat_prefix_kernel, dim3(blocks_for((long)n.N * 192, 1024), n.B), dim3(256), 0, s, pack, x, n.N, 512, 256, mx, arg, cat);
}

// ---------------------------------------------------------------- backward pieces
int bn_bwd(const Net &n, const float *dy, const float *y, const float *x, const float *res, const float *g, const BnSave &sv, int C, float slope,
           float *dx, float *dg, float *db, const ChainScratch &c, hipStream_t s) {
    return dvm_bn_act_train_bwd_pm_sync_f32(dy, y, x, res, g, sv.mean, sv.invstd, n.Rg, C, n.G, slope, dx, dg, db, 1, c.bnws, c.bn_bytes, n.coll, c.syncbuf, s);
}
// dX [R][K] = dY [R][Co] W [Co][K]  (+ res): dvm_linear_f32 with the operands' roles swapped
int dgrad(const Net &n, const float *dy, const float *W, int Co, int K, const float *res, float *dx, hipStream_t s) {
    return dvm_linear_f32(W, dy, 1, K, Co, (int)n.R, 1, nullptr, res, nullptr, nullptr, 1.f, dx, s);
}
int wgrad(const Net &n, const float *dy, const float *x, int Co, int K, float *dW, const ChainScratch &c, hipStream_t s) {
    return dvm_linear_wgrad_ws_f32(dy, x, n.R, Co, K, dW, c.wgws, c.wg_bytes, s);   // (the workspace is used in deterministic mode only)
}
void colsum_accum(const Net &n, const float *g, int C, float *out, const ChainScratch &c, hipStream_t s) {
    long chunks = (n.R + 255) / 256;        // (C in {64, 128}: checked by the callers' layer table)
    if (chunks > 128) chunks = 128;         // (the second launch is ONE workgroup walking the chunks: 512 of them cost it 37 us)
    const long rows_per = (n.R + chunks - 1) / chunks;
    chunks = (n.R + rows_per - 1) / rows_per;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)chunks), dim3(256), 0, s, g, n.R, C, rows_per, c.colpart);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(1), dim3(256), 0, s, c.colpart, (int)chunks, C, out);
}
void slice_add(const float *src, int ld, int off, const float *add, long rows, int C, float *dst, hipStream_t s) {
    hipLaunchKernelGGL(slice_add_kernel, dim3(blocks_for(rows * C / 4)), dim3(256), 0, s, src, ld, off, (const f32x4 *)add, rows, C / 4, (f32x4 *)dst);
}

// conv block i backward: dy -> (dx unless x needs none); scratch dz
int conv_bwd(const Net &n, int i, const float *dy, const float *y, const float *x, float *dz, const float *dx_res, float *dx, const ChainScratch &c,
             hipStream_t s) {
    const float *const *p = n.P + T_CONV0 + i * TC_N;
    float *const *g = n.GR + T_CONV0 + i * TC_N;
    const ConvSave &cv = n.w.cv[i];
    T_TRY(bn_bwd(n, dy, y, cv.z, nullptr, p[TC_G], cv.bn, CONV_CO[i], 0.2f, dz, g[TC_G], g[TC_B], c, s));
    T_TRY(wgrad(n, dz, x, CONV_CO[i], CONV_K[i], g[TC_W], c, s));
    if (dx) T_TRY(dgrad(n, dz, p[TC_W], CONV_CO[i], CONV_K[i], dx_res, dx, s));
    return DVM_OK;
}

// N2P block l backward: g_out (R,C) -> dx (R,C) (gradient w.r.t. the block's input); g_out may alias nothing of the scratch used here
int n2p_bwd(const Net &n, int l, const float *xin, const float *g_out, float *dx, const ChainScratch &c, hipStream_t s) {
    const float *const *p = n.P + T_NP0 + l * TN_N;
    float *const *g = n.GR + T_NP0 + l * TN_N;
    const NpSave &sv = n.w.np[l];
    const int C = NP_C[l];
    float *dz = c.gc, *dx1 = c.gd, *din = c.ge;
    T_TRY(bn_bwd(n, g_out, sv.out, sv.x1, sv.ffo, p[TN_G2], sv.bn2, C, 1.f, dz, g[TN_G2], g[TN_B2], c, s));
    T_TRY(wgrad(n, dz, sv.h, C, 4 * C, g[TN_FF2], c, s));
    T_TRY(dgrad(n, dz, p[TN_FF2], C, 4 * C, nullptr, c.dh, s));
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(n.R * C)), dim3(256), 0, s, (f32x4 *)c.dh, (const f32x4 *)sv.h, n.R * C, 0.2f);
    T_TRY(wgrad(n, c.dh, sv.x1, 4 * C, C, g[TN_FF0], c, s));
    T_TRY(dgrad(n, c.dh, p[TN_FF0], 4 * C, C, dz, dx1, s));                       // d x1 = dh W0 + dz (the residual path)
    T_TRY(bn_bwd(n, dx1, sv.x1, xin, sv.att, p[TN_G1], sv.bn1, C, 1.f, din, g[TN_G1], g[TN_B1], c, s));
    T_TRY(dvm_n2p_core_bwd_f32(sv.qkv, sv.idx, sv.attn, din, n.B, n.N, C, n.K, 4, c.dqkv, c.npbws, c.npb_bytes, s));
    const bool st = stacked(p[TN_WQ], p[TN_WK], p[TN_WV], C), gst = stacked(g[TN_WQ], g[TN_WK], g[TN_WV], C);
    T_TRY(wgrad(n, c.dqkv, xin, 3 * C, C, gst ? g[TN_WQ] : n.w.dwqkv[l], c, s));
    T_TRY(dgrad(n, c.dqkv, st ? p[TN_WQ] : n.w.wqkv[l], 3 * C, C, din, dx, s));   // d xin = dqkv Wqkv + din (the residual path)
    return DVM_OK;
}

// SA_Layer l backward: g_out (R,64) -> dx (R,64)
int sa_bwd(const Net &n, int l, const float *xin, const float *g_out, float *dx, const ChainScratch &c, hipStream_t s) {
    const float *const *p = n.P + T_SA0 + l * TS_N;
    float *const *g = n.GR + T_SA0 + l * TS_N;
    const SaSave &sv = n.w.sa[l];
    const long n4 = n.R * 16;
    float *dt = c.gc, *dd = c.gd, *ssum = c.ge, *ngx = c.gc;   // dt is dead once dd, dWt, dbt are formed
    T_TRY(bn_bwd(n, g_out, sv.y, sv.t, nullptr, p[TS_G], sv.bn, 64, 0.f, dt, g[TS_G], g[TS_B], c, s));
    T_TRY(wgrad(n, dt, sv.d, 64, 64, g[TS_WT], c, s));
    colsum_accum(n, dt, 64, g[TS_BT], c, s);
    T_TRY(dgrad(n, dt, p[TS_WT], 64, 64, nullptr, dd, s));
    hipLaunchKernelGGL(add_neg_kernel, dim3(blocks_for(n4)), dim3(256), 0, s, (const f32x4 *)g_out, (const f32x4 *)dd, n4, (f32x4 *)ssum, (f32x4 *)ngx);
    float *dv = dd;   // dd is dead after add_neg
    T_TRY(dvm_sa_attention_bwd_f32(sv.p, sv.v, sv.xr, sv.stats, sv.cinv, ngx, n.B, n.N, c.dp, dv, c.sabws, c.sab_bytes, s));
    T_TRY(wgrad(n, c.dp, xin, 16, 64, g[TS_WK], c, s));
    T_TRY(wgrad(n, dv, xin, 64, 64, g[TS_WV], c, s));
    colsum_accum(n, dv, 64, g[TS_BV], c, s);
    T_TRY(dgrad(n, c.dp, p[TS_WK], 16, 64, ssum, ngx, s));      // ngx is dead after the attention backward: reused as the partial sum
    T_TRY(dgrad(n, dv, p[TS_WV], 64, 64, ngx, dx, s));
    return DVM_OK;
}

// backward of [max-prefix | x] -> conv_hi, conv_lo(x) -> wide: given dcat (R,768) returns the gradient of x (R,256) in dx256
int prefix_bwd(const Net &n, int conv_wide, const float *dcat, const float *x256, const int32_t *arg, float *dx256, const ChainScratch &c, hipStream_t s) {
    const int S = n.N >= 512 ? 16 : 1, rows_per = (n.N + S - 1) / S;
    hipLaunchKernelGGL(prefix_colsum_kernel, dim3(512 / 64, n.B, S), dim3(256), 0, s, dcat, n.N, 768, 512, rows_per, c.dmax);
    hipLaunchKernelGGL(max_bwd_kernel, dim3(blocks_for((long)n.N * 128, 1024), n.B), dim3(256), 0, s, c.dmax, S, arg, n.N, 512, c.g512a);
    slice_add(dcat, 768, 512, nullptr, n.R, 256, c.g256a, s);
    // wide = blk(conv_wide, x256):  d x256 = dz W + (the slice above)
    return conv_bwd(n, conv_wide, c.g512a, n.w.cv[conv_wide].y, x256, c.g512b, c.g256a, dx256, c, s);
}

struct Fork {
    PairCtx *cx;
    hipStream_t main, side;
    void fork() const {
        if (cx) {
            (void)hipEventRecord(cx->ev_fork, main);
            (void)hipStreamWaitEvent(side, cx->ev_fork, 0);
        }
    }
};

int check_groups(const char *who, int B, int groups) {
    DVM_REQUIRE(groups >= 1 && groups <= MAXG && B % groups == 0, "%s: %d shapes cannot be split into %d groups (1..%d)", who, B, groups, MAXG);
    return DVM_OK;
}
int check_args(const char *who, int B, int N, int k, int nparams, const void *const *tab, bool grads) {
    DVM_REQUIRE(nparams == DVM_U3_TRAIN_NPARAMS, "%s: the table has %d entries, expected %d", who, nparams, DVM_U3_TRAIN_NPARAMS);
    DVM_REQUIRE(B >= 1 && N >= 1 && k >= 1 && k <= 64 && k <= N, "%s: bad sizes (B=%d N=%d k=%d)", who, B, N, k);
    for (int i = 0; i < nparams; ++i) {
        bool stat = false;   // running statistics: no gradient entry
        if (i < T_SA0) stat = (i % TC_N == TC_RM || i % TC_N == TC_RV);
        else if (i < T_NP0) stat = ((i - T_SA0) % TS_N == TS_RM || (i - T_SA0) % TS_N == TS_RV);
        else {
            const int j = (i - T_NP0) % TN_N;
            stat = (j == TN_RM1 || j == TN_RV1 || j == TN_RM2 || j == TN_RV2);
        }
        if (grads && stat) continue;
        DVM_REQUIRE(tab[i] != nullptr, "%s: table entry %d is null", who, i);
    }
    return DVM_OK;
}

}  // namespace
}  // namespace dvm

using namespace dvm;

DVM_EXPORT size_t dvm_uni3fc_train_workspace_bytes(int B, int N, int k) {
    Arena ar(nullptr, 0);
    TrainWs w;
    carve(ar, B, N, k, w);
    return ar.off;
}

DVM_EXPORT int dvm_uni3fc_train_fwd_f32(const float *xyz, const float *dino, int B, int N, const float *const *params, int nparams, int k, float eps,
                                        float momentum, int groups, int defer_running_stats, const int32_t *const *knn_forced,
                                        int32_t *const *knn_log, float *feat, float *tmp, void *arena, size_t arena_bytes, void *stream) {
    return dvm_uni3fc_train_fwd_sync_f32(xyz, dino, B, N, params, nparams, k, eps, momentum, groups, defer_running_stats, knn_forced, knn_log, feat, tmp,
                                         arena, arena_bytes, nullptr, stream);
}

DVM_EXPORT int dvm_uni3fc_train_fwd_sync_f32(const float *xyz, const float *dino, int B, int N, const float *const *params, int nparams, int k, float eps,
                                             float momentum, int groups, int defer_running_stats, const int32_t *const *knn_forced,
                                             int32_t *const *knn_log, float *feat, float *tmp, void *arena, size_t arena_bytes,
                                             const dvm_collective *coll, void *stream) {
    DVM_REQUIRE(xyz && dino && params && feat && tmp, "dvm_uni3fc_train_fwd_f32: null pointer");
    DVM_REQUIRE(!coll || coll->allreduce, "dvm_uni3fc_train_fwd_sync_f32: a collective without its function");
    T_TRY(check_args("dvm_uni3fc_train_fwd_f32", B, N, k, nparams, (const void *const *)params, false));
    Net n;
    T_TRY(check_groups("dvm_uni3fc_train_fwd_f32", B, groups));
    n.P = params, n.GR = nullptr, n.B = B, n.N = N, n.K = k, n.R = (long)B * N, n.eps = eps, n.momentum = momentum;
    n.G = groups, n.Rg = n.R / groups;
    n.coll = coll;
    n.knn_forced = knn_forced, n.knn_log = knn_log;
    n.defer_stats = defer_running_stats != 0;
    Arena ar(arena, arena_bytes);
    carve(ar, B, N, k, n.w);
    if (!ar.ok()) {
        set_error("dvm_uni3fc_train_fwd_f32: arena too small (%zu < %zu)", arena_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    TrainWs &w = n.w;
    w.cv[1].y = tmp, w.cv[7].y = feat;
    hipStream_t s = (hipStream_t)stream;
    const long R = n.R;
    PairCtx *cx = pair_ctx_find(s);
    const Fork fk{cx, s, cx ? cx->side : s};
    const ChainScratch &c0 = w.cs[0], &c1 = w.cs[1];

    // stacked q|k|v weights of the N2P blocks (skipped where the three parameters already lie back to back)
    {
        StackArgs a;
        a.count = 0;
        for (int l = 0; l < 7; ++l) {
            const float *const *p = params + T_NP0 + l * TN_N;
            const int C = NP_C[l];
            if (stacked(p[TN_WQ], p[TN_WK], p[TN_WV], C)) continue;
            for (int j = 0; j < 3; ++j) a.src[a.count] = p[TN_WQ + j], a.dst[a.count] = w.wqkv[l] + (size_t)j * C * C, a.n[a.count] = C * C, ++a.count;
        }
        if (a.count) hipLaunchKernelGGL(stack_copy_kernel, dim3(16, a.count), dim3(256), 0, s, a, 0);
    }
    // f = blk(conv, dino); tmp = blk(conv0, f + pos_encoding(x)^T)
    T_TRY(conv_fwd(n, 0, dino, w.cv[0].y, c0, s));
    for (int q = 0; q < n.G; ++q)   // the encoding normalises with the min / max over ONE call's tensor (models/model.py:548)
        if (n.coll)
            T_TRY(dvm_pos_encoding_sync_f32(xyz + (size_t)q * (B / n.G) * 3 * N, B / n.G, N, w.pe + (size_t)q * n.Rg * 384, w.pews,
                                            dvm_pos_encoding_workspace_bytes(), n.coll, w.pesync + 2 * q, s));
        else
            T_TRY(dvm_pos_encoding_f32(xyz + (size_t)q * (B / n.G) * 3 * N, B / n.G, N, w.pe + (size_t)q * n.Rg * 384, w.pews,
                                       dvm_pos_encoding_workspace_bytes(), s));
    (void)hipMemcpyAsync(w.fpe, w.cv[0].y, (size_t)R * 384 * sizeof(float), hipMemcpyDeviceToDevice, s);
    hipLaunchKernelGGL(add_transposed_kernel, dim3((N + 31) / 32, 384 / 32, B), dim3(256), 0, s, w.fpe, w.pe, N, 384);
    T_TRY(conv_fwd(n, 1, w.fpe, tmp, c0, s));

    fk.fork();
    auto global_chain = [&]() -> int {
        hipStream_t gs = fk.side;
        const float *in = tmp;
        for (int l = 0; l < 4; ++l) {
            T_TRY(sa_fwd(n, l, in, c1, gs));
            in = w.sa[l].out;
        }
        concat4(w.sa[0].out, w.sa[1].out, w.sa[2].out, w.sa[3].out, 4, 64, R, w.glo, gs);
        T_TRY(conv_fwd(n, 3, w.glo, w.cv[3].y, c1, gs));                               // conv2
        max_prefix(n, w.cv[3].y, w.glo, w.packG, w.mxG, w.argG, w.catG, gs);
        T_TRY(conv_fwd(n, 5, w.catG, w.cv[5].y, c1, gs));                              // conv4 over [max | glo]
        return DVM_OK;
    };
    auto local_chain = [&]() -> int {
        const float *in = tmp;
        for (int l = 0; l < 4; ++l) {
            T_TRY(n2p_fwd(n, l, in, c0, s));
            in = w.np[l].out;
        }
        concat4(w.np[0].out, w.np[1].out, w.np[2].out, w.np[3].out, 4, 64, R, w.loc, s);
        T_TRY(conv_fwd(n, 2, w.loc, w.cv[2].y, c0, s));                                // conv1
        max_prefix(n, w.cv[2].y, w.loc, w.packL, w.mxL, w.argL, w.catL, s);
        T_TRY(conv_fwd(n, 4, w.catL, w.cv[4].y, c0, s));                               // conv3 over [max | loc]
        return DVM_OK;
    };
    const int rcg = global_chain();
    if (cx) (void)hipEventRecord(cx->ev_join, fk.side);
    const int rcl = local_chain();
    if (cx) (void)hipStreamWaitEvent(s, cx->ev_join, 0);   // the caller's stream waits for the helper stream on EVERY path from here on
    if (rcg != DVM_OK) return rcg;
    if (rcl != DVM_OK) return rcl;

    concat4(w.cv[4].y, w.cv[5].y, nullptr, nullptr, 2, 128, R, w.ycat2, s);
    T_TRY(conv_fwd(n, 6, w.ycat2, w.cv[6].y, c0, s));                                  // conv5 -> y1
    const float *in = w.cv[6].y;
    for (int l = 4; l < 7; ++l) {
        T_TRY(n2p_fwd(n, l, in, c0, s));
        in = w.np[l].out;
    }
    concat4(w.cv[6].y, w.np[4].out, w.np[5].out, w.np[6].out, 4, 128, R, w.ycat4, s);
    T_TRY(conv_fwd(n, 7, w.ycat4, feat, c0, s));                                       // conv6
    DVM_CHECK_LAUNCH("uni3fc_train_fwd");
    return DVM_OK;
}

DVM_EXPORT int dvm_uni3fc_train_bwd_f32(const float *g_feat, const float *g_tmp, const float *dino, const float *feat, const float *tmp, int B, int N,
                                        const float *const *params, float *const *grads, int nparams, int k, int groups, void *arena,
                                        size_t arena_bytes, void *stream) {
    return dvm_uni3fc_train_bwd_sync_f32(g_feat, g_tmp, dino, feat, tmp, B, N, params, grads, nparams, k, groups, arena, arena_bytes, nullptr, stream);
}

DVM_EXPORT int dvm_uni3fc_train_bwd_sync_f32(const float *g_feat, const float *g_tmp, const float *dino, const float *feat, const float *tmp, int B, int N,
                                             const float *const *params, float *const *grads, int nparams, int k, int groups, void *arena,
                                             size_t arena_bytes, const dvm_collective *coll, void *stream) {
    DVM_REQUIRE(g_feat && dino && feat && tmp && params && grads, "dvm_uni3fc_train_bwd_f32: null pointer");
    DVM_REQUIRE(!coll || coll->allreduce, "dvm_uni3fc_train_bwd_sync_f32: a collective without its function");
    T_TRY(check_args("dvm_uni3fc_train_bwd_f32", B, N, k, nparams, (const void *const *)params, false));
    T_TRY(check_args("dvm_uni3fc_train_bwd_f32 (gradients)", B, N, k, nparams, (const void *const *)grads, true));
    Net n;
    T_TRY(check_groups("dvm_uni3fc_train_bwd_f32", B, groups));
    n.P = params, n.GR = grads, n.B = B, n.N = N, n.K = k, n.R = (long)B * N, n.eps = 0.f, n.momentum = 0.f;
    n.G = groups, n.Rg = n.R / groups;
    n.coll = coll;
    Arena ar(arena, arena_bytes);
    carve(ar, B, N, k, n.w);
    if (!ar.ok()) {
        set_error("dvm_uni3fc_train_bwd_f32: arena too small (%zu < %zu)", arena_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    TrainWs &w = n.w;
    w.cv[1].y = (float *)tmp, w.cv[7].y = (float *)feat;
    hipStream_t s = (hipStream_t)stream;
    const long R = n.R;
    PairCtx *cx = pair_ctx_find(s);
    const Fork fk{cx, s, cx ? cx->side : s};
    const ChainScratch &c0 = w.cs[0], &c1 = w.cs[1];

    bool any_unstacked = false;
    for (int l = 0; l < 7; ++l) {
        float *const *g = grads + T_NP0 + l * TN_N;
        any_unstacked = any_unstacked || !stacked(g[TN_WQ], g[TN_WK], g[TN_WV], NP_C[l]);
    }
    if (any_unstacked) (void)hipMemsetAsync((char *)arena + w.dwqkv_off, 0, w.dwqkv_bytes, s);

    // conv6: feat = blk(conv6, ycat4)
    T_TRY(conv_bwd(n, 7, g_feat, feat, w.ycat4, c0.ga, nullptr, c0.g512a, c0, s));     // g512a = d ycat4
    // trunk N2P blocks 7, 6, 5 (np[6], np[5], np[4]); their inputs: np[5].out, np[4].out, y1 = cv[6].y
    slice_add(c0.g512a, 512, 384, nullptr, R, 128, c0.ga, s);
    T_TRY(n2p_bwd(n, 6, w.np[5].out, c0.ga, c0.gb, c0, s));
    slice_add(c0.g512a, 512, 256, c0.gb, R, 128, c0.ga, s);
    T_TRY(n2p_bwd(n, 5, w.np[4].out, c0.ga, c0.gb, c0, s));
    slice_add(c0.g512a, 512, 128, c0.gb, R, 128, c0.ga, s);
    T_TRY(n2p_bwd(n, 4, w.cv[6].y, c0.ga, c0.gb, c0, s));
    slice_add(c0.g512a, 512, 0, c0.gb, R, 128, c0.ga, s);                              // ga = d y1
    // conv5: y1 = blk(conv5, ycat2) -> d ycat2 (R,256) = [d lout | d gout]
    T_TRY(conv_bwd(n, 6, c0.ga, w.cv[6].y, w.ycat2, c0.gb, nullptr, c0.g256b, c0, s));
    slice_add(c0.g256b, 256, 0, nullptr, R, 128, c0.ga, s);                            // d lout (local chain, caller's stream)
    slice_add(c0.g256b, 256, 128, nullptr, R, 128, c1.ga, s);                          // d gout (global chain; written before the fork)

    fk.fork();
    auto global_chain = [&]() -> int {
        hipStream_t gs = fk.side;
        // conv4: gout = blk(conv4, catG)
        T_TRY(conv_bwd(n, 5, c1.ga, w.cv[5].y, w.catG, c1.gb, nullptr, c1.g768, c1, gs));
        T_TRY(prefix_bwd(n, 3, c1.g768, w.glo, w.argG, c1.g256b, c1, gs));             // g256b = d glo
        slice_add(c1.g256b, 256, 192, nullptr, R, 64, c1.ga, gs);
        T_TRY(sa_bwd(n, 3, w.sa[2].out, c1.ga, c1.gb, c1, gs));
        slice_add(c1.g256b, 256, 128, c1.gb, R, 64, c1.ga, gs);
        T_TRY(sa_bwd(n, 2, w.sa[1].out, c1.ga, c1.gb, c1, gs));
        slice_add(c1.g256b, 256, 64, c1.gb, R, 64, c1.ga, gs);
        T_TRY(sa_bwd(n, 1, w.sa[0].out, c1.ga, c1.gb, c1, gs));
        slice_add(c1.g256b, 256, 0, c1.gb, R, 64, c1.ga, gs);
        T_TRY(sa_bwd(n, 0, tmp, c1.ga, c1.gb, c1, gs));                                // c1.gb = d tmp through the global chain
        return DVM_OK;
    };
    auto local_chain = [&]() -> int {
        T_TRY(conv_bwd(n, 4, c0.ga, w.cv[4].y, w.catL, c0.gb, nullptr, c0.g768, c0, s));
        T_TRY(prefix_bwd(n, 2, c0.g768, w.loc, w.argL, c0.g256b, c0, s));              // g256b = d loc
        slice_add(c0.g256b, 256, 192, nullptr, R, 64, c0.ga, s);
        T_TRY(n2p_bwd(n, 3, w.np[2].out, c0.ga, c0.gb, c0, s));
        slice_add(c0.g256b, 256, 128, c0.gb, R, 64, c0.ga, s);
        T_TRY(n2p_bwd(n, 2, w.np[1].out, c0.ga, c0.gb, c0, s));
        slice_add(c0.g256b, 256, 64, c0.gb, R, 64, c0.ga, s);
        T_TRY(n2p_bwd(n, 1, w.np[0].out, c0.ga, c0.gb, c0, s));
        slice_add(c0.g256b, 256, 0, c0.gb, R, 64, c0.ga, s);
        T_TRY(n2p_bwd(n, 0, tmp, c0.ga, c0.gb, c0, s));                                // c0.gb = d tmp through the local chain
        return DVM_OK;
    };
    const int rcg = global_chain();
    if (cx) (void)hipEventRecord(cx->ev_join, fk.side);
    const int rcl = local_chain();
    if (cx) (void)hipStreamWaitEvent(s, cx->ev_join, 0);
    if (rcg != DVM_OK) return rcg;
    if (rcl != DVM_OK) return rcl;

    // d tmp = local + global (+ the caller's gradient of the second output)
    hipLaunchKernelGGL(add3_kernel, dim3(blocks_for(R * 16)), dim3(256), 0, s, (const f32x4 *)c0.gb, (const f32x4 *)c1.gb, (const f32x4 *)g_tmp, R * 16,
                       (f32x4 *)c0.ga);
    // conv0: tmp = blk(conv0, fpe) -> d fpe = d f (the position encoding carries no gradient); conv: f = blk(conv, dino)
    T_TRY(conv_bwd(n, 1, c0.ga, tmp, w.fpe, c0.gb, nullptr, c0.g512a, c0, s));         // g512a (R,384) = d f
    T_TRY(conv_bwd(n, 0, c0.g512a, w.cv[0].y, dino, c0.g512b, nullptr, nullptr, c0, s));

    if (any_unstacked) {   // stacked q|k|v weight gradients -> the three parameters' gradient buffers
        StackArgs a;
        a.count = 0;
        for (int l = 0; l < 7; ++l) {
            float *const *g = grads + T_NP0 + l * TN_N;
            const int C = NP_C[l];
            if (stacked(g[TN_WQ], g[TN_WK], g[TN_WV], C)) continue;
            for (int j = 0; j < 3; ++j) a.src[a.count] = w.dwqkv[l] + (size_t)j * C * C, a.dst[a.count] = g[TN_WQ + j], a.n[a.count] = C * C, ++a.count;
        }
        hipLaunchKernelGGL(stack_copy_kernel, dim3(16, a.count), dim3(256), 0, s, a, 1);
    }
    DVM_CHECK_LAUNCH("uni3fc_train_bwd");
    return DVM_OK;
}

// The running-statistics updates of one network call whose forward ran with defer_running_stats: 26 BatchNorms, one launch
DVM_EXPORT int dvm_uni3fc_train_running_stats_f32(const float *const *params, int nparams, int B, int N, int k, int groups, float momentum,
                                                  void *arena, size_t arena_bytes, void *stream) {
    DVM_REQUIRE(params && arena, "dvm_uni3fc_train_running_stats_f32: null pointer");
    T_TRY(check_groups("dvm_uni3fc_train_running_stats_f32", B, groups));
    T_TRY(check_args("dvm_uni3fc_train_running_stats_f32", B, N, k, nparams, (const void *const *)params, false));
    TrainWs w;
    Arena ar(arena, arena_bytes);
    carve(ar, B, N, k, w);
    if (!ar.ok()) {
        set_error("dvm_uni3fc_train_running_stats_f32: arena too small (%zu < %zu)", arena_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    float *rm[26], *rv[26];
    const float *mean[26], *var[26];
    int C[26], n = 0;
    auto add = [&](const float *const *p, int irm, int irv, const BnSave &sv, int c) {
        rm[n] = (float *)p[irm], rv[n] = (float *)p[irv], mean[n] = sv.mean, var[n] = sv.var, C[n] = c, ++n;
    };
    for (int i = 0; i < 8; ++i) add(params + T_CONV0 + i * TC_N, TC_RM, TC_RV, w.cv[i].bn, CONV_CO[i]);
    for (int l = 0; l < 4; ++l) add(params + T_SA0 + l * TS_N, TS_RM, TS_RV, w.sa[l].bn, 64);
    for (int l = 0; l < 7; ++l) {
        add(params + T_NP0 + l * TN_N, TN_RM1, TN_RV1, w.np[l].bn1, NP_C[l]);
        add(params + T_NP0 + l * TN_N, TN_RM2, TN_RV2, w.np[l].bn2, NP_C[l]);
    }
    for (int q = 0; q < groups; ++q) {   // one launch per group, in order: the second group's update builds on the first's
        const float *mq[26], *vq[26];
        for (int i = 0; i < n; ++i) mq[i] = mean[i] + q * C[i], vq[i] = var[i] + q * C[i];
        launch_bn_running_update(rm, rv, mq, vq, C, n, momentum, (hipStream_t)stream);
    }
    DVM_CHECK_LAUNCH("uni3fc_train_running_stats");
    return DVM_OK;
}
