// dvm_uni3fc.hip — the eval-mode forward of LG-Net (`Uni3FC.forward`, reference models/model.py:680-761 with the layers of
// 506-529, the N2P blocks 325-395 and SA_Layer 97-123) behind ONE C-ABI call: dvm_uni3fc_fwd_f32.
//
// Nothing new is computed here: every layer is one of the library's own launches (dvm_linear_f32 and its prefix / residual
// forms, dvm_knn_neg_f32, dvm_n2p_core_fwd_f32, dvm_sa_attention_fwd_f32, dvm_pos_encoding_f32), in the order and with the
// operands of models/model.py::Uni3FC._forward_infer — the Python method this replaces enqueues ~250 launches through
// ~150 Python calls and is host-bound at 8 x 2048 points.  The five element-wise steps that method leaves to torch (the
// position encoding added to the first block's output, x - x_r, the eval-mode BatchNorm after the attention residual, the
// max over the points, the channel concatenations) are the small kernels below.  The local (kNN attention) and global
// (self-attention) chains share only `tmp`; with a context from dvm_pair_init for the caller's stream the global chain
// runs on that context's helper stream, as `_two_branches` does with torch streams.
//
// Weight table: DVM_U3_NWEIGHTS device pointers, order documented in include/dvm.h (dvm_uni3fc_fwd_f32); the eval-mode
// BatchNorms arrive folded to (alpha, beta) exactly as models/model.py::_bn_affine folds them on the host.
#include "dvm_uni3fc_kernels.h"

namespace dvm {

void launch_linear(const float *x, const float *w, int B, int N, int K, int Co, int channel_major, const float *bias,
                   const float *res, const float *alpha, const float *beta, float slope, float *y, hipStream_t s, const float *xg,
                   int Cg, const float *post_res, float post_scale);   // dvm_gemm.hip

namespace {

// indices into the weight table (include/dvm.h)
enum { CB_W = 0, CB_A, CB_B, CB_N };                                   // conv block: weight, BN alpha, BN beta
enum { SA_WK = 0, SA_WV, SA_BV, SA_WT, SA_BT, SA_A, SA_B, SA_N };       // SA_Layer
enum { NP_WQKV = 0, NP_A1, NP_B1, NP_FF0, NP_FF2, NP_A2, NP_B2, NP_N }; // N2P block
constexpr int U3_CONV0 = 0;                       // conv, conv0 ... conv6: 8 blocks
constexpr int U3_SA0 = U3_CONV0 + 8 * CB_N;       // sa1 ... sa4
constexpr int U3_NP0 = U3_SA0 + 4 * SA_N;         // n2p_attention1 ... 7
constexpr int U3_TOTAL = U3_NP0 + 7 * NP_N;
static_assert(U3_TOTAL == DVM_U3_NWEIGHTS, "weight table layout and include/dvm.h disagree");

struct U3Ws {
    float *f, *pe, *pews;                                           // first block
    // local chain + trunk (caller's stream)
    int32_t *idx;
    float *qkv, *att, *attn, *xt2, *h, *x[4], *cat4, *wide, *mx, *lout;
    void *knnws;
    size_t knn_bytes;
    // global chain (helper stream)
    float *p, *v, *xr, *d, *g[4], *gcat, *gwide, *gmx, *gout;
    void *saws;
    size_t sa_bytes;
    // trunk
    float *y, *y14[4], *ycat;
};

void carve(Arena &ar, int B, int N, int K, U3Ws &w) {
    const size_t R = (size_t)B * N;
    w.f = ar.take<float>(R * 384);
    w.pe = ar.take<float>(R * 384);
    w.pews = ar.take<float>(dvm_pos_encoding_workspace_bytes() / sizeof(float));
    w.idx = ar.take<int32_t>(R * K);
    w.qkv = ar.take<float>(R * 384);
    w.att = ar.take<float>(R * 128);
    w.attn = ar.take<float>(R * K * 4);
    w.xt2 = ar.take<float>(R * 128);
    w.h = ar.take<float>(R * 512);
    for (int i = 0; i < 4; ++i) w.x[i] = ar.take<float>(R * 64);
    w.cat4 = ar.take<float>(R * 256);
    w.wide = ar.take<float>(R * 512);
    w.mx = ar.take<float>((size_t)B * 512);
    w.lout = ar.take<float>(R * 128);
    w.knn_bytes = dvm_knn_neg_workspace_bytes(B, N, N, 128, K);
    w.knnws = ar.take<char>(w.knn_bytes);
    w.p = ar.take<float>(R * 16);
    w.v = ar.take<float>(R * 64);
    w.xr = ar.take<float>(R * 64);
    w.d = ar.take<float>(R * 64);
    for (int i = 0; i < 4; ++i) w.g[i] = ar.take<float>(R * 64);
    w.gcat = ar.take<float>(R * 256);
    w.gwide = ar.take<float>(R * 512);
    w.gmx = ar.take<float>((size_t)B * 512);
    w.gout = ar.take<float>(R * 128);
    w.sa_bytes = dvm_sa_attention_workspace_bytes(B, N);
    w.saws = ar.take<char>(w.sa_bytes);
    w.y = ar.take<float>(R * 256);
    for (int i = 0; i < 4; ++i) w.y14[i] = ar.take<float>(R * 128);
    w.ycat = ar.take<float>(R * 512);
}

#define U3_TRY(call)            \
    do {                        \
        const int rc_ = (call); \
        if (rc_ != DVM_OK) return rc_; \
    } while (0)

// conv + eval BatchNorm + LeakyReLU(0.2) — `blk` of _forward_infer; prefix: rows [g[b] | x[b][n]]
int conv_block(const float *const *W, int blk, const float *x, int B, int N, int K, int Co, float *y, hipStream_t s, const float *prefix = nullptr,
               int Cg = 0) {
    const float *const *c = W + U3_CONV0 + blk * CB_N;
    if (prefix) return dvm_linear_prefix_f32(prefix, Cg, x, c[CB_W], B, N, K, Co, nullptr, nullptr, c[CB_A], c[CB_B], 0.2f, y, s);
    return dvm_linear_f32(x, c[CB_W], B, N, K, Co, 0, nullptr, nullptr, c[CB_A], c[CB_B], 0.2f, y, s);
}

// SA_Layer.infer_pm: xt (B,N,64) -> out (B,N,64)
int sa_layer(const float *const *W, int l, const float *xt, int B, int N, U3Ws &w, float *out, hipStream_t s) {
    const float *const *c = W + U3_SA0 + l * SA_N;
    const long R = (long)B * N;
    U3_TRY(dvm_linear_f32(xt, c[SA_WK], B, N, 64, 16, 0, nullptr, nullptr, nullptr, nullptr, 1.f, w.p, s));
    U3_TRY(dvm_linear_f32(xt, c[SA_WV], B, N, 64, 64, 0, c[SA_BV], nullptr, nullptr, nullptr, 1.f, w.v, s));
    U3_TRY(dvm_sa_attention_fwd_f32(w.p, w.v, B, N, w.xr, w.saws, w.sa_bytes, s));
    hipLaunchKernelGGL(sub_kernel, dim3(blocks_for(R * 16)), dim3(256), 0, s, (const f32x4 *)xt, (const f32x4 *)w.xr, R * 16, (f32x4 *)w.d);
    // xt + relu(bn(trans_conv(xt - x_r)))
    launch_linear(w.d, c[SA_WT], B, N, 64, 64, 0, c[SA_BT], nullptr, c[SA_A], c[SA_B], 0.f, out, s, nullptr, 0, xt, 1.f);
    DVM_CHECK_LAUNCH("uni3fc: SA layer");
    return DVM_OK;
}

// _N2P.infer_pm: xt (B,N,C) -> out (B,N,C)
int n2p_layer(const float *const *W, int l, const float *xt, int B, int N, int C, int K, U3Ws &w, float *out, hipStream_t s) {
    const float *const *c = W + U3_NP0 + l * NP_N;
    const long R = (long)B * N;
    U3_TRY(dvm_knn_neg_f32(xt, xt, B, N, N, C, K, w.idx, w.knnws, w.knn_bytes, s));
    U3_TRY(dvm_linear_f32(xt, c[NP_WQKV], B, N, C, 3 * C, 0, nullptr, nullptr, nullptr, nullptr, 1.f, w.qkv, s));
    U3_TRY(dvm_n2p_core_fwd_f32(w.qkv, w.idx, B, N, C, K, 4, w.att, w.attn, s));
    hipLaunchKernelGGL(add_affine_kernel, dim3(blocks_for(R * C / 4)), dim3(256), 0, s, (const f32x4 *)xt, (const f32x4 *)w.att,
                       (const f32x4 *)c[NP_A1], (const f32x4 *)c[NP_B1], R * C / 4, C / 4, (f32x4 *)w.xt2);
    U3_TRY(dvm_linear_f32(w.xt2, c[NP_FF0], B, N, C, 4 * C, 0, nullptr, nullptr, nullptr, nullptr, 0.2f, w.h, s));
    // bn2(x + ff(x))
    U3_TRY(dvm_linear_f32(w.h, c[NP_FF2], B, N, 4 * C, C, 0, nullptr, w.xt2, c[NP_A2], c[NP_B2], 1.f, out, s));
    return DVM_OK;
}

void concat(const float *s0, const float *s1, const float *s2, const float *s3, int ns, int C, long rows, float *out, hipStream_t s) {
    CatArgs a;
    a.src[0] = (const f32x4 *)s0, a.src[1] = (const f32x4 *)s1, a.src[2] = (const f32x4 *)s2, a.src[3] = (const f32x4 *)s3;
    a.ns = ns, a.c4 = C / 4, a.rows = rows, a.out = (f32x4 *)out;
    hipLaunchKernelGGL(concat_kernel, dim3(blocks_for(rows * ns * C / 4)), dim3(256), 0, s, a);
}

void colmax(const float *x, int B, int N, int C, float *out, hipStream_t s) {
    hipLaunchKernelGGL(fill_kernel, dim3(blocks_for((long)B * C)), dim3(256), 0, s, out, (long)B * C, -INFINITY);
    const int splits = N >= 512 ? 16 : 1, rows_per = (N + splits - 1) / splits;
    hipLaunchKernelGGL(colmax_kernel, dim3((C + 63) / 64, B, splits), dim3(256), 0, s, x, N, C, rows_per, out);
}

}  // namespace
}  // namespace dvm

using namespace dvm;

DVM_EXPORT size_t dvm_uni3fc_fwd_workspace_bytes(int B, int N, int k) {
    Arena ar(nullptr, 0);
    U3Ws w;
    carve(ar, B, N, k, w);
    return ar.off;
}

DVM_EXPORT int dvm_uni3fc_fwd_f32(const float *xyz, const float *dino, int B, int N, const float *const *weights, int nweights, int k,
                                  float *feat, float *tmp, void *ws, size_t ws_bytes, void *stream) {
    DVM_REQUIRE(xyz && dino && weights && feat && tmp, "dvm_uni3fc_fwd_f32: null pointer");
    DVM_REQUIRE(nweights == DVM_U3_NWEIGHTS, "dvm_uni3fc_fwd_f32: the weight table has %d entries, expected %d", nweights, DVM_U3_NWEIGHTS);
    for (int i = 0; i < nweights; ++i) DVM_REQUIRE(weights[i] != nullptr, "dvm_uni3fc_fwd_f32: weight table entry %d is null", i);
    DVM_REQUIRE(B >= 1 && N >= 1 && k >= 1 && k <= 64 && k <= N, "dvm_uni3fc_fwd_f32: bad sizes (B=%d N=%d k=%d)", B, N, k);
    Arena ar(ws, ws_bytes);
    U3Ws w;
    carve(ar, B, N, k, w);
    if (!ar.ok()) {
        set_error("dvm_uni3fc_fwd_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const long R = (long)B * N;
    const float *const *W = weights;

    // f = blk(conv, dino); tmp = blk(conv0, f + pos_encoding(x)^T)
    U3_TRY(conv_block(W, 0, dino, B, N, 1152, 384, w.f, s));
    U3_TRY(dvm_pos_encoding_f32(xyz, B, N, w.pe, w.pews, dvm_pos_encoding_workspace_bytes(), s));
    hipLaunchKernelGGL(add_transposed_kernel, dim3((N + 31) / 32, 384 / 32, B), dim3(256), 0, s, w.f, w.pe, N, 384);
    U3_TRY(conv_block(W, 1, w.f, B, N, 384, 64, tmp, s));

    // the global chain on the helper stream of the caller's context (if one exists), the local chain on the caller's
    PairCtx *cx = pair_ctx_find(s);
    hipStream_t gs = cx ? cx->side : s;
    if (cx) {
        (void)hipEventRecord(cx->ev_fork, s);
        (void)hipStreamWaitEvent(gs, cx->ev_fork, 0);
    }
    int rc = DVM_OK;
    auto global_chain = [&]() -> int {
        const float *in = tmp;
        for (int l = 0; l < 4; ++l) {
            U3_TRY(sa_layer(W, l, in, B, N, w, w.g[l], gs));
            in = w.g[l];
        }
        concat(w.g[0], w.g[1], w.g[2], w.g[3], 4, 64, R, w.gcat, gs);
        U3_TRY(conv_block(W, 3, w.gcat, B, N, 256, 512, w.gwide, gs));          // conv2
        colmax(w.gwide, B, N, 512, w.gmx, gs);
        U3_TRY(conv_block(W, 5, w.gcat, B, N, 768, 128, w.gout, gs, w.gmx, 512));   // conv4 over [max | glo]
        return DVM_OK;
    };
    auto local_chain = [&]() -> int {
        const float *in = tmp;
        for (int l = 0; l < 4; ++l) {
            U3_TRY(n2p_layer(W, l, in, B, N, 64, k, w, w.x[l], s));
            in = w.x[l];
        }
        concat(w.x[0], w.x[1], w.x[2], w.x[3], 4, 64, R, w.cat4, s);
        U3_TRY(conv_block(W, 2, w.cat4, B, N, 256, 512, w.wide, s));              // conv1
        colmax(w.wide, B, N, 512, w.mx, s);
        U3_TRY(conv_block(W, 4, w.cat4, B, N, 768, 128, w.lout, s, w.mx, 512));     // conv3 over [max | loc]
        return DVM_OK;
    };
    const int rcg = global_chain();
    // the caller's stream waits for the helper stream on EVERY path from here on
    if (cx) (void)hipEventRecord(cx->ev_join, gs);
    rc = local_chain();
    if (cx) (void)hipStreamWaitEvent(s, cx->ev_join, 0);
    if (rcg != DVM_OK) return rcg;
    if (rc != DVM_OK) return rc;

    // trunk: y = [local | global]; y1 = blk(conv5, y); three 128-wide N2P blocks; out = blk(conv6, [y1 y2 y3 y4])
    concat(w.lout, w.gout, nullptr, nullptr, 2, 128, R, w.y, s);
    U3_TRY(conv_block(W, 6, w.y, B, N, 256, 128, w.y14[0], s));
    for (int l = 0; l < 3; ++l) U3_TRY(n2p_layer(W, 4 + l, w.y14[l], B, N, 128, k, w, w.y14[l + 1], s));
    concat(w.y14[0], w.y14[1], w.y14[2], w.y14[3], 4, 128, R, w.ycat, s);
    U3_TRY(conv_block(W, 7, w.ycat, B, N, 512, 128, feat, s));
    DVM_CHECK_LAUNCH("uni3fc_fwd");
    return DVM_OK;
}
