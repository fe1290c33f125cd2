// dvm_pair.hip — one direction of GraphDeformLoss_Neural.deform() for a batch of pairs as a
// single stream of launches with no host round trip (reference models/loss.py:1228-1296,
// 1401-1410; deform.py:232-257).  This is BASELINE config 2, "correspondence + deform forward".
#include "dvm_common.h"

namespace dvm {
// dvm_softcorr.hip / dvm_geom.hip / dvm_graph.hip / dvm_deformer.hip
int launch_mean(const float *in, int B, int n, float scale, float *out, int stride, int off, int accumulate, hipStream_t s);
int launch_reduce_partials(const double *partial, int B, int nparts, float scale, float *out, int stride, int off, hipStream_t s);
int map_term_blocks(int N, int k);
int launch_map_term(const float *verts12, const float *verts2, const int32_t *idx11, const int32_t *idx22, const float *pi_val,
                    const int32_t *pi_idx, int B, int N, int M, int k, int topk, double *partial, hipStream_t s);
int launch_dg_build(const float *xyz, int B, int N, const int32_t *start, int32_t *nodes_idx, int32_t *ring, int32_t *infl_idx,
                    float *dists, float *weights, double *sigma, double *nnd, hipStream_t s);
int launch_dg_warp(const float *xyz, int B, int N, const int32_t *nodes_idx, const int32_t *ring, const int32_t *infl_idx,
                   const float *weights, const float *def9, float *R, float *T, float *warped, float *arap, int arap_stride,
                   float *sr, hipStream_t s);
int launch_deformer(const float *feat1, const float *feat2, const float *verts1, const float *verts12, const int32_t *idx11,
                    const int32_t *idx22, const float *pi_val, const int32_t *pi_idx, const int32_t *fps1, int B, int N, int M,
                    int Nn, int k, int topk, const float *conv_w, const float *conv_b, const float *W0, const float *b0,
                    const float *W1, const float *b1, const float *W2, const float *b2, const float *W3, const float *b3,
                    float *out, int variant, void *ws, size_t ws_bytes, hipStream_t s);
size_t deformer_ws_bytes(int B, int M, int Nn);

__global__ void take_col0_kernel(const int32_t *__restrict__ in, int rows, int stride, int32_t *__restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) out[i] = in[(size_t)i * stride];
}
__global__ void fill_kernel(float *p, int n, float v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

struct PairWs {
    int32_t *nodes, *ring, *infl, *pidx, *idx11, *idx22;
    float *dists, *weights, *pval, *def9, *R, *T, *d1, *d2;
    double *nnd, *partial;
    void *sc_ws, *df_ws;
    size_t sc_bytes, df_bytes;
};

static size_t carve_pair(Arena &ar, int B, int N, int M, PairWs &w) {
    const int Nn = N / 2, k = 10, topk = 10;
    w.nodes = ar.take<int32_t>((size_t)B * Nn);
    w.ring = ar.take<int32_t>((size_t)B * Nn * 9);
    w.infl = ar.take<int32_t>((size_t)B * N * 3);
    w.dists = ar.take<float>((size_t)B * N * 3);
    w.weights = ar.take<float>((size_t)B * N * 3);
    w.nnd = ar.take<double>((size_t)B * N);
    w.pval = ar.take<float>((size_t)B * N * topk);
    w.pidx = ar.take<int32_t>((size_t)B * N * topk);
    w.idx11 = ar.take<int32_t>((size_t)B * N * k);
    w.idx22 = ar.take<int32_t>((size_t)B * M * k);
    w.def9 = ar.take<float>((size_t)B * Nn * 9);
    w.R = ar.take<float>((size_t)B * Nn * 9);
    w.T = ar.take<float>((size_t)B * Nn * 3);
    w.d1 = ar.take<float>((size_t)B * N);
    w.d2 = ar.take<float>((size_t)B * M);
    w.partial = ar.take<double>((size_t)B * map_term_blocks(N, k));
    w.sc_bytes = dvm_softcorr_workspace_bytes(B, N, M, 128);
    w.sc_ws = ar.take<char>(w.sc_bytes);
    w.df_bytes = deformer_ws_bytes(B, M, Nn);
    w.df_ws = ar.take<char>(w.df_bytes);
    return ar.off;
}
}  // namespace dvm

using namespace dvm;

DVM_EXPORT size_t dvm_pair_direction_workspace_bytes(int B, int N, int M) {
    Arena ar(nullptr, 0);
    PairWs w;
    return carve_pair(ar, B, N, M, w);
}

DVM_EXPORT int dvm_pair_direction_fwd_f32(const float *feat1, const float *feat2, const float *verts1, const float *verts2,
                                          int B, int N, int M, float neg_alpha, const int32_t *fps_start, const float *conv_w,
                                          const float *conv_b, const float *W0, const float *b0, const float *W1,
                                          const float *b1, const float *W2, const float *b2, const float *W3, const float *b3,
                                          int with_map, float *warped, float *verts12, int32_t *T12, float *losses, void *ws,
                                          size_t ws_bytes, void *stream) {
    DVM_REQUIRE(feat1 && feat2 && verts1 && verts2 && fps_start && warped && verts12 && T12 && losses,
                "dvm_pair_direction_fwd_f32: null tensor pointer");
    DVM_REQUIRE(conv_w && conv_b && W0 && b0 && W1 && b1 && W2 && b2 && W3 && b3,
                "dvm_pair_direction_fwd_f32: null weight pointer");
    DVM_REQUIRE(B >= 1 && N >= 20 && M >= 10, "dvm_pair_direction_fwd_f32: bad sizes (B=%d N=%d M=%d)", B, N, M);
    DVM_REQUIRE(neg_alpha < 0.f, "dvm_pair_direction_fwd_f32: neg_alpha must be negative");
    Arena ar(ws, ws_bytes);
    PairWs w;
    carve_pair(ar, B, N, M, w);
    if (!ar.ok()) {
        set_error("dvm_pair_direction_fwd_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int Nn = N / 2, k = 10, topk = 10;
    int rc;
    // graph of the source cloud (nodes, ring, skinning weights)
    launch_dg_build(verts1, B, N, fps_start, w.nodes, w.ring, w.infl, w.dists, w.weights, nullptr, w.nnd, s);
    // soft correspondence Pi_12 (top-10) and the arg-max map
    rc = dvm_softcorr_fwd_f32(feat1, feat2, B, N, M, 128, neg_alpha, topk, w.pval, w.pidx, nullptr, nullptr, 0, w.sc_ws,
                              w.sc_bytes, s);
    if (rc != DVM_OK) return rc;
    hipLaunchKernelGGL(take_col0_kernel, dim3((B * N + 255) / 256), dim3(256), 0, s, w.pidx, B * N, topk, T12);
    // verts12 = Pi_12 @ verts2
    rc = dvm_softcorr_apply_f32(w.pval, w.pidx, verts2, B, N, M, topk, 3, verts12, s);
    if (rc != DVM_OK) return rc;
    // xyz neighbourhoods
    rc = dvm_knn_cdist_f32(verts1, verts1, B, N, N, 3, k, w.idx11, s);
    if (rc != DVM_OK) return rc;
    rc = dvm_knn_cdist_f32(verts2, verts2, B, M, M, 3, k, w.idx22, s);
    if (rc != DVM_OK) return rc;
    // Deformer -> per-node [t, r6]
    rc = launch_deformer(feat1, feat2, verts1, verts12, w.idx11, w.idx22, w.pval, w.pidx, w.nodes, B, N, M, Nn, k, topk, conv_w,
                         conv_b, W0, b0, W1, b1, W2, b2, W3, b3, w.def9, 0, w.df_ws, w.df_bytes, s);
    if (rc != DVM_OK) return rc;
    // embedded-deformation warp + ARAP (losses[:,1])
    launch_dg_warp(verts1, B, N, w.nodes, w.ring, w.infl, w.weights, w.def9, w.R, w.T, warped, losses + 2, 6, nullptr, s);
    // chamfer(warped, verts2) -> losses[:,0:2]; chamfer(verts12, verts2) -> losses[:,3:5]
    rc = dvm_chamfer_fwd_f32(warped, verts2, B, N, M, w.d1, w.d2, nullptr, nullptr, s);
    if (rc != DVM_OK) return rc;
    launch_mean(w.d1, B, N, 1.f, losses, 6, 0, 0, s);
    launch_mean(w.d2, B, M, 1.f, losses, 6, 1, 0, s);
    rc = dvm_chamfer_fwd_f32(verts12, verts2, B, N, M, w.d1, w.d2, nullptr, nullptr, s);
    if (rc != DVM_OK) return rc;
    launch_mean(w.d1, B, N, 1.f, losses, 6, 3, 0, s);
    launch_mean(w.d2, B, M, 1.f, losses, 6, 4, 0, s);
    if (with_map) {
        launch_map_term(verts12, verts2, w.idx11, w.idx22, w.pval, w.pidx, B, N, M, k, topk, w.partial, s);
        launch_reduce_partials(w.partial, B, map_term_blocks(N, k), 1.f, losses, 6, 5, s);
    } else {
        // losses[:,5] = 0 — a strided fill through the mean kernel's overwrite path
        launch_mean(w.d1, B, 1, 0.f, losses, 6, 5, 0, s);
    }
    DVM_CHECK_LAUNCH("pair_direction");
    return DVM_OK;
}
