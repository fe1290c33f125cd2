// dvm_pair.hip — one direction of GraphDeformLoss_Neural.deform() for a batch of pairs as a
// single stream of launches with no host round trip (reference models/loss.py:1228-1296,
// 1401-1410; deform.py:232-257).  This is BASELINE config 2, "correspondence + deform forward".
#include <stdlib.h>

#include "dvm_common.h"

namespace dvm {
// dvm_softcorr.hip / dvm_geom.hip / dvm_graph.hip / dvm_deformer.hip
int launch_mean(const float *in, int B, int n, float scale, float *out, int stride, int off, int accumulate, hipStream_t s);
int launch_mean_grouped(const float *const *in, const int *n, float *const *out, const int *off, int ngroups, int B, float scale,
                        int stride, hipStream_t s);
int launch_reduce_partials(const double *partial, int B, int nparts, float scale, float *out, int stride, int off, hipStream_t s);
int map_term_blocks(int N, int k);
void launch_gather_nbr_xyz(const float *verts, const int32_t *idx, int B, int M, int k, float *nbr, hipStream_t s);
int launch_map_term_nbr(const float *verts12, const float *nbr2, const int32_t *idx11, const float *pi_val, const int32_t *pi_idx,
                        int B, int N, int M, int k, int topk, double *partial, hipStream_t s);
int launch_map_term(const float *verts12, const float *verts2, const int32_t *idx11, const int32_t *idx22, const float *pi_val,
                    const int32_t *pi_idx, int B, int N, int M, int k, int topk, double *partial, hipStream_t s, float *resid = nullptr);
bool launch_map_term_lds(const float *verts12, const float *verts2, const int32_t *idx11, const int32_t *idx22, const float *pi_val,
                         const int32_t *pi_idx, int B, int N, int M, int k, int topk, double *partial, hipStream_t s);
bool map_term_lds_applies(int N, int M, int k);
bool launch_map_term_lds_pair(const float *verts12, const float *verts21, const float *verts1, const float *verts2, const int32_t *idx11,
                              const int32_t *idx22, const float *val12, const int32_t *pidx12, const float *val21, const int32_t *pidx21, int B,
                              int N, int M, int k, int topk, double *partial12, double *partial21, hipStream_t s);
bool launch_apply3_pair(const float *val12, const int32_t *idx12, const float *verts2, float *verts12, int32_t *T12, const float *val21,
                        const int32_t *idx21, const float *verts1, float *verts21, int32_t *T21, int B, int N, int M, hipStream_t s);
int launch_dg_build(const float *xyz, int B, int N, const int32_t *start, int32_t *nodes_idx, int32_t *ring, int32_t *infl_idx,
                    float *dists, float *weights, double *sigma, double *nnd, const GridBuf &gverts, const GridBuf &gnodes,
                    bool build_gverts, hipStream_t s, hipEvent_t gverts_ready = nullptr);
int launch_dg_warp(const float *xyz, int B, int N, const int32_t *nodes_idx, const int32_t *ring, const int32_t *infl_idx,
                   const float *weights, const float *def9, float *R, float *T, float *warped, float *arap, int arap_stride,
                   float *sr, hipStream_t s);
bool launch_dg_warp_pair(const float *xyz1, const float *xyz2, int B, int N, int M, const int32_t *const nodes[2], const int32_t *const ring[2],
                         const int32_t *const infl[2], const float *const weights[2], const float *def9_12, const float *def9_21, float *R12,
                         float *R21, float *T12, float *T21, float *warped12, float *warped21, float *arap12, float *arap21, int arap_stride,
                         hipStream_t s);
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
    GridBuf gv1, gn1, gv2, gw, g12;
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
    w.gv1 = grid_carve(ar, B, N);
    w.gn1 = grid_carve(ar, B, Nn);
    w.gv2 = grid_carve(ar, B, M);
    w.gw = grid_carve(ar, B, N);
    w.g12 = grid_carve(ar, B, N);
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
    launch_dg_build(verts1, B, N, fps_start, w.nodes, w.ring, w.infl, w.dists, w.weights, nullptr, w.nnd, w.gv1, w.gn1, true, s);
    // soft correspondence Pi_12 (top-10) and the arg-max map
    rc = dvm_softcorr_fwd_f32(feat1, feat2, B, N, M, 128, neg_alpha, topk, w.pval, w.pidx, nullptr, nullptr, 0, w.sc_ws,
                              w.sc_bytes, s);
    if (rc != DVM_OK) return rc;
    hipLaunchKernelGGL(take_col0_kernel, dim3((B * N + 255) / 256), dim3(256), 0, s, w.pidx, B * N, topk, T12);
    // verts12 = Pi_12 @ verts2
    rc = dvm_softcorr_apply_f32(w.pval, w.pidx, verts2, B, N, M, topk, 3, verts12, s);
    if (rc != DVM_OK) return rc;
    // xyz neighbourhoods (grids: verts1's was built with the graph)
    launch_grid_knn_self(w.gv1, B, k, w.idx11, s);
    launch_grid_build(verts2, B, M, nullptr, w.gv2, s);
    launch_grid_knn_self(w.gv2, B, k, w.idx22, s);
    // Deformer -> per-node [t, r6]
    rc = launch_deformer(feat1, feat2, verts1, verts12, w.idx11, w.idx22, w.pval, w.pidx, w.nodes, B, N, M, Nn, k, topk, conv_w,
                         conv_b, W0, b0, W1, b1, W2, b2, W3, b3, w.def9, 0, w.df_ws, w.df_bytes, s);
    if (rc != DVM_OK) return rc;
    // embedded-deformation warp + ARAP (losses[:,1])
    launch_dg_warp(verts1, B, N, w.nodes, w.ring, w.infl, w.weights, w.def9, w.R, w.T, warped, losses + 2, 6, nullptr, s);
    // chamfer(warped, verts2) -> losses[:,0:2]; chamfer(verts12, verts2) -> losses[:,3:5]
    launch_grid_build(warped, B, N, nullptr, w.gw, s);
    launch_grid_build(verts12, B, N, nullptr, w.g12, s);
    {
        GridBuf qg[2] = {w.gw, w.gv2}, tg[2] = {w.gv2, w.gw};
        float *dout[2] = {w.d1, w.d2};
        launch_grid_chamfer(qg, tg, dout, nullptr, 2, B, s);
        launch_mean(w.d1, B, N, 1.f, losses, 6, 0, 0, s);
        launch_mean(w.d2, B, M, 1.f, losses, 6, 1, 0, s);
        GridBuf qg2[2] = {w.g12, w.gv2}, tg2[2] = {w.gv2, w.g12};
        launch_grid_chamfer(qg2, tg2, dout, nullptr, 2, B, s);
        launch_mean(w.d1, B, N, 1.f, losses, 6, 3, 0, s);
        launch_mean(w.d2, B, M, 1.f, losses, 6, 4, 0, s);
    }
    if (with_map) {
        if (!launch_map_term_lds(verts12, verts2, w.idx11, w.idx22, w.pval, w.pidx, B, N, M, k, topk, w.partial, s))
            launch_map_term(verts12, verts2, w.idx11, w.idx22, w.pval, w.pidx, B, N, M, k, topk, w.partial, s);
        launch_reduce_partials(w.partial, B, map_term_blocks(N, k), 1.f, losses, 6, 5, s);
    } else {
        // losses[:,5] = 0 — a strided fill through the mean kernel's overwrite path
        launch_mean(w.d1, B, 1, 0.f, losses, 6, 5, 0, s);
    }
    DVM_CHECK_LAUNCH("pair_direction");
    return DVM_OK;
}

// ================================================================ both directions at once
// GraphDeformLoss_Neural.forward's deformation part for B pairs (reference models/loss.py:1401-1411):
// graphs of both clouds, Pi_12 and Pi_21 (one launch), xyz kNN and pooled features once per cloud,
// one Deformer-MLP launch over all nodes of both directions, warps, 4 Chamfer terms in one grouped
// launch, map terms.  When N == M the per-cloud stages run as single launches over 2B shapes.
namespace dvm {
int launch_softcorr_both(const float *f1, const float *f2, const float *n1, const float *n2, int B, int N, int M,
                         float neg_alpha, float *val12, int32_t *idx12, float *val21, int32_t *idx21, hipStream_t s);
void launch_rownorm2(const float *x, int rows, int K, float *out, hipStream_t s);
size_t softcorr_pair_ws_bytes(int B, int N, int M);
int launch_softcorr_pair(const float *f1, const float *f2, float *n1, float *n2, int B, int N, int M, float neg_alpha, float *val12,
                         int32_t *idx12, float *val21, int32_t *idx21, void *ws, size_t ws_bytes, hipStream_t s);
void launch_pool_all(const float *feat, const int32_t *idx, int B, int P, int k, const float *cw, const float *cb, float *out,
                     hipStream_t s, const int32_t *order = nullptr);
void launch_assemble_pooled(const float *vsrc, const float *vcorr, const float *gsrc, const float *gtgt, const float *pi_val,
                            const int32_t *pi_idx, const int32_t *fps, int B, int N, int M, int Nn, float *z, hipStream_t s, const int *gate = nullptr);
void launch_assemble_pooled_planes(const float *vsrc, const float *vcorr, const float *gsrc, const float *gtgt, const float *pi_val,
                                   const int32_t *pi_idx, const int32_t *fps, int B, int N, int M, int Nn, void *zp, hipStream_t s);
void launch_assemble_pooled_pair(const float *verts1, const float *verts2, const float *verts12, const float *verts21, const float *g1,
                                 const float *g2, const float *val12, const int32_t *idx12, const float *val21, const int32_t *idx21,
                                 const int32_t *nodes1, const int32_t *nodes2, int B, int N, int M, void *z12, void *z21, bool planes,
                                 const int *gate, hipStream_t s);
size_t mlp_pack_floats();
size_t mlp_zplane_bytes(int rows);
size_t mlp_zplane_row_bytes();
void launch_mlp_rows(const float *z, int rows, const float *W0, const float *b0, const float *W1, const float *b1, const float *W2,
                     const float *b2, const float *W3, const float *b3, float *wp, float *out, hipStream_t s, int variant, void *zp);
const int *launch_mlp_planes(const void *zp, int rows, const float *W0, const float *b0, const float *W1, const float *b1, const float *W2,
                             const float *b2, const float *W3, const float *b3, float *wp, float *out, hipStream_t s);
void launch_mlp_fallback(const float *z, int rows, const float *W0, const float *b0, const float *W1, const float *b1, const float *W2,
                         const float *b2, const float *W3, const float *b3, float *wp, float *out, hipStream_t s, const int *flag);
int launch_chamfer_grouped(const float *const *a, const float *const *b, const int *Na, const int *Nb, float *const *dout,
                           int ngroups, int B, hipStream_t s);

struct Pair2Ws {
    float *vcat;                                  // [2B][N][3] when N == M
    int32_t *startcat;                            // [2B]
    int32_t *nodes[2], *ring[2], *infl[2], *idxk[2], *pidx[2];
    float *dists[2], *weights[2], *pval[2], *nrm[2], *gall[2];
    double *nnd[2], *partial[2];
    float *z, *def9, *R, *T, *wp;
    char *zp;          // the Deformer rows in the plane form (dvm_mlp_f16.h)
    float *nbrxyz[2];  // [B][P][10][3] coordinates of every point's xyz neighbours (map term)
    char *k1ws;  // soft-correspondence scratch (fp16 planes, candidates, flags)
    size_t k1ws_bytes;
    float *cd[8];
    GridBuf gv[2], gn[2], gw[2], gc[2];  // vertices, nodes, warped clouds, Pi-mapped clouds
    GridBuf gvcat, gncat;                // 2B-shape grids when N == M
};

static size_t carve_pair2(Arena &ar, int B, int N, int M, Pair2Ws &w) {
    const int P[2] = {N, M};
    w.vcat = ar.take<float>((size_t)2 * B * (N > M ? N : M) * 3);
    w.startcat = ar.take<int32_t>((size_t)2 * B);
    // per-cloud arrays are carved back to back so that for N == M side 1 follows side 0 contiguously
    for (int sd = 0; sd < 2; ++sd) w.nodes[sd] = ar.take<int32_t>((size_t)B * (P[sd] / 2));
    for (int sd = 0; sd < 2; ++sd) w.ring[sd] = ar.take<int32_t>((size_t)B * (P[sd] / 2) * 9);
    for (int sd = 0; sd < 2; ++sd) w.infl[sd] = ar.take<int32_t>((size_t)B * P[sd] * 3);
    for (int sd = 0; sd < 2; ++sd) w.dists[sd] = ar.take<float>((size_t)B * P[sd] * 3);
    for (int sd = 0; sd < 2; ++sd) w.weights[sd] = ar.take<float>((size_t)B * P[sd] * 3);
    for (int sd = 0; sd < 2; ++sd) w.nnd[sd] = ar.take<double>((size_t)B * P[sd]);
    for (int sd = 0; sd < 2; ++sd) w.idxk[sd] = ar.take<int32_t>((size_t)B * P[sd] * 10);
    for (int sd = 0; sd < 2; ++sd) w.pval[sd] = ar.take<float>((size_t)B * P[sd] * 10);
    for (int sd = 0; sd < 2; ++sd) w.pidx[sd] = ar.take<int32_t>((size_t)B * P[sd] * 10);
    for (int sd = 0; sd < 2; ++sd) w.nrm[sd] = ar.take<float>((size_t)B * P[sd]);
    for (int sd = 0; sd < 2; ++sd) w.gall[sd] = ar.take<float>((size_t)B * P[sd] * 128);
    for (int sd = 0; sd < 2; ++sd) w.partial[sd] = ar.take<double>((size_t)B * map_term_blocks(P[sd], 10));
    const size_t rows = (size_t)B * (N / 2) + (size_t)B * (M / 2);
    w.z = ar.take<float>(rows * 264);
    w.zp = ar.take<char>(mlp_zplane_bytes((int)rows));
    w.def9 = ar.take<float>(rows * 9);
    w.R = ar.take<float>(rows * 9);
    w.T = ar.take<float>(rows * 3);
    w.wp = ar.take<float>(mlp_pack_floats());
    for (int sd = 0; sd < 2; ++sd) w.nbrxyz[sd] = ar.take<float>((size_t)B * P[sd] * 30);
    w.k1ws_bytes = softcorr_pair_ws_bytes(B, N, M);
    w.k1ws = ar.take<char>(w.k1ws_bytes);
    const int cdn[8] = {N, M, N, M, M, N, M, N};
    for (int q = 0; q < 8; ++q) w.cd[q] = ar.take<float>((size_t)B * cdn[q]);
    if (N == M) {
        w.gvcat = grid_carve(ar, 2 * B, N);
        w.gncat = grid_carve(ar, 2 * B, N / 2);
        for (int sd = 0; sd < 2; ++sd) {
            w.gv[sd] = grid_slice(w.gvcat, sd * B);
            w.gn[sd] = grid_slice(w.gncat, sd * B);
        }
    } else {
        for (int sd = 0; sd < 2; ++sd) {
            w.gv[sd] = grid_carve(ar, B, P[sd]);
            w.gn[sd] = grid_carve(ar, B, P[sd] / 2);
        }
    }
    for (int sd = 0; sd < 2; ++sd) {
        w.gw[sd] = grid_carve(ar, B, P[sd]);
        w.gc[sd] = grid_carve(ar, B, P[sd]);
    }
    return ar.off;
}

// arrays carved back to back are contiguous only if every per-side size is a multiple of the arena
// alignment (256 B); otherwise the N == M fast path is not used.
static bool contiguous_sides(int B, int N) {
    return ((size_t)B * (N / 2) * sizeof(int32_t)) % 256 == 0 && ((size_t)B * N * sizeof(float)) % 256 == 0;
}
}  // namespace dvm

DVM_EXPORT size_t dvm_pair_workspace_bytes(int B, int N, int M) {
    Arena ar(nullptr, 0);
    Pair2Ws w;
    return carve_pair2(ar, B, N, M, w);
}

// 1 (default): the geometry chain runs on a helper stream next to the soft-correspondence chain; 0: one stream
static int g_pair_overlap = options().pair_overlap;

DVM_EXPORT int dvm_pair_set_overlap(int on) {
    const int prev = g_pair_overlap;
    g_pair_overlap = on ? 1 : 0;
    return prev;
}

// The coordinate-only part of the pair path — both clouds' deformation graphs (FPS nodes, rings, skinning), their uniform grids, the
// xyz kNN and (without the LDS map term) the neighbours' coordinates — written into the workspace `w`.  `s` carries the FPS chain;
// with a context `cx` the vertex grid and the xyz kNN, which need no FPS, run beside it on cx->side2 (joined into `s` before this
// returns).  `pool` (feature pointers given): the Deformer's pooled features are made right behind the xyz kNN on that stream.
// Reference: lib/deformation_graph_point.py:18-33, 177-201; models/loss.py:1325-1337 (graphs), :97-101 (xyz kNN).
namespace dvm {
static bool pair_geometry(PairCtx *cx, hipStream_t s, const Pair2Ws &w, const float *verts1, const float *verts2, int B, int N, int M,
                          const int32_t *start1, const int32_t *start2, bool both, bool gather_nbr, const float *pool_feat1,
                          const float *pool_feat2, const float *conv_w, const float *conv_b) {
    const int P[2] = {N, M};
    const float *verts[2] = {verts1, verts2};
    const int32_t *start[2] = {start1, start2};
    bool pooled = false;
    if (both) {
        (void)hipMemcpyAsync(w.vcat, verts1, (size_t)B * N * 3 * sizeof(float), hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(w.vcat + (size_t)B * N * 3, verts2, (size_t)B * N * 3 * sizeof(float), hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(w.startcat, start1, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(w.startcat + B, start2, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, s);
        if (cx) {
            // FPS is N / 2 dependent steps on one workgroup per cloud (0.7 ms whatever the batch) and everything behind it on this
            // stream waits for it — but the vertex grid, the xyz kNN and the pooled features need the coordinates / features only:
            // they go on the SECOND helper stream, beside FPS
            (void)hipEventRecord(cx->ev_aux, s);                      // the concatenated coordinates are in place
            (void)hipStreamWaitEvent(cx->side2, cx->ev_aux, 0);
            launch_grid_build(w.vcat, 2 * B, N, nullptr, w.gvcat, cx->side2);
            (void)hipEventRecord(cx->ev_aux, cx->side2);              // (re-used: the vertex grid is built)
            launch_grid_knn_self(w.gvcat, 2 * B, 10, w.idxk[0], cx->side2);
            if (pool_feat1) {
                launch_pool_all(pool_feat1, w.idxk[0], B, N, 10, conv_w, conv_b, w.gall[0], cx->side2, w.gv[0].ids);
                launch_pool_all(pool_feat2, w.idxk[1], B, M, 10, conv_w, conv_b, w.gall[1], cx->side2, w.gv[1].ids);
                pooled = true;
            }
            (void)hipEventRecord(cx->ev_join2, cx->side2);
            launch_dg_build(w.vcat, 2 * B, N, w.startcat, w.nodes[0], w.ring[0], w.infl[0], w.dists[0], w.weights[0], nullptr,
                            w.nnd[0], w.gvcat, w.gncat, false, s, cx->ev_aux);
            (void)hipStreamWaitEvent(s, cx->ev_join2, 0);
        } else {
            launch_dg_build(w.vcat, 2 * B, N, w.startcat, w.nodes[0], w.ring[0], w.infl[0], w.dists[0], w.weights[0], nullptr,
                            w.nnd[0], w.gvcat, w.gncat, true, s);
            launch_grid_knn_self(w.gvcat, 2 * B, 10, w.idxk[0], s);
        }
    } else {
        // N != M (or sizes that do not tile the arena): one chain per cloud set.  FPS is a one-workgroup-per-cloud
        // sequential kernel, so the two chains go on two streams and overlap each other as well.
        if (cx) {
            (void)hipEventRecord(cx->ev_aux, s);
            (void)hipStreamWaitEvent(cx->side2, cx->ev_aux, 0);
        }
        for (int sd = 0; sd < 2; ++sd) {
            const hipStream_t cs = (cx && sd == 1) ? cx->side2 : s;
            launch_dg_build(verts[sd], B, P[sd], start[sd], w.nodes[sd], w.ring[sd], w.infl[sd], w.dists[sd], w.weights[sd],
                            nullptr, w.nnd[sd], w.gv[sd], w.gn[sd], true, cs);
            launch_grid_knn_self(w.gv[sd], B, 10, w.idxk[sd], cs);
        }
        if (cx) {
            (void)hipEventRecord(cx->ev_join2, cx->side2);
            (void)hipStreamWaitEvent(s, cx->ev_join2, 0);
        }
    }
    if (gather_nbr) {
        launch_gather_nbr_xyz(verts2, w.idxk[1], B, M, 10, w.nbrxyz[1], s);
        launch_gather_nbr_xyz(verts1, w.idxk[0], B, N, 10, w.nbrxyz[0], s);
    }
    return pooled;
}
}  // namespace dvm

// reuse_geometry != 0: the coordinate-only products of an earlier call on the SAME workspace with the SAME coordinates and FPS
// starts — dvm_pair_geometry_f32, or a dvm_pair_fwd[_cached]_f32 call — are still in `ws` and are used as they are (the per-shape
// graph cache of SURVEY 8f-2 / 8d and the geometry prefetch of a pipelined caller; the reference rebuilds them on every call,
// models/loss.py:1325-1337); everything that depends on the features runs as always
static int pair_fwd_impl(const float *feat1, const float *feat2, const float *verts1, const float *verts2, int B, int N, int M,
                         float neg_alpha, const int32_t *start1, const int32_t *start2, const float *conv_w, const float *conv_b,
                         const float *W0, const float *b0, const float *W1, const float *b1, const float *W2, const float *b2,
                         const float *W3, const float *b3, int with_map, float *warped12, float *verts12, int32_t *T12, float *losses12,
                         float *warped21, float *verts21, int32_t *T21, float *losses21, void *ws, size_t ws_bytes, void *stream,
                         int reuse_geometry) {
    DVM_REQUIRE(feat1 && feat2 && verts1 && verts2 && start1 && start2, "dvm_pair_fwd_f32: null input pointer");
    DVM_REQUIRE(warped12 && verts12 && T12 && losses12 && warped21 && verts21 && T21 && losses21,
                "dvm_pair_fwd_f32: null output pointer");
    DVM_REQUIRE(conv_w && conv_b && W0 && b0 && W1 && b1 && W2 && b2 && W3 && b3, "dvm_pair_fwd_f32: null weight pointer");
    DVM_REQUIRE(B >= 1 && N >= 20 && M >= 20, "dvm_pair_fwd_f32: bad sizes (B=%d N=%d M=%d)", B, N, M);
    DVM_REQUIRE(neg_alpha < 0.f, "dvm_pair_fwd_f32: neg_alpha must be negative");
    Arena ar(ws, ws_bytes);
    Pair2Ws w;
    carve_pair2(ar, B, N, M, w);
    if (!ar.ok()) {
        set_error("dvm_pair_fwd_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    int rc;
    const bool both = (N == M) && contiguous_sides(B, N);
    // ---- per-cloud geometry: graph + xyz kNN.  It depends on the coordinates only and is latency-bound (FPS: N/2
    // sequential steps per cloud), the soft correspondence on the features only and is ALU-bound: the two chains run
    // concurrently, the geometry on a helper stream forked from / joined back into the caller's stream by events.
    // The helper streams and events belong to a context made by dvm_pair_init for (current device, caller stream):
    // nothing is created here, two caller streams / host threads / devices never share an event, and without a context
    // (or with dvm_pair_set_overlap(0)) everything runs on the caller's stream.
    PairCtx *cx = g_pair_overlap != 0 ? pair_ctx_find(s) : nullptr;
    const bool overlap = cx != nullptr;
    const hipStream_t caller = s;
    if (overlap) {
        (void)hipEventRecord(cx->ev_fork, caller);
        (void)hipStreamWaitEvent(cx->side, cx->ev_fork, 0);
        s = cx->side;
    }
    const bool map_lds = map_term_lds_applies(N, M, 10) && map_term_lds_applies(M, N, 10);   // (the target side in LDS: no neighbour tables)
    bool pooled = false;   // the pooled features were made on the second helper stream
    if (!reuse_geometry)
        pooled = pair_geometry(cx, s, w, verts1, verts2, B, N, M, start1, start2, both, with_map && !map_lds, feat1, feat2, conv_w, conv_b);
    // still on the geometry side: what depends on the xyz kNN and the input features only — the Deformer's pooled features
    // (once per cloud, points in grid-cell order) — so that the L2-bound gathers run next to the ALU-bound sweep instead of after it
    if (!pooled) {
        launch_pool_all(feat1, w.idxk[0], B, N, 10, conv_w, conv_b, w.gall[0], s, w.gv[0].ids);
        launch_pool_all(feat2, w.idxk[1], B, M, 10, conv_w, conv_b, w.gall[1], s, w.gv[1].ids);
    }
    if (overlap) {
        (void)hipEventRecord(cx->ev_join, cx->side);
        s = caller;
    }
    // from here to the join the caller's stream must wait for the helper streams on EVERY exit path: the caller may reuse
    // the workspace as soon as this function returns
    auto fail = [&](int code) {
        if (overlap) (void)hipStreamWaitEvent(caller, cx->ev_join, 0);
        return code;
    };
    // ---- soft correspondence, both directions in one launch
    rc = launch_softcorr_pair(feat1, feat2, w.nrm[0], w.nrm[1], B, N, M, neg_alpha, w.pval[0], w.pidx[0], w.pval[1], w.pidx[1], w.k1ws,
                              w.k1ws_bytes, s);
    if (rc != DVM_OK) return fail(rc);
    if (!launch_apply3_pair(w.pval[0], w.pidx[0], verts2, verts12, T12, w.pval[1], w.pidx[1], verts1, verts21, T21, B, N, M, s)) {
        hipLaunchKernelGGL(take_col0_kernel, dim3((B * N + 255) / 256), dim3(256), 0, s, w.pidx[0], B * N, 10, T12);
        hipLaunchKernelGGL(take_col0_kernel, dim3((B * M + 255) / 256), dim3(256), 0, s, w.pidx[1], B * M, 10, T21);
        rc = dvm_softcorr_apply_f32(w.pval[0], w.pidx[0], verts2, B, N, M, 10, 3, verts12, s);
        if (rc != DVM_OK) return fail(rc);
        rc = dvm_softcorr_apply_f32(w.pval[1], w.pidx[1], verts1, B, M, N, 10, 3, verts21, s);
        if (rc != DVM_OK) return fail(rc);
    }
    if (overlap) (void)hipStreamWaitEvent(caller, cx->ev_join, 0);  // join: everything below needs the graphs / kNN
    // ---- second fork (round 6): what needs the soft correspondence but NOT the Deformer — the Chamfer terms of the Pi-mapped clouds
    // (verts12 / verts21 against the targets) and the map terms — runs on the helper stream beside assemble -> MLP -> warp -> the warped
    // clouds' Chamfer terms.  At a strong-scaling rank's batch (32 - 64 pairs) none of these kernels fills the chip and the step is
    // the LENGTH of this chain; at 512 pairs it is neutral.  Same kernels, same arguments per group: same bits as the one-stream order.
    hipStream_t s2 = s;
    if (overlap) {
        (void)hipEventRecord(cx->ev_fork, caller);
        (void)hipStreamWaitEvent(cx->side, cx->ev_fork, 0);
        s2 = cx->side;
    }
    {
        const float *const gx[2] = {verts12, verts21};
        const int gn[2] = {N, M};
        const GridBuf gg[2] = {w.gc[0], w.gc[1]};
        launch_grid_build_sets(gx, gn, gg, 2, B, s2);
        const GridBuf qg[4] = {w.gc[0], w.gv[1], w.gc[1], w.gv[0]};
        const GridBuf tg[4] = {w.gv[1], w.gc[0], w.gv[0], w.gc[1]};
        float *const cd[4] = {w.cd[2], w.cd[3], w.cd[6], w.cd[7]};
        launch_grid_chamfer(qg, tg, cd, nullptr, 4, B, s2);
    }
    // ---- map terms (losses[:,5])
    if (with_map) {
        if (map_lds) {
            launch_map_term_lds_pair(verts12, verts21, verts1, verts2, w.idxk[0], w.idxk[1], w.pval[0], w.pidx[0], w.pval[1], w.pidx[1], B, N, M, 10,
                                     10, w.partial[0], w.partial[1], s2);
        } else {
            launch_map_term_nbr(verts12, w.nbrxyz[1], w.idxk[0], w.pval[0], w.pidx[0], B, N, M, 10, 10, w.partial[0], s2);
            launch_map_term_nbr(verts21, w.nbrxyz[0], w.idxk[1], w.pval[1], w.pidx[1], B, M, N, 10, 10, w.partial[1], s2);
        }
        launch_reduce_partials(w.partial[0], B, map_term_blocks(N, 10), 1.f, losses12, 6, 5, s2);
        launch_reduce_partials(w.partial[1], B, map_term_blocks(M, 10), 1.f, losses21, 6, 5, s2);
    } else {
        launch_mean(w.cd[0], B, 1, 0.f, losses12, 6, 5, 0, s2);
        launch_mean(w.cd[0], B, 1, 0.f, losses21, 6, 5, 0, s2);
    }
    if (overlap) (void)hipEventRecord(cx->ev_join2, cx->side);
    // ---- Deformer: z for both directions from the pooled features (made above), one MLP launch
    const int Nn1 = N / 2, Nn2 = M / 2;
    float *z21 = w.z + (size_t)B * Nn1 * 264;
    const int rows = B * (Nn1 + Nn2);
    {
        // the rows straight in the plane form the MLP kernel stages by LDS-DMA; the fp32 rows are written (and the bf16x3 kernel
        // runs) only if that kernel's range flag comes up: three gated launches that return at once otherwise
        char *zp21 = w.zp + (size_t)B * Nn1 * mlp_zplane_row_bytes();
        // (both directions per launch: the plane rows, and — gated — the fp32 rows)
        launch_assemble_pooled_pair(verts1, verts2, verts12, verts21, w.gall[0], w.gall[1], w.pval[0], w.pidx[0], w.pval[1], w.pidx[1], w.nodes[0],
                                    w.nodes[1], B, N, M, w.zp, zp21, true, nullptr, s);
        const int *flag = launch_mlp_planes(w.zp, rows, W0, b0, W1, b1, W2, b2, W3, b3, w.wp, w.def9, s);
        launch_assemble_pooled_pair(verts1, verts2, verts12, verts21, w.gall[0], w.gall[1], w.pval[0], w.pidx[0], w.pval[1], w.pidx[1], w.nodes[0],
                                    w.nodes[1], B, N, M, w.z, z21, false, flag, s);
        launch_mlp_fallback(w.z, rows, W0, b0, W1, b1, W2, b2, W3, b3, w.wp, w.def9, s, flag);
    }
    // ---- ED warp + ARAP (losses[:,2])
    float *def21 = w.def9 + (size_t)B * Nn1 * 9, *R21 = w.R + (size_t)B * Nn1 * 9, *T21v = w.T + (size_t)B * Nn1 * 3;
    if (!launch_dg_warp_pair(verts1, verts2, B, N, M, w.nodes, w.ring, w.infl, w.weights, w.def9, def21, w.R, R21, w.T, T21v, warped12, warped21,
                             losses12 + 2, losses21 + 2, 6, s)) {
        launch_dg_warp(verts1, B, N, w.nodes[0], w.ring[0], w.infl[0], w.weights[0], w.def9, w.R, w.T, warped12, losses12 + 2, 6,
                       nullptr, s);
        launch_dg_warp(verts2, B, M, w.nodes[1], w.ring[1], w.infl[1], w.weights[1], def21, R21, T21v, warped21, losses21 + 2, 6,
                       nullptr, s);
    }
    // ---- the warped clouds' Chamfer terms, then the means of all eight
    {
        const float *const gx[2] = {warped12, warped21};
        const int gn[2] = {N, M};
        const GridBuf gg[2] = {w.gw[0], w.gw[1]};
        launch_grid_build_sets(gx, gn, gg, 2, B, s);
        const GridBuf qg[4] = {w.gw[0], w.gv[1], w.gw[1], w.gv[0]};
        const GridBuf tg[4] = {w.gv[1], w.gw[0], w.gv[0], w.gw[1]};
        float *const cd[4] = {w.cd[0], w.cd[1], w.cd[4], w.cd[5]};
        launch_grid_chamfer(qg, tg, cd, nullptr, 4, B, s);
        if (overlap) (void)hipStreamWaitEvent(caller, cx->ev_join2, 0);   // the helper stream's Chamfer and map terms
        const int Na[8] = {N, M, N, M, M, N, M, N};
        float *const L[8] = {losses12, losses12, losses12, losses12, losses21, losses21, losses21, losses21};
        const int off[8] = {0, 1, 3, 4, 0, 1, 3, 4};
        launch_mean_grouped(w.cd, Na, L, off, 8, B, 1.f, 6, s);   // (one launch for the eight means)
    }
    DVM_CHECK_LAUNCH("pair_fwd");
    return DVM_OK;
}

DVM_EXPORT int dvm_pair_fwd_f32(const float *feat1, const float *feat2, const float *verts1, const float *verts2, int B, int N,
                                int M, float neg_alpha, const int32_t *start1, const int32_t *start2, const float *conv_w,
                                const float *conv_b, const float *W0, const float *b0, const float *W1, const float *b1,
                                const float *W2, const float *b2, const float *W3, const float *b3, int with_map, float *warped12,
                                float *verts12, int32_t *T12, float *losses12, float *warped21, float *verts21, int32_t *T21,
                                float *losses21, void *ws, size_t ws_bytes, void *stream) {
    return pair_fwd_impl(feat1, feat2, verts1, verts2, B, N, M, neg_alpha, start1, start2, conv_w, conv_b, W0, b0, W1, b1, W2, b2, W3, b3,
                         with_map, warped12, verts12, T12, losses12, warped21, verts21, T21, losses21, ws, ws_bytes, stream, 0);
}

DVM_EXPORT int dvm_pair_fwd_cached_f32(const float *feat1, const float *feat2, const float *verts1, const float *verts2, int B, int N,
                                       int M, float neg_alpha, const int32_t *start1, const int32_t *start2, const float *conv_w,
                                       const float *conv_b, const float *W0, const float *b0, const float *W1, const float *b1,
                                       const float *W2, const float *b2, const float *W3, const float *b3, int with_map,
                                       float *warped12, float *verts12, int32_t *T12, float *losses12, float *warped21,
                                       float *verts21, int32_t *T21, float *losses21, void *ws, size_t ws_bytes, int reuse_geometry,
                                       void *stream) {
    return pair_fwd_impl(feat1, feat2, verts1, verts2, B, N, M, neg_alpha, start1, start2, conv_w, conv_b, W0, b0, W1, b1, W2, b2, W3, b3,
                         with_map, warped12, verts12, T12, losses12, warped21, verts21, T21, losses21, ws, ws_bytes, stream,
                         reuse_geometry ? 1 : 0);
}

// The coordinate-only half of dvm_pair_fwd_f32 as a call of its own: a pipelined caller enqueues the geometry of batch t + 1 on
// another stream while batch t's feature-dependent half (soft correspondence -> Deformer -> warp -> Chamfer) runs, and then calls
// dvm_pair_fwd_cached_f32(..., reuse_geometry = 1) on the SAME workspace.  Nothing is cached: every batch's graphs are still built
// once per step, only earlier (the reference builds them inside the criterion call, models/loss.py:1325-1337, from coordinates that
// are known as soon as the batch is loaded).  With a dvm_pair_init context for `stream` the vertex grid + xyz kNN run beside the FPS
// chain on the context's second helper stream; both are joined into `stream` before this returns.
DVM_EXPORT int dvm_pair_geometry_f32(const float *verts1, const float *verts2, int B, int N, int M, const int32_t *start1,
                                     const int32_t *start2, int with_map, void *ws, size_t ws_bytes, void *stream) {
    DVM_REQUIRE(verts1 && verts2 && start1 && start2, "dvm_pair_geometry_f32: null input pointer");
    DVM_REQUIRE(B >= 1 && N >= 20 && M >= 20, "dvm_pair_geometry_f32: bad sizes (B=%d N=%d M=%d)", B, N, M);
    Arena ar(ws, ws_bytes);
    Pair2Ws w;
    carve_pair2(ar, B, N, M, w);
    if (!ar.ok()) {
        set_error("dvm_pair_geometry_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    PairCtx *cx = g_pair_overlap != 0 ? pair_ctx_find(s) : nullptr;
    const bool both = (N == M) && contiguous_sides(B, N);
    const bool map_lds = map_term_lds_applies(N, M, 10) && map_term_lds_applies(M, N, 10);
    pair_geometry(cx, s, w, verts1, verts2, B, N, M, start1, start2, both, with_map && !map_lds, nullptr, nullptr, nullptr, nullptr);
    DVM_CHECK_LAUNCH("pair_geometry");
    return DVM_OK;
}
