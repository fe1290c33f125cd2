// dvm_criterion_train.hip — the deformation part of GraphDeformLoss_Neural.forward in TRAINING (reference models/loss.py:1228-1296
// deform() for both directions, 1401-1432 their weighting; the Deformer models/model.py:454-478 with its MLP 433-452) as ONE pair of
// C-ABI calls: dvm_criterion_train_fwd_f32 / dvm_criterion_train_bwd_f32.
//
// The B pairs of a batch enter as ONE batch of P = 2B directional pairs — [(1 -> 2) x B | (2 -> 1) x B], the layout LG-Net's merged
// training call and the batched geometry() produce anyway (features / coordinates / graphs / xyz-kNN of the B first shapes followed by
// those of the B second shapes); the TARGET of pair p is shape (p + B) mod P.  Every step is one of the library's launches or a small
// kernel of this file, enqueued without Python in between (the autograd path enqueued ~300 launches through ~40 autograd nodes per
// step).  Forward keeps what the backward needs in a caller-provided arena; the output is the table terms [P][7] =
// [map numerator, cd(warped -> target) side means (2), cd(verts12 -> target) side means (2), ARAP, dist term of shape p (models/loss.py:
// 1351-1396; optional, on a helper stream beside the deformation part)]; the caller's weighting of the table
// (sums / batch means, models/loss.py:1413-1432) stays with autograd: one small matrix product.  Backward takes d terms and writes
// d feat [P][N][C] and ADDS the Deformer's parameter gradients into caller-provided buffers.
//
//   forward                                                       backward (reverse order)
//   Pi~ = softcorr(feat, feat^T)         (K1, with row statistics)   dvm_softcorr_bwd_f32 on the summed d val
//   verts12 = Pi~ verts^T                                           d val += d verts12 . verts^T[col]
//   gp = pool(feat, kNN) (+ its half-swapped copy gp^T)             reversed kNN lists: d feat += w[s] d gp; d w, d bias
//   z[node] = [v, gp, verts12, Pi~ gp^T]                            rows scattered back; reversed Pi~ lists (node rows) -> d gp^T, d val
//   def9 = MLP(z)   (dvm_linear_f32, ELU in the epilogue)           dvm_linear_wgrad_f32 / dvm_linear_f32 with the roles swapped
//   warped, ARAP = warp(def9)            (one workgroup per shape)  dvm_dg_warp_arap_bwd + rot6d backward
//   cd = Chamfer side means; map = sum |lhs - rhs|^2                source-side Chamfer gradient; map residuals kept in the arena
//   dist[p] = sum_n 1 - |cos(x_n, y_n)| (x_j, y_j kept)             weights W from the kept x, y; d feat = diag(colsum W) feat - W^T fa,
//                                                                   d feat[a_n] += rowsum(W)_n fa_n - (W feat)_n on the library's GEMM kernels
#include "dvm_common.h"

namespace dvm {

// ---- the library's own launchers (dvm_gemm.hip, dvm_geom.hip, dvm_deformer.hip, dvm_graph.hip, dvm_loss_bwd.hip)
void launch_linear(const float *x, const float *w, int B, int N, int K, int Co, int channel_major, const float *bias, const float *res,
                   const float *alpha, const float *beta, float slope, float *y, hipStream_t s, const float *xg, int Cg, const float *post_res,
                   float post_scale);
void launch_pool_all(const float *feat, const int32_t *idx, int B, int P, int k, const float *cw, const float *cb, float *out, hipStream_t s,
                     const int32_t *order);
int launch_dg_warp(const float *xyz, int B, int N, const int32_t *nodes_idx, const int32_t *ring, const int32_t *infl_idx, const float *weights,
                   const float *def9, float *R, float *T, float *warped, float *arap, int arap_stride, float *sr, hipStream_t s);
int launch_mean_grouped(const float *const *in, const int *n, float *const *out, const int *off, int ngroups, int B, float scale, int stride,
                        hipStream_t s);
int launch_reduce_partials(const double *partial, int B, int nparts, float scale, float *out, int stride, int off, hipStream_t s);
int map_term_blocks(int N, int k);
int launch_map_term(const float *verts12, const float *verts2, const int32_t *idx11, const int32_t *idx22, const float *pi_val,
                    const int32_t *pi_idx, int B, int N, int M, int k, int topk, double *partial, hipStream_t s, float *resid);
void launch_rev_csr(const int32_t *idx, int B, long E, int M, int32_t *offs, int32_t *cursor, int32_t *edges, hipStream_t s);
void launch_apply_bwd_dval(const float *pi_val, const int32_t *pi_idx, const float *V, const float *g_out, int B, int N, int M, int topk, int C,
                           float *d_val, hipStream_t s);
void launch_apply_bwd_gather(const float *pi_val, const float *g_out, const int32_t *offs, const int32_t *edges, int B, int N, int M, int topk,
                             int C, float *d_V, hipStream_t s);
void launch_dg_warp_arap_bwd(const float *xyz, int B, int N, const int32_t *nodes_idx, const int32_t *ring, const int32_t *infl_idx,
                             const float *weights, const float *R, const float *T, const float *g_warped, const float *g_arap, int garap_stride,
                             float *d_R, float *d_T, hipStream_t s);
void launch_def9_bwd(const float *def9, const float *dR, const float *dT, int rows, float *ddef9, hipStream_t s);
void launch_chamfer_bwd_src2(const float *a0, const float *a1, const float *b0, const float *b1, const int32_t *i1a, const int32_t *i2a,
                             const int32_t *i1b, const int32_t *i2b, const float *gt, int gstride, int off0, int off1, int B, int N, int M,
                             float *da0, float *da1, hipStream_t s);

 int launch_dist_loss_fwd(const float *feat, const float *dist, const int32_t *anchors, int B, int N, int C, int nA, int k, float *out, int out_stride,
                         int out_off, int32_t *idx_out, float *xsave, float *fa_out, void *ws, size_t ws_bytes, hipStream_t s);
void launch_dist_loss_bwd_weights_saved(const float *xsave, const int32_t *idx, const float *gterm, int gstride, int B, int N, int nA, int k, float *W,
                                        float *rs, hipStream_t s);
void launch_wgrad_batched(const float *gy, const float *x, int nb, long R, int Co, int K, float *dW, hipStream_t s);
void launch_linear_bmm(const float *x, const float *w, int B, int N, int K, int Co, float *y, hipStream_t s);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CT_C = 128;                // feature width (the Deformer's pooled features, models/model.py:459)
constexpr int CT_Z = 2 * CT_C + 6;       // 262: the decoder's input row
constexpr int CT_TERMS = 7;
constexpr int CT_H[3] = {512, 256, 128};
enum { PW_CONV_W = 0, PW_CONV_B, PW_W0, PW_B0, PW_W1, PW_B1, PW_W2, PW_B2, PW_W3, PW_B3, PW_N };
static_assert(PW_N == DVM_CRIT_TRAIN_NPARAMS, "parameter table layout and include/dvm.h disagree");

#define CT_TRY(call)                  \
    do {                              \
        const int rc_ = (call);       \
        if (rc_ != DVM_OK) return rc_; \
    } while (0)

// Work forked onto the context's helper stream writes into the caller's arena: the caller's stream must wait for it on EVERY exit path,
// error returns included (the caller may free or reuse the arena as soon as the call returns).  Armed at the fork, disarmed by the
// regular join; an early return records a fresh join point behind whatever was enqueued on the helper stream and waits for it.
struct HelperJoin {
    PairCtx *cx;
    hipStream_t s;
    bool pending = false;
    HelperJoin(PairCtx *c, hipStream_t st) : cx(c), s(st) {}
    void fork() {
        if (!cx) return;
        (void)hipEventRecord(cx->ev_fork, s);
        (void)hipStreamWaitEvent(cx->side, cx->ev_fork, 0);
        pending = true;
    }
    void join() {
        if (!pending) return;
        (void)hipEventRecord(cx->ev_join, cx->side);
        (void)hipStreamWaitEvent(s, cx->ev_join, 0);
        pending = false;
    }
    ~HelperJoin() { join(); }
};

inline unsigned blocks_for(long n, int cap = 8192) { return (unsigned)((n + 255) / 256 < cap ? (n + 255) / 256 : cap); }

// ---------------------------------------------------------------- small kernels
// dst[a][q] = src[a][(q + B) mod 2B] for up to four arrays of 16-byte elements: the TARGET side of the 2B directional pairs
struct SwapArgs {
    const f32x4 *src[4];
    f32x4 *dst[4];
    long half[4];   // 16-byte elements per half (B shapes)
    int count;
};
__global__ __launch_bounds__(256) void swap_halves_kernel(const SwapArgs a) {
    const int m = blockIdx.y;
    if (m >= a.count) return;
    const long h = a.half[m];
    const f32x4 *s = a.src[m];
    f32x4 *d = a.dst[m];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * h; i += (long)gridDim.x * blockDim.x) d[i] = s[i < h ? i + h : i - h];
}

// z[p][a] = [verts[v] | gp[v] | verts12[v] | sum_t val[v][t] gpT[idx[v][t]]] for v = nodes[p][a] (stride 262), and the node rows of the sparse
// correspondence in compact form (pval_n, pidx_n [P][Nn][topk]) for the backward's reversed lists.  32 lanes per node, 4 channels each.
__global__ __launch_bounds__(256) void assemble_train_kernel(const float *__restrict__ verts, const float *__restrict__ verts12,
                                                             const float *__restrict__ gp, const float *__restrict__ gpT,
                                                             const float *__restrict__ pval, const int32_t *__restrict__ pidx,
                                                             const int32_t *__restrict__ nodes, int N, int M, int Nn, int topk, float *__restrict__ z,
                                                             float *__restrict__ pval_n, int32_t *__restrict__ pidx_n) {
    const int p = blockIdx.y;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)Nn * 32) return;
    const int a = (int)(g >> 5), c4 = (int)(g & 31);
    const int v = nodes[(size_t)p * Nn + a];
    const size_t row = (size_t)p * N + v, nrow = (size_t)p * Nn + a;
    const float *gt = gpT + (size_t)p * M * CT_C;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < topk; ++t) {
        const float w = pval[row * topk + t];
        const int col = pidx[row * topk + t];
        const f32x4 f = *(const f32x4 *)(gt + (size_t)col * CT_C + 4 * c4);
        acc.x = fmaf(w, f.x, acc.x), acc.y = fmaf(w, f.y, acc.y), acc.z = fmaf(w, f.z, acc.z), acc.w = fmaf(w, f.w, acc.w);
        if (c4 == t % 32) pval_n[nrow * topk + t] = w, pidx_n[nrow * topk + t] = col;
    }
    const f32x4 gs = *(const f32x4 *)(gp + row * CT_C + 4 * c4);
    float *zr = z + nrow * CT_Z;
    zr[3 + 4 * c4] = gs.x, zr[4 + 4 * c4] = gs.y, zr[5 + 4 * c4] = gs.z, zr[6 + 4 * c4] = gs.w;
    zr[134 + 4 * c4] = acc.x, zr[135 + 4 * c4] = acc.y, zr[136 + 4 * c4] = acc.z, zr[137 + 4 * c4] = acc.w;
    if (c4 < 3) zr[c4] = verts[row * 3 + c4], zr[131 + c4] = verts12[row * 3 + c4];
}

// g *= (y > 0 ? 1 : y + 1): ELU's derivative from its OUTPUT (y + 1 = exp(x) for x <= 0)
__global__ __launch_bounds__(256) void elu_bwd_kernel(f32x4 *__restrict__ g, const f32x4 *__restrict__ y, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 v = g[i];
        const f32x4 yv = y[i];
        v.x = yv.x > 0.f ? v.x : v.x * (yv.x + 1.f), v.y = yv.y > 0.f ? v.y : v.y * (yv.y + 1.f);
        v.z = yv.z > 0.f ? v.z : v.z * (yv.z + 1.f), v.w = yv.w > 0.f ? v.w : v.w * (yv.w + 1.f);
        g[i] = v;
    }
}

// out[c] += sum_r g[r][c] for any width C, fixed order: row chunks (blockIdx.y) x 64-column slabs (blockIdx.x), then one block per slab adds
// the chunks in order
__global__ __launch_bounds__(256) void colsum_any_partial_kernel(const float *__restrict__ g, long R, int C, long rows_per, float *__restrict__ partial) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), h = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.y * rows_per, r1 = r0 + rows_per < R ? r0 + rows_per : R;
    float s = 0.f;
    if (c < C)
        for (long r = r0 + h; r < r1; r += 4) s += g[r * C + c];
    part[h][threadIdx.x & 63] = s;
    __syncthreads();
    if (h == 0 && c < C) partial[(size_t)blockIdx.y * C + c] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
}
__global__ __launch_bounds__(64) void colsum_any_final_kernel(const float *__restrict__ partial, int chunks, int C, float *__restrict__ out) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    float t = 0.f;
    for (int k = 0; k < chunks; ++k) t += partial[(size_t)k * C + c];
    atomicAdd(out + c, t);   // (the caller's buffer may be the parameter's .grad, shared with other nodes of the step)
}
// out[0] += sum of n floats, fixed order (double partials)
__global__ __launch_bounds__(256) void sum_all_partial_kernel(const f32x4 *__restrict__ x, long n4, double *__restrict__ partial) {
    __shared__ double red[256];
    double s = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = x[i];
        s += (double)((v.x + v.y) + (v.z + v.w));
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void sum_all_final_kernel(const double *__restrict__ partial, int n, float *__restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out, (float)red[0]);
}

// d z [P][Nn][262] back to its sources, first half: d verts12[v] += dz[131:134], and the Pi~ gp^T part as compact rows g2t_c [P][Nn][128]
__global__ __launch_bounds__(256) void z_bwd_split_kernel(const float *__restrict__ dz, const int32_t *__restrict__ nodes, int N, int Nn,
                                                          float *__restrict__ dv12, float *__restrict__ g2t_c) {
    const int p = blockIdx.y;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)Nn * 32) return;
    const int a = (int)(g >> 5), c4 = (int)(g & 31);
    const size_t nrow = (size_t)p * Nn + a;
    const float *zr = dz + nrow * CT_Z;
    const f32x4 v = {zr[134 + 4 * c4], zr[135 + 4 * c4], zr[136 + 4 * c4], zr[137 + 4 * c4]};
    *(f32x4 *)(g2t_c + nrow * CT_C + 4 * c4) = v;
    if (c4 < 3) {   // (the nodes of a shape are distinct vertices: no two threads meet on a row)
        float *d = dv12 + ((size_t)p * N + nodes[nrow]) * 3 + c4;
        *d = *d + zr[131 + c4];
    }
}
// second half: d gp[p][v] += dz[3:131]
__global__ __launch_bounds__(256) void z_bwd_pool_kernel(const float *__restrict__ dz, const int32_t *__restrict__ nodes, int N, int Nn,
                                                         float *__restrict__ dgp) {
    const int p = blockIdx.y;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)Nn * 32) return;
    const int a = (int)(g >> 5), c4 = (int)(g & 31);
    const size_t nrow = (size_t)p * Nn + a;
    const float *zr = dz + nrow * CT_Z;
    f32x4 *d = (f32x4 *)(dgp + ((size_t)p * N + nodes[nrow]) * CT_C + 4 * c4);
    f32x4 v = *d;
    v.x += zr[3 + 4 * c4], v.y += zr[4 + 4 * c4], v.z += zr[5 + 4 * c4], v.w += zr[6 + 4 * c4];
    *d = v;
}

// wexp[p][i][s] = w[s]: the pooling conv's weights as the "values" of the xyz-kNN lists (the pooling is the same weighted gather-sum
// as Pi~ @ V with row-independent weights, and shares its backward kernels)
__global__ __launch_bounds__(256) void expand_w_kernel(const float *__restrict__ w, int k, long n, float *__restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = w[i % k];
}

// map term backward, the left-hand side: d verts12[j] += 2 g_p sum over the in-edges (i, s) of j in the xyz-kNN lists of resid[i][s]
__global__ __launch_bounds__(256) void map_bwd_gather_kernel(const float *__restrict__ resid, const int32_t *__restrict__ offs,
                                                             const int32_t *__restrict__ edges, const float *__restrict__ gt, int N, int k,
                                                             float *__restrict__ dv12) {
    const int p = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const long E = (long)N * k;
    const int beg = offs[(size_t)p * (N + 1) + j], end = offs[(size_t)p * (N + 1) + j + 1];
    const int32_t *ed = edges + (size_t)p * E;
    const float *rb = resid + (size_t)p * E * 3;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int e = beg; e < end; ++e) {
        const float *r = rb + (size_t)ed[e] * 3;
        a0 += r[0], a1 += r[1], a2 += r[2];
    }
    const float g2 = 2.f * gt[(size_t)p * CT_TERMS];
    float *d = dv12 + ((size_t)p * N + j) * 3;
    d[0] = fmaf(g2, a0, d[0]), d[1] = fmaf(g2, a1, d[1]), d[2] = fmaf(g2, a2, d[2]);
}

// d val[p][i][t] = d verts12[i] . vertsT[col]  -  2 g_p sum_s resid[i][s] . vertsT[idxT[col][s]]        (col = pidx[i][t]; the second
// part only with the map term)
__global__ __launch_bounds__(256) void gval_total_kernel(const float *__restrict__ dv12, const float *__restrict__ vertsT,
                                                         const int32_t *__restrict__ idxT, const int32_t *__restrict__ pidx,
                                                         const float *__restrict__ resid, const float *__restrict__ gt, int N, int M, int k, int topk,
                                                         float *__restrict__ gval) {
    const int p = blockIdx.y;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)N * topk) return;
    const int i = (int)(g / topk);
    const size_t row = (size_t)p * N + i;
    const int col = pidx[(size_t)p * N * topk + g];
    const float *vt = vertsT + (size_t)p * M * 3;
    const float *d = dv12 + row * 3, *q = vt + (size_t)col * 3;
    float acc = (d[0] * q[0] + d[1] * q[1]) + d[2] * q[2];
    if (resid) {
        const int32_t *nb = idxT + ((size_t)p * M + col) * k;
        const float *r = resid + row * k * 3;
        float m = 0.f;
        for (int s = 0; s < k; ++s) {
            const float *u = vt + (size_t)nb[s] * 3;
            m += (r[3 * s] * u[0] + r[3 * s + 1] * u[1]) + r[3 * s + 2] * u[2];
        }
        acc = fmaf(-2.f * gt[(size_t)p * CT_TERMS], m, acc);
    }
    gval[(size_t)p * N * topk + g] = acc;
}
// gval[p][nodes[a]][t] += dval_n[p][a][t]
__global__ __launch_bounds__(256) void gval_nodes_kernel(const float *__restrict__ dval_n, const int32_t *__restrict__ nodes, int N, int Nn, int topk,
                                                         float *__restrict__ gval) {
    const int p = blockIdx.y;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)Nn * topk) return;
    const int a = (int)(g / topk), t = (int)(g % topk);
    float *o = gval + ((size_t)p * N + nodes[(size_t)p * Nn + a]) * topk + t;
    *o = *o + dval_n[(size_t)p * Nn * topk + g];
}
// dfeat[q] = df1[q] + df2[(q + B) mod 2B] + dpool[q] (+ ddist[q])
__global__ __launch_bounds__(256) void combine_feat_kernel(const f32x4 *__restrict__ df1, const f32x4 *__restrict__ df2, const f32x4 *__restrict__ dpool,
                                                           const f32x4 *__restrict__ ddist, long half, f32x4 *__restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * half; i += (long)gridDim.x * blockDim.x) {
        f32x4 v = (df1[i] + df2[i < half ? i + half : i - half]) + dpool[i];
        if (ddist) v = v + ddist[i];
        out[i] = v;
    }
}
// out = a + b
__global__ __launch_bounds__(256) void add2_kernel(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b, long n4, f32x4 *__restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}
// dist term backward: cs[b][v] = sum_n W[b][n][v] (anchors in order)
__global__ __launch_bounds__(256) void dist_colsum_kernel(const float *__restrict__ W, int N, int nA, float *__restrict__ cs) {
    const int b = blockIdx.y, v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    const float *w = W + (size_t)b * nA * N + v;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int n = 0;
    for (; n + 3 < nA; n += 4) a0 += w[(size_t)n * N], a1 += w[(size_t)(n + 1) * N], a2 += w[(size_t)(n + 2) * N], a3 += w[(size_t)(n + 3) * N];
    for (; n < nA; ++n) a0 += w[(size_t)n * N];
    cs[(size_t)b * N + v] = (a0 + a1) + (a2 + a3);
}
// ddist[b][v] = cs[b][v] feat[b][v] - C1[b][v]
__global__ __launch_bounds__(256) void dist_combine_rows_kernel(const float *__restrict__ cs, const f32x4 *__restrict__ feat, const f32x4 *__restrict__ c1,
                                                                long rows, f32x4 *__restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * (CT_C / 4); i += (long)gridDim.x * blockDim.x) {
        const float c = cs[i / (CT_C / 4)];
        const f32x4 f = feat[i], g = c1[i];
        out[i] = f32x4{c * f.x - g.x, c * f.y - g.y, c * f.z - g.z, c * f.w - g.w};
    }
}
// ddist[b][a_n] += rs[b][n] fa[b][n] - C2[b][n].  The reference draws a shape's anchors with random.sample (distinct points), but
// the criterion's `anchors=` argument is public: atomic adds, so that a repeated anchor accumulates instead of losing an update
// (with distinct anchors each location receives exactly one add: same bits as a plain store)
__global__ __launch_bounds__(256) void dist_combine_anchors_kernel(const float *__restrict__ rs, const f32x4 *__restrict__ fa, const f32x4 *__restrict__ c2,
                                                                   const int32_t *__restrict__ anchors, int N, int nA, f32x4 *__restrict__ out) {
    const int b = blockIdx.y;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)nA * (CT_C / 4)) return;
    const int n = (int)(g / (CT_C / 4)), c4 = (int)(g % (CT_C / 4));
    const size_t src = ((size_t)b * nA + n) * (CT_C / 4) + c4, dst = ((size_t)b * N + anchors[n]) * (CT_C / 4) + c4;
    const float r = rs[(size_t)b * nA + n];
    const f32x4 f = fa[src], q = c2[src];
    float *o = (float *)(out + dst);
    unsafeAtomicAdd(o + 0, r * f.x - q.x);
    unsafeAtomicAdd(o + 1, r * f.y - q.y);
    unsafeAtomicAdd(o + 2, r * f.z - q.z);
    unsafeAtomicAdd(o + 3, r * f.w - q.w);
}

// ---------------------------------------------------------------- arena
// One batch of P directional pairs: sources with N points, targets with M.  `swapped`: the targets ARE the sources of the other half
// (N == M, P = 2B: target of pair p = source of pair (p + B) mod P) — their copies are made here; otherwise the caller hands them in.
struct Dims {
    int P, N, M, k, topk, nA, kd;
    bool swapped;
};
struct CritWs {
    // kept from the forward
    float *featT, *vertsT;
    int32_t *idxT;
    float *pval, *smax, *ssum;
    int32_t *pidx;
    float *verts12, *warped, *gp, *gpT, *z, *pval_n;
    int32_t *pidx_n;
    float *h[3], *def9, *Rm, *T;
    float *d1w, *d2w, *d1s, *d2s;
    int32_t *i1w, *i2w, *i1s, *i2s;
    float *resid;
    double *partial;
    void *scws, *chws;
    size_t sc_bytes, ch_bytes;
    // backward scratch
    float *dwarped, *dv12, *dR, *dT, *ddef9, *dh[3], *dz, *colpart, *g2t_c, *dval_n, *dgpT, *dgp, *wexp, *dvalw, *dpool, *dpoolT, *gval, *df1, *df2;
    double *sumpart;
    int32_t *offsA, *curA, *edgesA, *offsB, *curB, *edgesB, *offsC, *curC, *edgesC;
    void *sbws, *wgws;
    size_t sb_bytes, wg_bytes;
    // dist term (nA > 0; swapped form only)
    int32_t *didx;
    float *xsave, *fa, *W, *rs, *cs, *c1, *c2, *ddist;
    void *dws;
    size_t d_bytes;
};
constexpr int COLSUM_CHUNKS = 128, SUM_BLOCKS = 256;

void carve(Arena &ar, const Dims &d, CritWs &w) {
    const int P = d.P, N = d.N, M = d.M, k = d.k, topk = d.topk;
    const size_t Nn = (size_t)N / 2, R = (size_t)P * Nn, PN = (size_t)P * N, PM = (size_t)P * M, PX = PN > PM ? PN : PM;
    w.featT = w.vertsT = nullptr, w.idxT = nullptr;
    if (d.swapped) w.featT = ar.take<float>(PM * CT_C), w.vertsT = ar.take<float>(PM * 3), w.idxT = ar.take<int32_t>(PM * k);
    w.pval = ar.take<float>(PN * topk), w.pidx = ar.take<int32_t>(PN * topk), w.smax = ar.take<float>(PN), w.ssum = ar.take<float>(PN);
    w.verts12 = ar.take<float>(PN * 3), w.warped = ar.take<float>(PN * 3), w.gp = ar.take<float>(PN * CT_C), w.gpT = ar.take<float>(PM * CT_C);
    w.z = ar.take<float>(R * CT_Z), w.pval_n = ar.take<float>(R * topk), w.pidx_n = ar.take<int32_t>(R * topk);
    for (int l = 0; l < 3; ++l) w.h[l] = ar.take<float>(R * CT_H[l]);
    w.def9 = ar.take<float>(R * 9), w.Rm = ar.take<float>(R * 9), w.T = ar.take<float>(R * 3);
    w.d1w = ar.take<float>(PN), w.d2w = ar.take<float>(PM), w.d1s = ar.take<float>(PN), w.d2s = ar.take<float>(PM);
    w.i1w = ar.take<int32_t>(PN), w.i2w = ar.take<int32_t>(PM), w.i1s = ar.take<int32_t>(PN), w.i2s = ar.take<int32_t>(PM);
    w.resid = ar.take<float>(PN * k * 3);
    w.partial = ar.take<double>((size_t)P * map_term_blocks(N, k));
    w.sc_bytes = dvm_softcorr_workspace_bytes(P, N, M, CT_C), w.scws = ar.take<char>(w.sc_bytes);
    w.ch_bytes = dvm_chamfer_workspace_bytes(P, N, M), w.chws = ar.take<char>(w.ch_bytes);
    // backward
    w.dwarped = ar.take<float>(PN * 3), w.dv12 = ar.take<float>(PN * 3), w.dR = ar.take<float>(R * 9), w.dT = ar.take<float>(R * 3);
    w.ddef9 = ar.take<float>(R * 9);
    for (int l = 0; l < 3; ++l) w.dh[l] = ar.take<float>(R * CT_H[l]);
    w.dz = ar.take<float>(R * CT_Z), w.colpart = ar.take<float>((size_t)COLSUM_CHUNKS * 512), w.sumpart = ar.take<double>(SUM_BLOCKS);
    w.g2t_c = ar.take<float>(R * CT_C), w.dval_n = ar.take<float>(R * topk), w.dgpT = ar.take<float>(PM * CT_C), w.dgp = ar.take<float>(PN * CT_C);
    w.wexp = ar.take<float>(PX * k), w.dvalw = ar.take<float>(PX * k), w.dpool = ar.take<float>(PN * CT_C), w.gval = ar.take<float>(PN * topk);
    w.dpoolT = d.swapped ? nullptr : ar.take<float>(PM * CT_C);
    w.df1 = ar.take<float>(PN * CT_C), w.df2 = ar.take<float>(PM * CT_C);
    w.offsA = ar.take<int32_t>((size_t)P * (M + 1)), w.curA = ar.take<int32_t>(PM), w.edgesA = ar.take<int32_t>(R * topk);
    w.offsB = ar.take<int32_t>((size_t)P * (N + 1)), w.curB = ar.take<int32_t>(PN), w.edgesB = ar.take<int32_t>(PN * k);
    w.offsC = w.curC = w.edgesC = nullptr;
    if (!d.swapped) w.offsC = ar.take<int32_t>((size_t)P * (M + 1)), w.curC = ar.take<int32_t>(PM), w.edgesC = ar.take<int32_t>(PM * k);
    w.sb_bytes = dvm_softcorr_bwd_workspace_bytes(P, N, M, CT_C), w.sbws = ar.take<char>(w.sb_bytes);
    w.wg_bytes = dvm_linear_wgrad_workspace_bytes((long)R, 512, CT_Z), w.wgws = ar.take<char>(w.wg_bytes);
    if (d.nA > 0) {
        const size_t B = (size_t)P / 2;
        const int nA = d.nA, kd = d.kd;
        w.didx = ar.take<int32_t>((size_t)P * nA * kd), w.xsave = ar.take<float>((size_t)P * nA * kd * 2), w.fa = ar.take<float>((size_t)P * nA * CT_C);
        w.d_bytes = dvm_dist_loss_workspace_bytes((int)B, N, CT_C, nA, kd), w.dws = ar.take<char>(w.d_bytes);
        w.W = ar.take<float>(B * nA * N), w.rs = ar.take<float>(B * nA), w.cs = ar.take<float>(B * N);
        w.c1 = ar.take<float>(B * N * CT_C), w.c2 = ar.take<float>(B * nA * CT_C), w.ddist = ar.take<float>(PN * CT_C);
    }
}

void swap_halves(const void *const *src, void *const *dst, const long *half_bytes, int count, hipStream_t s) {
    SwapArgs a;
    long mx = 1;
    for (int m = 0; m < 4; ++m) {
        const int q = m < count ? m : 0;
        a.src[m] = (const f32x4 *)src[q], a.dst[m] = (f32x4 *)dst[q], a.half[m] = half_bytes[q] / 16;
        if (m < count && a.half[m] > mx) mx = a.half[m];
    }
    a.count = count;
    hipLaunchKernelGGL(swap_halves_kernel, dim3(blocks_for(2 * mx, 2048), count), dim3(256), 0, s, a);
}
void colsum_add(const float *g, long R, int C, float *out, const CritWs &w, hipStream_t s) {
    long chunks = (R + 255) / 256;
    if (chunks > COLSUM_CHUNKS) chunks = COLSUM_CHUNKS;
    const long rows_per = (R + chunks - 1) / chunks;
    chunks = (R + rows_per - 1) / rows_per;
    const int slabs = (C + 63) / 64;
    hipLaunchKernelGGL(colsum_any_partial_kernel, dim3(slabs, (unsigned)chunks), dim3(256), 0, s, g, R, C, rows_per, w.colpart);
    hipLaunchKernelGGL(colsum_any_final_kernel, dim3(slabs), dim3(64), 0, s, w.colpart, (int)chunks, C, out);
}

int check_dist(const char *who, int N, const void *d1, const void *d2, const void *a1, const void *a2, int nA, int kd) {
    if (nA == 0) return DVM_OK;
    DVM_REQUIRE(d1 && d2 && a1 && a2, "%s: the dist term needs both distance matrices and both anchor lists", who);
    DVM_REQUIRE(nA >= 1 && nA <= N && kd >= 1 && kd <= 512 && kd <= N, "%s: dist term sizes out of range (anchors %d, neighbours %d)", who, nA, kd);
    return DVM_OK;
}

int check_common(const char *who, int P, int N, int M, bool swapped, int C, int k, int topk, const void *const *params, int nparams) {
    DVM_REQUIRE(P >= 1 && N >= 64 && N <= 8192 && M >= 64 && M <= 8192, "%s: bad sizes (pairs %d, N=%d, M=%d; point counts in 64..8192)", who, P, N, M);
    DVM_REQUIRE(!swapped || (N == M && N % 4 == 0), "%s: N=%d must be a multiple of 4", who, N);
    DVM_REQUIRE(C == CT_C, "%s: C=%d (the Deformer pools 128-wide features)", who, C);
    DVM_REQUIRE(k >= 1 && k <= 16 && topk >= 1 && topk <= 10, "%s: k=%d / topk=%d out of range (k <= 16, topk <= 10)", who, k, topk);
    DVM_REQUIRE(params && nparams == PW_N, "%s: the parameter table has %d entries, expected %d", who, nparams, (int)PW_N);
    for (int i = 0; i < PW_N; ++i) DVM_REQUIRE(params[i] != nullptr, "%s: parameter %d is null", who, i);
    return DVM_OK;
}

// what a pass reads of its sources and targets
struct Sides {
    const float *feat_s, *verts_s;
    const int32_t *nodes, *ring, *infl;
    const float *weights;
    const int32_t *knn_s;
    const float *feat_t, *verts_t;   // swapped form: filled from the arena's copies
    const int32_t *knn_t;
};
struct DistIn {
    const float *dist1, *dist2;
    const int32_t *anchors1, *anchors2;
};

int crit_fwd(const char *who, const Dims &d, Sides io, const DistIn &di, float neg_alpha, const float *const *params, int with_map, float *terms,
             void *arena, size_t arena_bytes, hipStream_t s) {
    const int P = d.P, N = d.N, M = d.M, k = d.k, topk = d.topk, Nn = N / 2, B = P / 2;
    Arena ar(arena, arena_bytes);
    CritWs w;
    carve(ar, d, w);
    if (!ar.ok()) {
        set_error("%s: arena too small (%zu < %zu)", who, arena_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    // the dist term of the 2B shapes: independent of the deformation part, on the helper stream of the caller's context (dvm_pair_init)
    PairCtx *cx = d.nA > 0 ? pair_ctx_find(s) : nullptr;
    HelperJoin hj(cx, s);
    if (d.nA > 0) {
        hipStream_t ds = cx ? cx->side : s;
        hj.fork();
        for (int side = 0; side < 2; ++side) {
            const size_t so = (size_t)side * B;
            CT_TRY(launch_dist_loss_fwd(io.feat_s + so * N * CT_C, side ? di.dist2 : di.dist1, side ? di.anchors2 : di.anchors1, B, N, CT_C, d.nA, d.kd,
                                        terms + so * CT_TERMS, CT_TERMS, 6, w.didx + so * d.nA * d.kd, w.xsave + so * d.nA * d.kd * 2,
                                        w.fa + so * d.nA * CT_C, w.dws, w.d_bytes, ds));
        }
    } else {
        (void)hipMemsetAsync(w.partial, 0, (size_t)P * sizeof(double), s);
        launch_reduce_partials(w.partial, P, 1, 1.f, terms, CT_TERMS, 6, s);
    }
    if (d.swapped) {   // the target side of every directional pair = the other half's sources
        const void *src[3] = {io.feat_s, io.verts_s, io.knn_s};
        void *dst[3] = {w.featT, w.vertsT, w.idxT};
        const long half[3] = {(long)B * N * CT_C * 4, (long)B * N * 3 * 4, (long)B * N * k * 4};
        swap_halves(src, dst, half, 3, s);
        io.feat_t = w.featT, io.verts_t = w.vertsT, io.knn_t = w.idxT;
    }
    CT_TRY(dvm_softcorr_fwd_f32(io.feat_s, io.feat_t, P, N, M, CT_C, neg_alpha, topk, w.pval, w.pidx, w.smax, w.ssum, 0, w.scws, w.sc_bytes, s));
    CT_TRY(dvm_softcorr_apply_f32(w.pval, w.pidx, io.verts_t, P, N, M, topk, 3, w.verts12, s));
    launch_pool_all(io.feat_s, io.knn_s, P, N, k, params[PW_CONV_W], params[PW_CONV_B], w.gp, s, nullptr);
    if (d.swapped) {
        const void *src[1] = {w.gp};
        void *dst[1] = {w.gpT};
        const long half[1] = {(long)B * N * CT_C * 4};
        swap_halves(src, dst, half, 1, s);
    } else {
        launch_pool_all(io.feat_t, io.knn_t, P, M, k, params[PW_CONV_W], params[PW_CONV_B], w.gpT, s, nullptr);
    }
    hipLaunchKernelGGL(assemble_train_kernel, dim3((unsigned)(((long)Nn * 32 + 255) / 256), P), dim3(256), 0, s, io.verts_s, w.verts12, w.gp, w.gpT,
                       w.pval, w.pidx, io.nodes, N, M, Nn, topk, w.z, w.pval_n, w.pidx_n);
    // the decoder MLP 262 -> 512 -> 256 -> 128 -> 9, ELU between the layers (models/model.py:433-452)
    launch_linear(w.z, params[PW_W0], P, Nn, CT_Z, 512, 0, params[PW_B0], nullptr, nullptr, nullptr, -1.f, w.h[0], s, nullptr, 0, nullptr, 1.f);
    launch_linear(w.h[0], params[PW_W1], P, Nn, 512, 256, 0, params[PW_B1], nullptr, nullptr, nullptr, -1.f, w.h[1], s, nullptr, 0, nullptr, 1.f);
    launch_linear(w.h[1], params[PW_W2], P, Nn, 256, 128, 0, params[PW_B2], nullptr, nullptr, nullptr, -1.f, w.h[2], s, nullptr, 0, nullptr, 1.f);
    launch_linear(w.h[2], params[PW_W3], P, Nn, 128, 9, 0, params[PW_B3], nullptr, nullptr, nullptr, 1.f, w.def9, s, nullptr, 0, nullptr, 1.f);
    // rot6d (+ identity) -> embedded-deformation warp -> ARAP (lib/deformation_graph_point.py:233-261); ARAP lands in terms[:, 5]
    launch_dg_warp(io.verts_s, P, N, io.nodes, io.ring, io.infl, io.weights, w.def9, w.Rm, w.T, w.warped, terms + 5, CT_TERMS, nullptr, s);
    CT_TRY(dvm_chamfer_fwd_f32(w.warped, io.verts_t, P, N, M, w.d1w, w.d2w, w.i1w, w.i2w, w.chws, w.ch_bytes, s));
    CT_TRY(dvm_chamfer_fwd_f32(w.verts12, io.verts_t, P, N, M, w.d1s, w.d2s, w.i1s, w.i2s, w.chws, w.ch_bytes, s));
    {
        const float *in[4] = {w.d1w, w.d2w, w.d1s, w.d2s};
        const int n[4] = {N, M, N, M}, off[4] = {1, 2, 3, 4};
        float *out[4] = {terms, terms, terms, terms};
        launch_mean_grouped(in, n, out, off, 4, P, 1.f, CT_TERMS, s);
    }
    if (with_map) {
        launch_map_term(w.verts12, io.verts_t, io.knn_s, io.knn_t, w.pval, w.pidx, P, N, M, k, topk, w.partial, s, w.resid);
        launch_reduce_partials(w.partial, P, map_term_blocks(N, k), 1.f, terms, CT_TERMS, 0, s);
    } else {
        (void)hipMemsetAsync(w.partial, 0, (size_t)P * sizeof(double), s);
        launch_reduce_partials(w.partial, P, 1, 1.f, terms, CT_TERMS, 0, s);
    }
    hj.join();
    return DVM_OK;
}

// d_feat_s [P][N][C] (and, directional form, d_feat_t [P][M][C]) overwritten; parameter gradients added into grads
int crit_bwd(const char *who, const Dims &d, Sides io, const DistIn &di, const float *g_terms, float neg_alpha, const float *const *params,
             float *const *grads, int with_map, float *d_feat_s, float *d_feat_t, void *arena, size_t arena_bytes, hipStream_t s) {
    const int P = d.P, N = d.N, M = d.M, k = d.k, topk = d.topk, Nn = N / 2, B = P / 2;
    const long R = (long)P * Nn, PN = (long)P * N, PM = (long)P * M;
    Arena ar(arena, arena_bytes);
    CritWs w;
    carve(ar, d, w);
    if (!ar.ok()) {
        set_error("%s: arena too small (%zu < %zu)", who, arena_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    if (d.swapped) io.feat_t = w.featT, io.verts_t = w.vertsT, io.knn_t = w.idxT;
    const dim3 node_grid((unsigned)(((long)Nn * 32 + 255) / 256), P);
    // the dist term's feature gradient (helper stream): W from the kept x, y, then the two products on the library's GEMM kernels
    PairCtx *cx = d.nA > 0 ? pair_ctx_find(s) : nullptr;
    HelperJoin hj(cx, s);
    if (d.nA > 0) {
        hipStream_t ds = cx ? cx->side : s;
        hj.fork();
        const int nA = d.nA;
        for (int side = 0; side < 2; ++side) {
            const size_t so = (size_t)side * B;
            const float *fs = io.feat_s + so * N * CT_C, *fa = w.fa + so * nA * CT_C;
            const int32_t *an = side ? di.anchors2 : di.anchors1;
            float *dd = w.ddist + so * N * CT_C;
            launch_dist_loss_bwd_weights_saved(w.xsave + so * nA * d.kd * 2, w.didx + so * nA * d.kd, g_terms + so * CT_TERMS + 6, CT_TERMS, B, N, nA, d.kd,
                                               w.W, w.rs, ds);
            (void)hipMemsetAsync(w.c1, 0, (size_t)B * N * CT_C * sizeof(float), ds);
            launch_wgrad_batched(w.W, fa, B, nA, N, CT_C, w.c1, ds);                 // C1[b] = W[b]^T fa[b]      [N][C]
            launch_linear_bmm(fs, w.W, B, CT_C, N, nA, w.c2, ds);                    // C2[b] = W[b] feat[b]      [nA][C]
            hipLaunchKernelGGL(dist_colsum_kernel, dim3((N + 255) / 256, B), dim3(256), 0, ds, w.W, N, nA, w.cs);
            hipLaunchKernelGGL(dist_combine_rows_kernel, dim3(blocks_for((long)B * N * (CT_C / 4))), dim3(256), 0, ds, w.cs, (const f32x4 *)fs,
                               (const f32x4 *)w.c1, (long)B * N, (f32x4 *)dd);
            hipLaunchKernelGGL(dist_combine_anchors_kernel, dim3((unsigned)(((long)nA * (CT_C / 4) + 255) / 256), B), dim3(256), 0, ds, w.rs,
                               (const f32x4 *)fa, (const f32x4 *)w.c2, an, N, nA, (f32x4 *)dd);
        }
    }
    // Chamfer side means -> d warped, d verts12 (source side only: the targets are inputs)
    launch_chamfer_bwd_src2(w.warped, w.verts12, io.verts_t, io.verts_t, w.i1w, w.i2w, w.i1s, w.i2s, g_terms, CT_TERMS, 1, 3, P, N, M, w.dwarped, w.dv12, s);
    // warp + ARAP -> (dR, dT) -> d def9
    launch_dg_warp_arap_bwd(io.verts_s, P, N, io.nodes, io.ring, io.infl, io.weights, w.Rm, w.T, w.dwarped, g_terms + 5, CT_TERMS, w.dR, w.dT, s);
    launch_def9_bwd(w.def9, w.dR, w.dT, (int)R, w.ddef9, s);
    // the decoder MLP, last layer first: bias gradient (column sums), weight gradient, input gradient x ELU'
    {
        const float *dy = w.ddef9;
        const int Co[4] = {512, 256, 128, 9}, K[4] = {CT_Z, 512, 256, 128};
        const float *x[4] = {w.z, w.h[0], w.h[1], w.h[2]};
        float *dx[4] = {w.dz, w.dh[0], w.dh[1], w.dh[2]};
        for (int l = 3; l >= 0; --l) {
            colsum_add(dy, R, Co[l], grads[PW_B0 + 2 * l], w, s);
            CT_TRY(dvm_linear_wgrad_ws_f32(dy, x[l], R, Co[l], K[l], grads[PW_W0 + 2 * l], w.wgws, w.wg_bytes, s));
            CT_TRY(dvm_linear_f32(params[PW_W0 + 2 * l], dy, 1, K[l], Co[l], (int)R, 1, nullptr, nullptr, nullptr, nullptr, 1.f, dx[l], s));
            if (l > 0) hipLaunchKernelGGL(elu_bwd_kernel, dim3(blocks_for(R * K[l] / 4)), dim3(256), 0, s, (f32x4 *)dx[l], (const f32x4 *)x[l], R * K[l] / 4);
            dy = dx[l];
        }
    }
    // z rows back to their sources
    hipLaunchKernelGGL(z_bwd_split_kernel, node_grid, dim3(256), 0, s, w.dz, io.nodes, N, Nn, w.dv12, w.g2t_c);
    launch_rev_csr(w.pidx_n, P, (long)Nn * topk, M, w.offsA, w.curA, w.edgesA, s);
    launch_apply_bwd_dval(w.pval_n, w.pidx_n, w.gpT, w.g2t_c, P, Nn, M, topk, CT_C, w.dval_n, s);
    launch_apply_bwd_gather(w.pval_n, w.g2t_c, w.offsA, w.edgesA, P, Nn, M, topk, CT_C, w.dgpT, s);
    if (d.swapped) {   // d gp[q] = d gpT[(q + B) mod P] (+ the node rows of dz)
        const void *src[1] = {w.dgpT};
        void *dst[1] = {w.dgp};
        const long half[1] = {(long)B * N * CT_C * 4};
        swap_halves(src, dst, half, 1, s);
    } else {
        (void)hipMemsetAsync(w.dgp, 0, (size_t)PN * CT_C * sizeof(float), s);
    }
    hipLaunchKernelGGL(z_bwd_pool_kernel, node_grid, dim3(256), 0, s, w.dz, io.nodes, N, Nn, w.dgp);
    // the pooling conv backward through the reversed xyz-kNN lists (also the map term's left-hand side)
    launch_rev_csr(io.knn_s, P, (long)N * k, N, w.offsB, w.curB, w.edgesB, s);
    hipLaunchKernelGGL(expand_w_kernel, dim3(blocks_for(PN * k)), dim3(256), 0, s, params[PW_CONV_W], k, PN * k, w.wexp);
    launch_apply_bwd_dval(w.wexp, io.knn_s, io.feat_s, w.dgp, P, N, N, k, CT_C, w.dvalw, s);
    launch_apply_bwd_gather(w.wexp, w.dgp, w.offsB, w.edgesB, P, N, N, k, CT_C, w.dpool, s);
    colsum_add(w.dvalw, PN, k, grads[PW_CONV_W], w, s);
    hipLaunchKernelGGL(sum_all_partial_kernel, dim3(SUM_BLOCKS), dim3(256), 0, s, (const f32x4 *)w.dgp, PN * CT_C / 4, w.sumpart);
    hipLaunchKernelGGL(sum_all_final_kernel, dim3(1), dim3(256), 0, s, w.sumpart, SUM_BLOCKS, grads[PW_CONV_B]);
    if (!d.swapped) {   // ... and of the targets' own pooling (the swapped form pooled them as the other half's sources)
        launch_rev_csr(io.knn_t, P, (long)M * k, M, w.offsC, w.curC, w.edgesC, s);
        hipLaunchKernelGGL(expand_w_kernel, dim3(blocks_for(PM * k)), dim3(256), 0, s, params[PW_CONV_W], k, PM * k, w.wexp);
        launch_apply_bwd_dval(w.wexp, io.knn_t, io.feat_t, w.dgpT, P, M, M, k, CT_C, w.dvalw, s);
        launch_apply_bwd_gather(w.wexp, w.dgpT, w.offsC, w.edgesC, P, M, M, k, CT_C, w.dpoolT, s);
        colsum_add(w.dvalw, PM, k, grads[PW_CONV_W], w, s);
        hipLaunchKernelGGL(sum_all_partial_kernel, dim3(SUM_BLOCKS), dim3(256), 0, s, (const f32x4 *)w.dgpT, PM * CT_C / 4, w.sumpart);
        hipLaunchKernelGGL(sum_all_final_kernel, dim3(1), dim3(256), 0, s, w.sumpart, SUM_BLOCKS, grads[PW_CONV_B]);
    }
    if (with_map)
        hipLaunchKernelGGL(map_bwd_gather_kernel, dim3((N + 255) / 256, P), dim3(256), 0, s, w.resid, w.offsB, w.edgesB, g_terms, N, k, w.dv12);
    // everything that reaches the correspondence values, then the soft correspondence itself
    hipLaunchKernelGGL(gval_total_kernel, dim3((unsigned)(((long)N * topk + 255) / 256), P), dim3(256), 0, s, w.dv12, io.verts_t, io.knn_t, w.pidx,
                       with_map ? w.resid : (const float *)nullptr, g_terms, N, M, k, topk, w.gval);
    hipLaunchKernelGGL(gval_nodes_kernel, dim3((unsigned)(((long)Nn * topk + 255) / 256), P), dim3(256), 0, s, w.dval_n, io.nodes, N, Nn, topk, w.gval);
    CT_TRY(dvm_softcorr_bwd_f32(io.feat_s, io.feat_t, P, N, M, CT_C, neg_alpha, topk, w.pval, w.pidx, w.smax, w.ssum, w.gval, w.df1, w.df2, 0, w.sbws,
                                w.sb_bytes, s));
    hj.join();
    if (d.swapped) {
        hipLaunchKernelGGL(combine_feat_kernel, dim3(blocks_for(PN * CT_C / 4)), dim3(256), 0, s, (const f32x4 *)w.df1, (const f32x4 *)w.df2,
                           (const f32x4 *)w.dpool, d.nA > 0 ? (const f32x4 *)w.ddist : (const f32x4 *)nullptr, (long)B * N * CT_C / 4, (f32x4 *)d_feat_s);
    } else {
        hipLaunchKernelGGL(add2_kernel, dim3(blocks_for(PN * CT_C / 4)), dim3(256), 0, s, (const f32x4 *)w.df1, (const f32x4 *)w.dpool, PN * CT_C / 4,
                           (f32x4 *)d_feat_s);
        hipLaunchKernelGGL(add2_kernel, dim3(blocks_for(PM * CT_C / 4)), dim3(256), 0, s, (const f32x4 *)w.df2, (const f32x4 *)w.dpoolT, PM * CT_C / 4,
                           (f32x4 *)d_feat_t);
    }
    return DVM_OK;
}

}  // namespace
}  // namespace dvm

using namespace dvm;

DVM_EXPORT size_t dvm_criterion_train_workspace_bytes(int B, int N, int k, int topk, int n_anchors, int k_dist) {
    if (B < 1 || N < 2 || k < 1 || topk < 1 || n_anchors < 0 || (n_anchors > 0 && k_dist < 1)) return 0;
    Arena ar(nullptr, 0);
    CritWs w;
    carve(ar, Dims{2 * B, N, N, k, topk, n_anchors, k_dist, true}, w);
    return ar.off;
}

DVM_EXPORT int dvm_criterion_train_fwd_f32(const float *feat, const float *verts, const int32_t *nodes_idx, const int32_t *ring,
                                           const int32_t *infl_idx, const float *weights, const int32_t *knn_idx, int B, int N, int C, int k,
                                           int topk, float neg_alpha, const float *const *params, int nparams, int with_map, const float *dist1,
                                           const float *dist2, const int32_t *anchors1, const int32_t *anchors2, int n_anchors, int k_dist,
                                           float *terms, void *arena, size_t arena_bytes, void *stream) {
    const char *who = "dvm_criterion_train_fwd_f32";
    DVM_REQUIRE(feat && verts && nodes_idx && ring && infl_idx && weights && knn_idx && terms, "%s: null pointer", who);
    CT_TRY(check_common(who, 2 * B, N, N, true, C, k, topk, (const void *const *)params, nparams));
    CT_TRY(check_dist(who, N, dist1, dist2, anchors1, anchors2, n_anchors, k_dist));
    DVM_REQUIRE(neg_alpha < 0.f, "%s: neg_alpha must be negative", who);
    const Sides io{feat, verts, nodes_idx, ring, infl_idx, weights, knn_idx, nullptr, nullptr, nullptr};
    CT_TRY(crit_fwd(who, Dims{2 * B, N, N, k, topk, n_anchors, k_dist, true}, io, DistIn{dist1, dist2, anchors1, anchors2}, neg_alpha, params, with_map,
                    terms, arena, arena_bytes, (hipStream_t)stream));
    DVM_CHECK_LAUNCH("criterion_train_fwd");
    return DVM_OK;
}

DVM_EXPORT int dvm_criterion_train_bwd_f32(const float *g_terms, const float *feat, const float *verts, const int32_t *nodes_idx,
                                           const int32_t *ring, const int32_t *infl_idx, const float *weights, const int32_t *knn_idx, int B,
                                           int N, int C, int k, int topk, float neg_alpha, const float *const *params, float *const *grads,
                                           int nparams, int with_map, const int32_t *anchors1, const int32_t *anchors2, int n_anchors, int k_dist,
                                           float *d_feat, void *arena, size_t arena_bytes, void *stream) {
    const char *who = "dvm_criterion_train_bwd_f32";
    DVM_REQUIRE(g_terms && feat && verts && nodes_idx && ring && infl_idx && weights && knn_idx && d_feat && grads, "%s: null pointer", who);
    CT_TRY(check_common(who, 2 * B, N, N, true, C, k, topk, (const void *const *)params, nparams));
    DVM_REQUIRE(n_anchors == 0 || (anchors1 && anchors2 && n_anchors <= N && k_dist >= 1 && k_dist <= 512),
                "%s: the dist term needs both anchor lists (anchors %d, neighbours %d)", who, n_anchors, k_dist);
    for (int i = 0; i < PW_N; ++i) DVM_REQUIRE(grads[i] != nullptr, "%s: gradient buffer %d is null", who, i);
    const Sides io{feat, verts, nodes_idx, ring, infl_idx, weights, knn_idx, nullptr, nullptr, nullptr};
    CT_TRY(crit_bwd(who, Dims{2 * B, N, N, k, topk, n_anchors, k_dist, true}, io, DistIn{nullptr, nullptr, anchors1, anchors2}, g_terms, neg_alpha, params,
                    grads, with_map, d_feat, nullptr, arena, arena_bytes, (hipStream_t)stream));
    DVM_CHECK_LAUNCH("criterion_train_bwd");
    return DVM_OK;
}

// ONE direction of deform() for P pairs with sources of N and targets of M points (the partial-shape configs: models/loss.py:986-1073,
// train_partial.py:93-112): the same passes with the targets handed in by the caller and their own pooling / reversed lists.
DVM_EXPORT size_t dvm_criterion_dir_train_workspace_bytes(int P, int N, int M, int k, int topk) {
    if (P < 1 || N < 2 || M < 2 || k < 1 || topk < 1) return 0;
    Arena ar(nullptr, 0);
    CritWs w;
    carve(ar, Dims{P, N, M, k, topk, 0, 0, false}, w);
    return ar.off;
}

DVM_EXPORT int dvm_criterion_dir_train_fwd_f32(const float *feat_s, const float *feat_t, const float *verts_s, const float *verts_t,
                                               const int32_t *nodes_idx, const int32_t *ring, const int32_t *infl_idx, const float *weights,
                                               const int32_t *knn_s, const int32_t *knn_t, int P, int N, int M, int C, int k, int topk,
                                               float neg_alpha, const float *const *params, int nparams, int with_map, float *terms, void *arena,
                                               size_t arena_bytes, void *stream) {
    const char *who = "dvm_criterion_dir_train_fwd_f32";
    DVM_REQUIRE(feat_s && feat_t && verts_s && verts_t && nodes_idx && ring && infl_idx && weights && knn_s && knn_t && terms, "%s: null pointer", who);
    CT_TRY(check_common(who, P, N, M, false, C, k, topk, (const void *const *)params, nparams));
    DVM_REQUIRE(neg_alpha < 0.f, "%s: neg_alpha must be negative", who);
    const Sides io{feat_s, verts_s, nodes_idx, ring, infl_idx, weights, knn_s, feat_t, verts_t, knn_t};
    CT_TRY(crit_fwd(who, Dims{P, N, M, k, topk, 0, 0, false}, io, DistIn{nullptr, nullptr, nullptr, nullptr}, neg_alpha, params, with_map, terms, arena,
                    arena_bytes, (hipStream_t)stream));
    DVM_CHECK_LAUNCH("criterion_dir_train_fwd");
    return DVM_OK;
}

DVM_EXPORT int dvm_criterion_dir_train_bwd_f32(const float *g_terms, const float *feat_s, const float *feat_t, const float *verts_s,
                                               const float *verts_t, const int32_t *nodes_idx, const int32_t *ring, const int32_t *infl_idx,
                                               const float *weights, const int32_t *knn_s, const int32_t *knn_t, int P, int N, int M, int C, int k,
                                               int topk, float neg_alpha, const float *const *params, float *const *grads, int nparams, int with_map,
                                               float *d_feat_s, float *d_feat_t, void *arena, size_t arena_bytes, void *stream) {
    const char *who = "dvm_criterion_dir_train_bwd_f32";
    DVM_REQUIRE(g_terms && feat_s && feat_t && verts_s && verts_t && nodes_idx && ring && infl_idx && weights && knn_s && knn_t && d_feat_s && d_feat_t && grads,
                "%s: null pointer", who);
    CT_TRY(check_common(who, P, N, M, false, C, k, topk, (const void *const *)params, nparams));
    for (int i = 0; i < PW_N; ++i) DVM_REQUIRE(grads[i] != nullptr, "%s: gradient buffer %d is null", who, i);
    const Sides io{feat_s, verts_s, nodes_idx, ring, infl_idx, weights, knn_s, feat_t, verts_t, knn_t};
    CT_TRY(crit_bwd(who, Dims{P, N, M, k, topk, 0, 0, false}, io, DistIn{nullptr, nullptr, nullptr, nullptr}, g_terms, neg_alpha, params, grads, with_map,
                    d_feat_s, d_feat_t, arena, arena_bytes, (hipStream_t)stream));
    DVM_CHECK_LAUNCH("criterion_dir_train_bwd");
    return DVM_OK;
}
