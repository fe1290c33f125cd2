/*
 * dvm.h — C ABI of libdvm_hip.so: the MI355X (gfx950) implementation of
 * DV-Matcher's correspondence hot path.
 *
 * The reference has no FFI layer (it is pure Python/PyTorch); its boundary
 * for this path is the Python module API.  Each entry point
 * below names the reference function(s) it replaces (paths relative to the
 * reference checkout).  The Python mirror of the reference's modules in
 * dv-matcher_amd/{models,lib}/ binds these through ctypes; INTEGRATION.md
 * shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *  - every function returns 0 on success or a negative DVM_E* code;
 *    dvm_last_error() returns a thread-local message for the last failure;
 *  - all tensor pointers are caller-owned DEVICE pointers, contiguous,
 *    row-major, fp32 / int32 (double where stated), 16-byte aligned;
 *  - `stream` is a hipStream_t (NULL = default stream); calls are
 *    asynchronous on it, never synchronise, never allocate (graph-capturable);
 *    scratch comes from the caller: query the size with *_workspace_bytes.
 *    The library keeps no per-call global state: kernel attributes are set once
 *    per (device, kernel) under a lock, and the only HIP objects it owns are the
 *    helper streams of dvm_pair_init / the events of dvm_profile_enable;
 *  - calls on distinct streams or devices may come from different host threads;
 *  - B is the batch (pairs or shapes); per-batch tensors are stacked on dim 0;
 *  - indices are 0-based int32.
 */
#ifndef DVM_H
#define DVM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVM_ABI_VERSION 1

#define DVM_OK 0
#define DVM_EINVAL (-1)    /* bad argument (shape, null pointer, unsupported size) */
#define DVM_ELAUNCH (-2)   /* HIP launch / runtime error */
#define DVM_ENOSPACE (-3)  /* workspace too small */

/* The caller's collective for data-parallel statistics (the *_sync_f32 entry points): `allreduce` is called on the HOST while
 * the library enqueues its launches and must enqueue, ordered on `stream`, an IN-PLACE all-reduce over all ranks of `count`
 * elements at the device pointer `buf` (inside memory the caller handed to that entry point); dtype: 0 = float32, 1 = float64;
 * op: 0 = SUM, 1 = MIN, 2 = MAX.  Returns 0 on success.  (RCCL: ncclAllReduce(buf, buf, count, type, op, comm, stream);
 * torch.distributed: all_reduce of a view of the caller's tensor under torch.cuda.stream(ExternalStream(stream)).) */
typedef int (*dvm_allreduce_fn)(void *user, void *buf, size_t count, int dtype, int op, void *stream);
typedef struct dvm_collective {
    dvm_allreduce_fn allreduce;
    void *user;
} dvm_collective;

/* ---------------------------------------------------------------------------------------------------------------------
 * Environment options of the library: SIX variables, read once when the library is loaded (csrc/dvm_api.cpp::options()).
 * Nothing else in the library reads the environment; every option selects among HIP paths (there is no CPU path).
 * (The Python host side, dv-matcher_amd/, reads one variable of its own — DVM_RANK_CPUS, the host cores of a rank, set by
 * `bench.py --gpus N` — and no execution-path switches: those are module attributes and driver flags, listed in README.md.)
 *
 *   DVM_DETERMINISTIC=1        initial value of the dvm_set_deterministic flag (fixed summation order in LG-Net's backward)
 *   DVM_K1_ROUTE=0|1|3         force pass A of the soft correspondence at alpha >= 32: 0 full first form, 1 lean first form,
 *                              3 coarse screen; default: a probe picks per launch and direction, and a device-side gate sends a
 *                              direction the coarse screen serves badly through the lean form (results never depend on the route)
 *   DVM_K1_ROUTE_P="pc,pl"     the probe's thresholds (fractions of a row within the softmax cut; default 0.0014,0.02)
 *   DVM_LINEAR_CFG=0..7        tile configuration of dvm_linear_f32 (default: chosen per shape; all give the same bits)
 *   DVM_PAIR_OVERLAP=0         initial value of dvm_pair_set_overlap (helper-stream overlap of the pair forward)
 *   DVM_DEBUG=<bits>           synchronous diagnostics on stderr: 1 K1 routes per launch, 2 rows sent to the exact-rows kernel,
 *                              4 grid-Chamfer query statistics, 8 K1 coarse-screen cycle stamps, 16 Deformer-MLP cycle stamps
 * --------------------------------------------------------------------------------------------------------------------- */

int dvm_abi_version(void);
const char *dvm_last_error(void);
/* number of visible HIP devices (<=0: none) — lets callers fail loudly. */
int dvm_device_count(void);

/* Launch timing of the pair path's kernels with HIP events recorded on the stream each kernel is
 * launched on (bench.py's roofline legs; the reference has no counterpart: its profiling is
 * wall-clock around train.py's step, train.py:93-112).  enable pre-creates 2*max_launches events
 * and opens a window; the read functions synchronise on the window's events and return the summed
 * kernel time and bracket count of one slot (dvm_profile_read = slot DVM_PROF_K1_SWEEP); disable
 * closes the window.  dvm_profile_select chooses which slots record (bit k = slot k; default:
 * the sweep only, so the timed region of bench.py carries two event records per step). */
#define DVM_PROF_K1_SWEEP 0   /* pass A of K1: softcorr_sweep2_kernel, or softcorr_sweep_f16_kernel where the probe routes */
#define DVM_PROF_K1_REFINE 1  /* softcorr_refine_kernel (pass B: exact re-evaluation) */
#define DVM_PROF_MLP 2        /* mlp_f16x2_kernel (Deformer MLP) */
#define DVM_PROF_CHAMFER 3    /* grid_chamfer_kernel */
#define DVM_PROF_POOL 4       /* pool_kernel (Deformer Conv2d(k->1) pooling) */
#define DVM_PROF_KNN_XYZ 5    /* grid_knn_self_kernel (xyz kNN) */
#define DVM_PROF_FPS 6        /* fps_kernel */
#define DVM_PROF_ASSEMBLE 7   /* assemble_pooled_kernel (Deformer input rows) */
#define DVM_PROF_COUNT 8
int dvm_profile_enable(int max_launches);
int dvm_profile_select(unsigned kernel_mask);
int dvm_profile_read(double *total_ms, int *launches);
int dvm_profile_read_kernel(int kernel, double *total_ms, int *launches);
const char *dvm_profile_kernel_name(int kernel);
int dvm_profile_disable(void);

/* x.pow(2).sum(-1) in ATen's summation order (the |a|^2 terms of torch.cdist's
 * matmul form and of knn_new/knn: models/model.py:274-275, models/loss.py:458-459).
 * x [rows,K] -> out [rows]. */
int dvm_rownorm2_f32(const float *x, int rows, int K, float *out, void *stream);

/* 1x1 convolution / linear layer with its epilogue: nn.Conv1d(kernel_size=1) (+bias) -> [+ residual] ->
 * eval-mode nn.BatchNorm1d -> (Leaky)ReLU of LG-Net (models/model.py:506-529 conv..conv6 blocks, the ff / q / k / v
 * projections of N2PAttention[_DIM] 325-395, SA_Layer's q/k/v/trans_conv 97-123).
 * The contraction is the k-ordered fp32 fma chain acc = fma(w[co][k], x[k], acc) on v_mfma_f32_32x32x2_f32, restarted
 * and summed per K-block exactly like the CPU sgemm behind the reference's Conv1d / matmul (blocks of 384 while more
 * than 768 remain, then one block or two halves): bit-identical to the reference evaluated by one CPU thread,
 * independent of B, N and of the tile configuration.  Then, in this order:
 *   y += bias[co];  y += res;  y = fma(y, bn_alpha[co], bn_beta[co]);  y = y > 0 ? y : y * slope
 * (each step skipped when its pointer is NULL / slope == 1; slope 0 = ReLU; slope < 0 = ELU with alpha 1: y > 0 ? y : exp(y) - 1,
 * the Deformer's decoder MLP, models/model.py:433-452).
 *   channel_major == 0: x [B*N, K], y / res [B*N, Co]          (inference, activations point-major)
 *   channel_major == 1: x [B, K, N], y / res [B, Co, N]        (the reference's Conv1d layout, training forward)
 * w [Co, K]; K <= 8448 (16-byte aligned rows, K % 4 == 0, take the vector-load path). */
int dvm_linear_f32(const float *x, const float *w, int B, int N, int K, int Co, int channel_major, const float *bias,
                   const float *res, const float *bn_alpha, const float *bn_beta, float slope, float *y, void *stream);

/* The same layer (point-major) on rows [xg[b] (Cg) | x[b,n] (K - Cg)]: a 1x1 conv over torch.cat((g.repeat(1,1,N), x), 1)
 * without building the concatenation — conv3 / conv4 of Uni3FC.forward (models/model.py:735-747: the max-pooled
 * 512-vector of a shape in front of its 256 per-point channels), same K-blocked chain over the K = Cg + (K - Cg)
 * concatenated channels.  xg [B, Cg], x [B*N, K - Cg], w [Co, K], y / res [B*N, Co]; Cg % 4 == 0. */
int dvm_linear_prefix_f32(const float *xg, int Cg, const float *x, const float *w, int B, int N, int K, int Co,
                          const float *bias, const float *res, const float *bn_alpha, const float *bn_beta, float slope,
                          float *y, void *stream);

/* The "fixup" form of FeatUp's JBU stages (featup/upsamplers.py: `x + 0.1 * conv1x1(x')`):  y = res + scale * (x . w^T + bias)
 * in one launch, `res` laid out like y; the product is the same chain as dvm_linear_f32, then one multiply and one add
 * (two roundings, as the torch expression). */
int dvm_linear_scaled_residual_f32(const float *x, const float *w, int B, int N, int K, int Co, int channel_major, const float *bias,
                                   float scale, const float *res, float *y, void *stream);

/* Weight gradient of the point-major layer above (what autograd computes for nn.Conv1d.weight in `loss.backward()`,
 * train.py:110):  dW[co][k] += sum_r gy[r][co] * x[r][k]  over the R = B*N rows; gy [R, Co], x [R, K], dW [Co, K] must be
 * ZEROED by the caller (row chunks are combined with fp32 atomics: summation order not fixed).  The input gradient of the
 * same layer needs no entry point of its own: dX = gy W is dvm_linear_f32(x := W as [1, Co, K], w := gy, channel_major = 1). */
int dvm_linear_wgrad_f32(const float *gy, const float *x, long R, int Co, int K, float *dW, void *stream);
/* The same weight gradient with an optional workspace (dvm_linear_wgrad_workspace_bytes): in the deterministic mode (below) every
 * row chunk writes its partial tile there and the tiles are added to dW in chunk order — bit-reproducible from run to run;
 * otherwise (or with ws == NULL) identical to dvm_linear_wgrad_f32. */
size_t dvm_linear_wgrad_workspace_bytes(long R, int Co, int K);
int dvm_linear_wgrad_ws_f32(const float *gy, const float *x, long R, int Co, int K, float *dW, void *ws, size_t ws_bytes, void *stream);

/* Deterministic gradient sums for LG-Net's backward (the reference's `loss.backward()`, train.py:110, sums in whatever order
 * its CUDA kernels' atomics land): on = 1 fixes the summation order of the three places that combine partial sums of different
 * workgroups with fp32 atomics — the row chunks of dvm_linear_wgrad_ws_f32, the split inner loop of dvm_sa_attention_bwd_f32, the
 * in-edge order of dvm_n2p_core_bwd_f32 — so that dvm_uni3fc_train_bwd_f32 returns the same bits on every run (measured:
 * LG-Net forward + backward, 2 x 8 x 2048 points, 18.5 ms instead of 14.3 ms).  Returns the previous setting; environment default DVM_DETERMINISTIC=1.  Process-wide. */
int dvm_set_deterministic(int on);
/* the current setting, read-only (a reader must not toggle the process-wide flag to learn it: another thread's backward could run in between) */
int dvm_get_deterministic(void);

/* Diagnostic, SYNCHRONOUS: which pass-A kernel of knnsearch_t_grad + topk_pi (models/loss.py:110-114, 1339-1347) the most recent
 * soft-correspondence launch of the calling host thread sent its (direction, pair) entries to — the probe and the gate decide on
 * the device, nothing else reads them back.  counts[0] full first form (3 fp16 products), [1] lean first form (3), [2] coarse
 * screen (1), [3] entries swept again by the lean form behind the coarse screen, [4] entries in all.  bench.py prices the
 * roofline line from it.  Waits for that launch's stream; valid only until its workspace is rewritten. */
int dvm_k1_last_routes(int *counts);

/* knnsearch_t_grad + topk_pi (+ the argmax map)  —  models/loss.py:110-114,
 * 1339-1347, 1404-1407.   D = cdist(f1,f2) (matmul form, bit-identical squared
 * distances); P = softmax(D*neg_alpha) over M; keep the `topk` largest of each
 * row, no renormalisation.  Sparse result, rows ordered by descending P
 * (ascending distance, ties -> lowest column):
 *   pi_val [B,N,topk], pi_idx [B,N,topk];
 *   row_smax [B,N] = max_j s_ij, row_sum [B,N] = sum_j exp(s_ij - smax)  (s = D*neg_alpha)
 * f1 [B,N,d], f2 [B,M,d]; neg_alpha = (float)(-alpha) < 0; 1 <= topk <= 16; d % 4 == 0, d <= 512.
 * variant: 0 = auto, 1 = scalar-FMA kernel, 2 = fp32-MFMA kernel (d == 128), 3 = sweep on the 16-bit matrix cores
 * over an exact 2-way fp16 split of the scaled features (3 partial products, fp32 accumulate) with exact fp32
 * re-evaluation of the 12 best columns per row and exact recompute of uncertified rows (d == 128, topk <= 10);
 * all variants give identical columns and row_smax. */
size_t dvm_softcorr_workspace_bytes(int B, int N, int M, int d);
int dvm_softcorr_fwd_f32(const float *f1, const float *f2, int B, int N, int M, int d, float neg_alpha, int topk,
                         float *pi_val, int32_t *pi_idx, float *row_smax, float *row_sum, int variant, void *ws,
                         size_t ws_bytes, void *stream);

/* Backward of dvm_softcorr_fwd_f32 (autograd through models/loss.py:110-114 + the top-k keep of
 * 1339-1347): given g_val [B,N,topk] = dL/d pi_val and the forward's outputs (pi_val, pi_idx, row_smax,
 * row_sum), writes d_f1 [B,N,d] and d_f2 [B,M,d] (overwritten, not accumulated).  The dense N x M term is
 * recomputed tile by tile (never stored); rows with distance exactly 0 contribute 0, like cdist's backward.
 * Sums over rows/columns use fp32 atomics (order not fixed).  variant as in the forward. */
size_t dvm_softcorr_bwd_workspace_bytes(int B, int N, int M, int d);
int dvm_softcorr_bwd_f32(const float *f1, const float *f2, int B, int N, int M, int d, float neg_alpha, int topk,
                         const float *pi_val, const int32_t *pi_idx, const float *row_smax, const float *row_sum,
                         const float *g_val, float *d_f1, float *d_f2, int variant, void *ws, size_t ws_bytes,
                         void *stream);

/* knnsearch_t / search_t — models/loss.py:91-95,121-124; test.py:19-28.
 * T[b,i] = argmin_j cdist(f1,f2, 'donot_use_mm_for_euclid_dist') (0-based; the
 * test scripts add 1), ties -> lowest j; dmin [B,N] optional (may be NULL).
 * With a workspace and d == 128 the columns are first screened by the matrix-core sweep of the soft correspondence
 * and only those that can still be the minimum are evaluated in the exact-difference form (same result, ~50x
 * faster at N = M = 4995); ws == NULL evaluates every column.  dvm_argmin_pair_f32 returns both maps of a pair
 * (T12 [B,N] into f2, T21 [B,M] into f1) from ONE sweep launch, as test.py needs them. */
size_t dvm_argmin_workspace_bytes(int B, int N, int M, int d, int both_directions);
int dvm_argmin_exact_f32(const float *f1, const float *f2, int B, int N, int M, int d, int32_t *T, float *dmin,
                         void *ws, size_t ws_bytes, void *stream);
int dvm_argmin_pair_f32(const float *f1, const float *f2, int B, int N, int M, int d, int32_t *T12, int32_t *T21,
                        float *dmin12, float *dmin21, void *ws, size_t ws_bytes, void *stream);

/* knn_grad — models/loss.py:97-101: the k smallest of cdist(x,y) (matmul form)
 * per row, ascending.  x [B,N,C], y [B,M,C] -> idx [B,N,k]; C <= 16, k <= 16.  A 3-D cloud against
 * itself (x == y) is searched exactly on a uniform grid; otherwise brute force (ws may be NULL). */
size_t dvm_knn_cdist_workspace_bytes(int B, int N, int M, int C);
int dvm_knn_cdist_f32(const float *x, const float *y, int B, int N, int M, int C, int k, int32_t *idx, void *ws,
                      size_t ws_bytes, void *stream);

/* knn_new / knn — models/model.py:267-278, models/loss.py:451-462: the k largest of
 * (-|a|^2 - (-2 a.b)) - |b|^2 per row, descending (ties -> lowest column).
 * a [B,N,C], b [B,M,C] -> idx [B,N,k]; k <= min(512, M), M <= 8192.  C in {64,128} uses the
 * fp32 matrix cores; any other C a scalar kernel. */
size_t dvm_knn_neg_workspace_bytes(int B, int N, int M, int C, int k);
int dvm_knn_neg_f32(const float *a, const float *b, int B, int N, int M, int C, int k, int32_t *idx, void *ws,
                    size_t ws_bytes, void *stream);

/* knnsearch_t_grad as a dense matrix — models/loss.py:110-114 (compatibility entry; the
 * criterion uses dvm_softcorr_fwd_f32).  P [B,N,M] = softmax(cdist(f1,f2) * neg_alpha). */
size_t dvm_softcorr_dense_workspace_bytes(int B, int N, int M, int d);
int dvm_softcorr_dense_f32(const float *f1, const float *f2, int B, int N, int M, int d, float neg_alpha, float *P,
                           void *ws, size_t ws_bytes, void *stream);

/* Pi~ @ V for the sparse Pi~ — models/loss.py:1408-1409 (verts), models/model.py:471
 * (pooled features).  out[b,i,:] = sum_t val[b,i,t] * V[b, idx[b,i,t], :], summed in
 * ascending column order.  V [B,M,C] -> out [B,N,C]. */
int dvm_softcorr_apply_f32(const float *pi_val, const int32_t *pi_idx, const float *V, int B, int N, int M, int topk,
                           int C, float *out, void *stream);

/* Backward of dvm_softcorr_apply_f32: g_out [B,N,C] -> d_val [B,N,topk] = g_out[i] . V[idx[i,t]] and
 * d_V [B,M,C] (overwritten).  With a workspace the (row, slot) -> target lists are reversed by a counting sort
 * and every target row gathers its in-edges (no float atomics); ws == NULL scatter-accumulates with fp32 atomics.
 * C <= 256, topk <= 64. */
size_t dvm_softcorr_apply_bwd_workspace_bytes(int B, int N, int M, int topk);
int dvm_softcorr_apply_bwd_f32(const float *pi_val, const int32_t *pi_idx, const float *V, const float *g_out, int B,
                               int N, int M, int topk, int C, float *d_val, float *d_V, void *ws, size_t ws_bytes,
                               void *stream);

/* farthest_point_sample — lib/deformation_graph_point.py:18-33 with the random
 * start index made an input.  xyz [B,N,3], start [B] -> out [B,npoint]. */
int dvm_fps_f32(const float *xyz, int B, int N, int npoint, const int32_t *start, int32_t *out, void *stream);

/* DeformationGraph_geod.construct_graph_euclidean via deformation_graph_node —
 * lib/deformation_graph_point.py:177-201, models/loss.py:1325-1337.
 * xyz [B,N,3], start [B] -> nodes_idx [B,Nn] (Nn = N/2), ring [B,Nn,9] (node-local),
 * infl_idx [B,N,3] (node-local), dists [B,N,3], weights [B,N,3], sigma [B] (double). */
size_t dvm_dg_build_workspace_bytes(int B, int N);
int dvm_dg_build_f32(const float *xyz, int B, int N, const int32_t *start, int32_t *nodes_idx, int32_t *ring,
                     int32_t *infl_idx, float *dists, float *weights, double *sigma, void *ws, size_t ws_bytes,
                     void *stream);

/* rotation_6d_to_matrix — models/loss.py:39-45.  d6 [rows,6] -> R [rows,9] (rows b1,b2,b1xb2). */
int dvm_rot6d_f32(const float *d6, int rows, float *R, void *stream);
/* Backward of dvm_rot6d_f32: g_R [rows,9] -> g_d6 [rows,6] (Gram-Schmidt differentiated). */
int dvm_rot6d_bwd_f32(const float *d6, const float *g_R, int rows, float *g_d6, void *stream);

/* DeformationGraph_geod.forward — lib/deformation_graph_point.py:233-261.
 * R [B,Nn,9], T [B,Nn,3] -> warped [B,N,3], arap [B], sr [B] (optional). */
int dvm_dg_warp_arap_fwd_f32(const float *xyz, int B, int N, const int32_t *nodes_idx, const int32_t *ring,
                             const int32_t *infl_idx, const float *weights, const float *R, const float *T,
                             float *warped, float *arap, float *sr, void *stream);
/* The same forward for a graph with an explicit node count and ring width — the mesh-mode graph of
 * DeformationGraph_geod.construct_graph (lib/deformation_graph_point.py:203-231): Nn decimated vertices, rings of
 * max_neigh_num = 18 adjacent nodes padded with the node itself.  nodes_idx [B,Nn], ring [B,Nn,ring_width]. */
int dvm_dg_warp_arap_graph_f32(const float *xyz, int B, int N, int Nn, int ring_width, const int32_t *nodes_idx,
                               const int32_t *ring, const int32_t *infl_idx, const float *weights, const float *R,
                               const float *T, float *warped, float *arap, float *sr, void *stream);
/* Backward of dvm_dg_warp_arap_fwd_f32 w.r.t. the node transforms: g_warped [B,N,3], g_arap [B] ->
 * d_R [B,Nn,9], d_T [B,Nn,3] (overwritten; fp32 atomics).  The vertices carry no gradient (inputs). */
int dvm_dg_warp_arap_bwd_f32(const float *xyz, int B, int N, const int32_t *nodes_idx, const int32_t *ring,
                             const int32_t *infl_idx, const float *weights, const float *R, const float *T,
                             const float *g_warped, const float *g_arap, float *d_R, float *d_T, void *stream);

/* chamfer_3DDist — third-party ChamferDistancePytorch (un-vendored); call sites
 * models/loss.py:1120,1223,874.  a [B,N,3], b [B,M,3] -> d1 [B,N], d2 [B,M] squared NN
 * distances, i1 [B,N], i2 [B,M] (optional). */
size_t dvm_chamfer_workspace_bytes(int B, int N, int M);
int dvm_chamfer_fwd_f32(const float *a, const float *b, int B, int N, int M, float *d1, float *d2, int32_t *i1,
                        int32_t *i2, void *ws, size_t ws_bytes, void *stream);
/* Backward of dvm_chamfer_fwd_f32 through the arg-min indices held fixed (as the upstream CUDA extension
 * does): g_d1 [B,N], g_d2 [B,M] -> d_a [B,N,3], d_b [B,M,3] (overwritten; fp32 atomics). */
int dvm_chamfer_bwd_f32(const float *a, const float *b, const int32_t *idx1, const int32_t *idx2, const float *g_d1,
                        const float *g_d2, int B, int N, int M, float *d_a, float *d_b, void *stream);

/* Deformer.forward — models/model.py:464-478 (+ MLP 433-452), fed the raw features and
 * kNN indices instead of the (B,N,k,128) gathers of models/loss.py:1254-1255.
 * feat1 [B,N,128], feat2 [B,M,128], verts1 [B,N,3], verts12 [B,N,3], idx11 [B,N,k],
 * idx22 [B,M,k], pi_val/pi_idx [B,N,topk], fps1 [B,Nn]; weights: conv_w [k], conv_b [1],
 * W0 [512,262] b0, W1 [256,512] b1, W2 [128,256] b2, W3 [9,128] b3 -> out [B,Nn,9].
 * variant: 0 = fp16x2-split matrix-core MLP, 64 nodes per workgroup (fp32-accurate; a value outside fp16's range
 * raises a device flag and the bf16x3 kernel, gated on it, overwrites the result), 3 = bf16x3-split matrix-core MLP
 * (fp32-accurate, any range), 2 = fp32-MFMA MLP (k-ordered fma chain, bit-identical to the oracle's),
 * 1 = scalar-FMA MLP (cross-check). */
size_t dvm_deformer_workspace_bytes(int B, int N, int M, int Nn);
int dvm_deformer_fwd_f32(const float *feat1, const float *feat2, const float *verts1, const float *verts12,
                         const int32_t *idx11, const int32_t *idx22, const float *pi_val, const int32_t *pi_idx,
                         const int32_t *fps1, int B, int N, int M, int Nn, int k, int topk, const float *conv_w,
                         const float *conv_b, const float *W0, const float *b0, const float *W1, const float *b1,
                         const float *W2, const float *b2, const float *W3, const float *b3, float *out, int variant,
                         void *ws, size_t ws_bytes, void *stream);

/* The Deformer's decoder MLP alone — models/model.py:433-452, 476-477.
 * z [rows,262] -> out [rows,9] (262 -> 512 -> 256 -> 128 -> 9, ELU). */
size_t dvm_deformer_mlp_workspace_bytes(int rows);
int dvm_deformer_mlp_fwd_f32(const float *z, int rows, const float *W0, const float *b0, const float *W1,
                             const float *b1, const float *W2, const float *b2, const float *W3, const float *b3,
                             float *out, void *ws, size_t ws_bytes, void *stream);

/* Uni3FC.pos_encoding_sin_wave — models/model.py:544-561.  x [B,3,N] -> out [B,384,N]:
 * nc = 2*((x-min)/(max-min))-1 with the min/max of the WHOLE tensor, 64 octaves pi*2^j. */
size_t dvm_pos_encoding_workspace_bytes(void);
int dvm_pos_encoding_f32(const float *x, int B, int N, float *out, void *ws, size_t ws_bytes, void *stream);
/* Same encoding with the normalisation range supplied: minmax[0] = min, minmax[1] = max (device pointer).  For a
 * pair batch sharded over ranks the reference's whole-tensor min/max (models/model.py:548) is the min/max over
 * ALL ranks' shards; the host reduces the two numbers (MIN / MAX all-reduce) and passes them here. */
int dvm_pos_encoding_minmax_f32(const float *x, const float *minmax, int B, int N, float *out, void *stream);
/* ... with the range over all ranks: this rank's (min, max) -> minmax2 (two floats of caller-owned device memory), the caller's
 * collective (MIN on the first, MAX on the second), then the encoding */
int dvm_pos_encoding_sync_f32(const float *x, int B, int N, float *out, void *ws, size_t ws_bytes, const dvm_collective *coll,
                              float *minmax2, void *stream);

/* Training-mode nn.BatchNorm1d over [B,C,N] fused with the residual add in front of it and the (Leaky)ReLU behind it
 * (models/model.py:97-123 SA_Layer, 325-395 N2PAttention, 506-529 conv blocks):
 *   z = x (+ res);  y = act(gamma (z - mean_c) / sqrt(var_c + eps) + beta),  act(t) = t > 0 ? t : slope t
 * (slope 1: none, 0: ReLU, 0.2: LeakyReLU).  save_mean / save_invstd [C] feed the backward; running_mean / running_var
 * (may be NULL) get PyTorch's momentum update with the unbiased variance.  Backward: dx (= d res) [B,C,N],
 * dgamma / dbeta [C] (may be NULL); the activation's derivative is taken from the sign of y. */
size_t dvm_bn_workspace_bytes(int B, int C, int N);
/* The same fused BatchNorm (forward and backward below) on POINT-MAJOR activations x [R, C] (R = B*N rows, C % 4 == 0), the
 * layout of the training path: statistics per column, in a fixed summation order (row chunk by row chunk).  The backward
 * with accumulate = 1 ADDS the parameter gradients to dgamma / dbeta (autograd's `p.grad += g` without the extra launches). */
size_t dvm_bn_pm_workspace_bytes(long R, int C);
int dvm_bn_act_train_fwd_pm_f32(const float *x, const float *res, const float *gamma, const float *beta, long R, int C, float eps,
                                float slope, float momentum, float *y, float *save_mean, float *save_invstd, float *running_mean,
                                float *running_var, void *ws, size_t ws_bytes, void *stream);
/* The same forward that also returns the UNBIASED batch variance (save_var_unbiased [C], may be NULL): with running_mean ==
 * running_var == NULL the running statistics are left alone and can be updated later from (save_mean, save_var_unbiased) with
 * exactly the fused kernel's expression — dvm_uni3fc_train_running_stats_f32 does so for a whole network call. */
int dvm_bn_act_train_fwd_pm_var_f32(const float *x, const float *res, const float *gamma, const float *beta, long R, int C, int groups,
                                    float eps, float slope, float momentum, float *y, float *save_mean, float *save_invstd,
                                    float *save_var_unbiased, float *running_mean, float *running_var, void *ws, size_t ws_bytes,
                                    void *stream);
/* groups > 1 (here and in dvm_bn_act_train_bwd_pm_groups_f32): x holds `groups` consecutive blocks of R rows, each normalised
 * with ITS OWN batch statistics (several network calls merged into one batch: the reference normalises per call); save_* are
 * [groups][C]; the running statistics take the groups' updates one after the other; dgamma / dbeta sum over the groups.
 * Workspace: dvm_bn_pm_groups_workspace_bytes(R, C, groups). */
size_t dvm_bn_pm_groups_workspace_bytes(long R, int C, int groups);
/* Cross-rank statistics for a data-parallel step (train.py under DDP: the reference's single-process batch normalises over ALL
 * pairs, models/model.py:496-503).  With a collective, the per-(group, channel) totals (sum x, sum x^2 — backward: sum dz,
 * sum dz xhat) and the groups' row counts are written to `sync_buf` (dvm_bn_pm_sync_bytes(C, groups) bytes of caller-owned device
 * memory), handed to the caller's all-reduce (float64, SUM, in place, enqueued on `stream`), and the normalisation uses the global
 * totals; dgamma / dbeta stay this rank's sums (the gradient all-reduce adds the ranks').  coll == NULL: the plain calls above. */
size_t dvm_bn_pm_sync_bytes(int C, int groups);
int dvm_bn_act_train_fwd_pm_sync_f32(const float *x, const float *res, const float *gamma, const float *beta, long R, int C, int groups,
                                     float eps, float slope, float momentum, float *y, float *save_mean, float *save_invstd,
                                     float *save_var_unbiased, float *running_mean, float *running_var, void *ws, size_t ws_bytes,
                                     const dvm_collective *coll, void *sync_buf, void *stream);
int dvm_bn_act_train_bwd_pm_sync_f32(const float *dy, const float *y, const float *x, const float *res, const float *gamma,
                                     const float *save_mean, const float *save_invstd, long R, int C, int groups, float slope,
                                     float *dx, float *dgamma, float *dbeta, int accumulate, void *ws, size_t ws_bytes,
                                     const dvm_collective *coll, void *sync_buf, void *stream);
int dvm_bn_act_train_bwd_pm_groups_f32(const float *dy, const float *y, const float *x, const float *res, const float *gamma,
                                       const float *save_mean, const float *save_invstd, long R, int C, int groups, float slope,
                                       float *dx, float *dgamma, float *dbeta, int accumulate, void *ws, size_t ws_bytes, void *stream);
int dvm_bn_act_train_bwd_pm_f32(const float *dy, const float *y, const float *x, const float *res, const float *gamma,
                                const float *save_mean, const float *save_invstd, long R, int C, float slope, float *dx,
                                float *dgamma, float *dbeta, int accumulate, void *ws, size_t ws_bytes, void *stream);
int dvm_bn_act_train_fwd_f32(const float *x, const float *res, const float *gamma, const float *beta, int B, int C, int N,
                             float eps, float slope, float momentum, float *y, float *save_mean, float *save_invstd,
                             float *running_mean, float *running_var, void *ws, size_t ws_bytes, void *stream);
int dvm_bn_act_train_bwd_f32(const float *dy, const float *y, const float *x, const float *res, const float *gamma,
                             const float *save_mean, const float *save_invstd, int B, int C, int N, float slope, float *dx,
                             float *dgamma, float *dbeta, void *ws, size_t ws_bytes, void *stream);

/* cal_geo — models/dataset.py:49-54 (potpourri3d heat method in the reference; graph shortest paths here, as in the
 * reference's eval/geo_mat.py:15-41).  All-pairs shortest paths of an undirected graph given as padded neighbour lists:
 * nbr [N,K] (-1 ends a list), w [N,K] fp64 edge lengths -> D [N,N] fp64 (+inf between components).  N <= 19200. */
int dvm_graph_geodesics_f64(const int32_t *nbr, const double *w, int N, int K, double *D, void *stream);

/* Uni3FC.proj2img — models/model.py:584-650 (+ get_colored_depth_maps 563-581).  pts [B,N,3] (one of the three
 * axis-permuted views) -> img [B,3,224,224]: every point adds its depth (3rd coordinate) to the 5x5 pixels around its
 * cell, sigmoid, ImageNet-normalise, min-max rescale per image, 'PiYG' colour table, pixels whose depth sum is exactly 0
 * -> -1.  Also returns what I2P needs: pc_min [B,2], grid_size [B], offsets [B,2] (float-valued integers).  The depth
 * sums are accumulated in 64-bit fixed point, so the image is bit-reproducible from run to run. */
size_t dvm_proj2img_workspace_bytes(int B);
int dvm_proj2img_f32(const float *pts, int B, int N, float *img, float *pc_min, float *grid_size, float *offsets, void *ws,
                     size_t ws_bytes, void *stream);

/* The per-pixel filtering step of the FeatUp joint-bilateral upsampler that stands between proj2img and I2P (the
 * `upsampler` of models/model.py:693, train.py:72; FeatUp's `AdaptiveConv` CUDA op):
 *   out[b,c,h,w] = sum_{i,j<d} in[b,c,h+i,w+j] * kern[b,h,w,i,j]
 * in [B,C,H+d-1,W+d-1] (the padded, bicubically upsampled features), kern [B,H,W,d,d] — or [B,d*d,H,W] when
 * kern_tap_major != 0 —, out [B,C,H,W]; d odd, <= 15. */
int dvm_adaptive_conv_f32(const float *in, const float *kern, int B, int C, int H, int W, int d, int kern_tap_major, float *out,
                          void *stream);

/* F.interpolate(x, size=(Ho,Wo), mode='bicubic', align_corners=False) followed by F.pad(.., [pad]*4, mode='reflect') in
 * one pass (the high-resolution source of a joint-bilateral stage): in [BC,Hi,Wi] -> out [BC,Ho+2*pad,Wo+2*pad]. */
int dvm_bicubic_resize_pad_f32(const float *in, int BC, int Hi, int Wi, int Ho, int Wo, int pad, float *out, void *stream);

/* The combined range x spatial kernel of one joint-bilateral stage, before its learned correction (FeatUp
 * JBULearnedRange: get_range_kernel, get_spatial_kernel and their normalised product): for pixel p and tap t of the
 * d x d window (reflect padding)  k[p,t] ~ softmax_t(exp(range_temp) * <proj[p], proj[p+t]>) * exp(-|t|^2 / (2 sigma^2)),
 * normalised over t.  proj [B,key_dim,H,W] (the guidance image through range_proj); range_temp, sigma_spatial: device
 * scalars (the module's parameters); out [B,d*d,H,W].  key_dim = 32, d = 7 (FeatUp's JBU stack). */
int dvm_jbu_kernel_f32(const float *proj, const float *range_temp, const float *sigma_spatial, int B, int key_dim, int H, int W, int d,
                       float *out, void *stream);

/* Uni3FC.I2P (+ F.normalize) — models/model.py:653-678, 701-708.  f [B,C,H,W] image features; for every point the
 * bicubic (A=-0.75, align_corners=False) resample of f to 224x224 is evaluated ONLY at the point's pixel — the
 * (B,C,224,224) resized tensor is never built — and written to out[b,i,0:C] (row stride ldo >= C floats, so the three
 * views can land side by side in one (B,N,3C) tensor); normalize != 0 divides by max(|row|_2, 1e-12). */
int dvm_i2p_f32(const float *pts, const float *f, const float *pc_min, const float *grid_size, const float *offsets, int B,
                int N, int C, int H, int W, int normalize, float *out, int ldo, void *stream);

/* SA_Layer attention core — models/model.py:113-121.  p [B,N,16] = Wqk x (q and k share the
 * weight), v [B,N,64] = Wv x + bv, point-major.  energy = p p^T, row softmax, every column
 * divided by (1e-9 + its sum over rows), x_r = attention^T-weighted sum of v -> xr [B,N,64]. */
size_t dvm_sa_attention_workspace_bytes(int B, int N);
int dvm_sa_attention_fwd_f32(const float *p, const float *v, int B, int N, float *xr, void *ws, size_t ws_bytes,
                             void *stream);

/* Training twins of the SA attention core.  train_fwd: as dvm_sa_attention_fwd_f32, also returning what the
 * backward recomputes from: stats [B,N,2] = (row max m_i, 1/l_i) and cinv [B,N] = 1/(1e-9 + column sum).
 * bwd: g_xr [B,N,64] -> d_p [B,N,16], d_v [B,N,64] (overwritten).  Nothing N x N is stored: two
 * tile-recompute passes on the fp32 matrix cores (E is symmetric, so both dE_ij and dE_ji come from one tile). */
size_t dvm_sa_attention_train_fwd_workspace_bytes(int B, int N);
int dvm_sa_attention_train_fwd_f32(const float *p, const float *v, int B, int N, float *xr, float *stats, float *cinv,
                                   void *ws, size_t ws_bytes, void *stream);
size_t dvm_sa_attention_bwd_workspace_bytes(int B, int N);
int dvm_sa_attention_bwd_f32(const float *p, const float *v, const float *xr, const float *stats, const float *cinv,
                             const float *g_xr, int B, int N, float *d_p, float *d_v, void *ws, size_t ws_bytes,
                             void *stream);

/* N2PAttention[_DIM] attention core — models/model.py:339-350, 375-386.  q/kp/vp [B,N,C] =
 * Wq x, Wk x, Wv x (point-major; k(x_j - x_i) = kp_j - kp_i by linearity), idx [B,N,K] the
 * feature-space neighbours -> out [B,N,C] = sum_j softmax_j(q.(kp_j-kp_i)/sqrt(D)) (vp_j - vp_i)
 * per head.  C in {64,128}, heads = 4, K <= 64. */
int dvm_n2p_attention_fwd_f32(const float *q, const float *kp, const float *vp, const int32_t *idx, int B, int N,
                              int C, int K, int heads, float *out, void *stream);

/* Training twins of the N2P attention core.  qkv [B,N,3C] = [Wq x | Wk x | Wv x] per point (one GEMM's
 * output, used in place), idx [B,N,K].  fwd: out [B,N,C] as dvm_n2p_attention_fwd_f32, plus the attention
 * weights attn [B,N,K,heads] kept for the backward.  bwd: g_out [B,N,C] -> d_qkv [B,N,3C] (overwritten; the
 * scatter onto neighbour rows runs as a gather over the reversed neighbour lists — a counting sort of idx
 * in the workspace — so no float atomics; idx entries must lie in [0,N)).  C in {64,128}, heads = 4, K <= 64. */
int dvm_n2p_core_fwd_f32(const float *qkv, const int32_t *idx, int B, int N, int C, int K, int heads, float *out,
                         float *attn, void *stream);
size_t dvm_n2p_core_bwd_workspace_bytes(int B, int N, int K);
int dvm_n2p_core_bwd_f32(const float *qkv, const int32_t *idx, const float *attn, const float *g_out, int B, int N,
                         int C, int K, int heads, float *d_qkv, void *ws, size_t ws_bytes, void *stream);

/* dist-loss term — models/loss.py:1351-1396 for one shape batch: anchors [nA] (shared by the
 * batch), idx = knn(feat[:,anchors], feat, k); x = |feat[idx] - feat[anchor]|, y = dist[b, idx, anchor];
 * out[b] = sum_n (1 - |cos(x_n, y_n)|).  feat [B,N,C], dist [B,N,N]; idx_out [B,nA,k] optional. */
size_t dvm_dist_loss_workspace_bytes(int B, int N, int C, int nA, int k);
int dvm_dist_loss_fwd_f32(const float *feat, const float *dist, const int32_t *anchors, int B, int N, int C, int nA,
                          int k, float *out, int32_t *idx_out, void *ws, size_t ws_bytes, void *stream);

/* Backward of dvm_dist_loss_fwd_f32, first half: with idx [B,nA,k] from the forward (idx_out) and
 * g_out [B] = dL/d out, writes the dense weights W [B,nA,N] (overwritten; zero outside idx):
 *   W[b,n,v] = g_out[b] * d term_n/d x_v / x_v   (x_v = |feat[v] - feat[a_n]|, 0 where x_v = 0).
 * The feature gradient follows with library GEMMs:
 *   d feat = diag(colsum W) feat - W^T feat[anchors];  d feat[a_n] += rowsum(W)_n feat[a_n] - (W feat)_n. */
int dvm_dist_loss_bwd_weights_f32(const float *feat, const float *dist, const int32_t *anchors, const int32_t *idx,
                                  const float *g_out, int B, int N, int C, int nA, int k, float *W, void *stream);

/* map-loss numerator — models/loss.py:1232-1238 + FrobeniusLoss 476-482:
 * out[b] = sum_{i,s,c} (verts12[idx11[i,s],c] - sum_t P[i,t] verts2[idx22[pidx[i,t],s],c])^2. */
size_t dvm_map_term_workspace_bytes(int B, int N);
int dvm_map_term_f32(const float *verts12, const float *verts2, const int32_t *idx11, const int32_t *idx22,
                     const float *pi_val, const int32_t *pi_idx, int B, int N, int M, int k, int topk, float *out,
                     void *ws, size_t ws_bytes, void *stream);

/* One direction of GraphDeformLoss_Neural.deform() for B pairs, forward only, with no
 * host round trip — models/loss.py:1228-1296, 1401-1410; deform.py:232-257:
 *   graph(verts1) -> Pi_12 -> verts12 -> kNN(verts1), kNN(verts2) -> Deformer -> rot6d ->
 *   ED warp + ARAP -> chamfer(warped,verts2), chamfer(verts12,verts2) (+ map term).
 * Inputs: feat1 [B,N,128], feat2 [B,M,128], verts1 [B,N,3], verts2 [B,M,3], fps_start [B],
 * Deformer weights as in dvm_deformer_fwd_f32 (k = topk = 10).
 * Outputs: warped [B,N,3], verts12 [B,N,3], T12 [B,N] (argmax of Pi),
 *          losses [B,6] = {mean d(warped->verts2), mean d(verts2->warped), arap,
 *                          mean d(verts12->verts2), mean d(verts2->verts12), map_sum}
 *          (squared NN distances; chamfer_loss = [0]+[1], the partial variant keeps one side). */
size_t dvm_pair_direction_workspace_bytes(int B, int N, int M);
int dvm_pair_direction_fwd_f32(const float *feat1, const float *feat2, const float *verts1, const float *verts2,
                               int B, int N, int M, float neg_alpha, const int32_t *fps_start, const float *conv_w,
                               const float *conv_b, const float *W0, const float *b0, const float *W1,
                               const float *b1, const float *W2, const float *b2, const float *W3, const float *b3,
                               int with_map, float *warped, float *verts12, int32_t *T12, float *losses, void *ws,
                               size_t ws_bytes, void *stream);

/* Both directions of the deformation part of GraphDeformLoss_Neural.forward for B pairs —
 * models/loss.py:1401-1411 (graphs of both clouds, Pi_12 and Pi_21, deform() twice) with the work
 * shared between the directions (xyz kNN, pooled features, one soft-correspondence launch, one MLP
 * launch, one grouped Chamfer launch).  Outputs as in dvm_pair_direction_fwd_f32, once per direction
 * (`12`: cloud 1 deformed towards cloud 2; `21`: the reverse). */
size_t dvm_pair_workspace_bytes(int B, int N, int M);
int dvm_pair_fwd_f32(const float *feat1, const float *feat2, const float *verts1, const float *verts2, int B, int N,
                     int M, float neg_alpha, const int32_t *start1, const int32_t *start2, const float *conv_w,
                     const float *conv_b, const float *W0, const float *b0, const float *W1, const float *b1,
                     const float *W2, const float *b2, const float *W3, const float *b3, int with_map, float *warped12,
                     float *verts12, int32_t *T12, float *losses12, float *warped21, float *verts21, int32_t *T21,
                     float *losses21, void *ws, size_t ws_bytes, void *stream);

/* dvm_pair_fwd_f32 with the per-shape graph cache of SURVEY 8f-2: the coordinate-only products of a call — both clouds'
 * deformation graphs (FPS nodes, node rings, skinning weights: lib/deformation_graph_point.py:177-201 via
 * models/loss.py:1325-1337), their uniform grids, the xyz kNN (models/loss.py:97-101) and the neighbour coordinates of the map
 * term — stay in the caller's workspace.  reuse_geometry = 0 builds them (identical to dvm_pair_fwd_f32); reuse_geometry != 0
 * uses what an earlier call left in THIS workspace: valid iff that call had the same B, N, M, coordinates and FPS starts and
 * nothing else wrote the workspace since (the caller keeps one workspace per cache key).  Outputs are bit-identical to the
 * uncached call.  The reference rebuilds the graphs on every call; the cache is the caller's opt-in. */
int dvm_pair_fwd_cached_f32(const float *feat1, const float *feat2, const float *verts1, const float *verts2, int B, int N,
                            int M, float neg_alpha, const int32_t *start1, const int32_t *start2, const float *conv_w,
                            const float *conv_b, const float *W0, const float *b0, const float *W1, const float *b1,
                            const float *W2, const float *b2, const float *W3, const float *b3, int with_map, float *warped12,
                            float *verts12, int32_t *T12, float *losses12, float *warped21, float *verts21, int32_t *T21,
                            float *losses21, void *ws, size_t ws_bytes, int reuse_geometry, void *stream);

/* The coordinate-only half of dvm_pair_fwd_f32 as its own call, for a pipelined caller: both clouds' deformation graphs (FPS nodes,
 * node rings, skinning weights: lib/deformation_graph_point.py:18-33, 177-201 via models/loss.py:1325-1337), their uniform grids,
 * the xyz kNN (models/loss.py:97-101) and, when the map term needs them, the neighbour coordinates are written into `ws`
 * (dvm_pair_workspace_bytes(B, N, M)); a later dvm_pair_fwd_cached_f32(..., reuse_geometry = 1) on the SAME workspace, B, N, M,
 * coordinates, starts and with_map consumes them.  Typical use: the geometry of batch t + 1 on a second stream while batch t's
 * feature-dependent half runs (dv-matcher_amd/dvm/ops.py::PairPipeline; the caller orders the two calls with its own events and
 * keeps one workspace per batch in flight).  Nothing is cached: each batch's graphs are built once, only earlier; outputs are
 * bit-identical to dvm_pair_fwd_f32.  With a dvm_pair_init context for `stream`, vertex grid + xyz kNN run on the context's second
 * helper stream beside the FPS chain and are joined into `stream` before the call returns. */
int dvm_pair_geometry_f32(const float *verts1, const float *verts2, int B, int N, int M, const int32_t *start1,
                          const int32_t *start2, int with_map, void *ws, size_t ws_bytes, void *stream);

/* ---- LG-Net, the whole eval-mode forward in one call (reference models/model.py:680-761 `Uni3FC.forward` with
 * torch.no_grad() / model.eval(); layers 506-529, N2P blocks 325-395, SA_Layer 97-123).  xyz [B][3][N] coordinates, dino
 * [B][N][1152] per-point visual features -> feat [B][N][128], tmp [B][N][64] (the second return value of the reference's
 * forward).  Every layer is one of this library's own launches, enqueued natively in the order of the reference's forward:
 * the same kernels and operands as dv-matcher_amd/models/model.py::Uni3FC._forward_infer, ~250 launches without a Python call
 * in between.  k = neighbours of the N2P blocks (the reference's 40).  With a context from dvm_pair_init(stream) the global
 * (self-attention) chain runs on the context's helper stream next to the local (kNN attention) chain.
 * `weights`: DVM_U3_NWEIGHTS device pointers, fp32, in this order (W = conv weight [Co][K]; alpha, beta = the eval-mode
 * BatchNorm folded as ATen's CPU kernel evaluates it, y = fma(x, alpha, beta), alpha = w / sqrt(var + eps), beta =
 * fma(-mean, alpha, b) — dv-matcher_amd/models/model.py::_bn_affine):
 *   8 conv blocks {W, alpha, beta}: conv (1152->384), conv0 (384->64), conv1 (256->512), conv2 (256->512),
 *     conv3 (768->128), conv4 (768->128), conv5 (256->128), conv6 (512->128)                                  [0 .. 23]
 *   4 SA layers sa1..sa4 {k_conv W [16][64], v_conv W [64][64], v_conv bias, trans_conv W [64][64], trans_conv bias,
 *     after_norm alpha, after_norm beta}                                                                      [24 .. 51]
 *   7 N2P blocks n2p_attention1..7 (C = 64 x 4, 128 x 3) {stacked q|k|v W [3C][C], bn1 alpha, bn1 beta, ff[0] W [4C][C],
 *     ff[2] W [C][4C], bn2 alpha, bn2 beta}                                                                   [52 .. 100] */
#define DVM_U3_NWEIGHTS 101
size_t dvm_uni3fc_fwd_workspace_bytes(int B, int N, int k);
int dvm_uni3fc_fwd_f32(const float *xyz, const float *dino, int B, int N, const float *const *weights, int nweights, int k,
                       float *feat, float *tmp, void *ws, size_t ws_bytes, void *stream);

/* ---- LG-Net, the TRAINING forward and backward in two calls (reference `Uni3FC.forward` in train mode under autograd,
 * models/model.py:680-761 — conv blocks 506-529 with batch-statistics BatchNorm, N2PAttention[_DIM] 325-395, SA_Layer
 * 97-123 — and the part of `loss.backward()`, train.py:110, that runs through it).  Same launches as
 * dv-matcher_amd/models/model.py::Uni3FC._forward_train_pm and the autograd graph it records, enqueued natively.
 * fwd: xyz [B][3][N], dino [B][N][1152] -> feat [B][N][128], tmp [B][N][64]; every activation the backward needs is
 *      written into `arena` (dvm_uni3fc_train_workspace_bytes(B, N, k) bytes, caller-owned, must stay untouched until
 *      the matching bwd); running_mean / running_var get PyTorch's momentum update (unbiased variance) — or, with
 *      defer_running_stats != 0, are left alone: the batch statistics stay in the arena and
 *      dvm_uni3fc_train_running_stats_f32(params, arena) applies the 26 updates later in one launch (two network calls
 *      that run side by side on two streams share every BatchNorm; their updates are applied afterwards in the
 *      reference's call order, train.py:100-101); the num_batches_tracked counters are the caller's business.  knn_forced / knn_log (each NULL or 7 device pointers
 *      [B][N][k] int32, entries may be NULL) are the parity tests' handles on the 7 feature-space kNN layers: the
 *      library's own neighbour sets are copied to knn_log[l], and knn_forced[l], when given, is what the layer and
 *      its backward then use (teacher forcing: the discrete sets of the reference, tests/test_gpu_network.py).
 * bwd: g_feat [B][N][128], g_tmp [B][N][64] (may be NULL = zero), the forward's dino / feat / tmp and arena ->
 *      parameter gradients ADDED into grads[i] (same indexing as params; entries of running statistics ignored, may
 *      be NULL).  Inputs carry no gradient (the reference feeds data tensors).  Weight gradients are combined with
 *      fp32 atomics over row chunks (dvm_linear_wgrad_f32); bias / BatchNorm gradients in a fixed order.
 * groups (1 or 2): SEVERAL network calls of the reference in one — the B shapes are `groups` consecutive blocks of B / groups
 *      shapes, each with its own BatchNorm batch statistics and position-encoding range (and the running statistics updated
 *      block after block), i.e. exactly the results of `groups` separate calls (the criterion calls the network once per shape
 *      of a pair, train.py:100-101: same point count -> ONE call with groups = 2, every row-wise kernel over twice the rows).
 * With a context from dvm_pair_init(stream) the global (self-attention) chain runs on the helper stream in both passes.
 * `params`: DVM_U3_TRAIN_NPARAMS device pointers, fp32, in this order:
 *   8 conv blocks {W [Co][K], bn gamma, bn beta, running_mean, running_var}: conv, conv0 .. conv6            [0 .. 39]
 *   4 SA layers sa1..sa4 {k_conv W [16][64] (tied with q_conv), v_conv W [64][64], v_conv bias, trans_conv W [64][64],
 *     trans_conv bias, after_norm gamma, beta, running_mean, running_var}                                     [40 .. 75]
 *   7 N2P blocks n2p_attention1..7 (C = 64 x 4, 128 x 3) {q_conv W [C][C], k_conv W, v_conv W, bn1 gamma, beta,
 *     running_mean, running_var, ff[0] W [4C][C], ff[2] W [C][4C], bn2 gamma, beta, running_mean, running_var} [76 .. 166]
 * (q / k / v weights — and their gradient buffers — that lie back to back in memory are used as one stacked [3C][C]
 * matrix in place; otherwise a packed copy is made per call.) */
#define DVM_U3_TRAIN_NPARAMS 167
size_t dvm_uni3fc_train_workspace_bytes(int B, int N, int k);
int dvm_uni3fc_train_fwd_f32(const float *xyz, const float *dino, int B, int N, const float *const *params, int nparams, int k,
                             float eps, float momentum, int groups, int defer_running_stats, const int32_t *const *knn_forced,
                             int32_t *const *knn_log, float *feat, float *tmp, void *arena, size_t arena_bytes, void *stream);
int dvm_uni3fc_train_running_stats_f32(const float *const *params, int nparams, int B, int N, int k, int groups, float momentum,
                                       void *arena, size_t arena_bytes, void *stream);
int dvm_uni3fc_train_bwd_f32(const float *g_feat, const float *g_tmp, const float *dino, const float *feat, const float *tmp,
                             int B, int N, const float *const *params, float *const *grads, int nparams, int k, int groups,
                             void *arena, size_t arena_bytes, void *stream);
/* The same two calls inside a DATA-PARALLEL step (train.py:30-33 picks one device; the 8-GPU run shards the pair batch): with
 * `coll` every batch statistic the reference takes over its whole batch is combined over the ranks by the caller's collective —
 * the 26 BatchNorms' totals in both passes (dvm_bn_act_train_*_pm_sync_f32) and the position encoding's min / max
 * (dvm_pos_encoding_sync_f32, models/model.py:548) — so the sharded step reproduces the single-process one on THIS native node.
 * The buffers handed to the collective lie inside `arena`.  coll == NULL: identical to the calls above. */
int dvm_uni3fc_train_fwd_sync_f32(const float *xyz, const float *dino, int B, int N, const float *const *params, int nparams, int k,
                                  float eps, float momentum, int groups, int defer_running_stats, const int32_t *const *knn_forced,
                                  int32_t *const *knn_log, float *feat, float *tmp, void *arena, size_t arena_bytes,
                                  const dvm_collective *coll, void *stream);
int dvm_uni3fc_train_bwd_sync_f32(const float *g_feat, const float *g_tmp, const float *dino, const float *feat, const float *tmp,
                                  int B, int N, const float *const *params, float *const *grads, int nparams, int k, int groups,
                                  void *arena, size_t arena_bytes, const dvm_collective *coll, void *stream);

/* The deformation part of GraphDeformLoss_Neural.forward in TRAINING, natively — models/loss.py:1228-1296 (deform(), both
 * directions), the Deformer models/model.py:454-478 and its MLP 433-452; the weighting 1413-1432 stays with the caller.
 * The B pairs of a step are handled as ONE batch of P = 2B directional pairs [(1 -> 2) x B | (2 -> 1) x B]:
 *   feat [P][N][C=128], verts [P][N][3]  : the B first shapes followed by the B second shapes (LG-Net's merged training call
 *                                          and the batched graph build lay them out like this); pair p maps shape p onto
 *                                          shape (p + B) mod P
 *   nodes_idx [P][N/2], ring [P][N/2][9], infl_idx [P][N][3], weights [P][N][3] : dvm_dg_build_f32 of verts
 *   knn_idx [P][N][k]                    : dvm_knn_cdist_f32(verts, verts, k)
 *   params: DVM_CRIT_TRAIN_NPARAMS device pointers — Deformer.conv_layer.weight [k], .bias [1], then the decoder's
 *           (weight, bias) x 4: [512][262], [512], [256][512], [256], [128][256], [128], [9][128], [9]
 *   dist term (models/loss.py:1351-1396), n_anchors > 0: dist1 / dist2 [B][N][N] the geodesic matrices of the first / second shapes,
 *           anchors1 / anchors2 [n_anchors] the anchor draws, k_dist <= 512 neighbours — what dvm_dist_loss_fwd_f32 takes, for all 2B
 *           shapes; runs on the helper stream of the caller's dvm_pair_init context beside the deformation part.  n_anchors == 0: off.
 * fwd -> terms [P][7] = [map numerator (0 when with_map == 0) | mean d(warped -> target), mean d(target -> warped) |
 *        mean d(verts12 -> target), mean d(target -> verts12) | ARAP | dist term of SHAPE p (0 when off)]; everything the backward
 *        needs stays in `arena` (dvm_criterion_train_workspace_bytes bytes, caller-owned, untouched until the matching bwd).
 * bwd: g_terms [P][7] = dL / d terms -> d_feat [P][N][C] (overwritten; the coordinates carry no gradient), the parameter
 *      gradients ADDED into grads[i] (fp32 atomics; same indexing as params).
 * N a multiple of 4 in 64..8192, k <= 16, topk <= 10, neg_alpha < 0 as in dvm_softcorr_fwd_f32. */
#define DVM_CRIT_TRAIN_NPARAMS 10
size_t dvm_criterion_train_workspace_bytes(int B, int N, int k, int topk, int n_anchors, int k_dist);
int dvm_criterion_train_fwd_f32(const float *feat, const float *verts, const int32_t *nodes_idx, const int32_t *ring,
                                const int32_t *infl_idx, const float *weights, const int32_t *knn_idx, int B, int N, int C, int k,
                                int topk, float neg_alpha, const float *const *params, int nparams, int with_map, const float *dist1,
                                const float *dist2, const int32_t *anchors1, const int32_t *anchors2, int n_anchors, int k_dist,
                                float *terms, void *arena, size_t arena_bytes, void *stream);
int dvm_criterion_train_bwd_f32(const float *g_terms, const float *feat, const float *verts, const int32_t *nodes_idx,
                                const int32_t *ring, const int32_t *infl_idx, const float *weights, const int32_t *knn_idx, int B,
                                int N, int C, int k, int topk, float neg_alpha, const float *const *params, float *const *grads,
                                int nparams, int with_map, const int32_t *anchors1, const int32_t *anchors2, int n_anchors, int k_dist,
                                float *d_feat, void *arena, size_t arena_bytes, void *stream);

/* The same node for ONE direction of deform() with sources of N and targets of M points (the partial-shape configs:
 * GraphDeformLoss_Neural_Partial, models/loss.py:986-1073; train_partial.py:93-112): P pairs, feat_s [P][N][128] / verts_s [P][N][3] /
 * knn_s [P][N][k] and the graph of the SOURCES, feat_t [P][M][128] / verts_t [P][M][3] / knn_t [P][M][k] of the targets.
 * -> terms [P][7] as above (column 6 = 0: the dist term stays with the caller); bwd -> d_feat_s [P][N][128], d_feat_t [P][M][128]
 * (overwritten), parameter gradients ADDED.  N, M in 64..8192 (any parity). */
size_t dvm_criterion_dir_train_workspace_bytes(int P, int N, int M, int k, int topk);
int dvm_criterion_dir_train_fwd_f32(const float *feat_s, const float *feat_t, const float *verts_s, const float *verts_t,
                                    const int32_t *nodes_idx, const int32_t *ring, const int32_t *infl_idx, const float *weights,
                                    const int32_t *knn_s, const int32_t *knn_t, int P, int N, int M, int C, int k, int topk,
                                    float neg_alpha, const float *const *params, int nparams, int with_map, float *terms, void *arena,
                                    size_t arena_bytes, void *stream);
int dvm_criterion_dir_train_bwd_f32(const float *g_terms, const float *feat_s, const float *feat_t, const float *verts_s,
                                    const float *verts_t, const int32_t *nodes_idx, const int32_t *ring, const int32_t *infl_idx,
                                    const float *weights, const int32_t *knn_s, const int32_t *knn_t, int P, int N, int M, int C, int k,
                                    int topk, float neg_alpha, const float *const *params, float *const *grads, int nparams, int with_map,
                                    float *d_feat_s, float *d_feat_t, void *arena, size_t arena_bytes, void *stream);

/* dvm_pair_fwd_f32 can run its coordinate-only chain (FPS, graph, xyz kNN: latency-bound) on helper streams, forked
 * from and joined back into `stream` by events, next to the feature-only soft-correspondence chain.  The helper streams
 * and events are NOT created by the compute call: dvm_pair_init(stream) makes them for (current device, `stream`) —
 * idempotent, the one call of this library that allocates HIP objects — and dvm_pair_destroy() frees every context of
 * the process.  A dvm_pair_fwd_f32 on a (device, stream) without a context runs everything on `stream`: same results,
 * no overlap.  Contexts are per caller stream, so calls on distinct streams / devices / host threads never share an
 * event; calls on the SAME stream must come from one thread at a time (as for any stream-ordered API).
 * dvm_pair_set_overlap(0) ignores the contexts (A/B measurements); returns the previous setting.
 * Environment default: DVM_PAIR_OVERLAP=0. */
int dvm_pair_init(void *stream);
int dvm_pair_destroy(void);
int dvm_pair_set_overlap(int on);

#ifdef __cplusplus
}
#endif
#endif /* DVM_H */
