// dvm_geodesic.hip — all-pairs shortest paths on a sparse undirected graph (the k-nearest-neighbour graph of a point
// cloud, or a mesh's edges): the N x N "geodesic" matrices the dataset caches for the dist-loss term.
// The reference gets them from potpourri3d's heat-method solver (models/dataset.py:49-54, a C++ dependency outside its
// tree); its own evaluation code uses graph shortest paths (eval/geo_mat.py:15-41), which is what this computes.
// Every source is an independent single-source problem: one workgroup per source keeps the N tentative distances in
// LDS (fp64) and relaxes all edges until nothing changes — Bellman-Ford, but the values only ever decrease towards the
// unique fixed point, so in-place updates within a sweep are harmless and the result is the exact minimum over paths of
// the left-to-right fp64 path sums, i.e. what Dijkstra returns.
#include "dvm_common.h"

namespace dvm {
namespace {

// nbr [N][K] neighbour ids (-1: none), w [N][K] edge lengths; D [N][N]
__global__ __launch_bounds__(256) void graph_sssp_kernel(const int32_t *__restrict__ nbr, const double *__restrict__ w, int N, int K,
                                                         double *__restrict__ D) {
    extern __shared__ double dist[];  // [N]
    __shared__ int changed;
    const int src = blockIdx.x, tid = threadIdx.x;
    for (int v = tid; v < N; v += 256) dist[v] = v == src ? 0.0 : INFINITY;
    __syncthreads();
    for (int sweep = 0; sweep < N; ++sweep) {  // at most N - 1 sweeps are ever needed
        if (tid == 0) changed = 0;
        __syncthreads();
        int any = 0;
        for (int v = tid; v < N; v += 256) {
            double best = dist[v];
            const int32_t *nv = nbr + (size_t)v * K;
            const double *wv = w + (size_t)v * K;
            for (int k = 0; k < K; ++k) {
                const int u = nv[k];
                if (u < 0) break;
                const double cand = dist[u] + wv[k];
                best = cand < best ? cand : best;
            }
            if (best < dist[v]) {
                dist[v] = best;
                any = 1;
            }
        }
        if (any) changed = 1;
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
    for (int v = tid; v < N; v += 256) D[(size_t)src * N + v] = dist[v];
}

}  // namespace
}  // namespace dvm

using namespace dvm;

DVM_EXPORT int dvm_graph_geodesics_f64(const int32_t *nbr, const double *w, int N, int K, double *D, void *stream) {
    DVM_REQUIRE(nbr && w && D, "dvm_graph_geodesics_f64: null pointer");
    DVM_REQUIRE(N >= 1 && K >= 1, "dvm_graph_geodesics_f64: empty graph (N=%d K=%d)", N, K);
    DVM_REQUIRE((size_t)N * sizeof(double) <= 150 * 1024, "dvm_graph_geodesics_f64: N=%d exceeds the %d nodes that fit in LDS", N,
                150 * 1024 / 8);
    ensure_dyn_lds((const void *)graph_sssp_kernel, 152 * 1024);
    hipLaunchKernelGGL(graph_sssp_kernel, dim3(N), dim3(256), (size_t)N * sizeof(double), (hipStream_t)stream, nbr, w, N, K, D);
    DVM_CHECK_LAUNCH("graph_geodesics");
    return DVM_OK;
}
