// dvm_graph.hip — K7/K8: embedded-deformation graph on a point cloud, entirely on the GPU.
//
// Replaces DeformationGraph_geod.construct_graph_euclidean / .forward and their driver
// deformation_graph_node (reference lib/deformation_graph_point.py:18-33,177-201,233-261;
// models/loss.py:1325-1337, 39-45): the reference copies an N x N distance matrix to the host
// and runs a Python FPS loop + scipy KDTree per shape per step; here FPS runs as one workgroup
// per shape with the cloud and the running distances resident in registers/LDS, the KDTree
// queries become brute-force fp64 sweeps (exactly scipy's arithmetic), and nothing leaves HBM.
#include "dvm_common.h"
#include <stdlib.h>

namespace dvm {

// ---------------------------------------------------------------- farthest point sampling
// One workgroup per shape; thread t owns points t, t+T, t+2T, ... (PPT of them) with their
// running min-distance in registers; the cloud is also kept in LDS for the centroid broadcast.
// threads per cloud: 256 / 512 / 1024 (launch_fps), points per thread matched to N

struct ArgMax {
    float v;
    int i;
};
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {  // larger value, then lower index
    bool tb = (b.v > a.v) || (b.v == a.v && b.i < a.i);
    return tb ? b : a;
}

template <int PPT, int FPS_T>
__global__ __launch_bounds__(FPS_T) void fps_kernel(const float *__restrict__ xyz, int N, int npoint,
                                                    const int32_t *__restrict__ start, int32_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [N*3] cloud + [2][4] partial argmax
    float *cloud = smem;
    ArgMax *part = (ArgMax *)(smem + ((N * 3 + 3) & ~3));
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *p = xyz + (size_t)b * N * 3;
    for (int e = tid; e < N * 3; e += FPS_T) cloud[e] = p[e];
    float px[PPT], py[PPT], pz[PPT], dist[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        int j = tid + q * FPS_T;
        bool ok = j < N;
        px[q] = ok ? p[3 * j] : 0.f;
        py[q] = ok ? p[3 * j + 1] : 0.f;
        pz[q] = ok ? p[3 * j + 2] : 0.f;
        dist[q] = ok ? 1e10f : -INFINITY;  // padding can never be the arg-max
    }
    __syncthreads();
    int far = start[b];
    int32_t *o = out + (size_t)b * npoint;
    for (int it = 0; it < npoint; ++it) {
        if (tid == 0) o[it] = far;
        const float cx = cloud[3 * far], cy = cloud[3 * far + 1], cz = cloud[3 * far + 2];
        // the running minima (one v_min each: fminf keeps dist where dv is a NaN, as `dv < dist ? dv : dist` did), the thread's largest
        // one (v_max), then ITS FIRST position from a descending pass of compare + select: 4 instructions per point instead of the 5
        // of a (value, index) pair carried through the loop; same arg-max (largest value, lowest index)
        float bv = -INFINITY;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const float dv = d2_diff3(px[q], py[q], pz[q], cx, cy, cz);
            dist[q] = fminf(dist[q], dv);
            bv = fmaxf(bv, dist[q]);
        }
        int bi = 0x7fffffff;
#pragma unroll
        for (int q = PPT - 1; q >= 0; --q) bi = (dist[q] == bv) ? tid + q * FPS_T : bi;
        ArgMax best = {bv, bi};
        // wave arg-max by two DPP reductions (distances are >= 0 or -inf: order-preserving as sign-flipped integers):
        // the largest value, then the lowest index among its holders — instead of six dependent (value, index)
        // shuffle pairs through the LDS crossbar, which dominated this latency-bound loop
        {
            const int vb = __float_as_int(best.v);
            const unsigned key = (unsigned)(vb >= 0 ? vb | 0x80000000 : ~vb);
            const unsigned kmax = __reduce_max_sync(~0ull, key);
            const unsigned imin = __reduce_min_sync(~0ull, key == kmax ? (unsigned)best.i : 0xffffffffu);
            best.v = __int_as_float((int)(kmax & 0x80000000u ? kmax & 0x7fffffffu : ~kmax));
            best.i = (int)imin;
        }
        ArgMax *slot = part + (it & 1) * (FPS_T / 64);
        if ((tid & 63) == 0) slot[tid >> 6] = best;
        __syncthreads();
        ArgMax r = slot[0];
#pragma unroll
        for (int w = 1; w < FPS_T / 64; ++w) r = better(r, slot[w]);
        far = r.i;
    }
}

// ---------------------------------------------------------------- node ring: 9-NN among nodes
// scipy.spatial.KDTree(nodes).query(nodes, 9): fp64 squared distances of the fp32 coordinates,
// ascending, self first.  Thread per node, nodes staged through LDS.
constexpr int DG_TILE = 256;

__device__ __forceinline__ double d2_f64(float ax, float ay, float az, float bx, float by, float bz) {
    double dx = (double)ax - (double)bx, dy = (double)ay - (double)by, dz = (double)az - (double)bz;
    double s = 0.0;
    s = s + dx * dx;
    s = s + dy * dy;
    s = s + dz * dz;
    return s;
}

__global__ __launch_bounds__(128) void dg_ring_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ nodes_idx,
                                                      int N, int Nn, int32_t *__restrict__ ring) {
    __shared__ float tx[DG_TILE], ty[DG_TILE], tz[DG_TILE];
    const int b = blockIdx.y;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    const float *p = xyz + (size_t)b * N * 3;
    const int32_t *nid = nodes_idx + (size_t)b * Nn;
    const int ac = a < Nn ? a : Nn - 1;
    const int va = nid[ac];
    const float ax = p[3 * va], ay = p[3 * va + 1], az = p[3 * va + 2];
    KBest<9, double> kb;
    kb.init((double)INFINITY);
    for (int j0 = 0; j0 < Nn; j0 += DG_TILE) {
        __syncthreads();
        for (int e = threadIdx.x; e < DG_TILE; e += blockDim.x) {
            int v = (j0 + e < Nn) ? nid[j0 + e] : 0;
            tx[e] = p[3 * v], ty[e] = p[3 * v + 1], tz[e] = p[3 * v + 2];
        }
        __syncthreads();
        int lim = Nn - j0 < DG_TILE ? Nn - j0 : DG_TILE;
        for (int j = 0; j < lim; ++j) kb.insert(d2_f64(ax, ay, az, tx[j], ty[j], tz[j]), j0 + j);
    }
    if (a < Nn)
        for (int t = 0; t < 9; ++t) ring[((size_t)b * Nn + a) * 9 + t] = t < Nn ? kb.idx[t] : a;
}

// ---------------------------------------------------------------- influence nodes + 1-NN distance
// (dists, infl) = 3 smallest of cdist(verts,verts)[nodes_idx] (matmul form, node = row operand);
// nnd[i] = distance to the nearest other vertex in fp64 (KDTree(vertices).query(vertices,2)[:,1]).
__global__ __launch_bounds__(128) void dg_infl_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ nodes_idx,
                                                      int N, int Nn, int32_t *__restrict__ infl, float *__restrict__ dists,
                                                      double *__restrict__ nnd) {
    __shared__ float4 tp[DG_TILE];
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float *p = xyz + (size_t)b * N * 3;
    const int32_t *nid = nodes_idx + (size_t)b * Nn;
    const int ic = i < N ? i : N - 1;
    const float vx = p[3 * ic], vy = p[3 * ic + 1], vz = p[3 * ic + 2];
    const float nv = sumsq3(vx, vy, vz);
    KBest<3, float> kb;
    kb.init(INFINITY);
    for (int j0 = 0; j0 < Nn; j0 += DG_TILE) {
        __syncthreads();
        for (int e = threadIdx.x; e < DG_TILE; e += blockDim.x) {
            int v = (j0 + e < Nn) ? nid[j0 + e] : 0;
            float4 q = {p[3 * v], p[3 * v + 1], p[3 * v + 2], 0.f};
            q.w = sumsq3(q.x, q.y, q.z);
            tp[e] = q;
        }
        __syncthreads();
        int lim = Nn - j0 < DG_TILE ? Nn - j0 : DG_TILE;
#pragma unroll 4
        for (int j = 0; j < lim; ++j) {
            float4 q = tp[j];
            kb.insert(sqrt_rn(d2_mm3(q.x, q.y, q.z, q.w, vx, vy, vz, nv)), j0 + j);
        }
    }
    double m1 = (double)INFINITY, m2 = (double)INFINITY;
    for (int j0 = 0; j0 < N; j0 += DG_TILE) {
        __syncthreads();
        for (int e = threadIdx.x; e < DG_TILE; e += blockDim.x) {
            int v = (j0 + e < N) ? j0 + e : 0;
            float4 q = {p[3 * v], p[3 * v + 1], p[3 * v + 2], 0.f};
            tp[e] = q;
        }
        __syncthreads();
        int lim = N - j0 < DG_TILE ? N - j0 : DG_TILE;
#pragma unroll 4
        for (int j = 0; j < lim; ++j) {
            float4 q = tp[j];
            double v = d2_f64(vx, vy, vz, q.x, q.y, q.z);
            bool lt1 = v < m1, lt2 = v < m2;
            m2 = lt1 ? m1 : (lt2 ? v : m2);
            m1 = lt1 ? v : m1;
        }
    }
    if (i < N) {
        size_t row = (size_t)b * N + i;
        for (int t = 0; t < 3; ++t) {
            infl[row * 3 + t] = t < Nn ? kb.idx[t] : 0;
            dists[row * 3 + t] = kb.key[t];
        }
        nnd[row] = sqrt(m2);
    }
}

// sigma = 20 * mean(nnd) (fp64, fixed order); weights = exp(-d^2 / float(2 sigma^2)) row-normalised
__global__ __launch_bounds__(256) void dg_weights_kernel(const double *__restrict__ nnd, const float *__restrict__ dists,
                                                         int N, float *__restrict__ weights, double *__restrict__ sigma) {
    __shared__ double red[256];
    const int b = blockIdx.x;
    double s = 0.0;
    for (int q = threadIdx.x; q < N; q += blockDim.x) s += nnd[(size_t)b * N + q];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const double sg = 20.0 * (red[0] / (double)N);
    if (threadIdx.x == 0 && sigma) sigma[b] = sg;
    const float den = (float)(2.0 * sg * sg);
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        size_t row = (size_t)b * N + i;
        float w[3], tot = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            float dv = dists[row * 3 + t];
            w[t] = expf(-((dv * dv) / den));
            tot = tot + w[t];
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) weights[row * 3 + t] = w[t] / tot;
    }
}

// ---------------------------------------------------------------- rot6d, warp, ARAP
// def9 = [t(3), r6(6)]: r6 += [1,0,0,0,1,0]; Gram-Schmidt; rows (b1,b2,b1 x b2)
__device__ __forceinline__ void rot6d(const float *__restrict__ d, float (&r)[9], float (&t)[3]) {
    float a1x = d[3] + 1.f, a1y = d[4] + 0.f, a1z = d[5] + 0.f;
    float a2x = d[6] + 0.f, a2y = d[7] + 1.f, a2z = d[8] + 0.f;
    float n1 = sqrt_rn((a1x * a1x + a1y * a1y) + a1z * a1z);
    n1 = n1 > 1e-12f ? n1 : 1e-12f;
    float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    float dot = (b1x * a2x + b1y * a2y) + b1z * a2z;
    float b2x = a2x - dot * b1x, b2y = a2y - dot * b1y, b2z = a2z - dot * b1z;
    float n2 = sqrt_rn((b2x * b2x + b2y * b2y) + b2z * b2z);
    n2 = n2 > 1e-12f ? n2 : 1e-12f;
    b2x = b2x / n2, b2y = b2y / n2, b2z = b2z / n2;
    r[0] = b1x, r[1] = b1y, r[2] = b1z;
    r[3] = b2x, r[4] = b2y, r[5] = b2z;
    r[6] = b1y * b2z - b1z * b2y, r[7] = b1z * b2x - b1x * b2z, r[8] = b1x * b2y - b1y * b2x;
    t[0] = d[0], t[1] = d[1], t[2] = d[2];
}

// rotation_6d_to_matrix proper (models/loss.py:39-45): d6 [rows,6] -> R [rows,9], no identity offset
__global__ void rot6d_plain_kernel(const float *__restrict__ d6, int total, float *__restrict__ R) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= total) return;
    float r[9];
    float a1x = d6[(size_t)n * 6 + 0], a1y = d6[(size_t)n * 6 + 1], a1z = d6[(size_t)n * 6 + 2];
    float a2x = d6[(size_t)n * 6 + 3], a2y = d6[(size_t)n * 6 + 4], a2z = d6[(size_t)n * 6 + 5];
    float n1 = sqrt_rn((a1x * a1x + a1y * a1y) + a1z * a1z);
    n1 = n1 > 1e-12f ? n1 : 1e-12f;
    float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    float dot = (b1x * a2x + b1y * a2y) + b1z * a2z;
    float b2x = a2x - dot * b1x, b2y = a2y - dot * b1y, b2z = a2z - dot * b1z;
    float n2 = sqrt_rn((b2x * b2x + b2y * b2y) + b2z * b2z);
    n2 = n2 > 1e-12f ? n2 : 1e-12f;
    b2x = b2x / n2, b2y = b2y / n2, b2z = b2z / n2;
    r[0] = b1x, r[1] = b1y, r[2] = b1z;
    r[3] = b2x, r[4] = b2y, r[5] = b2z;
    r[6] = b1y * b2z - b1z * b2y, r[7] = b1z * b2x - b1x * b2z, r[8] = b1x * b2y - b1y * b2x;
#pragma unroll
    for (int c = 0; c < 9; ++c) R[(size_t)n * 9 + c] = r[c];
}

__global__ void rot6d_kernel(const float *__restrict__ def9, int total, float *__restrict__ R, float *__restrict__ T) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= total) return;
    float r[9], t[3];
    rot6d(def9 + (size_t)n * 9, r, t);
#pragma unroll
    for (int c = 0; c < 9; ++c) R[(size_t)n * 9 + c] = r[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) T[(size_t)n * 3 + c] = t[c];
}

// v' = sum_s w_s (R_s (v - g_s) + g_s + t_s)
__global__ __launch_bounds__(256) void dg_warp_kernel(const float *__restrict__ xyz, int N, int Nn,
                                                      const int32_t *__restrict__ nodes_idx, const int32_t *__restrict__ infl,
                                                      const float *__restrict__ weights, const float *__restrict__ R,
                                                      const float *__restrict__ T, float *__restrict__ warped) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float *p = xyz + (size_t)b * N * 3;
    const size_t row = (size_t)b * N + i;
    const float vx = p[3 * i], vy = p[3 * i + 1], vz = p[3 * i + 2];
    float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        int nb = infl[row * 3 + s];
        int gv = nodes_idx[(size_t)b * Nn + nb];
        const float *r = R + ((size_t)b * Nn + nb) * 9, *t = T + ((size_t)b * Nn + nb) * 3;
        float gx = p[3 * gv], gy = p[3 * gv + 1], gz = p[3 * gv + 2];
        float dx = vx - gx, dy = vy - gy, dz = vz - gz;
        float w = weights[row * 3 + s];
        float g[3] = {gx, gy, gz};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float rv = (r[3 * c] * dx + r[3 * c + 1] * dy) + r[3 * c + 2] * dz;
            o[c] = o[c] + ((rv + g[c]) + t[c]) * w;
        }
    }
    warped[row * 3] = o[0], warped[row * 3 + 1] = o[1], warped[row * 3 + 2] = o[2];
}

// arap = sum_{a, b in ring(a)} |(g_a+t_a) - (g_b+t_b) - R_a (g_a-g_b)|^2 / Nn ; sr = mean (R_a-R_b)^2
__global__ __launch_bounds__(256) void dg_arap_kernel(const float *__restrict__ xyz, int N, int Nn,
                                                      const int32_t *__restrict__ nodes_idx, const int32_t *__restrict__ ring,
                                                      const float *__restrict__ R, const float *__restrict__ T,
                                                      float *__restrict__ arap, int arap_stride, float *__restrict__ sr, int K) {
    __shared__ double red[2][256];
    const int b = blockIdx.x;
    const float *p = xyz + (size_t)b * N * 3;
    double sa = 0.0, ss = 0.0;
    for (int a = threadIdx.x; a < Nn; a += blockDim.x) {
        size_t na = (size_t)b * Nn + a;
        int va = nodes_idx[na];
        const float *ra = R + na * 9, *ta = T + na * 3;
        float gax = p[3 * va], gay = p[3 * va + 1], gaz = p[3 * va + 2];
        float ga[3] = {gax, gay, gaz};
        for (int q = 0; q < K; ++q) {  // ring width: 9 (point-cloud graph) or 18 (mesh graph, padded with the node itself)
            int nb = ring[na * K + q];
            size_t nbg = (size_t)b * Nn + nb;
            int vb = nodes_idx[nbg];
            const float *rb = R + nbg * 9, *tb = T + nbg * 3;
            float gb[3] = {p[3 * vb], p[3 * vb + 1], p[3 * vb + 2]};
            float dx = ga[0] - gb[0], dy = ga[1] - gb[1], dz = ga[2] - gb[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float rv = (ra[3 * c] * dx + ra[3 * c + 1] * dy) + ra[3 * c + 2] * dz;
                float e = ((ga[c] + ta[c]) - (gb[c] + tb[c])) - rv;
                sa += (double)(e * e);
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                float e = ra[c] - rb[c];
                ss += (double)(e * e);
            }
        }
    }
    red[0][threadIdx.x] = sa;
    red[1][threadIdx.x] = ss;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        arap[(size_t)b * arap_stride] = (float)(red[0][0] / (double)Nn);
        if (sr) sr[b] = (float)(red[1][0] / ((double)Nn * K * 9.0));
    }
}

template <int PPT, int T>
static void launch_fps_t(const float *xyz, int B, int N, int npoint, const int32_t *start, int32_t *out, hipStream_t s) {
    ensure_dyn_lds((const void *)fps_kernel<PPT, T>, 160 * 1024);
    const size_t lds = (size_t)((N * 3 + 3) & ~3) * sizeof(float) + 2 * (T / 64) * sizeof(ArgMax);
    prof_begin(s, DVM_PROF_FPS);
    hipLaunchKernelGGL((fps_kernel<PPT, T>), dim3(B), dim3(T), lds, s, xyz, N, npoint, start, out);
    prof_end(s, DVM_PROF_FPS);
}

// Each step is a dependent chain (distance update -> arg-max -> next centre), so its latency is what counts: the
// per-thread part shrinks with more threads per cloud (and with PPT matched to N instead of rounded up to a power of
// two), the cross-wave part grows with the wave count.
int launch_fps(const float *xyz, int B, int N, int npoint, const int32_t *start, int32_t *out, hipStream_t s) {
    // measured (us/step at 256 / 512 / 1024 threads): N = 2048: 0.65 / 0.71 / 1.18; 4995: 0.93 / 0.89 / 1.31; 12000: 1.63 / 1.22 / 1.63
    int T = N <= 4096 ? 256 : 512;
    if (N > 48 * T) T = N > 48 * 512 ? 1024 : 512;
    const int ppt = (N + T - 1) / T;
#define DVM_FPS_CASE(P)                                                                          \
    if (ppt <= P) {                                                                              \
        if (T == 256) launch_fps_t<P, 256>(xyz, B, N, npoint, start, out, s);                    \
        else if (T == 512) launch_fps_t<P, 512>(xyz, B, N, npoint, start, out, s);               \
        else launch_fps_t<P, 1024>(xyz, B, N, npoint, start, out, s);                            \
        return DVM_OK;                                                                           \
    }
    DVM_FPS_CASE(2) DVM_FPS_CASE(4) DVM_FPS_CASE(6) DVM_FPS_CASE(8) DVM_FPS_CASE(10) DVM_FPS_CASE(12) DVM_FPS_CASE(16)
    DVM_FPS_CASE(20) DVM_FPS_CASE(24) DVM_FPS_CASE(32) DVM_FPS_CASE(48)
#undef DVM_FPS_CASE
    return DVM_EINVAL;
}

// graph build on uniform grids (dvm_grid.hip): gverts = grid over all vertices (built here, reusable by
// the caller), gnodes = grid over the FPS nodes
int launch_dg_build(const float *xyz, int B, int N, const int32_t *start, int32_t *nodes_idx, int32_t *ring,
                    int32_t *infl_idx, float *dists, float *weights, double *sigma, double *nnd, const GridBuf &gverts,
                    const GridBuf &gnodes, bool build_gverts, hipStream_t s, hipEvent_t gverts_ready) {
    // gverts_ready: the vertex grid is built by another stream (which records this event behind it); waited for in front of the
    // influence search, the first kernel here that reads it
    const int Nn = N / 2;
    launch_fps(xyz, B, N, Nn, start, nodes_idx, s);
    if (build_gverts) launch_grid_build(xyz, B, N, nullptr, gverts, s);
    launch_grid_build(xyz, B, N, nodes_idx, gnodes, s);
    launch_grid_ring(gnodes, B, ring, s);
    if (gverts_ready) (void)hipStreamWaitEvent(s, gverts_ready, 0);
    launch_grid_infl(xyz, B, N, gnodes, gverts, infl_idx, dists, nnd, s);
    hipLaunchKernelGGL(dg_weights_kernel, dim3(B), dim3(256), 0, s, nnd, dists, N, weights, sigma);
    return DVM_OK;
}

// Round 4: rot6d -> warp -> ARAP of one cloud in ONE workgroup with the node tables in LDS — node positions (p[nodes_idx]), R, T:
// 15 floats per node, 60 KB at 1024 nodes.  The three kernels above chase nodes_idx -> coordinates and the R / T rows through L2
// for every (vertex, influence node) and (node, ring neighbour) pair (62 + 105 us per launch of 1024 clouds, latency-bound: 256
// threads per cloud in the ARAP kernel); here every such lookup is an LDS read.  Same expressions in the same order; the ARAP sum
// keeps the 256-thread accumulation and reduction tree of dg_arap_kernel (thread t: nodes t, t + 256, ...), hence the same bits.
// (round 6: blockIdx.y = direction — both warps of the pair path in one launch of 2 B workgroups)
struct WarpSide {
    const float *xyz;
    int N, Nn;
    const int32_t *nodes_idx, *ring, *infl;
    const float *weights, *def9;
    float *R, *T, *warped, *arap;
};
struct WarpArgs {
    WarpSide d[2];
    int arap_stride;
};
__global__ __launch_bounds__(256) void dg_warp_arap_fused_kernel(const WarpArgs args) {
    extern __shared__ __attribute__((aligned(16))) float wa_lds[];   // gpos [Nn][3] | R [Nn][9] | T [Nn][3]
    __shared__ double red[256];
    const WarpSide &A = args.d[blockIdx.y];
    const float *__restrict__ xyz = A.xyz, *__restrict__ weights = A.weights, *__restrict__ def9 = A.def9;
    const int32_t *__restrict__ nodes_idx = A.nodes_idx, *__restrict__ ring = A.ring, *__restrict__ infl = A.infl;
    float *__restrict__ R = A.R, *__restrict__ T = A.T, *__restrict__ warped = A.warped, *__restrict__ arap = A.arap;
    const int N = A.N, Nn = A.Nn, arap_stride = args.arap_stride;
    float *gp = wa_lds, *lr = wa_lds + (size_t)Nn * 3, *lt = lr + (size_t)Nn * 9;
    const int b = blockIdx.x;
    const float *p = xyz + (size_t)b * N * 3;
    for (int a = threadIdx.x; a < Nn; a += 256) {
        const size_t na = (size_t)b * Nn + a;
        float r[9], t[3];
        rot6d(def9 + na * 9, r, t);
        const int va = nodes_idx[na];
#pragma unroll
        for (int c = 0; c < 9; ++c) lr[a * 9 + c] = r[c], R[na * 9 + c] = r[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) lt[a * 3 + c] = t[c], T[na * 3 + c] = t[c], gp[a * 3 + c] = p[3 * va + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 256) {   // v' = sum_s w_s (R_s (v - g_s) + g_s + t_s)
        const size_t row = (size_t)b * N + i;
        const float vx = p[3 * i], vy = p[3 * i + 1], vz = p[3 * i + 2];
        float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int nb = infl[row * 3 + s];
            const float *r = lr + nb * 9, *t = lt + nb * 3;
            const float gx = gp[nb * 3], gy = gp[nb * 3 + 1], gz = gp[nb * 3 + 2];
            const float dx = vx - gx, dy = vy - gy, dz = vz - gz;
            const float w = weights[row * 3 + s];
            const float g[3] = {gx, gy, gz};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float rv = (r[3 * c] * dx + r[3 * c + 1] * dy) + r[3 * c + 2] * dz;
                o[c] = o[c] + ((rv + g[c]) + t[c]) * w;
            }
        }
        warped[row * 3] = o[0], warped[row * 3 + 1] = o[1], warped[row * 3 + 2] = o[2];
    }
    double sa = 0.0;
    for (int a = threadIdx.x; a < Nn; a += 256) {
        const float *ra = lr + a * 9, *ta = lt + a * 3;
        const float ga[3] = {gp[a * 3], gp[a * 3 + 1], gp[a * 3 + 2]};
        for (int q = 0; q < 9; ++q) {
            const int nb = ring[((size_t)b * Nn + a) * 9 + q];
            const float *tb = lt + nb * 3;
            const float gb[3] = {gp[nb * 3], gp[nb * 3 + 1], gp[nb * 3 + 2]};
            const float dx = ga[0] - gb[0], dy = ga[1] - gb[1], dz = ga[2] - gb[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float rv = (ra[3 * c] * dx + ra[3 * c + 1] * dy) + ra[3 * c + 2] * dz;
                const float e = ((ga[c] + ta[c]) - (gb[c] + tb[c])) - rv;
                sa += (double)(e * e);
            }
        }
    }
    red[threadIdx.x] = sa;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) arap[(size_t)b * arap_stride] = (float)(red[0] / (double)Nn);
}

int launch_dg_warp_rt(const float *xyz, int B, int N, const int32_t *nodes_idx, const int32_t *ring,
                      const int32_t *infl_idx, const float *weights, const float *R, const float *T, float *warped,
                      float *arap, int arap_stride, float *sr, hipStream_t s) {
    const int Nn = N / 2;
    hipLaunchKernelGGL(dg_warp_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, xyz, N, Nn, nodes_idx, infl_idx, weights, R, T,
                       warped);
    hipLaunchKernelGGL(dg_arap_kernel, dim3(B), dim3(256), 0, s, xyz, N, Nn, nodes_idx, ring, R, T, arap, arap_stride, sr, 9);
    return DVM_OK;
}

int launch_dg_warp(const float *xyz, int B, int N, const int32_t *nodes_idx, const int32_t *ring, const int32_t *infl_idx,
                   const float *weights, const float *def9, float *R, float *T, float *warped, float *arap, int arap_stride,
                   float *sr, hipStream_t s) {
    const int Nn = N / 2;
    const size_t lds = (size_t)Nn * 15 * sizeof(float);
    if (!sr && lds <= 150 * 1024) {   // (sr — the unused smooth-rotation term — only through the separate kernels)
        ensure_dyn_lds((const void *)dg_warp_arap_fused_kernel, (int)lds);
        WarpArgs a;
        a.d[0] = a.d[1] = WarpSide{xyz, N, Nn, nodes_idx, ring, infl_idx, weights, def9, R, T, warped, arap};
        a.arap_stride = arap_stride;
        hipLaunchKernelGGL(dg_warp_arap_fused_kernel, dim3(B, 1), dim3(256), lds, s, a);
        return DVM_OK;
    }
    hipLaunchKernelGGL(rot6d_kernel, dim3((B * Nn + 255) / 256), dim3(256), 0, s, def9, B * Nn, R, T);
    hipLaunchKernelGGL(dg_warp_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, xyz, N, Nn, nodes_idx, infl_idx, weights, R, T,
                       warped);
    hipLaunchKernelGGL(dg_arap_kernel, dim3(B), dim3(256), 0, s, xyz, N, Nn, nodes_idx, ring, R, T, arap, arap_stride, sr, 9);
    return DVM_OK;
}

// both directions' rot6d + warp + ARAP in one launch (the fused form only; false: the caller launches the directions one by one)
bool launch_dg_warp_pair(const float *xyz1, const float *xyz2, int B, int N, int M, const int32_t *const nodes[2], const int32_t *const ring[2],
                         const int32_t *const infl[2], const float *const weights[2], const float *def9_12, const float *def9_21, float *R12,
                         float *R21, float *T12, float *T21, float *warped12, float *warped21, float *arap12, float *arap21, int arap_stride,
                         hipStream_t s) {
    const int nn = N / 2 > M / 2 ? N / 2 : M / 2;
    const size_t lds = (size_t)nn * 15 * sizeof(float);
    if (lds > 150 * 1024) return false;
    ensure_dyn_lds((const void *)dg_warp_arap_fused_kernel, (int)lds);
    WarpArgs a;
    a.d[0] = WarpSide{xyz1, N, N / 2, nodes[0], ring[0], infl[0], weights[0], def9_12, R12, T12, warped12, arap12};
    a.d[1] = WarpSide{xyz2, M, M / 2, nodes[1], ring[1], infl[1], weights[1], def9_21, R21, T21, warped21, arap21};
    a.arap_stride = arap_stride;
    hipLaunchKernelGGL(dg_warp_arap_fused_kernel, dim3(B, 2), dim3(256), lds, s, a);
    return true;
}

}  // namespace dvm

using namespace dvm;

constexpr int DVM_MAX_POINTS = 48 * 256;  // fps_kernel<48>; LDS: 12288*12 B = 144 KiB

DVM_EXPORT int dvm_fps_f32(const float *xyz, int B, int N, int npoint, const int32_t *start, int32_t *out, void *stream) {
    DVM_REQUIRE(xyz && start && out, "dvm_fps_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1 && npoint >= 1 && npoint <= N, "dvm_fps_f32: bad sizes (B=%d N=%d npoint=%d)", B, N, npoint);
    DVM_REQUIRE(N <= DVM_MAX_POINTS, "dvm_fps_f32: N=%d exceeds %d", N, DVM_MAX_POINTS);
    launch_fps(xyz, B, N, npoint, start, out, (hipStream_t)stream);
    DVM_CHECK_LAUNCH("fps");
    return DVM_OK;
}

DVM_EXPORT size_t dvm_dg_build_workspace_bytes(int B, int N) {
    return align_up((size_t)B * N * sizeof(double)) + grid_bytes(B, N) + grid_bytes(B, N / 2);
}

DVM_EXPORT int dvm_dg_build_f32(const float *xyz, int B, int N, const int32_t *start, int32_t *nodes_idx, int32_t *ring,
                                int32_t *infl_idx, float *dists, float *weights, double *sigma, void *ws, size_t ws_bytes,
                                void *stream) {
    DVM_REQUIRE(xyz && start && nodes_idx && ring && infl_idx && dists && weights, "dvm_dg_build_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 2, "dvm_dg_build_f32: bad sizes (B=%d N=%d)", B, N);
    DVM_REQUIRE(N <= DVM_MAX_POINTS, "dvm_dg_build_f32: N=%d exceeds %d", N, DVM_MAX_POINTS);
    Arena ar(ws, ws_bytes);
    double *nnd = ar.take<double>((size_t)B * N);
    GridBuf gv = grid_carve(ar, B, N), gn = grid_carve(ar, B, N / 2);
    if (!ar.ok()) {
        set_error("dvm_dg_build_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    launch_dg_build(xyz, B, N, start, nodes_idx, ring, infl_idx, dists, weights, sigma, nnd, gv, gn, true, (hipStream_t)stream, nullptr);
    DVM_CHECK_LAUNCH("dg_build");
    return DVM_OK;
}

DVM_EXPORT int dvm_rot6d_f32(const float *d6, int rows, float *R, void *stream) {
    DVM_REQUIRE(d6 && R && rows >= 1, "dvm_rot6d_f32: bad arguments");
    hipLaunchKernelGGL(rot6d_plain_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, d6, rows, R);
    DVM_CHECK_LAUNCH("rot6d");
    return DVM_OK;
}

DVM_EXPORT int dvm_dg_warp_arap_fwd_f32(const float *xyz, int B, int N, const int32_t *nodes_idx, const int32_t *ring,
                                        const int32_t *infl_idx, const float *weights, const float *R, const float *T,
                                        float *warped, float *arap, float *sr, void *stream) {
    DVM_REQUIRE(xyz && nodes_idx && ring && infl_idx && weights && R && T && warped && arap,
                "dvm_dg_warp_arap_fwd_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 2, "dvm_dg_warp_arap_fwd_f32: bad sizes (B=%d N=%d)", B, N);
    launch_dg_warp_rt(xyz, B, N, nodes_idx, ring, infl_idx, weights, R, T, warped, arap, 1, sr, (hipStream_t)stream);
    DVM_CHECK_LAUNCH("dg_warp_arap");
    return DVM_OK;
}

DVM_EXPORT int dvm_dg_warp_arap_graph_f32(const float *xyz, int B, int N, int Nn, int ring_width, const int32_t *nodes_idx,
                                          const int32_t *ring, const int32_t *infl_idx, const float *weights, const float *R,
                                          const float *T, float *warped, float *arap, float *sr, void *stream) {
    DVM_REQUIRE(xyz && nodes_idx && ring && infl_idx && weights && R && T && warped && arap,
                "dvm_dg_warp_arap_graph_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1 && Nn >= 1 && Nn <= N && ring_width >= 1 && ring_width <= 64,
                "dvm_dg_warp_arap_graph_f32: bad sizes (B=%d N=%d Nn=%d ring=%d)", B, N, Nn, ring_width);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(dg_warp_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, xyz, N, Nn, nodes_idx, infl_idx, weights, R, T,
                       warped);
    hipLaunchKernelGGL(dg_arap_kernel, dim3(B), dim3(256), 0, s, xyz, N, Nn, nodes_idx, ring, R, T, arap, 1, sr, ring_width);
    DVM_CHECK_LAUNCH("dg_warp_arap_graph");
    return DVM_OK;
}
