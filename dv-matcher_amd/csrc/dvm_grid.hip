// dvm_grid.hip — exact 3-D nearest-neighbour queries on a uniform grid.
//
// The reference solves four small 3-D neighbour problems per shape and step by brute force /
// KDTree: xyz kNN (knn_grad, models/loss.py:97-101), the node ring and the 1-NN distance
// (scipy KDTree, lib/deformation_graph_point.py:181-191), the 3 nearest graph nodes of every vertex
// (:186-187) and Chamfer's nearest neighbours (models/loss.py:1216-1226).  Here each cloud is
// binned once into a G^3 grid (counting sort in LDS, one workgroup per shape) and every query walks
// cubes of growing radius until its k-th best distance is certified against the distance to the
// cube's faces.  Distances are evaluated in the reference's own rounding (matmul-form fp32, exact
// fp64, or difference-form fp32) and ranked by (distance, index), so results equal the brute-force
// kernels bit for bit whatever order the cells are visited in.
#include <stdlib.h>

#include "dvm_common.h"

namespace dvm {

constexpr int GRID_T = 256;

struct GridView {          // one shape's grid (device pointers already offset to the shape)
    const float4 *pts;     // sorted points: x, y, z, |p|^2 (ATen order)
    const int32_t *ids;    // original (candidate-local) index of each sorted point
    const int32_t *start;  // [G^3 + 1]
    float ox, oy, oz, h, scale2;
    int G;
};

// GridBuf (dvm_common.h): batched storage — pts [B][P], ids [B][P], start [B][G^3+1], params [B][8]

static int grid_dim_for(int P) {
    // ~1.2 points per cell (measured on uniform clouds at P = 2048: G = 12 beats 8 by 4 % end to end, 16 loses again);
    // results do not depend on G (exact search)
    int g = 2;
    while (g < 16 && (double)(g + 1) * (g + 1) * (g + 1) * 1.2 <= (double)P * 1.15) ++g;
    return g < 4 ? 4 : g;
}

// One workgroup per shape.  src points are xyz[sel[j]] (sel == nullptr: identity), j < P.
// up to four (cloud set, grid) pairs built by ONE launch (blockIdx.y = set): the pair path builds the grids of its four Chamfer
// clouds back to back on the main stream, one workgroup per cloud each — four launches of B workgroups were four launch gaps
struct GridBuildSets {
    const float *xyz[4];
    int Nsrc[4];
    GridBuf gb[4];
};
template <bool SETS>
__device__ __forceinline__ void grid_build_body(const float *__restrict__ xyz, int Nsrc, const int32_t *__restrict__ sel, const GridBuf &gb);
__global__ __launch_bounds__(GRID_T) void grid_build_sets_kernel(const GridBuildSets sets) {
    const int q = blockIdx.y;
    grid_build_body<true>(sets.xyz[q], sets.Nsrc[q], nullptr, sets.gb[q]);
}
__global__ __launch_bounds__(GRID_T) void grid_build_kernel(const float *__restrict__ xyz, int Nsrc,
                                                            const int32_t *__restrict__ sel, GridBuf gb) {
    grid_build_body<false>(xyz, Nsrc, sel, gb);
}
template <bool SETS>
__device__ __forceinline__ void grid_build_body(const float *__restrict__ xyz, int Nsrc, const int32_t *__restrict__ sel, const GridBuf &gb) {
    extern __shared__ int lds[];  // [G^3 + 1] counts/starts, then cursors [G^3]
    __shared__ float red[6][GRID_T / 64];
    __shared__ float par[8];
    __shared__ int wsum[GRID_T / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = gb.P, G = gb.G, G3 = G * G * G;
    const float *p = xyz + (size_t)b * Nsrc * 3;
    const int32_t *sl = sel ? sel + (size_t)b * P : nullptr;
    // Up to GC points per thread (P <= 2048: every cloud and node set of the pair path) are read ONCE, all selections and then
    // all coordinates in flight together; the three passes below (bounding box, cell counts, scatter) run on the registers.
    // (As written first, each pass walked its points one dependent selection -> coordinate round trip at a time: ~48 round trips
    // per workgroup, 46 us per call and 22 calls per step of the pair path.)
    constexpr int GC = 8;
    const bool cached = P <= GC * GRID_T;   // (uniform)
    float px[GC], py[GC], pz[GC];
    if (cached) {
        int vs[GC];
#pragma unroll
        for (int q = 0; q < GC; ++q) {
            const int j = tid + q * GRID_T, jc = j < P ? j : P - 1;
            vs[q] = sl ? sl[jc] : jc;
        }
#pragma unroll
        for (int q = 0; q < GC; ++q) px[q] = p[3 * vs[q]], py[q] = p[3 * vs[q] + 1], pz[q] = p[3 * vs[q] + 2];
    }
    auto each_point = [&](auto &&f) __attribute__((always_inline)) {   // f(j, x, y, z) for this thread's points j
        if (cached) {
#pragma unroll
            for (int q = 0; q < GC; ++q) {
                const int j = tid + q * GRID_T;
                if (j < P) f(j, px[q], py[q], pz[q]);
            }
        } else {
            for (int j = tid; j < P; j += GRID_T) {
                const int v = sl ? sl[j] : j;
                f(j, p[3 * v], p[3 * v + 1], p[3 * v + 2]);
            }
        }
    };
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    each_point([&](int, float x, float y, float z) {
        mn[0] = fminf(mn[0], x), mn[1] = fminf(mn[1], y), mn[2] = fminf(mn[2], z);
        mx[0] = fmaxf(mx[0], x), mx[1] = fmaxf(mx[1], y), mx[2] = fmaxf(mx[2], z);
    });
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
        }
        if (lane == 0) {
            red[a][wave] = mn[a];
            red[3 + a][wave] = mx[a];
        }
    }
    __syncthreads();
    if (tid == 0) {
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(fminf(red[a][0], red[a][1]), fminf(red[a][2], red[a][3]));
            hi[a] = fmaxf(fmaxf(red[3 + a][0], red[3 + a][1]), fmaxf(red[3 + a][2], red[3 + a][3]));
        }
        float ext = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
        float h = ext > 0.f ? (ext / (float)G) * 1.000001f : 1.f;
        float m = 0.f;
        for (int a = 0; a < 3; ++a) m = fmaxf(m, fmaxf(fabsf(lo[a]), fabsf(hi[a])));
        par[0] = lo[0], par[1] = lo[1], par[2] = lo[2], par[3] = h, par[4] = 3.f * m * m + 1e-30f;
        for (int q = 0; q < 5; ++q) gb.params[(size_t)b * 8 + q] = par[q];
    }
    for (int c = tid; c <= G3; c += GRID_T) lds[c] = 0;
    __syncthreads();
    const float ox = par[0], oy = par[1], oz = par[2], inv = 1.0f / par[3];
    auto cell_of = [&](float x, float y, float z) {
        int cx = (int)((x - ox) * inv), cy = (int)((y - oy) * inv), cz = (int)((z - oz) * inv);
        cx = cx < 0 ? 0 : (cx > G - 1 ? G - 1 : cx);
        cy = cy < 0 ? 0 : (cy > G - 1 ? G - 1 : cy);
        cz = cz < 0 ? 0 : (cz > G - 1 ? G - 1 : cz);
        return (cz * G + cy) * G + cx;
    };
    each_point([&](int, float x, float y, float z) { atomicAdd(&lds[cell_of(x, y, z)], 1); });
    __syncthreads();
    // exclusive scan of G3 counts (each thread owns a contiguous chunk)
    const int chunk = (G3 + GRID_T - 1) / GRID_T;
    int local = 0;
    for (int c = tid * chunk; c < (tid + 1) * chunk && c < G3; ++c) local += lds[c];
    int inc = local;
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = inc - local;
    for (int w2 = 0; w2 < wave; ++w2) base += wsum[w2];
    __syncthreads();
    int *cursor = lds + G3 + 1;
    int run = base;
    for (int c = tid * chunk; c < (tid + 1) * chunk && c < G3; ++c) {
        int cnt = lds[c];
        lds[c] = run;
        cursor[c] = run;
        run += cnt;
    }
    if (tid == 0) lds[G3] = P;
    __syncthreads();
    int32_t *st = gb.start + (size_t)b * (G3 + 1);
    for (int c = tid; c <= G3; c += GRID_T) st[c] = lds[c];
    float4 *op = gb.pts + (size_t)b * P;
    int32_t *oi = gb.ids + (size_t)b * P;
    each_point([&](int j, float x, float y, float z) {
        const int pos = atomicAdd(&cursor[cell_of(x, y, z)], 1);
        op[pos] = make_float4(x, y, z, sumsq3(x, y, z));
        oi[pos] = j;
    });
}

__device__ __forceinline__ GridView grid_view(const GridBuf &gb, int b) {
    GridView g;
    const int G3 = gb.G * gb.G * gb.G;
    g.pts = gb.pts + (size_t)b * gb.P;
    g.ids = gb.ids + (size_t)b * gb.P;
    g.start = gb.start + (size_t)b * (G3 + 1);
    const float *pr = gb.params + (size_t)b * 8;
    g.ox = pr[0], g.oy = pr[1], g.oz = pr[2], g.h = pr[3], g.scale2 = pr[4];
    g.G = gb.G;
    return g;
}

// distance functors: return the RANKING key of candidate c (x,y,z,|c|^2) for query q
struct MetricMMQueryRow {  // torch.cdist matmul form, query is the row operand (knn_grad)
    typedef float key_t;
    float qx, qy, qz, nq;
    __device__ __forceinline__ void set(float x, float y, float z) { qx = x, qy = y, qz = z, nq = sumsq3(x, y, z); }
    __device__ __forceinline__ float operator()(const float4 &c) const {
        return sqrt_rn(d2_mm3(qx, qy, qz, nq, c.x, c.y, c.z, c.w));
    }
    __device__ __forceinline__ static float to_d2(float k) { return k * k; }
};
struct MetricMMCandRow {  // matmul form with the CANDIDATE as row operand (geod[nodes_idx] transposed)
    typedef float key_t;
    float qx, qy, qz, nq;
    __device__ __forceinline__ void set(float x, float y, float z) { qx = x, qy = y, qz = z, nq = sumsq3(x, y, z); }
    __device__ __forceinline__ float operator()(const float4 &c) const {
        return sqrt_rn(d2_mm3(c.x, c.y, c.z, c.w, qx, qy, qz, nq));
    }
    __device__ __forceinline__ static float to_d2(float k) { return k * k; }
};
struct MetricF64 {  // scipy KDTree: exact fp64 squared distance of the fp32 coordinates
    typedef double key_t;
    float qx, qy, qz;
    __device__ __forceinline__ void set(float x, float y, float z) { qx = x, qy = y, qz = z; }
    __device__ __forceinline__ double operator()(const float4 &c) const {
        double dx = (double)qx - (double)c.x, dy = (double)qy - (double)c.y, dz = (double)qz - (double)c.z;
        double s = 0.0;
        s = s + dx * dx;
        s = s + dy * dy;
        s = s + dz * dz;
        return s;
    }
    __device__ __forceinline__ static float to_d2(double k) { return (float)k; }
};
struct MetricF64Rounded {  // the same distance rounded to fp32: a screening key (weakly monotone in the fp64 value)
    typedef float key_t;
    MetricF64 m;
    __device__ __forceinline__ void set(float x, float y, float z) { m.set(x, y, z); }
    __device__ __forceinline__ float operator()(const float4 &c) const { return (float)m(c); }
    __device__ __forceinline__ static float to_d2(float k) { return k; }
};
struct MetricDiff {  // chamfer: (dx^2 + dy^2) + dz^2 in fp32, no contraction
    typedef float key_t;
    float qx, qy, qz;
    __device__ __forceinline__ void set(float x, float y, float z) { qx = x, qy = y, qz = z; }
    __device__ __forceinline__ float operator()(const float4 &c) const { return d2_diff3(qx, qy, qz, c.x, c.y, c.z); }
    __device__ __forceinline__ static float to_d2(float k) { return k; }
};

// Walk cubes of radius R = 1, 2, ... around the query's cell until the K-th best key is certified:
// every unvisited point is at least `face` away (true distance), the reference-rounded squared
// distance of such a point is >= face^2 - margin, so once kth_d2 < face^2*(1-1e-4) - margin nothing
// outside can rank before the current K-th.
// Returns true once the list is certified (always, unless Rmax stops the walk first).
template <int K, class Metric, class List>
__device__ __forceinline__ bool grid_search(const GridView &g, float qx, float qy, float qz, Metric &met, List &kb,
                                            int R0 = 1 /* cubes below R0: already in kb */, int Rmax = 1 << 30) {
    const int G = g.G;
    const float inv = 1.0f / g.h;
    int cx = (int)((qx - g.ox) * inv), cy = (int)((qy - g.oy) * inv), cz = (int)((qz - g.oz) * inv);
    cx = cx < 0 ? 0 : (cx > G - 1 ? G - 1 : cx);
    cy = cy < 0 ? 0 : (cy > G - 1 ? G - 1 : cy);
    cz = cz < 0 ? 0 : (cz > G - 1 ? G - 1 : cz);
    const float q2 = sumsq3(qx, qy, qz);
    const float margin = 64.f * 1.1920929e-7f * (g.scale2 + q2) + 1e-30f;
    int Rprev = R0 > 1 ? R0 - 1 : -1;
    // (Measured, round 3: the first cube with all 18 row bounds requested together and the candidates four to a round trip —
    // the shape of grid_chamfer_kernel's fast path — made the xyz kNN SLOWER, 1.30 -> 1.48 ms per launch of 1024 clouds, and
    // left the ring / influence searches where they were: with 16+ waves per SIMD the dependent loads are already covered, and
    // the extra registers and list insertions behind predicates cost more than the round trips they save.)
    for (int R = R0; R <= G && R <= Rmax; ++R) {
        const int x0 = cx - R < 0 ? 0 : cx - R, x1 = cx + R > G - 1 ? G - 1 : cx + R;
        const int y0 = cy - R < 0 ? 0 : cy - R, y1 = cy + R > G - 1 ? G - 1 : cy + R;
        const int z0 = cz - R < 0 ? 0 : cz - R, z1 = cz + R > G - 1 ? G - 1 : cz + R;
        // cells that are adjacent in x are adjacent in memory: each (z, y) row of the cube is one
        // contiguous candidate range (two when the row crosses the already visited inner cube)
        // A (z, y) row of cells whose slab lies farther from the query than the list's current worst entry cannot contribute (the
        // same inequality, with the same margins, as the certification below): it is skipped without a load.  The first cube
        // of an empty list skips nothing; from the second shell on — and in walks that start from a seeded list — most rows go.
        // (DVM_CHAMFER_STATS: 6 - 8 % of the bench's Chamfer queries need a second shell, and a wave with one such lane used to
        // walk all 98 cells of it.)
        for (int z = z0; z <= z1; ++z) {
            const float zlo = g.oz + (float)z * g.h;
            const float dz = fmaxf(0.f, fmaxf(zlo - qz, qz - (zlo + g.h)));
            for (int y = y0; y <= y1; ++y) {
                const float ylo = g.oy + (float)y * g.h;
                const float dy = fmaxf(0.f, fmaxf(ylo - qy, qy - (ylo + g.h)));
                if (Metric::to_d2(kb.worst()) < (dy * dy + dz * dz) * 0.9999f - margin) continue;
                const int rowbase = (z * G + y) * G;
                const bool inner_zy = (abs(z - cz) <= Rprev) && (abs(y - cy) <= Rprev);
                int sa0, sa1, sb0 = 0, sb1 = 0;
                if (inner_zy) {
                    const int ix0 = cx - Rprev < 0 ? 0 : cx - Rprev;  // visited run, clipped like the cube
                    const int ix1 = cx + Rprev > G - 1 ? G - 1 : cx + Rprev;
                    sa0 = g.start[rowbase + x0];
                    sa1 = g.start[rowbase + ix0];
                    sb0 = g.start[rowbase + ix1 + 1];
                    sb1 = g.start[rowbase + x1 + 1];
                } else {
                    sa0 = g.start[rowbase + x0];
                    sa1 = g.start[rowbase + x1 + 1];
                }
                for (int s = sa0; s < sa1; ++s) kb.insert_lex(met(g.pts[s]), g.ids[s]);
                for (int s = sb0; s < sb1; ++s) kb.insert_lex(met(g.pts[s]), g.ids[s]);
            }
        }
        // certification.  An unvisited point lies inside the grid's box but beyond one of the cube's
        // (unclipped) faces, say along axis a:  |p - q|^2 >= f_a^2 + sum_{b != a} e_b^2, with f_a the
        // distance from q to that face and e_b the distance from q to the box along axis b (0 inside).
        const float ext = (float)G * g.h;
        const float ex = fmaxf(0.f, fmaxf(g.ox - qx, qx - (g.ox + ext)));
        const float ey = fmaxf(0.f, fmaxf(g.oy - qy, qy - (g.oy + ext)));
        const float ez = fmaxf(0.f, fmaxf(g.oz - qz, qz - (g.oz + ext)));
        const float exx = ex * ex, eyy = ey * ey, ezz = ez * ez;
        float bound2 = INFINITY;
        auto face_x = [&](float f) { f = fmaxf(f, 0.f); bound2 = fminf(bound2, f * f + eyy + ezz); };
        auto face_y = [&](float f) { f = fmaxf(f, 0.f); bound2 = fminf(bound2, f * f + exx + ezz); };
        auto face_z = [&](float f) { f = fmaxf(f, 0.f); bound2 = fminf(bound2, f * f + exx + eyy); };
        if (cx - R >= 1) face_x(qx - (g.ox + (float)(cx - R) * g.h));
        if (cx + R < G - 1) face_x((g.ox + (float)(cx + R + 1) * g.h) - qx);
        if (cy - R >= 1) face_y(qy - (g.oy + (float)(cy - R) * g.h));
        if (cy + R < G - 1) face_y((g.oy + (float)(cy + R + 1) * g.h) - qy);
        if (cz - R >= 1) face_z(qz - (g.oz + (float)(cz - R) * g.h));
        if (cz + R < G - 1) face_z((g.oz + (float)(cz + R + 1) * g.h) - qz);
        if (bound2 == INFINITY) return true;  // the cube covers the whole grid
        const float kth = Metric::to_d2(kb.worst());
        if (kth < bound2 * 0.9999f - margin) return true;
        Rprev = R;
    }
    return Rmax >= G;   // (walked the whole grid: certified by exhaustion)
}

// The workgroup copies a grid's sorted points and cell table into LDS and repoints the view at the copy: every candidate and
// range-bound read of a search is then an LDS read instead of an L2 round trip (the original indices stay in global memory).
// Returns the bytes used.  All threads of the workgroup must call it; the caller synchronises afterwards.
__device__ __forceinline__ size_t grid_stage_lds(GridView &g, int P, char *lds, int threads) {
    const int G3 = g.G * g.G * g.G;
    float4 *lp = (float4 *)lds;
    int32_t *ls = (int32_t *)(lds + (size_t)P * sizeof(float4));
    for (int i = threadIdx.x; i < P; i += threads) lp[i] = g.pts[i];
    for (int i = threadIdx.x; i <= G3; i += threads) ls[i] = g.start[i];
    g.pts = lp;
    g.start = ls;
    return ((size_t)P * sizeof(float4) + (size_t)(G3 + 1) * sizeof(int32_t) + 15) / 16 * 16;
}
static size_t grid_lds_bytes(const GridBuf &gb) {
    return ((size_t)gb.P * sizeof(float4) + ((size_t)gb.G * gb.G * gb.G + 1) * sizeof(int32_t) + 15) / 16 * 16;
}

// ---------------------------------------------------------------- kernels on top of grid_search
// xyz kNN of a cloud against itself (knn_grad): thread t handles the t-th point in cell order.
template <int K, int THREADS = 128, bool LDS_GRID = false>
__global__ __launch_bounds__(THREADS) void grid_knn_self_kernel(GridBuf gb, int k, int32_t *__restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) char kn_lds[];
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int P = gb.P;
    GridView g = grid_view(gb, b);
    if (LDS_GRID) {
        grid_stage_lds(g, P, kn_lds, THREADS);
        __syncthreads();
    }
    if (t >= P) return;
    const float4 qp = g.pts[t];
    MetricMMQueryRow met;
    met.set(qp.x, qp.y, qp.z);
    KBestPacked<K> kb;   // (keys: matmul-form squared distances clamped at 0)
    kb.init(INFINITY);
    grid_search<K, MetricMMQueryRow>(g, qp.x, qp.y, qp.z, met, kb);
    int32_t *o = idx + ((size_t)b * P + g.ids[t]) * k;
    for (int q = 0; q < K; ++q)
        if (q < k) o[q] = q < P ? kb.idx_at(q) : 0;
}

// node ring: 9-NN among nodes in fp64 (grid over the nodes, queries = nodes in cell order)
template <int THREADS = 128, bool LDS_GRID = false>
__global__ __launch_bounds__(THREADS) void grid_ring_kernel(GridBuf gb, int32_t *__restrict__ ring) {
    extern __shared__ __attribute__((aligned(16))) char rg_lds[];
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int P = gb.P;
    GridView g = grid_view(gb, b);
    if (LDS_GRID) {
        grid_stage_lds(g, P, rg_lds, THREADS);
        __syncthreads();
    }
    if (t >= P) return;
    const float4 qp = g.pts[t];
    // Screened: the search runs on the fp64 distance ROUNDED to fp32, packed with the index (KBestPacked: two instructions per
    // compare-swap instead of the fourteen of a (double, index) pair), on a list of 10 certified against its 9th entry.  Rounding
    // keeps the order except between values it merges: if two neighbours of the first ten share their fp32 key (a relative gap
    // below 6e-8: about one node in 10^5) the node is searched again with the fp64 keys themselves.  Same rings, bit for bit.
    const int a = g.ids[t];
    int32_t *o = ring + ((size_t)b * P + a) * 9;
    MetricF64Rounded mr;
    mr.set(qp.x, qp.y, qp.z);
    KBestPacked<10, 8> ks;
    ks.init(INFINITY);
    grid_search<9, MetricF64Rounded>(g, qp.x, qp.y, qp.z, mr, ks);
    bool ambiguous = false;
#pragma unroll
    for (int q = 0; q < 9; ++q) ambiguous = ambiguous || (ks.key_at(q) == ks.key_at(q + 1) && ks.key_at(q) < INFINITY);
    if (!ambiguous) {
        for (int q = 0; q < 9; ++q) o[q] = q < P ? ks.idx_at(q) : a;
        return;
    }
    MetricF64 met;
    met.set(qp.x, qp.y, qp.z);
    KBest<9, double> kb;
    kb.init((double)INFINITY);
    grid_search<9, MetricF64>(g, qp.x, qp.y, qp.z, met, kb);
    for (int q = 0; q < 9; ++q) o[q] = q < P ? kb.idx[q] : a;
}

// influence nodes (3 nearest nodes, matmul form with the node as row operand) on the node grid, and
// the fp64 distance to the nearest other vertex on the vertex grid
template <int THREADS = 128, int LDS_GRID = 0>   // LDS_GRID: bit 0 = the node grid, bit 1 = the vertex grid staged in LDS
__global__ __launch_bounds__(THREADS) void grid_infl_kernel(const float *__restrict__ xyz, int N, GridBuf gnodes, GridBuf gverts,
                                                            int32_t *__restrict__ infl, float *__restrict__ dists,
                                                            double *__restrict__ nnd) {
    extern __shared__ __attribute__((aligned(16))) char in_lds[];
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    GridView gv = grid_view(gverts, b);
    GridView gn = grid_view(gnodes, b);
    if (LDS_GRID) {   // the nodes' grid (three nearest nodes) and / or the vertices' (nearest other vertex)
        size_t used = 0;
        if (LDS_GRID & 2) used = grid_stage_lds(gv, gverts.P, in_lds, THREADS);
        if (LDS_GRID & 1) grid_stage_lds(gn, gnodes.P, in_lds + used, THREADS);
        __syncthreads();
    }
    if (t >= N) return;
    const float4 qp = gv.pts[t];  // vertices in cell order (coherent waves)
    const int i = gv.ids[t];
    {
        MetricMMCandRow met;
        met.set(qp.x, qp.y, qp.z);
        KBestPacked<3> kb;
        kb.init(INFINITY);
        grid_search<3, MetricMMCandRow>(gn, qp.x, qp.y, qp.z, met, kb);
        const size_t row = (size_t)b * N + i;
        for (int q = 0; q < 3; ++q) {
            infl[row * 3 + q] = q < gnodes.P ? kb.idx_at(q) : 0;
            dists[row * 3 + q] = kb.key_at(q);
        }
    }
    {
        MetricF64 met;
        met.set(qp.x, qp.y, qp.z);
        KBest<2, double> kb;
        kb.init((double)INFINITY);
        grid_search<2, MetricF64>(gv, qp.x, qp.y, qp.z, met, kb);
        nnd[(size_t)b * N + i] = sqrt(kb.key[1]);
    }
    (void)xyz;
}

// Chamfer: nearest neighbour of every a-point in cloud b (difference form), grouped launches
struct ChGridGroup {
    GridBuf gq;  // grid of the QUERY cloud: queries are taken in its cell order so that the lanes of a
                 // wave walk the same target cells (coalesced / broadcast loads)
    GridBuf gb;  // grid of the target cloud
    float *dout;
    int32_t *iout;
};
struct ChGridArgs {
    ChGridGroup g[8];
    int scan_min;   // uncertified lanes in a wave from which the whole target is scanned instead of walked
    unsigned long long *stats;   // diagnostic (DVM_DEBUG & 4): per group [8]: queries, radius-1 candidates, certified at radius 1,
                                 // walked, certified by the walk, waves that scanned, lanes served by a scan, exact fallback lanes
};
typedef float f32x16_g __attribute__((ext_vector_type(16)));

// the target's points [s_begin, s_end) in storage order with the reference's arithmetic (difference form, lower original index on
// exact ties)
__device__ __forceinline__ void chamfer_exact_scan(const GridView &g, const MetricDiff &met, int s_begin, int s_end, float &best, int &bs) {
    for (int s0 = s_begin; s0 < s_end; s0 += 4) {
        float4 pc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pc[u] = g.pts[s0 + u < s_end ? s0 + u : s_end - 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int sidx = s0 + u;
            const float d = sidx < s_end ? met(pc[u]) : INFINITY;
            if (d < best) {
                best = d, bs = sidx;
            } else if (d == best && d < INFINITY && g.ids[sidx] < g.ids[bs]) {  // exact tie: lower original index
                bs = sidx;
            }
        }
    }
}

// One WAVE of queries against the WHOLE target, screened on the matrix cores (described in grid_chamfer_kernel); lanes with `done`
// take part in the matrix instructions and write nothing.
__device__ __forceinline__ void chamfer_scan_wave(const ChGridGroup &G, const GridView &g, int b, int Na, bool done, const float4 qp, int i,
                                                  const MetricDiff &met, float margin, unsigned long long *stats, int grp) {
    const int P = G.gb.P;
    const int lane = threadIdx.x & 63, j32 = lane & 31, hh = lane >> 5;
    if (stats) {
        if (lane == 0) atomicAdd(stats + grp * 8 + 5, 1ull);
        if (!done) atomicAdd(stats + grp * 8 + 6, 1ull);
    }
    float bq[2][2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        const int src = j32 + 32 * tl;
        const float x = __shfl(qp.x, src, 64), y = __shfl(qp.y, src, 64), z = __shfl(qp.z, src, 64);
        bq[tl][0] = hh ? y : x;
        bq[tl][1] = hh ? 1.f : z;
    }
    // per half of the wave's queries: the two smallest tile minima with their tiles, and the smallest of all the others
    float tb[2] = {INFINITY, INFINITY}, ts[2] = {INFINITY, INFINITY}, th[2] = {INFINITY, INFINITY};
    int tt[2] = {0, 0}, tu[2] = {0, 0};
    // (four tiles per trip, the next four requested before these are used: one exposed round trip per scan instead of one per tile -
    // in the retry kernel few waves scan at a time and nothing else covers a load per tile)
    constexpr int TB = 4;
    float4 nx[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) {
        const int pi = 32 * u + j32;
        nx[u] = g.pts[pi < P ? pi : P - 1];
    }
    for (int s00 = 0; s00 < P; s00 += 32 * TB) {
        float4 cur[TB];
#pragma unroll
        for (int u = 0; u < TB; ++u) {
            cur[u] = nx[u];
            const int pi = s00 + 32 * (TB + u) + j32;
            nx[u] = g.pts[pi < P ? pi : P - 1];
        }
#pragma unroll
        for (int u = 0; u < TB; ++u) {
            const int s0 = s00 + 32 * u, pi = s0 + j32;
            const float4 pc = cur[u];
            const float pn = pi < P ? pc.w : INFINITY;   // (rows past the end: +inf, never the minimum)
            const float a0 = hh ? -2.f * pc.y : -2.f * pc.x, a1 = hh ? pn : -2.f * pc.z;
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                f32x16_g acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq[tl][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq[tl][1], acc, 0, 0, 0);
                float m = fminf(fminf(fminf(acc[0], acc[1]), fminf(acc[2], acc[3])), fminf(fminf(acc[4], acc[5]), fminf(acc[6], acc[7])));
                m = fminf(m, fminf(fminf(fminf(acc[8], acc[9]), fminf(acc[10], acc[11])), fminf(fminf(acc[12], acc[13]), fminf(acc[14], acc[15]))));
                const bool lt1 = m < tb[tl], lt2 = m < ts[tl];   // (tiles past the end are all +inf: never smaller)
                th[tl] = lt2 ? ts[tl] : fminf(th[tl], m);
                tu[tl] = lt1 ? tt[tl] : (lt2 ? s0 : tu[tl]);
                ts[tl] = lt1 ? tb[tl] : (lt2 ? m : ts[tl]);
                tt[tl] = lt1 ? s0 : tt[tl];
                tb[tl] = lt1 ? m : tb[tl];
            }
        }
    }
    // a query's 32 points per tile sit in two lanes (its own and lane ^ 32): merge the halves at the query's own lane
    // (each lane saw HALF of every tile's points: a tile's minimum is the smaller of the two lanes' values for it.  Of the four
    // (value, tile) pairs the two smallest with different tiles name the tiles that are evaluated exactly; every other tile's
    // minimum is at least `sother` - a pair's value, or one of the two "all the others" minima.  A tile that ranks third or
    // lower in one lane is represented there by that lane's th: the bound can only be too careful, never too lax.)
    const float b1 = hh ? tb[1] : tb[0], s1 = hh ? ts[1] : ts[0], h1 = hh ? th[1] : th[0];
    const int t1 = hh ? tt[1] : tt[0], u1 = hh ? tu[1] : tu[0];
    const float b2 = __shfl_xor(hh ? tb[0] : tb[1], 32, 64), s2 = __shfl_xor(hh ? ts[0] : ts[1], 32, 64), h2 = __shfl_xor(hh ? th[0] : th[1], 32, 64);
    const int t2 = __shfl_xor(hh ? tt[0] : tt[1], 32, 64), u2 = __shfl_xor(hh ? tu[0] : tu[1], 32, 64);
    const float sbest = fminf(b1, b2);
    const int stile = b2 < b1 ? t2 : t1;
    const float c1 = t1 != stile ? b1 : INFINITY, c2 = u1 != stile ? s1 : INFINITY, c3 = t2 != stile ? b2 : INFINITY, c4 = u2 != stile ? s2 : INFINITY;
    const float ssec = fminf(fminf(c1, c2), fminf(c3, c4));
    const int stile2 = ssec == c1 ? t1 : (ssec == c2 ? u1 : (ssec == c3 ? t2 : u2));
    const float o1 = t1 != stile2 ? c1 : INFINITY, o2 = u1 != stile2 ? c2 : INFINITY, o3 = t2 != stile2 ? c3 : INFINITY, o4 = u2 != stile2 ? c4 : INFINITY;
    const float sother = fminf(fminf(h1, h2), fminf(fminf(o1, o2), fminf(o3, o4)));
    float best = INFINITY;
    int bs = 0;
    bool cert = false;
    if (!done) {
        chamfer_exact_scan(g, met, stile, stile + 32 < P ? stile + 32 : P, best, bs);
        if (ssec < INFINITY) chamfer_exact_scan(g, met, stile2, stile2 + 32 < P ? stile2 + 32 : P, best, bs);
        cert = sother - sbest > 2.f * margin;
    }
    if (__ballot(!done && !cert) != 0) {   // (rare: a near-tie between tiles, within the screening's error)
        if (!done && !cert) {
            if (stats) atomicAdd(stats + grp * 8 + 7, 1ull);
            best = INFINITY, bs = 0;
            chamfer_exact_scan(g, met, 0, P, best, bs);
        }
    }
    if (!done) {
        G.dout[(size_t)b * Na + i] = best;
        if (G.iout) G.iout[(size_t)b * Na + i] = g.ids[bs];
    }
}

// LDS_TARGET (round 4): the workgroup first copies the TARGET cloud's sorted points and cell table into LDS (32 KB + 7 KB at 2048
// points) and every candidate / range-bound read of the searches below is an LDS read instead of an L2 round trip — the kernel
// spent 58 % of its wave cycles parked on those (profiles/r3_pmc_summary.txt, r4_pmc_grid.txt).  Same arithmetic, same order,
// same results; the original indices (one read per query, more on exact ties) stay in global memory.
template <int THREADS, bool LDS_TARGET>
__global__ __launch_bounds__(THREADS) void grid_chamfer_kernel(const ChGridArgs args) {
    extern __shared__ __attribute__((aligned(16))) char ch_lds[];
    const ChGridGroup &G = args.g[blockIdx.z];
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int Na = G.gq.P;
    // (lanes past the end stay in the kernel, with the last query and nothing to write: the matrix-core scan below needs whole waves)
    const bool inrange = t < Na;
    GridView g = grid_view(G.gb, b);
    if (LDS_TARGET) {
        const int P = G.gb.P, G3 = G.gb.G * G.gb.G * G.gb.G;
        float4 *lp = (float4 *)ch_lds;
        int32_t *ls = (int32_t *)(ch_lds + (size_t)P * sizeof(float4));
        for (int i = threadIdx.x; i < P; i += THREADS) lp[i] = g.pts[i];
        for (int i = threadIdx.x; i <= G3; i += THREADS) ls[i] = g.start[i];
        __syncthreads();
        g.pts = lp;
        g.start = ls;
    }
    GridView q = grid_view(G.gq, b);
    const int tq = inrange ? t : Na - 1;
    const float4 qp = q.pts[tq];
    const int i = q.ids[tq];
    MetricDiff met;
    met.set(qp.x, qp.y, qp.z);
    const float q2 = sumsq3(qp.x, qp.y, qp.z);
    const float margin = 64.f * 1.1920929e-7f * (g.scale2 + q2) + 1e-30f;
    bool done = !inrange;
    float fbest = INFINITY;   // the radius-1 cube's result, kept for the walk below
    int fid = 0x7fffffff;
    // K = 1 fast path: the radius-1 cube is 9 contiguous (z, y) rows.  All 18 range bounds are requested together
    // (the generic walk chases start -> pts -> ids one row at a time: a dependent-load chain per row), only the
    // coordinates are read per candidate, the original index once at the end (and on exact ties).
    {
        const int Gd = g.G;
        const float inv = 1.0f / g.h;
        int cx = (int)((qp.x - g.ox) * inv), cy = (int)((qp.y - g.oy) * inv), cz = (int)((qp.z - g.oz) * inv);
        cx = cx < 0 ? 0 : (cx > Gd - 1 ? Gd - 1 : cx);
        cy = cy < 0 ? 0 : (cy > Gd - 1 ? Gd - 1 : cy);
        cz = cz < 0 ? 0 : (cz > Gd - 1 ? Gd - 1 : cz);
        const int x0 = cx - 1 < 0 ? 0 : cx - 1, x1 = cx + 1 > Gd - 1 ? Gd - 1 : cx + 1;
        int rs[9], re[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
            const bool in = z >= 0 && z < Gd && y >= 0 && y < Gd;
            const int rowbase = ((in ? z : cz) * Gd + (in ? y : cy)) * Gd;
            rs[r] = g.start[rowbase + x0];
            // (loaded unconditionally — the row is clamped above — and selected: a predicated load is a branch around a load, and
            // the compiler waits for rs[r] in front of it: the 18 bounds arrived in nine dependent round trips)
            const int rend = g.start[rowbase + x1 + 1];
            re[r] = in ? rend : rs[r];
        }
        float best = INFINITY;
        int bs = -1;
        bool tie = false;   // another candidate at exactly the best distance: resolved after the rows (lower original index wins)
        // the query's own row first, then the four rows beside it, then the corners: a row whose slab lies beyond the best found
        // so far (the inequality and margins of the certification) is skipped
        const float dlo[2] = {fmaxf(0.f, qp.y - (g.oy + (float)cy * g.h)), fmaxf(0.f, qp.z - (g.oz + (float)cz * g.h))};
        const float dhi[2] = {fmaxf(0.f, (g.oy + (float)(cy + 1) * g.h) - qp.y), fmaxf(0.f, (g.oz + (float)(cz + 1) * g.h) - qp.z)};
        constexpr int order[9] = {4, 1, 3, 5, 7, 0, 2, 6, 8};
        float rowb[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const float dyr = r % 3 == 1 ? 0.f : (r % 3 == 0 ? dlo[0] : dhi[0]), dzr = r / 3 == 1 ? 0.f : (r / 3 == 0 ? dlo[1] : dhi[1]);
            rowb[r] = (dyr * dyr + dzr * dzr) * 0.9999f - margin;
        }
        // The candidate update is straight-line: strictly smaller -> take it (a select and a min), equal -> remember that there
        // was a tie.  (As `if (d < best) ... else if (d == best && ids[s] < ids[bs])` every candidate cost two branches around
        // the tie path - ~22 instructions per candidate, a third of them scalar mask bookkeeping.)
#pragma unroll
        for (int ri = 0; ri < 9; ++ri) {
            const int r = order[ri];
            if (best < rowb[r]) continue;
            const int rend = re[r];
            for (int s0 = rs[r]; s0 < rend; s0 += 4) {  // four candidates in flight (each iteration otherwise waits a full load)
                float4 pc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) pc[u] = g.pts[s0 + u < rend ? s0 + u : rend - 1];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int s = s0 + u;
                    float dm = met(pc[u]);
                    asm volatile("" : "+v"(dm));   // (evaluated for every lane, then selected: no branch around eight instructions)
                    const bool inb = s < rend;
                    const float d = inb ? dm : INFINITY;
                    const bool lt = d < best, eq = d == best && inb && d < INFINITY;   // (inf == inf while bs is still -1 is NOT a tie: there is no candidate to resolve)
                    tie = lt ? false : (tie || eq);
                    bs = lt ? s : bs;
                    best = lt ? d : best;
                }
            }
        }
        if (tie && bs >= 0) {   // exact ties (duplicated target points): the lowest original index among the candidates at the best distance
            int bid = g.ids[bs];
#pragma unroll 1
            for (int r = 0; r < 9; ++r) {
                if (best < rowb[r]) continue;
                for (int s = rs[r]; s < re[r]; ++s) {
                    if (met(g.pts[s]) == best) {
                        const int id = g.ids[s];
                        if (id < bid) bid = id, bs = s;
                    }
                }
            }
        }
        // certification of the radius-1 cube (same bound as grid_search)
        const float ext = (float)Gd * g.h;
        const float ex = fmaxf(0.f, fmaxf(g.ox - qp.x, qp.x - (g.ox + ext)));
        const float ey = fmaxf(0.f, fmaxf(g.oy - qp.y, qp.y - (g.oy + ext)));
        const float ez = fmaxf(0.f, fmaxf(g.oz - qp.z, qp.z - (g.oz + ext)));
        const float exx = ex * ex, eyy = ey * ey, ezz = ez * ez;
        float bound2 = INFINITY;
        auto face = [&](float f, float o2) { f = fmaxf(f, 0.f); bound2 = fminf(bound2, f * f + o2); };
        if (cx - 1 >= 1) face(qp.x - (g.ox + (float)(cx - 1) * g.h), eyy + ezz);
        if (cx + 1 < Gd - 1) face((g.ox + (float)(cx + 2) * g.h) - qp.x, eyy + ezz);
        if (cy - 1 >= 1) face(qp.y - (g.oy + (float)(cy - 1) * g.h), exx + ezz);
        if (cy + 1 < Gd - 1) face((g.oy + (float)(cy + 2) * g.h) - qp.y, exx + ezz);
        if (cz - 1 >= 1) face(qp.z - (g.oz + (float)(cz - 1) * g.h), exx + eyy);
        if (cz + 1 < Gd - 1) face((g.oz + (float)(cz + 2) * g.h) - qp.z, exx + eyy);
        if (args.stats && inrange) {
            int ncand = 0;
#pragma unroll
            for (int r = 0; r < 9; ++r) ncand += re[r] - rs[r];
            atomicAdd(args.stats + blockIdx.z * 8 + 0, 1ull);
            atomicAdd(args.stats + blockIdx.z * 8 + 1, (unsigned long long)ncand);
        }
        if (bs >= 0) fbest = best, fid = g.ids[bs];
        if (!done && bs >= 0 && (bound2 == INFINITY || best < bound2 * 0.9999f - margin)) {
            G.dout[(size_t)b * Na + i] = best;
            if (G.iout) G.iout[(size_t)b * Na + i] = fid;
            done = true;
            if (args.stats) atomicAdd(args.stats + blockIdx.z * 8 + 2, 1ull);
        }
    }
    // not certified within the radius-1 cube.  (Deferring these queries to a compacted second launch — marked here, gathered into
    // full waves and scanned by a second kernel — made this one 2.5x faster, 0.81 ms, and the second launch cost 0.79 ms: 1.60 vs
    // 1.51 ms.  The scans are bound by their own arithmetic, 2.5 G pair distances per launch, not by how full the scanning waves
    // are; queries far outside the target's box, as the bench's untrained warps produce, need most of the grid either way.)
    const int nfall = __popcll(__ballot(!done));   // (wave-uniform)
    if (nfall == 0) return;
    // When many lanes of the wave are in that position — a query cloud far from, or much larger than, the target: a COLLAPSED
    // correspondence image (flat soft-max rows at small alpha) makes every query of the other cloud such a one — the wave goes
    // through the WHOLE target instead of walking cells.  Round 3: that scan is screened on the matrix cores.  Per 32 target
    // points, two v_mfma_f32_32x32x2_f32 per half of the wave's queries give |p|^2 - 2 p.q — the squared distance less the
    // query's own |q|^2, which does not move a query's minimum — for 32 x 32 pairs (rows [-2 px, -2 py, -2 pz, |p|^2] against
    // columns [qx, qy, qz, 1]); a lane keeps, per query, the smallest value
    // it has seen, the tile it came from and the smallest value of any OTHER tile.  At the end the best tile's 32 points are
    // evaluated with the reference's difference form (ties by original index), and the result stands if every other tile's
    // minimum lies more than twice the screening's error bound above the best — otherwise that query is scanned exactly.
    // (The vector form of this scan was 12 instructions per pair and the whole of the kernel's time in the bench's regime:
    // 2.57 ms per launch of 4 x 512 clouds.)
    bool scan = nfall >= args.scan_min;   // (wave-uniform)
    if (!scan) {
        // few open lanes: they walk on from the radius-1 cube (seeded with its best) — but only two more shells: a lane still
        // open after radius 3 is a FAR query, for which the walk would visit most of the grid one dependent cell row at a
        // time (tens of thousands of instructions for the whole wave); the wave scans the target for it instead
        if (!done) {
            if (args.stats) atomicAdd(args.stats + blockIdx.z * 8 + 3, 1ull);
            KBest<1, float> kb;
            kb.key[0] = fbest, kb.idx[0] = fid;
            if (grid_search<1, MetricDiff>(g, qp.x, qp.y, qp.z, met, kb, 2, 3)) {
                if (args.stats) atomicAdd(args.stats + blockIdx.z * 8 + 4, 1ull);
                G.dout[(size_t)b * Na + i] = kb.key[0];
                // (non-finite coordinates leave the list empty: keep the index a valid row, the backward pass gathers through it)
                if (G.iout) G.iout[(size_t)b * Na + i] = (unsigned)kb.idx[0] < (unsigned)G.gb.P ? kb.idx[0] : 0;
                done = true;
            }
        }
        scan = __ballot(!done) != 0;
        if (!scan) return;
    }
    chamfer_scan_wave(G, g, b, Na, done, qp, i, met, margin, args.stats, blockIdx.z);
}

// ---------------------------------------------------------------- host side
size_t grid_bytes(int B, int P) {
    const int G = grid_dim_for(P), G3 = G * G * G;
    return align_up((size_t)B * P * sizeof(float4)) + align_up((size_t)B * P * sizeof(int32_t)) +
           align_up((size_t)B * (G3 + 1) * sizeof(int32_t)) + align_up((size_t)B * 8 * sizeof(float));
}

GridBuf grid_carve(Arena &ar, int B, int P) {
    GridBuf gb;
    gb.P = P;
    gb.G = grid_dim_for(P);
    const int G3 = gb.G * gb.G * gb.G;
    gb.pts = ar.take<float4>((size_t)B * P);
    gb.ids = ar.take<int32_t>((size_t)B * P);
    gb.start = ar.take<int32_t>((size_t)B * (G3 + 1));
    gb.params = ar.take<float>((size_t)B * 8);
    return gb;
}

void launch_grid_build_sets(const float *const *xyz, const int *Nsrc, const GridBuf *gb, int nsets, int B, hipStream_t s) {
    GridBuildSets sets;
    size_t lds = 0;
    for (int q = 0; q < 4; ++q) {
        const int r = q < nsets ? q : 0;
        sets.xyz[q] = xyz[r], sets.Nsrc[q] = Nsrc[r], sets.gb[q] = gb[r];
        const int G3 = gb[r].G * gb[r].G * gb[r].G;
        const size_t nb = (size_t)(2 * G3 + 1) * sizeof(int);
        lds = nb > lds ? nb : lds;
    }
    hipLaunchKernelGGL(grid_build_sets_kernel, dim3(B, nsets), dim3(GRID_T), lds, s, sets);
}
void launch_grid_build(const float *xyz, int B, int Nsrc, const int32_t *sel, const GridBuf &gb, hipStream_t s) {
    const int G3 = gb.G * gb.G * gb.G;
    size_t lds = (size_t)(2 * G3 + 1) * sizeof(int);
    hipLaunchKernelGGL(grid_build_kernel, dim3(B), dim3(GRID_T), lds, s, xyz, Nsrc, sel, gb);
}

// LDS-resident grids for the searches (round 4) where the grid fits 64 KB, 512 threads per workgroup (256 / 1024 measured no
// faster); larger clouds take the global-memory form with 128 threads
template <int K>
static void launch_knn_self_k(const GridBuf &gb, int B, int k, int32_t *idx, hipStream_t s) {
    const size_t lds = grid_lds_bytes(gb);
    if (lds <= 64 * 1024) {
        ensure_dyn_lds((const void *)grid_knn_self_kernel<K, 512, true>, (int)lds);
        hipLaunchKernelGGL((grid_knn_self_kernel<K, 512, true>), dim3((gb.P + 511) / 512, B), dim3(512), lds, s, gb, k, idx);
    } else {
        hipLaunchKernelGGL((grid_knn_self_kernel<K, 128, false>), dim3((gb.P + 127) / 128, B), dim3(128), 0, s, gb, k, idx);
    }
}

void launch_grid_knn_self(const GridBuf &gb, int B, int k, int32_t *idx, hipStream_t s) {
    prof_begin(s, DVM_PROF_KNN_XYZ);
    if (k <= 3)
        launch_knn_self_k<3>(gb, B, k, idx, s);
    else if (k <= 10)
        launch_knn_self_k<10>(gb, B, k, idx, s);
    else
        launch_knn_self_k<16>(gb, B, k, idx, s);
    prof_end(s, DVM_PROF_KNN_XYZ);
}

void launch_grid_ring(const GridBuf &gnodes, int B, int32_t *ring, hipStream_t s) {
    const size_t lds = grid_lds_bytes(gnodes);
    if (lds <= 64 * 1024) {
        ensure_dyn_lds((const void *)grid_ring_kernel<512, true>, (int)lds);
        hipLaunchKernelGGL((grid_ring_kernel<512, true>), dim3((gnodes.P + 511) / 512, B), dim3(512), lds, s, gnodes, ring);
    } else {
        hipLaunchKernelGGL((grid_ring_kernel<128, false>), dim3((gnodes.P + 127) / 128, B), dim3(128), 0, s, gnodes, ring);
    }
}

void launch_grid_infl(const float *xyz, int B, int N, const GridBuf &gnodes, const GridBuf &gverts, int32_t *infl, float *dists,
                      double *nnd, hipStream_t s) {
    // the NODE grid in LDS (with the vertex grid staged as well the kernel was slower: 0.59 vs 0.41 ms)
    const size_t lds = grid_lds_bytes(gnodes);
    if (lds <= 64 * 1024) {
        ensure_dyn_lds((const void *)grid_infl_kernel<512, 1>, (int)lds);
        hipLaunchKernelGGL((grid_infl_kernel<512, 1>), dim3((N + 511) / 512, B), dim3(512), lds, s, xyz, N, gnodes, gverts, infl, dists, nnd);
    } else {
        hipLaunchKernelGGL((grid_infl_kernel<128, 0>), dim3((N + 127) / 128, B), dim3(128), 0, s, xyz, N, gnodes, gverts, infl, dists, nnd);
    }
}

void launch_grid_chamfer(const GridBuf *gq, const GridBuf *gb, float *const *dout, int32_t *const *iout, int ngroups, int B,
                         hipStream_t s) {
    ChGridArgs args;
    int maxN = 1;
    for (int q = 0; q < 8; ++q) {
        int r = q < ngroups ? q : 0;
        args.g[q] = ChGridGroup{gq[r], gb[r], dout[r], iout ? iout[r] : nullptr};
        if (q < ngroups && gq[r].P > maxN) maxN = gq[r].P;
    }
    args.scan_min = 24;   // (round 3, matrix-core scan + seeded walk, bench regime: 8 / 16 / 24 / 32 / never = 2.75 / 2.32 / 2.22 / 2.30 / 2.81 ms)
    args.stats = nullptr;
    if ((options().debug & DVM_DEBUG_CHAMFER_STATS) && hipMalloc(&args.stats, 64 * sizeof(unsigned long long)) == hipSuccess)   // diagnostic: synchronous, allocates
        (void)hipMemsetAsync(args.stats, 0, 64 * sizeof(unsigned long long), s);
    // LDS-resident target, 512 threads per workgroup: every group's target must fit
    size_t lds_bytes = 0;
    for (int q = 0; q < ngroups; ++q) {
        const size_t nb = (size_t)gb[q].P * sizeof(float4) + ((size_t)gb[q].G * gb[q].G * gb[q].G + 1) * sizeof(int32_t);
        lds_bytes = nb > lds_bytes ? nb : lds_bytes;
    }
    lds_bytes = (lds_bytes + 15) / 16 * 16;
    prof_begin(s, DVM_PROF_CHAMFER);
    if (lds_bytes <= 64 * 1024) {
        ensure_dyn_lds((const void *)grid_chamfer_kernel<512, true>, (int)lds_bytes);
        hipLaunchKernelGGL((grid_chamfer_kernel<512, true>), dim3((maxN + 511) / 512, B, ngroups), dim3(512), lds_bytes, s, args);
    } else
        hipLaunchKernelGGL((grid_chamfer_kernel<128, false>), dim3((maxN + 127) / 128, B, ngroups), dim3(128), 0, s, args);
    prof_end(s, DVM_PROF_CHAMFER);
    if (args.stats) {
        unsigned long long h[64];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, args.stats, sizeof(h), hipMemcpyDeviceToHost);
        for (int q = 0; q < ngroups; ++q)
            fprintf(stderr, "chamfer group %d: %llu queries, %.1f radius-1 candidates each, %.1f%% certified there, %.1f%% walked (%.1f%% certified by the walk), "
                            "%llu waves scanned for %llu lanes, %llu exact fallbacks\n", q, h[q * 8], (double)h[q * 8 + 1] / (double)(h[q * 8] ? h[q * 8] : 1),
                    100.0 * h[q * 8 + 2] / (h[q * 8] ? h[q * 8] : 1), 100.0 * h[q * 8 + 3] / (h[q * 8] ? h[q * 8] : 1),
                    100.0 * h[q * 8 + 4] / (h[q * 8] ? h[q * 8] : 1), h[q * 8 + 5], h[q * 8 + 6], h[q * 8 + 7]);
        (void)hipFree(args.stats);
    }
}

}  // namespace dvm
