// dvm_knn_f16.hip — the feature-space self-kNN of the N2P blocks (reference models/model.py:267-278 `knn_new`: top-k of
// -|x_i - x_j|^2 per point, nearest first) WITHOUT the N x N score matrix: the sweep of the soft correspondence on the 16-bit
// matrix pipe finds the candidates, the reference's fp32 arithmetic is run only where the approximate order is not certain.
//
//   1. the features are split into two fp16 planes (dvm_softcorr_f16.h), the keys of a shape cut into S <= 8 slices, and the
//      second form of K1's sweep (dvm_softcorr_sweep2.hip: exact-split fp16 products, fp32 accumulate, the 12 smallest
//      approximate squared distances per row and launch entry, complete by construction) runs once over (shape, slice)
//      entries: up to 96 candidates per row, each with |approximate - exact| <= delta = HB_ERR (|q|^2 + max |k|^2);
//   2. one wave per row (knn_select_kernel) ranks the candidates by approximate distance.  With t = the k-th smallest:
//      every column that can be among the k nearest has approximate distance <= t + 2 delta, and a slice's list holds all
//      of its columns up to its own 12th entry — so the row is COMPLETE if no full list ends at or below t + 2 delta;
//      the ORDER of two candidates is certain if their approximate distances differ by more than 2 delta.  Only candidates
//      in a run of neighbours closer than that (a few per cent of the entries; always the band around the k-th) are
//      evaluated exactly — the reference's score (-|q|^2 - (-2 q.k)) - |k|^2 with the k-ordered fp32 fma chain, bit for bit
//      what knn_scores_mfma_kernel computes — and ranked inside their run by (score, column);
//   3. rows that are not complete, or hold runs longer than a wave can rank (heavy ties: duplicate points), are listed and
//      recomputed from all M exact scores by knn_exact_rows_kernel.
// Indices are bit-identical to the N x N path (tools/bench_knn.py: random, half-zero, clustered and duplicated features;
// tests/test_gpu_backbone.py::test_knn_f16_sweep_path_equals_dense_path).
//
// MEASURED, and therefore NOT the default (DVM_KNN_F16=1 selects it; profiles/r3_knn_f16.txt): 8 x 2048 points, C = 128, per
// call: sweep 113 us + selection 49 us + exact rows 92 us + preparation ~25 us against 176 us for the N x N path (fp32-MFMA
// scores 113 us + wave top-k 63 us).  The sweep's candidate machinery is built for 12 of 2048 (the two smallest of a lane's 16
// keys per sub-tile enter the list, a sub-tile whose third-smallest lies within the final bound is re-done): with the keys
// cut into 8 slices it selects 12 of 256 — a tenth of what it sees — so most sub-tiles are re-done from global memory
// (113 us instead of the 28 us the same products take inside K1), and 0.8 % of the rows overflow a slice's list and go
// through the exact kernel, whose per-thread row streaming is latency-bound.  What a k = 40 selection on this pipe needs is
// a sweep that emits everything below a per-row threshold (estimated from one slice) instead of maintaining lists.
#include <stdlib.h>

#include "dvm_common.h"
#include "dvm_softcorr_f16.h"

namespace dvm {

void launch_rownorm2(const float *x, int rows, int K, float *out, hipStream_t s);   // dvm_softcorr.hip

namespace {

using namespace k1;

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int KF_MAXS = 8;                    // key slices per shape
constexpr int KF_MAXRUN = 16;                 // longest run of uncertain neighbours ranked in the wave

__global__ __launch_bounds__(256) void kf_absmax_kernel(const float *__restrict__ x, long n4, int *__restrict__ out) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *(const f32x4 *)(x + 4 * i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // (out[1] mirrors out[0]: the sweep takes one scale per side; the plain read first keeps 8 000 atomics off two words)
    if ((threadIdx.x & 63) == 0 && __float_as_int(m) > __atomic_load_n(out + 1, __ATOMIC_RELAXED)) {
        atomicMax(out, __float_as_int(m));
        atomicMax(out + 1, __float_as_int(m));
    }
}

// rows of C (64 or 128) floats -> the two fp16 planes of a 128-wide row (the upper half zero at C = 64)
template <int C>
__global__ __launch_bounds__(256) void kf_split_kernel(const float *__restrict__ x, long rows, const int *__restrict__ maxbits,
                                                       char *__restrict__ planes) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one group of 4 columns per thread
    if (g >= rows * (HB_D / 4)) return;
    const float sc = pow2i(scale_exp(*maxbits));
    const long row = g / (HB_D / 4);
    const int c = (int)(g % (HB_D / 4));
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (4 * c < C) v = *(const f32x4 *)(x + row * C + 4 * c);
    f16x4 h, m;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xv = v[e] * sc;   // exact
        const _Float16 hh = (_Float16)xv;
        h[e] = hh, m[e] = (_Float16)(xv - (float)hh);
    }
    char *p = planes + row * HB_ROWB + 8 * c;
    *(f16x4 *)(p) = h;
    *(f16x4 *)(p + 256) = m;
}

__global__ void kf_norm_max_kernel(const float *__restrict__ nrm, int rows_per_batch, float *__restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < rows_per_batch ? nrm[(size_t)b * rows_per_batch + i] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax((int *)out + b, __float_as_int(v));
}

// order-preserving map float -> uint, ascending
__device__ __forceinline__ unsigned asc_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// the reference's score of (query row q, key row kk): (-|q|^2 - (-2 q.k)) - |k|^2, q.k a k-ordered fp32 fma chain from 0
template <int C>
__device__ __forceinline__ float exact_score(const float *__restrict__ q, const float *__restrict__ kk, float nq, float nk) {
    float acc = 0.f;
#pragma unroll 8
    for (int c = 0; c < C; c += 4) {
        const f32x4 qv = *(const f32x4 *)(q + c), kv = *(const f32x4 *)(kk + c);
        acc = fmaf(qv.x, kv.x, acc);
        acc = fmaf(qv.y, kv.y, acc);
        acc = fmaf(qv.z, kv.z, acc);
        acc = fmaf(qv.w, kv.w, acc);
    }
    const float inner = -2.f * acc;
    return (-nq - inner) - nk;
}

struct KFSel {
    const float *x;          // [B][N][C]
    const float *nrm;        // [B][N]
    const float *nmax;       // [B]
    const int32_t *cidx;     // [B][S][N][HB_KC] columns relative to the slice
    const float *cd2;        // [B][S][N][HB_KC] ascending
    int B, N, S, Ms, k;
    int32_t *idx;            // [B][N][k]
    int32_t *flagged;        // rows for knn_exact_rows_kernel
    int32_t *nflagged;
};

// one wave per row
template <int C>
__global__ __launch_bounds__(256) void knn_select_kernel(const KFSel a) {
    __shared__ unsigned long long sorted[4][2 * 64];   // (ascending key of the approximate distance << 32 | column)
    __shared__ unsigned char full_tail[4][2 * 64];      // 1: the entry is the 12th of a full list
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= (long)a.B * a.N) return;
    const int b = (int)(row / a.N), i = (int)(row - (long)b * a.N);
    const int nc = a.S * HB_KC, k = a.k;
    // candidate c = slice * 12 + t  ->  lanes hold c = lane and lane + 64
    unsigned long long ent[2];
    bool tail[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = lane + 64 * u;
        ent[u] = ~0ull;
        tail[u] = false;
        if (c < nc) {
            const int sl = c / HB_KC, t = c - sl * HB_KC;
            const size_t o = (((size_t)b * a.S + sl) * a.N + i) * HB_KC + t;
            const int j = a.cidx[o];
            const float d = a.cd2[o];
            if (j >= 0 && j < a.Ms && sl * a.Ms + j < a.N) {
                ent[u] = ((unsigned long long)asc_key(d) << 32) | (unsigned)(sl * a.Ms + j);
                tail[u] = t == HB_KC - 1;
            }
        }
    }
    // rank by counting (entries are distinct: the column breaks ties), scatter into sorted order
    int rank[2] = {0, 0};
    for (int c = 0; c < nc; ++c) {
        const unsigned lo0 = __builtin_amdgcn_readlane((unsigned)ent[0], c & 63), hi0 = __builtin_amdgcn_readlane((unsigned)(ent[0] >> 32), c & 63);
        const unsigned lo1 = __builtin_amdgcn_readlane((unsigned)ent[1], c & 63), hi1 = __builtin_amdgcn_readlane((unsigned)(ent[1] >> 32), c & 63);
        const unsigned long long o = c < 64 ? (((unsigned long long)hi0 << 32) | lo0) : (((unsigned long long)hi1 << 32) | lo1);
        rank[0] += o < ent[0] ? 1 : 0;
        rank[1] += o < ent[1] ? 1 : 0;
    }
    unsigned long long *so = sorted[wave];
    unsigned char *ft = full_tail[wave];
    so[lane] = ~0ull, so[lane + 64] = ~0ull;
    ft[lane] = 0, ft[lane + 64] = 0;
    // (LDS operations of one wave execute in order; invalid entries — all equal to ~0 — stay where the fill put them)
    if (ent[0] != ~0ull) so[rank[0]] = ent[0], ft[rank[0]] = tail[0];
    if (ent[1] != ~0ull) so[rank[1]] = ent[1], ft[rank[1]] = tail[1];
    const unsigned long long e0 = so[lane], e1 = so[lane + 64];
    const bool t0 = ft[lane] != 0, t1 = ft[lane + 64] != 0;
    // approximate distances back from the keys
    auto key_float = [](unsigned long long e) {
        const unsigned u = (unsigned)(e >> 32);
        return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
    };
    const float d0 = e0 != ~0ull ? key_float(e0) : INFINITY, d1 = e1 != ~0ull ? key_float(e1) : INFINITY;
    const float nq = a.nrm[row];
    const float delta2 = 2.f * HB_ERR * (nq + a.nmax[b]);
    const float tk = __shfl(d0, k - 1, 64);   // k <= 64: the k-th smallest sits in slot 0
    const float thr = tk + delta2;
    bool flag = !(tk < INFINITY);             // fewer than k candidates
    // complete?  no full list may end at or below the threshold
    flag = flag || __ballot((t0 && d0 <= thr) || (t1 && d1 <= thr)) != 0;
    // everything at or below the threshold has to fit the wave
    const int E = __popcll(__ballot(d0 <= thr));
    flag = flag || __ballot(d1 <= thr) != 0;
    // runs: entry p (sorted position = lane) starts a run unless it lies within 2 delta of its predecessor
    const float dprev = __shfl_up(d0, 1, 64);
    const bool in = lane < E;
    const bool starts = lane == 0 || !(d0 - dprev <= delta2);
    int rs = in && starts ? lane : 0;   // run start: prefix maximum over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(rs, o, 64);
        if (lane >= o) rs = max(rs, up);
    }
    const int rs_next = __shfl_down(rs, 1, 64);
    const bool uncertain = in && (rs != lane || (lane + 1 < E && rs_next == rs));   // the run has more than one member
    // run lengths: last member's position - start + 1
    const bool last = in && (lane + 1 >= E || rs_next != rs);
    const int len = last ? lane - rs + 1 : 0;
    const int maxlen = __reduce_max_sync(~0ull, len);
    flag = flag || maxlen > KF_MAXRUN;
    if (flag) {   // (wave-uniform)
        if (lane == 0) a.flagged[atomicAdd(a.nflagged, 1)] = (int32_t)row;
        return;
    }
    // exact scores where the order is not certain
    const int col = (int)(unsigned)e0;
    unsigned long long fin = e0;   // certain entries keep their approximate key: only (run start, rank in run) matters below
    if (uncertain) {
        const float *xb = a.x + (size_t)b * a.N * C;
        const float s = exact_score<C>(xb + (size_t)i * C, xb + (size_t)col * C, nq, a.nrm[(size_t)b * a.N + col]);
        fin = ((unsigned long long)(~asc_key(s)) << 32) | (unsigned)col;   // larger score first, then the lower column
    }
    int pos = lane;
    if (maxlen > 1) {   // rank inside the run
        int less = 0;
        for (int o = 1; o < maxlen; ++o) {
            const unsigned lu = __shfl_up((unsigned)fin, o, 64), hu = __shfl_up((unsigned)(fin >> 32), o, 64);
            const unsigned ld = __shfl_down((unsigned)fin, o, 64), hd = __shfl_down((unsigned)(fin >> 32), o, 64);
            const int ru = __shfl_up(rs, o, 64), rd = __shfl_down(rs, o, 64);
            const bool uin = lane >= o && ru == rs, din = lane + o < E && rd == rs;
            less += (uin && (((unsigned long long)hu << 32) | lu) < fin) ? 1 : 0;
            less += (din && (((unsigned long long)hd << 32) | ld) < fin) ? 1 : 0;
        }
        if (uncertain) pos = rs + less;
    }
    if (in && pos < k) a.idx[row * k + pos] = col;
}

// flagged rows from all M exact scores: one workgroup per row (persistent over the list)
template <int C>
__global__ __launch_bounds__(256) void knn_exact_rows_kernel(const float *__restrict__ x, const float *__restrict__ nrm, int B, int N, int k,
                                                             const int32_t *__restrict__ flagged, const int32_t *__restrict__ nflagged,
                                                             int32_t *__restrict__ idx) {
    extern __shared__ unsigned long long keys[];   // [N] (descending-score key << 32 | column); ~0 = taken
    __shared__ unsigned long long wbest[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = *nflagged;
    for (int f = blockIdx.x; f < n; f += gridDim.x) {
        const long row = flagged[f];
        const int b = (int)(row / N), i = (int)(row - (long)b * N);
        const float *xb = x + (size_t)b * N * C;
        const float nq = nrm[row];
        __syncthreads();
        for (int j = tid; j < N; j += 256) {
            const float s = exact_score<C>(xb + (size_t)i * C, xb + (size_t)j * C, nq, nrm[(size_t)b * N + j]);
            keys[j] = ((unsigned long long)(~asc_key(s)) << 32) | (unsigned)j;
        }
        __syncthreads();
        for (int r = 0; r < k; ++r) {
            unsigned long long m = ~0ull;
            for (int j = tid; j < N; j += 256) m = keys[j] < m ? keys[j] : m;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned long long y = ((unsigned long long)__shfl_xor((unsigned)(m >> 32), o, 64) << 32) | __shfl_xor((unsigned)m, o, 64);
                m = y < m ? y : m;
            }
            if (lane == 0) wbest[wave] = m;
            __syncthreads();
            unsigned long long w = wbest[0];
#pragma unroll
            for (int q = 1; q < 4; ++q) w = wbest[q] < w ? wbest[q] : w;
            const int j = (int)(unsigned)w;
            if (tid == 0) {
                idx[row * k + r] = j;
                keys[j] = ~0ull;
            }
            __syncthreads();
        }
    }
}

struct KFWs {
    char *planes, *nf;
    float *nrm, *nmax, *cd2, *lsum;
    int *amax;
    int32_t *cidx, *flag;
    int S, Ms, Np;
};
void kf_carve(Arena &ar, int B, int N, KFWs &w) {
    const int per = (N + KF_MAXS - 1) / KF_MAXS;
    w.Ms = (per + HB_KT - 1) / HB_KT * HB_KT;
    w.S = (N + w.Ms - 1) / w.Ms;
    w.Np = (N + HB_KT - 1) / HB_KT * HB_KT;
    const size_t R = (size_t)B * N;
    w.planes = ar.take<char>(R * HB_ROWB);
    w.nf = ar.take<char>((size_t)B * w.Np * 32);
    w.nrm = ar.take<float>(R);
    w.nmax = ar.take<float>(B);
    w.amax = ar.take<int>(2);
    w.cidx = ar.take<int32_t>(R * w.S * HB_KC);
    w.cd2 = ar.take<float>(R * w.S * HB_KC);
    w.lsum = ar.take<float>(R * w.S * 2);
    w.flag = ar.take<int32_t>(R + 1);
}

}  // namespace

bool knn_f16_applies(int B, int N, int M, int C, int k, bool self) {
    static const bool on = [] { const char *e = getenv("DVM_KNN_F16"); return e && atoi(e) != 0; }();   // opt-in: see the header
    (void)B;
    return on && self && N == M && (C == 64 || C == 128) && k <= 48 && N >= 512 && N <= 8192;
}

size_t knn_f16_ws_bytes(int B, int N) {
    Arena ar(nullptr, 0);
    KFWs w;
    kf_carve(ar, B, N, w);
    return ar.off;
}

// self-kNN of x [B][N][C]: idx [B][N][k], nearest first (ties: lower column first)
int launch_knn_f16(const float *x, int B, int N, int C, int k, int32_t *idx, void *ws, size_t ws_bytes, hipStream_t s) {
    Arena ar(ws, ws_bytes);
    KFWs w;
    kf_carve(ar, B, N, w);
    if (!ar.ok()) {
        set_error("knn (fp16 path): workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    const long R = (long)B * N;
    (void)hipMemsetAsync(w.nmax, 0, align_up((size_t)B * sizeof(float)) + align_up(2 * sizeof(int)), s);   // nmax, amax (adjacent)
    (void)hipMemsetAsync(w.flag, 0, sizeof(int32_t), s);
    launch_rownorm2(x, (int)R, C, w.nrm, s);
    hipLaunchKernelGGL(kf_absmax_kernel, dim3(1024), dim3(256), 0, s, x, R * C / 4, w.amax);
    hipLaunchKernelGGL(kf_norm_max_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, w.nrm, N, w.nmax);
    const unsigned sb = (unsigned)((R * (HB_D / 4) + 255) / 256);
    if (C == 64)
        hipLaunchKernelGGL(kf_split_kernel<64>, dim3(sb), dim3(256), 0, s, x, R, w.amax, w.planes);
    else
        hipLaunchKernelGGL(kf_split_kernel<128>, dim3(sb), dim3(256), 0, s, x, R, w.amax, w.planes);
    launch_norm_frags(w.nrm, B, N, w.Np, w.amax, w.nf, s);
    HBArgs a;
    a.g[0] = HBGroup{w.planes, w.planes, w.amax, w.amax + 1, w.nrm, nullptr, N, N, w.Np, (N + HB_QB - 1) / HB_QB, w.cidx, w.cd2, w.lsum, w.S, w.Ms};
    a.g[1] = a.g[0];
    a.blocks0 = B * w.S * a.g[0].tiles;
    a.neg_alpha = -100.f;   // only the candidate lists are used
    a.cutw = 0.f;
    a.route = nullptr;
    a.nb = B;
    launch_sweep2(a, w.nf, w.nf, w.amax, a.blocks0, 2, s);
    KFSel sel{x, w.nrm, w.nmax, w.cidx, w.cd2, B, N, w.S, w.Ms, k, idx, w.flag + 1, w.flag};
    const unsigned gb = (unsigned)((R + 3) / 4);
    const size_t lds = (size_t)N * sizeof(unsigned long long);
    if (C == 64) {
        hipLaunchKernelGGL(knn_select_kernel<64>, dim3(gb), dim3(256), 0, s, sel);
        ensure_dyn_lds((const void *)knn_exact_rows_kernel<64>, (int)lds);
        hipLaunchKernelGGL(knn_exact_rows_kernel<64>, dim3(512), dim3(256), lds, s, x, w.nrm, B, N, k, w.flag + 1, w.flag, idx);
    } else {
        hipLaunchKernelGGL(knn_select_kernel<128>, dim3(gb), dim3(256), 0, s, sel);
        ensure_dyn_lds((const void *)knn_exact_rows_kernel<128>, (int)lds);
        hipLaunchKernelGGL(knn_exact_rows_kernel<128>, dim3(512), dim3(256), lds, s, x, w.nrm, B, N, k, w.flag + 1, w.flag, idx);
    }
    return DVM_OK;
}

}  // namespace dvm
