// dvm_backbone.hip — LG-Net (Uni3FC) hot operators on gfx950:
//   K3  feature-space kNN (knn_new / knn): fp32-MFMA score tiles + per-row radix select
//   K4  neighbour-to-point attention (N2PAttention[_DIM]) on gathered projections
//   K5  offset self-attention of SA_Layer: symmetric energy, row softmax, column re-normalisation
//   positional sin/cos encoding, and the dist-loss term (K10).
// Reference: models/model.py:267-278 (knn_new), 325-395 (N2PAttention), 97-123 (SA_Layer),
// 544-561 (pos_encoding_sin_wave); models/loss.py:451-462 (knn), 1351-1396 (dist loss).
// Layout: all activations are point-major [B][N][C] fp32 in HBM (the host transposes the
// reference's (B,C,N) once per block), so neighbour rows are contiguous 256/512-B gathers.
#include "dvm_common.h"
#include <stdlib.h>

namespace dvm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ============================================================== K3: scores + radix select
// s_ij = (-|a_i|^2 - (-2 * a_i.b_j)) - |b_j|^2, the dot product a k-ordered fma chain.
// MFMA orientation: A = queries, B = keys, so a lane owns one key column and a register one
// query row: rows of S are written as contiguous 128-B segments.
constexpr int KS_KT = 64, KS_QB = 128, KS_THREADS = 256;  // 4 waves x 32 query rows per workgroup

template <int D>
__global__ __launch_bounds__(KS_THREADS, 2) void knn_scores_mfma_kernel(const float *__restrict__ a, const float *__restrict__ bq,
                                                                       const float *__restrict__ na, const float *__restrict__ nb,
                                                                       int N, int M, int kchunk, float *__restrict__ S) {
    constexpr int LDK = D + 4, H = D / 2;
    __shared__ __attribute__((aligned(16))) float kt[KS_KT * LDK];
    __shared__ float kn[KS_KT];
    const int b = blockIdx.y, qt = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const int qrow = qt * KS_QB + wave * 32 + r32;
    const int qrc = qrow < N ? qrow : N - 1;
    const float *qp = a + ((size_t)b * N + qrc) * D;
    float q[H];
#pragma unroll
    for (int c = 0; c < D / 4; ++c) {
        f32x4 v = *(const f32x4 *)(qp + 4 * c);
        q[2 * c] = h ? v.y : v.x;
        q[2 * c + 1] = h ? v.w : v.z;
    }
    // norms of the 16 query rows this lane's accumulator registers cover
    float nq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int row = qt * KS_QB + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        nq[r] = na[(size_t)b * N + (row < N ? row : N - 1)];
    }
    const float *kb = bq + (size_t)b * M * D;
    const int jbeg = blockIdx.z * kchunk, jend = jbeg + kchunk < M ? jbeg + kchunk : M;  // this workgroup's key range
    for (int j0 = jbeg; j0 < jend; j0 += KS_KT) {
        __syncthreads();
        for (int e = tid; e < KS_KT * D / 4; e += KS_THREADS) {
            int r = e / (D / 4), c = e % (D / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j0 + r < M) v = *(const f32x4 *)(kb + (size_t)(j0 + r) * D + 4 * c);
            float2 ev = {v.x, v.z}, od = {v.y, v.w};
            *(float2 *)(kt + r * LDK + 2 * c) = ev;
            *(float2 *)(kt + r * LDK + H + 2 * c) = od;
        }
        if (tid < KS_KT) kn[tid] = (j0 + tid < M) ? nb[(size_t)b * M + j0 + tid] : 0.f;
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const float *krow = kt + (sub * 32 + r32) * LDK + h * H;
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // the key fragments of the sub-tile are requested 8 at a time ahead of their matrix instructions (the scheduler
            // otherwise issues each read right before its four MFMAs and the wave waits out the LDS latency every time)
            // (D = 128 only: measured 122 -> 111 us per layer at 8 x 2048; at D = 64 the same change costs 30 %)
            constexpr int FR = D >= 128 ? 8 : 1;
#pragma unroll
            for (int c0 = 0; c0 < H / 4; c0 += FR) {
                f32x4 kv[FR];
#pragma unroll
                for (int c = 0; c < FR; ++c) kv[c] = *(const f32x4 *)(krow + 4 * (c0 + c));
                if (FR > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < FR; ++c) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * (c0 + c)], kv[c].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * (c0 + c) + 1], kv[c].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * (c0 + c) + 2], kv[c].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[4 * (c0 + c) + 3], kv[c].w, acc, 0, 0, 0);
                }
            }
            const int j = j0 + sub * 32 + r32;
            const float nbj = kn[sub * 32 + r32];
            if (j < M) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = qt * KS_QB + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    float inner = -2.f * acc[r];
                    float s = (-nq[r] - inner) - nbj;
                    if (row < N) S[((size_t)b * N + row) * M + j] = s;
                }
            }
        }
    }
}

// The same scores with the key tiles brought by LDS-DMA (round 3; shipped for C = 128, the form above for C = 64):
// two alternating buffers of 64 keys in their natural [key][D] layout, the 16-byte chunks of a row XOR-swizzled with the row
// number on the SOURCE address, the next tile requested before this tile's matrix instructions; a lane reads its two operands
// of a chunk (k = 4 c + h and 4 c + 2 + h) with one ds_read2_b32.  No staging registers, no ds_write, one barrier per tile.
// Same k-ordered chain per (query, key), same bits.
template <int D>
__global__ __launch_bounds__(KS_THREADS, 2) void knn_scores_dma_kernel(const float *__restrict__ a, const float *__restrict__ bq,
                                                                      const float *__restrict__ na, const float *__restrict__ nb,
                                                                      int N, int M, int kchunk, float *__restrict__ S) {
    constexpr int H = D / 2, ROWB = D * 4, TILEB = KS_KT * ROWB, NI = TILEB / 1024 / 4;   // 1-KiB pieces per wave and tile
    constexpr int RPI = 1024 / ROWB, CPR = ROWB / 16;                                     // rows per piece, chunks per row
    extern __shared__ __attribute__((aligned(16))) char ks_lds[];   // [2][64 keys][D] + [2][64] norms
    float *const kn0 = (float *)(ks_lds + 2 * TILEB);
    const int b = blockIdx.y, qt = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const int qrow = qt * KS_QB + wave * 32 + r32;
    const int qrc = qrow < N ? qrow : N - 1;
    const float *qp = a + ((size_t)b * N + qrc) * D;
    float q[H];
#pragma unroll
    for (int c = 0; c < D / 4; ++c) {
        f32x4 v = *(const f32x4 *)(qp + 4 * c);
        q[2 * c] = h ? v.y : v.x;
        q[2 * c + 1] = h ? v.w : v.z;
    }
    float nq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int row = qt * KS_QB + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        nq[r] = na[(size_t)b * N + (row < N ? row : N - 1)];
    }
    const float *kb = bq + (size_t)b * M * D;
    const float *nbb = nb + (size_t)b * M;
    const int jbeg = blockIdx.z * kchunk, jend = jbeg + kchunk < M ? jbeg + kchunk : M;
    // this lane's part of a piece: row lr of the piece, source chunk (position ^ row) — the row number's low bits are the
    // same for every piece (a piece starts on a multiple of RPI rows, RPI | 8 ... only RPI <= 8 rows per piece)
    const int lr = lane / CPR, lc = lane % CPR;
    auto stage = [&](int j0, int buf) {
        char *dst = ks_lds + buf * TILEB + wave * NI * 1024;
#pragma unroll
        for (int e = 0; e < NI; ++e) {
            const int r = (wave * NI + e) * RPI + lr;                 // key row of the tile
            const int j = j0 + r < M ? j0 + r : M - 1;                // (rows past the end: the last key again, never stored)
            const float *src = kb + (size_t)j * D + 4 * (lc ^ (r & 7));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + e * 1024), 16, 0, 0);
        }
        if (wave == 0) {
            const int j = j0 + lane < M ? j0 + lane : M - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nbb + j),
                                             (__attribute__((address_space(3))) void *)(kn0 + buf * KS_KT), 4, 0, 0);
        }
    };
    int roff[8];   // byte offset of chunk c (mod 8) in this lane's key row, first operand
#pragma unroll
    for (int c = 0; c < 8; ++c) roff[c] = ((c ^ (r32 & 7)) << 4) + 4 * h;
    int cur = 0;
    stage(jbeg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j0 = jbeg; j0 < jend; j0 += KS_KT) {
        if (j0 + KS_KT < jend) stage(j0 + KS_KT, cur ^ 1);   // (the other buffer was last read before the previous barrier)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const char *krow = ks_lds + cur * TILEB + (sub * 32 + r32) * ROWB;
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c0 = 0; c0 < D / 4; c0 += 8) {
                float k0[8], k1[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float *p = (const float *)(krow + (c0 / 8) * 128 + roff[c]);
                    k0[c] = p[0], k1[c] = p[2];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[2 * (c0 + c)], k0[c], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[2 * (c0 + c) + 1], k1[c], acc, 0, 0, 0);
                }
            }
            const int j = j0 + sub * 32 + r32;
            const float nbj = kn0[cur * KS_KT + sub * 32 + r32];
            if (j < M) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = qt * KS_QB + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    float inner = -2.f * acc[r];
                    float s = (-nq[r] - inner) - nbj;
                    if (row < N) S[((size_t)b * N + row) * M + j] = s;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
}

// generic-C scalar scores (any C): thread per (query, key tile)
__global__ __launch_bounds__(128) void knn_scores_scalar_kernel(const float *__restrict__ a, const float *__restrict__ bq,
                                                                const float *__restrict__ na, const float *__restrict__ nb, int N,
                                                                int M, int C, float *__restrict__ S) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float *qp = a + ((size_t)b * N + i) * C;
    const float nqi = na[(size_t)b * N + i];
    for (int j = 0; j < M; ++j) {
        const float *kp = bq + ((size_t)b * M + j) * C;
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot = fmaf(qp[c], kp[c], dot);
        float inner = -2.f * dot;
        S[((size_t)b * N + i) * M + j] = (-nqi - inner) - nb[(size_t)b * M + j];
    }
}

// order-preserving map float -> uint (ascending), then inverted: smaller key == larger score
__device__ __forceinline__ unsigned desc_key(float f) {
    unsigned u = __float_as_uint(f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~u;
}

__device__ __forceinline__ int block_exclusive_scan(int v, int *smem /* [8] */, int &total) {
    // 256 threads; returns exclusive prefix of v in thread order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) smem[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += smem[w];
    total = smem[0] + smem[1] + smem[2] + smem[3];
    return base + inc - v;
}

// One workgroup per row: exact k-th key by 4 radix passes, compaction in index order (ties ->
// lowest index), bitonic sort of the k winners by (key, index).
template <int EPT>
__global__ __launch_bounds__(256) void topk_select_kernel(const float *__restrict__ S, int M, int k, int32_t *__restrict__ idx) {
    __shared__ int hist[256];
    __shared__ int sc[8];
    __shared__ int bc[4];
    __shared__ unsigned long long sel[512];
    const size_t row = blockIdx.x;
    const float *s = S + row * M;
    const int tid = threadIdx.x;
    unsigned key[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        int j = tid * EPT + e;
        key[e] = j < M ? desc_key(s[j]) : 0xffffffffu;  // padding sorts last (NaN-free inputs assumed)
    }
    unsigned prefix = 0, mask = 0;
    int remaining = k;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            int j = tid * EPT + e;
            if (j < M && (key[e] & mask) == prefix) atomicAdd(&hist[(key[e] >> shift) & 255], 1);
        }
        __syncthreads();
        int total;
        int mine = hist[tid];
        int excl = block_exclusive_scan(mine, sc, total);
        if (excl < remaining && excl + mine >= remaining) {  // exactly one bin
            bc[0] = tid;
            bc[1] = excl;
        }
        __syncthreads();
        prefix |= (unsigned)bc[0] << shift;
        mask |= 255u << shift;
        remaining -= bc[1];
        __syncthreads();
    }
    // prefix == k-th smallest key; take all keys < prefix and the first `remaining` keys == prefix
    int nlt = 0, neq = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        int j = tid * EPT + e;
        if (j < M) {
            nlt += key[e] < prefix;
            neq += key[e] == prefix;
        }
    }
    int tot_lt, tot_eq;
    int plt = block_exclusive_scan(nlt, sc, tot_lt);
    int peq = block_exclusive_scan(neq, sc, tot_eq);
    int kp2 = 1;
    while (kp2 < k) kp2 <<= 1;
    for (int e = tid; e < kp2; e += 256) sel[e] = ~0ull;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        int j = tid * EPT + e;
        if (j < M) {
            if (key[e] < prefix) {
                sel[plt++] = ((unsigned long long)key[e] << 32) | (unsigned)j;
            } else if (key[e] == prefix) {
                if (peq < remaining) sel[tot_lt + peq] = ((unsigned long long)key[e] << 32) | (unsigned)j;
                peq++;
            }
        }
    }
    __syncthreads();
    for (int size = 2; size <= kp2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < kp2 / 2; t += 256) {
                int lo = (t / stride) * stride * 2 + (t % stride), hi = lo + stride;
                bool up = ((lo & size) == 0);
                unsigned long long x = sel[lo], y = sel[hi];
                if ((x > y) == up) {
                    sel[lo] = y;
                    sel[hi] = x;
                }
            }
            __syncthreads();
        }
    }
    for (int t = tid; t < k; t += 256) idx[row * k + t] = (int32_t)(sel[t] & 0xffffffffu);
}

// Tail shared by the wave top-k kernels: the (key << 32 | column) words compacted into cw[0..base) (rest ~0) are
// bitonic-sorted across the wave, element i = lane + 64 * slot (LDS operations of one wave execute in order), and the
// first k columns written out.
__device__ __forceinline__ void wave_sort_and_store(const unsigned long long *cw, int base, int lane, int k, int32_t *__restrict__ out) {
    unsigned long long a0 = cw[lane], a1 = cw[lane + 64];
    const bool small = base <= 64;  // everything sits in slot 0 then (slot 1 is all ~0)
    for (int size = 2; size <= (small ? 64 : 128); size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride == 64) {  // partner is the other slot of this lane; i = lane (slot 0) is the lower index, ascending
                const unsigned long long lo = a0 < a1 ? a0 : a1, hi = a0 < a1 ? a1 : a0;
                a0 = lo, a1 = hi;
            } else {
                const bool lower = (lane & stride) == 0;
#pragma unroll
                for (int slot = 0; slot < 2; ++slot) {
                    if (slot == 1 && small) break;
                    unsigned long long &x = slot ? a1 : a0;
                    const int i = lane + 64 * slot;
                    const bool up = (i & size) == 0 || size == 128;
                    const unsigned long long y =
                        ((unsigned long long)__shfl_xor((unsigned)(x >> 32), stride, 64) << 32) | __shfl_xor((unsigned)x, stride, 64);
                    const bool keep_min = (lower == up);
                    x = keep_min ? (x < y ? x : y) : (x < y ? y : x);
                }
            }
        }
    }
    if (lane < k) out[lane] = (int32_t)(unsigned)a0;
    if (!small && lane + 64 < k) out[lane + 64] = (int32_t)(unsigned)a1;
}

// k <= 64: one WAVE per row, no workgroup barriers (the block kernel above spends ~50 of them per row).
//   1. every lane keeps its EPL keys (columns e*64 + lane) and their two smallest;
//   2. Tc = the k-th smallest of those 128 lane minima, by a 32-step bit search with ballots: an upper bound of the
//      row's k-th smallest key, and tight (it is exact unless some lane holds three of the row's k best);
//   3. the keys <= Tc (about k of them) are compacted in column order into (key << 32 | column) words and bitonic-
//      sorted across the wave; the first k are the answer, ties already in column order.
//   If more than 128 keys pass (heavy ties) the exact k-th key is searched over all keys instead and only the
//   winners are compacted.
template <int EPL>
__global__ __launch_bounds__(256) void topk_wave_kernel(const float *__restrict__ S, int M, int k, long rows,
                                                        int32_t *__restrict__ idx) {
    __shared__ unsigned long long cand[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float *s = S + row * M;
    unsigned key[EPL];
    unsigned m1 = 0xffffffffu, m2 = 0xffffffffu;
    // all EPL loads are requested before the first is used: the index is clamped instead of the load being predicated (a
    // conditional load is a branch around a load + wait: the row arrived in 32 dependent round trips, 63 us per layer)
    float sv[EPL];
    if (M == EPL * 64) {   // (wave-uniform) full rows: one base register, immediate offsets, EPL loads back to back
#pragma unroll
        for (int e = 0; e < EPL; ++e) sv[e] = s[e * 64 + lane];
    } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int j = e * 64 + lane;
            sv[e] = s[j < M ? j : M - 1];
        }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int j = e * 64 + lane;
        const unsigned x = j < M ? desc_key(sv[e]) : 0xffffffffu;
        key[e] = x;
        m2 = min(m2, max(m1, x));
        m1 = min(m1, x);
    }
    unsigned T = 0;
    for (int bit = 31; bit >= 0; --bit) {  // largest T with count(minima < T) < k  ==  their k-th smallest
        const unsigned c = T | (1u << bit);
        const int cnt = __popcll(__ballot(m1 < c)) + __popcll(__ballot(m2 < c));
        if (cnt < k) T = c;
    }
    unsigned long long *cw = cand[wave];
    cw[lane] = ~0ull;
    cw[lane + 64] = ~0ull;
    int base = 0;
    // Each lane marks its own keys <= Tc (about one per lane); a wave prefix sum gives every lane its slots and the lane
    // writes its few winners itself — the final sort orders by (key, column), so the compaction order is free.  (The
    // previous form ranked all EPL key slots with two ballots + two lane counts each: 12 VALU x 32 slots.)
    static_assert(EPL <= 32, "one 32-bit pass mask per lane");
    unsigned pm = 0;
#pragma unroll
    for (int e = 0; e < EPL; ++e) pm |= key[e] <= T ? (1u << e) : 0u;
    const int mine = __popc(pm);
    int incl = mine;  // inclusive prefix over lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    const int C = __shfl(incl, 63, 64);
    if (C <= 128) {
        int pos = incl - mine;
        const int iters = __reduce_max_sync(~0ull, mine);
        for (int it = 0; it < iters; ++it) {
            if (pm) {
                const int e = __ffs(pm) - 1;
                pm &= pm - 1;
                const int j = e * 64 + lane;
                const unsigned x = j < M ? desc_key(s[j < M ? j : M - 1]) : 0xffffffffu;  // re-read (L1 hit): no dynamic register index
                cw[pos++] = ((unsigned long long)x << 32) | (unsigned)j;
            }
        }
        base = C;
    } else {  // heavy ties: exact k-th key over all keys, then only the winners (ties in column order)
        T = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned c = T | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int e = 0; e < EPL; ++e) cnt += __popcll(__ballot(key[e] < c));
            if (cnt < k) T = c;
        }
        int lt = 0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) lt += __popcll(__ballot(key[e] < T));
        const int ties = k - lt;
        int tbase = 0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const bool eq = key[e] == T;
            const unsigned long long meq = __ballot(eq);
            const int teq = tbase + __popcll(meq & ((1ull << lane) - 1ull));  // this key's rank among the ties, column order
            const bool take = key[e] < T || (eq && teq < ties);
            const unsigned long long mt = __ballot(take);
            if (take) cw[base + __popcll(mt & ((1ull << lane) - 1ull))] = ((unsigned long long)key[e] << 32) | (unsigned)(e * 64 + lane);
            base += __popcll(mt);
            tbase += __popcll(meq);
        }
    }
    wave_sort_and_store(cw, base, lane, k, idx + row * k);
}

// Rows longer than 2048 columns: the same algorithm without the per-lane key array (128 VGPRs at EPL = 128 left two
// waves per SIMD and ran 7x slower per byte than EPL = 32).  The row is streamed three times instead — lane minima,
// then count + optimistic compaction of the keys <= Tc — and stays L2-resident between the passes (20 KB at M = 4995).
__global__ __launch_bounds__(256) void topk_wave_stream_kernel(const float *__restrict__ S, int M, int k, long rows,
                                                               int32_t *__restrict__ idx) {
    __shared__ unsigned long long cand[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float *s = S + row * M;
    const int E = (M + 63) / 64;
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned m1 = 0xffffffffu, m2 = 0xffffffffu;
    for (int e0 = 0; e0 < E; e0 += 8) {   // eight loads in flight (clamped, not predicated: see topk_wave_kernel)
        float sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = (e0 + u) * 64 + lane;
            sv[u] = s[j < M ? j : M - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = (e0 + u) * 64 + lane;
            const unsigned x = j < M ? desc_key(sv[u]) : 0xffffffffu;
            m2 = min(m2, max(m1, x));
            m1 = min(m1, x);
        }
    }
    unsigned T = 0;
    for (int bit = 31; bit >= 0; --bit) {  // largest T with count(minima < T) < k  ==  their k-th smallest
        const unsigned c = T | (1u << bit);
        const int cnt = __popcll(__ballot(m1 < c)) + __popcll(__ballot(m2 < c));
        if (cnt < k) T = c;
    }
    unsigned long long *cw = cand[wave];
    cw[lane] = ~0ull;
    cw[lane + 64] = ~0ull;
    int base = 0;
    for (int e0 = 0; e0 < E; e0 += 8) {  // optimistic: at most 128 keys pass unless the row has heavy ties
        float sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = (e0 + u) * 64 + lane;
            sv[u] = s[j < M ? j : M - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = (e0 + u) * 64 + lane;
            const unsigned x = j < M ? desc_key(sv[u]) : 0xffffffffu;
            const bool take = x <= T && j < M;
            const unsigned long long mt = __ballot(take);
            const int pos = base + __popcll(mt & below);
            if (take && pos < 128) cw[pos] = ((unsigned long long)x << 32) | (unsigned)j;
            base += __popcll(mt);
        }
    }
    if (base > 128) {  // heavy ties: the exact k-th key over all keys, then only the winners (ties in column order)
        T = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned c = T | (1u << bit);
            int cnt = 0;
            for (int e = 0; e < E; ++e) {
                const int j = e * 64 + lane;
                const unsigned x = j < M ? desc_key(s[j < M ? j : M - 1]) : 0xffffffffu;   // (clamped, not predicated: see topk_wave_kernel)
                cnt += __popcll(__ballot(x < c));
            }
            if (cnt < k) T = c;
        }
        int lt = 0;
        for (int e = 0; e < E; ++e) {
            const int j = e * 64 + lane;
            const unsigned x = j < M ? desc_key(s[j < M ? j : M - 1]) : 0xffffffffu;   // (clamped, not predicated: see topk_wave_kernel)
            lt += __popcll(__ballot(x < T));
        }
        const int ties = k - lt;
        cw[lane] = ~0ull;
        cw[lane + 64] = ~0ull;
        base = 0;
        int tbase = 0;
        for (int e = 0; e < E; ++e) {
            const int j = e * 64 + lane;
            const unsigned x = j < M ? desc_key(s[j < M ? j : M - 1]) : 0xffffffffu;   // (clamped, not predicated: see topk_wave_kernel)
            const bool eq = x == T && j < M;
            const unsigned long long meq = __ballot(eq);
            const int teq = tbase + __popcll(meq & below);
            const bool take = (x < T && j < M) || (eq && teq < ties);
            const unsigned long long mt = __ballot(take);
            if (take) cw[base + __popcll(mt & below)] = ((unsigned long long)x << 32) | (unsigned)j;
            base += __popcll(mt);
            tbase += __popcll(meq);
        }
    }
    wave_sort_and_store(cw, base, lane, k, idx + row * k);
}

// ============================================================== positional encoding
__global__ void minmax_partial_kernel(const float *__restrict__ x, long n, float *__restrict__ part) {
    __shared__ float smn[256], smx[256];
    float mn = INFINITY, mx = -INFINITY;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = x[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    smn[threadIdx.x] = mn;
    smx[threadIdx.x] = mx;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + o]);
            smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = smn[0];
        part[2 * blockIdx.x + 1] = smx[0];
    }
}

// out[b, c*128 + j, n] = sin(nc * f_j), out[b, c*128 + 64 + j, n] = cos(nc * f_j);
// nc = 2*((x - mn)/(mx - mn)) - 1, f_j = fl32(pi) * 2^j.   x, out are (B,3,N) / (B,384,N).
__global__ void posenc_kernel(const float *__restrict__ x, const float *__restrict__ part, int nparts, int B, int N,
                              float *__restrict__ out) {
    float mn = INFINITY, mx = -INFINITY;
    for (int q = 0; q < nparts; ++q) {
        mn = fminf(mn, part[2 * q]);
        mx = fmaxf(mx, part[2 * q + 1]);
    }
    const long total = (long)B * 3 * 64 * N;
    const float range = mx - mn;
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        int n = (int)(g % N);
        int j = (int)((g / N) % 64);
        int c = (int)((g / ((long)N * 64)) % 3);
        int b = (int)(g / ((long)N * 64 * 3));
        float v = x[((size_t)b * 3 + c) * N + n];
        float t = __fdiv_rn(v - mn, range);
        float nc = 2.f * t - 1.f;
        float f = __uint_as_float(0x40490fdbu) * __uint_as_float((unsigned)(127 + j) << 23);  // fl32(pi) * 2^j, exact
        float kk = nc * f;
        size_t o = ((size_t)b * 384 + c * 128 + j) * N + n;
        out[o] = sinf(kk);
        out[o + (size_t)64 * N] = cosf(kk);
    }
}

// ============================================================== K5: SA_Layer attention
// p [B][N][16] = Wqk x, v [B][N][64] = Wv x + bv (host GEMMs).  E_ij = p_i . p_j is symmetric.
//   pass 1: row statistics (m_i, l_i) of softmax_j(E_ij)
//   pass 2: for every column j:  x_r[j,:] = sum_i v_i w_ij / (1e-9 + sum_i w_ij),  w_ij = exp(E_ij - m_i)/l_i
constexpr int SA_P = 16, SA_C = 64, SA_LDP = 20;

// Both passes: one workgroup owns 32 query rows / output columns; its 4 waves take the four 32-key tiles of every
// staged 128-key block and their partial results are merged through LDS at the end, so that B*N/32 workgroups
// (512 at B = 8, N = 2048) keep all 256 CUs busy where 128-row workgroups left half of them idle.
constexpr int SA_KB = 128;  // keys staged per iteration

// grid.z = key chunks of `kchunk` keys (multiple of SA_KB).  With one chunk the kernels write the final results; with
// more (small B * N: B = 1, N = 4995 has only 157 column groups) they write per-chunk partials that two tiny merge
// kernels combine.
__global__ __launch_bounds__(256) void sa_rowstats_kernel(const float *__restrict__ p, int N, int kchunk, float *__restrict__ stats) {
    __shared__ __attribute__((aligned(16))) float pt[SA_KB * SA_LDP];
    __shared__ float red[4][32][2];
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const int irow = blockIdx.x * 32 + r32;
    const int irc = irow < N ? irow : N - 1;
    const float *pb = p + (size_t)b * N * SA_P;
    float pi[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        f32x4 v = *(const f32x4 *)(pb + (size_t)irc * SA_P + 4 * c);
        pi[2 * c] = h ? v.y : v.x;
        pi[2 * c + 1] = h ? v.w : v.z;
    }
    float m = -INFINITY, l = 0.f;
    const int zbeg = blockIdx.z * kchunk, zend = zbeg + kchunk < N ? zbeg + kchunk : N;
    const bool partial = gridDim.z > 1;
    // the next 128-key block travels in registers while the current one is consumed
    f32x4 pre[2];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int e = tid + q * 256, r = e >> 2, c = e & 3;  // 128 rows x 4 float4
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j0 + r < N) v = *(const f32x4 *)(pb + (size_t)(j0 + r) * SA_P + 4 * c);
            pre[q] = v;
        }
    };
    fetch(zbeg);
    for (int j0 = zbeg; j0 < zend; j0 += SA_KB) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int e = tid + q * 256, r = e >> 2, c = e & 3;
            float2 ev = {pre[q].x, pre[q].z}, od = {pre[q].y, pre[q].w};
            *(float2 *)(pt + r * SA_LDP + 2 * c) = ev;
            *(float2 *)(pt + r * SA_LDP + 8 + 2 * c) = od;
        }
        __syncthreads();
        if (j0 + SA_KB < zend) fetch(j0 + SA_KB);
        const int jt = j0 + wave * 32;
        if (jt >= N) continue;
        const float *jr = pt + (wave * 32 + r32) * SA_LDP + h * 8;
        f32x4 a0 = *(const f32x4 *)(jr), a1 = *(const f32x4 *)(jr + 4);
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, pi[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, pi[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, pi[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, pi[3], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, pi[4], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, pi[5], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, pi[6], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, pi[7], acc, 0, 0, 0);
        // this lane: row i = irow, 16 columns j = jt + (r&3) + 8*(r>>2) + 4*h
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int j = jt + (r & 3) + 8 * (r >> 2) + 4 * h;
            float e = j < N ? acc[r] : -INFINITY;
            acc[r] = e;
            tmax = fmaxf(tmax, e);
        }
        if (tmax > m) {
            l = l * __expf(m - tmax);  // m = -inf, l = 0 -> 0 * 0
            m = tmax;
        }
        if (tmax != -INFINITY) {
#pragma unroll
            for (int r = 0; r < 16; ++r) l += __expf(acc[r] - m);
        }
    }
    float mo = __shfl_xor(m, 32, 64), lo = __shfl_xor(l, 32, 64);
    float mm = fmaxf(m, mo);
    float ll = (m == -INFINITY ? 0.f : l * __expf(m - mm)) + (mo == -INFINITY ? 0.f : lo * __expf(mo - mm));
    if (h == 0) {
        red[wave][r32][0] = mm;
        red[wave][r32][1] = ll;
    }
    __syncthreads();
    if (tid < 32 && irow < N) {
        float gm = fmaxf(fmaxf(red[0][tid][0], red[1][tid][0]), fmaxf(red[2][tid][0], red[3][tid][0]));
        float gl = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            float wm = red[w][tid][0];
            if (wm != -INFINITY) gl += red[w][tid][1] * __expf(wm - gm);
        }
        const size_t o = (((size_t)blockIdx.z * gridDim.y + b) * N + irow) * 2;  // chunk-major partials
        stats[o] = gm;
        stats[o + 1] = partial ? gl : 1.0f / gl;
    }
}

// (m, l) of S key chunks -> (m, 1 / l)
__global__ void sa_stats_merge_kernel(const float *__restrict__ part, long rows, int S, float *__restrict__ stats) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float gm = -INFINITY;
    for (int z = 0; z < S; ++z) gm = fmaxf(gm, part[((size_t)z * rows + r) * 2]);
    float gl = 0.f;
    for (int z = 0; z < S; ++z) {
        const float wm = part[((size_t)z * rows + r) * 2];
        if (wm != -INFINITY) gl += part[((size_t)z * rows + r) * 2 + 1] * __expf(wm - gm);
    }
    stats[r * 2] = gm;
    stats[r * 2 + 1] = 1.0f / gl;
}

// sum of S partial (unnormalised x_r, column sum) -> x_r, 1 / (1e-9 + column sum)
__global__ void sa_apply_merge_kernel(const float *__restrict__ po, const float *__restrict__ pc, long rows, int S,
                                      float *__restrict__ xr, float *__restrict__ cinv_out) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= rows * SA_C) return;
    const long r = g / SA_C;
    float cs = 0.f, acc = 0.f;
    for (int z = 0; z < S; ++z) {
        cs += pc[(size_t)z * rows + r];
        acc += po[(size_t)z * rows * SA_C + g];
    }
    const float inv = 1.0f / (1e-9f + cs);
    xr[g] = acc * inv;
    if (cinv_out && (g % SA_C) == 0) cinv_out[r] = inv;
}

__global__ __launch_bounds__(256) void sa_apply_kernel(const float *__restrict__ p, const float *__restrict__ v,
                                                       const float *__restrict__ stats, int N, int kchunk, float *__restrict__ xr,
                                                       float *__restrict__ cinv_out /* partial mode: column sums */) {
    __shared__ __attribute__((aligned(16))) float pt[SA_KB * SA_LDP];
    __shared__ __attribute__((aligned(16))) float vt[SA_KB * SA_C];  // reused as the [wave][reg][lane] merge buffer
    __shared__ float st[SA_KB * 2];
    __shared__ float csum[4][32];
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const int jcol = blockIdx.x * 32 + r32;
    const int jc = jcol < N ? jcol : N - 1;
    const float *pb = p + (size_t)b * N * SA_P;
    const float *vb = v + (size_t)b * N * SA_C;
    float pj[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        f32x4 q = *(const f32x4 *)(pb + (size_t)jc * SA_P + 4 * c);
        pj[2 * c] = h ? q.y : q.x;
        pj[2 * c + 1] = h ? q.w : q.z;
    }
    f32x16 o0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, o1 = o0;
    float colsum = 0.f;
    const int zbeg = blockIdx.z * kchunk, zend = zbeg + kchunk < N ? zbeg + kchunk : N;
    const bool partial = gridDim.z > 1;
    // the next 128-key block (p, v, stats) travels in registers while the current one is consumed
    f32x4 prep[2], prev[8];
    float pres;
    auto fetch = [&](int i0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int e = tid + q * 256, r = e >> 2, c = e & 3;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (i0 + r < N) t = *(const f32x4 *)(pb + (size_t)(i0 + r) * SA_P + 4 * c);
            prep[q] = t;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int e = tid + q * 256, r = e >> 4, c = e & 15;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (i0 + r < N) t = *(const f32x4 *)(vb + (size_t)(i0 + r) * SA_C + 4 * c);
            prev[q] = t;
        }
        pres = (i0 + (tid >> 1) < N) ? stats[((size_t)b * N + i0 + (tid >> 1)) * 2 + (tid & 1)] : 0.f;  // invl = 0 kills padding
    };
    fetch(zbeg);
    for (int i0 = zbeg; i0 < zend; i0 += SA_KB) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int e = tid + q * 256, r = e >> 2, c = e & 3;
            float2 ev = {prep[q].x, prep[q].z}, od = {prep[q].y, prep[q].w};
            *(float2 *)(pt + r * SA_LDP + 2 * c) = ev;
            *(float2 *)(pt + r * SA_LDP + 8 + 2 * c) = od;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int e = tid + q * 256, r = e >> 4, c = e & 15;
            *(f32x4 *)(vt + r * SA_C + 4 * c) = prev[q];
        }
        st[tid] = pres;
        __syncthreads();
        if (i0 + SA_KB < zend) fetch(i0 + SA_KB);
        if (i0 + wave * 32 >= N) continue;
        // E tile: rows = this wave's 32 keys i (A operand from LDS), cols = this lane's column j
        const float *ir = pt + (wave * 32 + r32) * SA_LDP + h * 8;
        f32x4 a0 = *(const f32x4 *)(ir), a1 = *(const f32x4 *)(ir + 4);
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, pj[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, pj[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, pj[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, pj[3], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, pj[4], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, pj[5], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, pj[6], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, pj[7], acc, 0, 0, 0);
        // w_r = exp(E - m_i) / l_i for the 16 key rows of this lane; then x_r += V^T w on the matrix cores:
        // MFMA step r pairs key rho_r (lanes h=0) with key rho_r+4 (lanes h=1) in both operands.
        const float *stw = st + wave * 64, *vtw = vt + wave * 32 * SA_C;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int key = (r & 3) + 8 * (r >> 2) + 4 * h;
            float w = __expf(acc[r] - stw[key * 2]) * stw[key * 2 + 1];
            colsum += w;
            float va = vtw[key * SA_C + r32], vb2 = vtw[key * SA_C + 32 + r32];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va, w, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vb2, w, o1, 0, 0, 0);
        }
    }
    colsum += __shfl_xor(colsum, 32, 64);
    __syncthreads();  // every wave is done with vt
    if (h == 0) csum[wave][r32] = colsum;
    float *red = vt + wave * (32 * 64);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        red[r * 64 + lane] = o0[r];
        red[(16 + r) * 64 + lane] = o1[r];
    }
    __syncthreads();
    const float cs = (csum[0][r32] + csum[1][r32]) + (csum[2][r32] + csum[3][r32]);
    const float inv = partial ? 1.0f : 1.0f / (1e-9f + cs);
    const size_t orow = ((size_t)blockIdx.z * gridDim.y + b) * N + jcol;  // chunk-major partials (chunk 0 = the output itself)
    if (cinv_out && wave == 0 && h == 0 && jcol < N) cinv_out[orow] = partial ? cs : inv;
    if (jcol < N) {
        float *o = xr + orow * SA_C;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = wave * 8 + q;  // merged register index: 0..15 -> o0, 16..31 -> o1
            const float *src = vt + r * 64 + lane;
            float sum = (src[0] + src[32 * 64]) + (src[2 * 32 * 64] + src[3 * 32 * 64]);
            const int rr = r & 15;
            o[(r >> 4) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h] = sum * inv;
        }
    }
}

// ============================================================== K4: N2P attention
// q, kp, vp [B][N][C] (= Wq x, Wk x, Wv x), idx [B][N][K].  One wave per point:
//   e_hj = q_h . (kp_j - kp_i)_h / sqrt(D),  a = softmax_j,  out_h = sum_j a_hj (vp_j - vp_i)_h
template <int C>
__global__ __launch_bounds__(256) void n2p_attention_kernel(const float *__restrict__ q, const float *__restrict__ kp,
                                                            const float *__restrict__ vp, const int32_t *__restrict__ idx,
                                                            int N, int K, int heads, float *__restrict__ out) {
    constexpr int CPL = C / 64;  // channels per lane
    const int lane = threadIdx.x & 63;
    const long pt = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (pt >= N) return;
    const size_t base = (size_t)b * N;
    const int D = C / heads;                 // 16 or 32
    const int lanes_per_head = D / CPL;      // 16
    const float scale = sqrtf((float)D);
    float qv[CPL], ki[CPL], vi[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        qv[c] = q[(base + pt) * C + lane * CPL + c];
        ki[c] = kp[(base + pt) * C + lane * CPL + c];
        vi[c] = vp[(base + pt) * C + lane * CPL + c];
    }
    const int32_t *nb = idx + (base + pt) * K;
    float m = -INFINITY, l = 0.f, acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
    for (int j = 0; j < K; ++j) {
        const size_t nrow = (base + nb[j]) * C + lane * CPL;
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) part = fmaf(qv[c], kp[nrow + c] - ki[c], part);
        for (int o = 1; o < lanes_per_head; o <<= 1) part += __shfl_xor(part, o, 64);
        const float e = part / scale;
        const float mn = fmaxf(m, e);
        const float sc = __expf(m - mn);  // first neighbour: exp(-inf) = 0
        const float w = __expf(e - mn);
        l = l * sc + w;
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c] = fmaf(w, vp[nrow + c] - vi[c], acc[c] * sc);
        m = mn;
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int c = 0; c < CPL; ++c) out[(base + pt) * C + lane * CPL + c] = acc[c] * inv;
}

// ============================================================== K10: dist loss term
// per (b, anchor n): x_j = |feat[idx_j] - feat[a_n]|_2, y_j = dist[b, idx_j, a_n], j < k;
// term = 1 - |cos(x, y)|;  out[b] = sum_n term.   One wave per (b, n).
__global__ __launch_bounds__(256) void dist_loss_generic_kernel(const float *__restrict__ feat, const float *__restrict__ dist,
                                                        const int32_t *__restrict__ anchors, const int32_t *__restrict__ idx,
                                                        int N, int C, int nA, int k, double *__restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * (blockDim.x >> 6) + wave;
    const int b = blockIdx.y;
    __shared__ double red[4];
    double term = 0.0;
    if (n < nA) {
        const int a = anchors[n];
        const float *fa = feat + ((size_t)b * N + a) * C;
        const int32_t *ix = idx + ((size_t)b * nA + n) * k;
        float sxy = 0.f, sxx = 0.f, syy = 0.f;
        for (int j = lane; j < k; j += 64) {
            const int v = ix[j];
            const float *fv = feat + ((size_t)b * N + v) * C;
            float s2 = 0.f;
            for (int c = 0; c < C; c += 4) {
                f32x4 p = *(const f32x4 *)(fv + c), q = *(const f32x4 *)(fa + c);
                float d0 = p.x - q.x, d1 = p.y - q.y, d2 = p.z - q.z, d3 = p.w - q.w;
                s2 = fmaf(d0, d0, s2);
                s2 = fmaf(d1, d1, s2);
                s2 = fmaf(d2, d2, s2);
                s2 = fmaf(d3, d3, s2);
            }
            float x = sqrt_rn(s2);
            float y = dist[((size_t)b * N + v) * N + a];
            sxy = fmaf(x, y, sxy);
            sxx = fmaf(x, x, sxx);
            syy = fmaf(y, y, syy);
        }
        sxy = wave_sum(sxy);
        sxx = wave_sum(sxx);
        syy = wave_sum(syy);
        float nx = fmaxf(sqrt_rn(sxx), 1e-8f), ny = fmaxf(sqrt_rn(syy), 1e-8f);
        float cosv = sxy / (nx * ny);
        term = 1.0 - (double)fabsf(cosv);
    }
    if (lane == 0) red[wave] = term;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Backward of the dist-loss term, first half: the (anchor, point) weights
//     W[b, n, idx_j] = g[b] * d term_n / d x_j / x_j          (0 where x_j = 0; idx rows are distinct points)
// with d term / d x_j = -sgn(cos) (y_j / (|x||y|) - cos x_j / |x|^2).  The caller finishes with plain GEMMs:
//     d feat      = diag(colsum W) feat - W^T feat[anchors]
//     d feat[a_n] += rowsum(W)_n feat[a_n] - (W feat)_n
// (k = 500 neighbours x 128 channels per anchor as atomics would be 0.5 G atomics per shape batch).
// W [B][nA][N] must be zero-filled by the caller.  One wave per (b, n), k <= 512.
__global__ __launch_bounds__(256) void dist_loss_bwd_weights_generic_kernel(const float *__restrict__ feat, const float *__restrict__ dist,
                                                                    const int32_t *__restrict__ anchors,
                                                                    const int32_t *__restrict__ idx, const float *__restrict__ gterm,
                                                                    int N, int C, int nA, int k, float *__restrict__ W) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * (blockDim.x >> 6) + wave;
    const int b = blockIdx.y;
    if (n >= nA) return;
    const int a = anchors[n];
    const float *fa = feat + ((size_t)b * N + a) * C;
    const int32_t *ix = idx + ((size_t)b * nA + n) * k;
    float xs[8], ys[8];
    float sxy = 0.f, sxx = 0.f, syy = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int j = lane + 64 * u;
        xs[u] = ys[u] = 0.f;
        if (j < k) {
            const int v = ix[j];
            const float *fv = feat + ((size_t)b * N + v) * C;
            float s2 = 0.f;
            for (int c = 0; c < C; c += 4) {
                f32x4 p = *(const f32x4 *)(fv + c), q = *(const f32x4 *)(fa + c);
                float d0 = p.x - q.x, d1 = p.y - q.y, d2 = p.z - q.z, d3 = p.w - q.w;
                s2 = fmaf(d0, d0, s2);
                s2 = fmaf(d1, d1, s2);
                s2 = fmaf(d2, d2, s2);
                s2 = fmaf(d3, d3, s2);
            }
            xs[u] = sqrt_rn(s2);
            ys[u] = dist[((size_t)b * N + v) * N + a];
            sxy = fmaf(xs[u], ys[u], sxy);
            sxx = fmaf(xs[u], xs[u], sxx);
            syy = fmaf(ys[u], ys[u], syy);
        }
    }
    sxy = wave_sum(sxy);
    sxx = wave_sum(sxx);
    syy = wave_sum(syy);
    const float nx = fmaxf(sqrt_rn(sxx), 1e-8f), ny = fmaxf(sqrt_rn(syy), 1e-8f);
    const float cosv = sxy / (nx * ny);
    const float sg = cosv > 0.f ? -1.f : (cosv < 0.f ? 1.f : 0.f);  // d(1 - |cos|)/d cos
    const float g = gterm[b] * sg;
    const float inv = 1.f / (nx * ny), cx = cosv / (nx * nx);
    float *Wr = W + ((size_t)b * nA + n) * N;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int j = lane + 64 * u;
        if (j < k && xs[u] > 0.f) Wr[ix[j]] = g * (ys[u] * inv - cx * xs[u]) / xs[u];
    }
}

// C = 128 (LG-Net's feature width), k <= 512: the rows above are read by 64 lanes streaming 64 DIFFERENT rows, 16 bytes
// each per instruction — 64 cache lines per load, every line fetched 8 times (0.55 ms per call at 8 x 1000 anchors x 500
// neighbours).  Here a 32-lane half reads ONE whole row per instruction (512 contiguous bytes), the half's partial sums
// are combined on the DPP network, and lane j % 64 keeps x_j, so that what follows is the arithmetic of the kernels above.
__device__ __forceinline__ void dist_rows128(const float *__restrict__ featb, const float *__restrict__ distb, int N, int a,
                                             const int32_t *__restrict__ ix, int k, int lane, float (&xs)[8], float (&ys)[8]) {
    const int half = lane >> 5, l = lane & 31;
    const f32x4 q = *(const f32x4 *)(featb + (size_t)a * 128 + 4 * l);
    int ixl[8];   // this lane's neighbour indices: j = lane + 64 u (clamped, not predicated: the loads go out together)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int j = lane + 64 * u;
        ixl[u] = ix[j < k ? j : k - 1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {   // the geodesic column entries: scattered 4-byte reads, all requested up front
        const int j = lane + 64 * u;
        const float y = distb[(size_t)ixl[u] * N + a];
        xs[u] = 0.f;
        ys[u] = j < k ? y : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (64 * u >= k) break;
#pragma unroll 4
        for (int jj = 0; jj < 64; jj += 2) {
            if (64 * u + jj >= k) break;
            const int j = 64 * u + jj + half;
            // neighbour j's index from the lane that holds it (no load in the loop: the row gathers of an unrolled group are
            // independent and go out together)
            const int v0 = __builtin_amdgcn_readlane(ixl[u], jj), v1 = __builtin_amdgcn_readlane(ixl[u], jj + 1);
            const int v = half ? v1 : v0;
            const f32x4 p = *(const f32x4 *)(featb + (size_t)v * 128 + 4 * l);
            const float d0 = p.x - q.x, d1 = p.y - q.y, d2 = p.z - q.z, d3 = p.w - q.w;
            float s2 = d0 * d0;
            s2 = fmaf(d1, d1, s2);
            s2 = fmaf(d2, d2, s2);
            s2 = fmaf(d3, d3, s2);
            s2 = sum32(s2);
            const float x = j < k ? sqrt_rn(s2) : 0.f;
            const float o = lane_xor32(x);
            if (lane == jj) xs[u] = half ? o : x;
            if (lane == jj + 1) xs[u] = half ? x : o;
        }
    }
}

__global__ __launch_bounds__(256) void dist_loss_kernel(const float *__restrict__ feat, const float *__restrict__ dist,
                                                        const int32_t *__restrict__ anchors, const int32_t *__restrict__ idx,
                                                        int N, int nA, int k, double *__restrict__ partial,
                                                        float *__restrict__ xsave /* nullptr, or [B][nA][k] x 2: x_j, y_j kept for the backward */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * (blockDim.x >> 6) + wave;
    const int b = blockIdx.y;
    __shared__ double red[4];
    double term = 0.0;
    if (n < nA) {
        float xs[8], ys[8];
        dist_rows128(feat + (size_t)b * N * 128, dist + (size_t)b * N * N, N, anchors[n], idx + ((size_t)b * nA + n) * k, k, lane, xs, ys);
        if (xsave) {
            float *xr = xsave + ((size_t)b * nA + n) * k * 2;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (lane + 64 * u < k) xr[lane + 64 * u] = xs[u], xr[k + lane + 64 * u] = ys[u];
        }
        float sxy = 0.f, sxx = 0.f, syy = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            sxy = fmaf(xs[u], ys[u], sxy);
            sxx = fmaf(xs[u], xs[u], sxx);
            syy = fmaf(ys[u], ys[u], syy);
        }
        sxy = wave_sum(sxy);
        sxx = wave_sum(sxx);
        syy = wave_sum(syy);
        const float nx = fmaxf(sqrt_rn(sxx), 1e-8f), ny = fmaxf(sqrt_rn(syy), 1e-8f);
        term = 1.0 - (double)fabsf(sxy / (nx * ny));
    }
    if (lane == 0) red[wave] = term;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void dist_loss_bwd_weights_kernel(const float *__restrict__ feat, const float *__restrict__ dist,
                                                                    const int32_t *__restrict__ anchors,
                                                                    const int32_t *__restrict__ idx, const float *__restrict__ gterm,
                                                                    int N, int nA, int k, float *__restrict__ W) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * (blockDim.x >> 6) + wave;
    const int b = blockIdx.y;
    if (n >= nA) return;
    const int32_t *ix = idx + ((size_t)b * nA + n) * k;
    float xs[8], ys[8];
    dist_rows128(feat + (size_t)b * N * 128, dist + (size_t)b * N * N, N, anchors[n], ix, k, lane, xs, ys);
    float sxy = 0.f, sxx = 0.f, syy = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        sxy = fmaf(xs[u], ys[u], sxy);
        sxx = fmaf(xs[u], xs[u], sxx);
        syy = fmaf(ys[u], ys[u], syy);
    }
    sxy = wave_sum(sxy);
    sxx = wave_sum(sxx);
    syy = wave_sum(syy);
    const float nx = fmaxf(sqrt_rn(sxx), 1e-8f), ny = fmaxf(sqrt_rn(syy), 1e-8f);
    const float cosv = sxy / (nx * ny);
    const float sg = cosv > 0.f ? -1.f : (cosv < 0.f ? 1.f : 0.f);  // d(1 - |cos|)/d cos
    const float g = gterm[b] * sg;
    const float inv = 1.f / (nx * ny), cx = cosv / (nx * nx);
    float *Wr = W + ((size_t)b * nA + n) * N;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int j = lane + 64 * u;
        if (j < k && xs[u] > 0.f) Wr[ix[j]] = g * (ys[u] * inv - cx * xs[u]) / xs[u];
    }
}

// the same weights from the x_j, y_j the forward kept (dist_loss_kernel's xsave): no second pass over the feature rows; also the row
// sums rs [B][nA] = sum_v W[b, n, v] (in lane order + a wave reduction).  gterm read at gterm[b * gstride].
__global__ __launch_bounds__(256) void dist_loss_bwd_weights_saved_kernel(const float *__restrict__ xsave, const int32_t *__restrict__ idx,
                                                                          const float *__restrict__ gterm, int gstride, int N, int nA, int k,
                                                                          float *__restrict__ W, float *__restrict__ rs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * (blockDim.x >> 6) + wave;
    const int b = blockIdx.y;
    if (n >= nA) return;
    const int32_t *ix = idx + ((size_t)b * nA + n) * k;
    const float *xr = xsave + ((size_t)b * nA + n) * k * 2;
    float xs[8], ys[8];
    float sxy = 0.f, sxx = 0.f, syy = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int j = lane + 64 * u;
        xs[u] = j < k ? xr[j] : 0.f, ys[u] = j < k ? xr[k + j] : 0.f;
        sxy = fmaf(xs[u], ys[u], sxy);
        sxx = fmaf(xs[u], xs[u], sxx);
        syy = fmaf(ys[u], ys[u], syy);
    }
    sxy = wave_sum(sxy);
    sxx = wave_sum(sxx);
    syy = wave_sum(syy);
    const float nx = fmaxf(sqrt_rn(sxx), 1e-8f), ny = fmaxf(sqrt_rn(syy), 1e-8f);
    const float cosv = sxy / (nx * ny);
    const float sg = cosv > 0.f ? -1.f : (cosv < 0.f ? 1.f : 0.f);
    const float g = gterm[(size_t)b * gstride] * sg;
    const float inv = 1.f / (nx * ny), cx = cosv / (nx * nx);
    float *Wr = W + ((size_t)b * nA + n) * N;
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int j = lane + 64 * u;
        if (j < k && xs[u] > 0.f) {
            const float wv = g * (ys[u] * inv - cx * xs[u]) / xs[u];
            Wr[ix[j]] = wv;
            acc += wv;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) rs[(size_t)b * nA + n] = acc;
}

__global__ void gather_rows_kernel(const float *__restrict__ src, const int32_t *__restrict__ rows, int N, int C, int nR,
                                   float *__restrict__ dst) {
    const int b = blockIdx.y;
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)nR * C) return;
    int r = (int)(g / C), c = (int)(g % C);
    dst[((size_t)b * nR + r) * C + c] = src[((size_t)b * N + rows[r]) * C + c];
}

void launch_rownorm2(const float *x, int rows, int K, float *out, hipStream_t s);  // dvm_softcorr.hip

int launch_reduce_partials(const double *partial, int B, int nparts, float scale, float *out, int stride, int off, hipStream_t s);

int launch_knn_neg(const float *a, const float *bq, int B, int N, int M, int C, int k, int32_t *idx, float *na, float *nb,
                   float *S, hipStream_t s) {
    launch_rownorm2(a, B * N, C, na, s);
    if (a == bq && N == M) nb = na;   // self-kNN (every N2P layer): one set of norms
    else launch_rownorm2(bq, B * M, C, nb, s);
    // key range split over grid.z until >= 1024 workgroups are in flight (B * N/128 alone is 128 at B = 8, N = 2048)
    const int qtiles = (N + KS_QB - 1) / KS_QB, ktiles = (M + KS_KT - 1) / KS_KT;
    int nz = 1;
    while (B * qtiles * nz < 1024 && nz * 2 <= ktiles) nz *= 2;
    const int kchunk = ((ktiles + nz - 1) / nz) * KS_KT;
    nz = (M + kchunk - 1) / kchunk;
    // measured at 8 x 2048 / 1 x 4995 points, scores + top-k per call: C = 128: 187 -> 171 us / 184 -> 163 us with the DMA form;
    // C = 64: 130 -> 137 us (half the bytes per key: the register path already hides them) — so C = 128 only
    if (C == 128) {
        constexpr int lds = 2 * KS_KT * 128 * 4 + 2 * KS_KT * 4;
        ensure_dyn_lds((const void *)knn_scores_dma_kernel<128>, lds);
        hipLaunchKernelGGL(knn_scores_dma_kernel<128>, dim3(qtiles, B, nz), dim3(KS_THREADS), lds, s, a, bq, na, nb, N, M, kchunk, S);
    } else if (C == 64)
        hipLaunchKernelGGL(knn_scores_mfma_kernel<64>, dim3(qtiles, B, nz), dim3(KS_THREADS), 0, s, a, bq, na, nb, N, M, kchunk, S);
    else
        hipLaunchKernelGGL(knn_scores_scalar_kernel, dim3((N + 127) / 128, B), dim3(128), 0, s, a, bq, na, nb, N, M, C, S);
    if (k <= 64 && M <= 8192) {
        const long rows = (long)B * N;
        const dim3 wgrid((unsigned)((rows + 3) / 4));
        if (M <= 2048)
            hipLaunchKernelGGL(topk_wave_kernel<32>, wgrid, dim3(256), 0, s, S, M, k, rows, idx);
        else
            hipLaunchKernelGGL(topk_wave_stream_kernel, wgrid, dim3(256), 0, s, S, M, k, rows, idx);
        return DVM_OK;
    }
    int ept = (M + 255) / 256;
    dim3 grid((unsigned)((size_t)B * N));
    if (ept <= 8)
        hipLaunchKernelGGL(topk_select_kernel<8>, grid, dim3(256), 0, s, S, M, k, idx);
    else if (ept <= 16)
        hipLaunchKernelGGL(topk_select_kernel<16>, grid, dim3(256), 0, s, S, M, k, idx);
    else
        hipLaunchKernelGGL(topk_select_kernel<32>, grid, dim3(256), 0, s, S, M, k, idx);
    return DVM_OK;
}

}  // namespace dvm

using namespace dvm;

DVM_EXPORT size_t dvm_knn_neg_workspace_bytes(int B, int N, int M, int C, int k) {
    return align_up((size_t)B * N * sizeof(float)) + align_up((size_t)B * M * sizeof(float)) + align_up((size_t)B * N * M * sizeof(float));
}

DVM_EXPORT int dvm_knn_neg_f32(const float *a, const float *b, int B, int N, int M, int C, int k, int32_t *idx, void *ws,
                               size_t ws_bytes, void *stream) {
    DVM_REQUIRE(a && b && idx, "dvm_knn_neg_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1 && M >= 1 && C >= 1, "dvm_knn_neg_f32: empty input (B=%d N=%d M=%d C=%d)", B, N, M, C);
    DVM_REQUIRE(k >= 1 && k <= 512 && k <= M, "dvm_knn_neg_f32: k=%d unsupported (1..min(512,M=%d))", k, M);
    DVM_REQUIRE(M <= 8192, "dvm_knn_neg_f32: M=%d exceeds 8192", M);
    Arena ar(ws, ws_bytes);
    float *na = ar.take<float>((size_t)B * N);
    float *nb = ar.take<float>((size_t)B * M);
    float *S = ar.take<float>((size_t)B * N * M);
    if (!ar.ok()) {
        set_error("dvm_knn_neg_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    launch_knn_neg(a, b, B, N, M, C, k, idx, na, nb, S, (hipStream_t)stream);
    DVM_CHECK_LAUNCH("knn_neg");
    return DVM_OK;
}

DVM_EXPORT size_t dvm_pos_encoding_workspace_bytes(void) { return align_up(2 * 256 * sizeof(float)); }

DVM_EXPORT int dvm_pos_encoding_f32(const float *x, int B, int N, float *out, void *ws, size_t ws_bytes, void *stream) {
    DVM_REQUIRE(x && out && B >= 1 && N >= 1, "dvm_pos_encoding_f32: bad arguments");
    Arena ar(ws, ws_bytes);
    float *part = ar.take<float>(2 * 256);
    if (!ar.ok()) {
        set_error("dvm_pos_encoding_f32: workspace too small");
        return DVM_ENOSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    long n = (long)B * 3 * N;
    int nparts = (int)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(nparts), dim3(256), 0, s, x, n, part);
    long total = (long)B * 3 * 64 * N;
    int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(posenc_kernel, dim3(blocks), dim3(256), 0, s, x, part, nparts, B, N, out);
    DVM_CHECK_LAUNCH("pos_encoding");
    return DVM_OK;
}

namespace dvm {
__global__ void minmax_final_kernel(const float *__restrict__ part, int nparts, float *__restrict__ out2) {
    float mn = INFINITY, mx = -INFINITY;
    for (int q = 0; q < nparts; ++q) mn = fminf(mn, part[2 * q]), mx = fmaxf(mx, part[2 * q + 1]);
    out2[0] = mn, out2[1] = mx;
}
}  // namespace dvm
// The same encoding with the range taken over ALL ranks' tensors (models/model.py:548 normalises with the min / max of the whole
// batch: under data parallelism that is the global batch): this rank's min / max go to minmax2 (two floats of caller-owned device
// memory), the caller's collective combines them (MIN on the first, MAX on the second, enqueued on `stream`), then the encoding.
DVM_EXPORT int dvm_pos_encoding_sync_f32(const float *x, int B, int N, float *out, void *ws, size_t ws_bytes, const dvm_collective *coll,
                                         float *minmax2, void *stream) {
    DVM_REQUIRE(x && out && B >= 1 && N >= 1 && coll && coll->allreduce && minmax2, "dvm_pos_encoding_sync_f32: bad arguments");
    Arena ar(ws, ws_bytes);
    float *part = ar.take<float>(2 * 256);
    if (!ar.ok()) {
        set_error("dvm_pos_encoding_sync_f32: workspace too small");
        return DVM_ENOSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    long n = (long)B * 3 * N;
    int nparts = (int)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(nparts), dim3(256), 0, s, x, n, part);
    hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(1), 0, s, part, nparts, minmax2);
    int rc = coll->allreduce(coll->user, minmax2, 1, 0, 1, stream);
    if (rc == 0) rc = coll->allreduce(coll->user, minmax2 + 1, 1, 0, 2, stream);
    DVM_REQUIRE(rc == 0, "dvm_pos_encoding_sync_f32: the caller's all-reduce failed (%d)", rc);
    return dvm_pos_encoding_minmax_f32(x, minmax2, B, N, out, stream);
}

DVM_EXPORT int dvm_pos_encoding_minmax_f32(const float *x, const float *minmax, int B, int N, float *out, void *stream) {
    DVM_REQUIRE(x && minmax && out && B >= 1 && N >= 1, "dvm_pos_encoding_minmax_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    long total = (long)B * 3 * 64 * N;
    int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(posenc_kernel, dim3(blocks), dim3(256), 0, s, x, minmax, 1, B, N, out);
    DVM_CHECK_LAUNCH("pos_encoding_minmax");
    return DVM_OK;
}

namespace dvm {
// key chunks per column group: enough workgroups to fill the chip (B * N/32 alone: 157 at B = 1, N = 4995)
static int sa_splits(int B, int N) {
    const int groups = B * ((N + 31) / 32);
    if (groups >= 400) return 1;  // measured: at 512 groups (B = 8, N = 2048) the merge passes cost more than the split gains
    int S = 1;
    while (groups * S < 1024 && S < 8 && (N + 2 * S - 1) / (2 * S) >= 4 * SA_KB) S *= 2;
    return S;
}
static size_t sa_partial_floats(int B, int N) {
    const int S = sa_splits(B, N);
    return S > 1 ? (size_t)S * B * N * (2 + SA_C + 1) : 0;
}
// fp16x2-split kernels (dvm_sa_f16.hip)
size_t sa_f16_ws_bytes(int B, int N);
void sa_f16_carve(void *ws, int B, int N, _Float16 *&pp, _Float16 *&vp);
void launch_sa_split_f16(const float *p, const float *v, int B, int N, _Float16 *pp, _Float16 *vp, hipStream_t s);
void launch_sa_rowstats_f16(const _Float16 *pp, int B, int N, int kchunk, int Z, float *stats, hipStream_t s);
void launch_sa_apply_f16(const _Float16 *pp, const _Float16 *vp, const float *stats, int B, int N, int kchunk, int Z, float *xr,
                         float *cinv, hipStream_t s);

// stats [B][N][2] and xr [B][N][64] (and cinv [B][N] when given) from p, v; `part` holds sa_partial_floats(B, N) floats,
// `f16ws` sa_f16_ws_bytes(B, N) bytes
static void launch_sa_forward(const float *p, const float *v, int B, int N, float *xr, float *stats, float *cinv, float *part,
                              void *f16ws, hipStream_t s) {
    const int S = sa_splits(B, N);
    const int kchunk = ((N + S - 1) / S + SA_KB - 1) / SA_KB * SA_KB;
    const int Z = (N + kchunk - 1) / kchunk;
    dim3 grid((N + 31) / 32, B, Z);
    const long rows = (long)B * N;
    float *pstats = part, *po = Z > 1 ? pstats + (size_t)Z * rows * 2 : nullptr, *pc = Z > 1 ? po + (size_t)Z * rows * SA_C : nullptr;
    const bool f16 = f16ws != nullptr;   // (a caller without the split planes' workspace — the training forward — keeps both contractions on the fp32 matrix instruction)
    _Float16 *pp = nullptr, *vp = nullptr;
    if (f16) {
        sa_f16_carve(f16ws, B, N, pp, vp);
        launch_sa_split_f16(p, v, B, N, pp, vp, s);
    }
    // pass 1
    float *st1 = Z == 1 ? stats : pstats;
    if (f16)
        launch_sa_rowstats_f16(pp, B, N, kchunk, Z, st1, s);
    else
        hipLaunchKernelGGL(sa_rowstats_kernel, grid, dim3(256), 0, s, p, N, kchunk, st1);
    if (Z > 1) hipLaunchKernelGGL(sa_stats_merge_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, pstats, rows, Z, stats);
    // pass 2
    float *xo = Z == 1 ? xr : po, *co = Z == 1 ? cinv : pc;
    if (f16)
        launch_sa_apply_f16(pp, vp, stats, B, N, kchunk, Z, xo, co, s);
    else
        hipLaunchKernelGGL(sa_apply_kernel, grid, dim3(256), 0, s, p, v, stats, N, kchunk, xo, co);
    if (Z > 1) hipLaunchKernelGGL(sa_apply_merge_kernel, dim3((unsigned)((rows * SA_C + 255) / 256)), dim3(256), 0, s, po, pc, rows, Z, xr, cinv);
}
}  // namespace dvm

DVM_EXPORT size_t dvm_sa_attention_workspace_bytes(int B, int N) {
    return align_up((size_t)B * N * 2 * sizeof(float)) + align_up(sa_partial_floats(B, N) * sizeof(float)) + sa_f16_ws_bytes(B, N);
}

DVM_EXPORT int dvm_sa_attention_fwd_f32(const float *p, const float *v, int B, int N, float *xr, void *ws, size_t ws_bytes,
                                        void *stream) {
    DVM_REQUIRE(p && v && xr && B >= 1 && N >= 1, "dvm_sa_attention_fwd_f32: bad arguments");
    Arena ar(ws, ws_bytes);
    float *stats = ar.take<float>((size_t)B * N * 2);
    float *part = ar.take<float>(sa_partial_floats(B, N));
    char *f16ws = ar.take<char>(sa_f16_ws_bytes(B, N));
    if (!ar.ok()) {
        set_error("dvm_sa_attention_fwd_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    launch_sa_forward(p, v, B, N, xr, stats, nullptr, part, f16ws, (hipStream_t)stream);
    DVM_CHECK_LAUNCH("sa_attention");
    return DVM_OK;
}

// The training forward stays on the fp32 matrix instruction: the backward kernels recompute E in fp32 and take the row
// statistics and column sums from here — with the fp16-split E (2.4e-7 * sum|p_i p_j| off) the two would disagree by
// up to 1e-4 relative in dp at large logits.  Inference (dvm_sa_attention_fwd_f32) has no such coupling.
DVM_EXPORT size_t dvm_sa_attention_train_fwd_workspace_bytes(int B, int N) { return align_up(sa_partial_floats(B, N) * sizeof(float)); }

DVM_EXPORT int dvm_sa_attention_train_fwd_f32(const float *p, const float *v, int B, int N, float *xr, float *stats, float *cinv,
                                              void *ws, size_t ws_bytes, void *stream) {
    DVM_REQUIRE(p && v && xr && stats && cinv && B >= 1 && N >= 1, "dvm_sa_attention_train_fwd_f32: bad arguments");
    const size_t need = sa_partial_floats(B, N) * sizeof(float);
    DVM_REQUIRE(need == 0 || (ws != nullptr && ws_bytes >= need), "dvm_sa_attention_train_fwd_f32: workspace too small (%zu < %zu)",
                ws_bytes, need);
    launch_sa_forward(p, v, B, N, xr, stats, cinv, (float *)ws, nullptr, (hipStream_t)stream);
    DVM_CHECK_LAUNCH("sa_attention_train_fwd");
    return DVM_OK;
}

DVM_EXPORT int dvm_n2p_attention_fwd_f32(const float *q, const float *kp, const float *vp, const int32_t *idx, int B, int N,
                                         int C, int K, int heads, float *out, void *stream) {
    DVM_REQUIRE(q && kp && vp && idx && out, "dvm_n2p_attention_fwd_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1, "dvm_n2p_attention_fwd_f32: empty input");
    DVM_REQUIRE((C == 64 || C == 128) && heads == 4, "dvm_n2p_attention_fwd_f32: C=%d heads=%d unsupported", C, heads);
    DVM_REQUIRE(K >= 1 && K <= 64, "dvm_n2p_attention_fwd_f32: K=%d unsupported (1..64)", K);
    dim3 grid((N + 3) / 4, B);
    hipStream_t s = (hipStream_t)stream;
    if (C == 64)
        hipLaunchKernelGGL(n2p_attention_kernel<64>, grid, dim3(256), 0, s, q, kp, vp, idx, N, K, heads, out);
    else
        hipLaunchKernelGGL(n2p_attention_kernel<128>, grid, dim3(256), 0, s, q, kp, vp, idx, N, K, heads, out);
    DVM_CHECK_LAUNCH("n2p_attention");
    return DVM_OK;
}

DVM_EXPORT int dvm_dist_loss_bwd_weights_f32(const float *feat, const float *dist, const int32_t *anchors, const int32_t *idx,
                                             const float *g_out, int B, int N, int C, int nA, int k, float *W, void *stream) {
    DVM_REQUIRE(feat && dist && anchors && idx && g_out && W, "dvm_dist_loss_bwd_weights_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1 && nA >= 1 && C % 4 == 0, "dvm_dist_loss_bwd_weights_f32: bad sizes");
    DVM_REQUIRE(k >= 1 && k <= 512 && k <= N, "dvm_dist_loss_bwd_weights_f32: k=%d unsupported", k);
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(W, 0, (size_t)B * nA * N * sizeof(float), s);
    if (C == 128)
        hipLaunchKernelGGL(dist_loss_bwd_weights_kernel, dim3((nA + 3) / 4, B), dim3(256), 0, s, feat, dist, anchors, idx, g_out, N, nA, k, W);
    else
        hipLaunchKernelGGL(dist_loss_bwd_weights_generic_kernel, dim3((nA + 3) / 4, B), dim3(256), 0, s, feat, dist, anchors, idx, g_out, N, C, nA,
                           k, W);
    DVM_CHECK_LAUNCH("dist_loss_bwd_weights");
    return DVM_OK;
}

DVM_EXPORT size_t dvm_dist_loss_workspace_bytes(int B, int N, int C, int nA, int k) {
    return align_up((size_t)B * nA * C * sizeof(float)) + align_up((size_t)B * nA * k * sizeof(int32_t)) +
           align_up((size_t)B * ((nA + 3) / 4) * sizeof(double)) + dvm_knn_neg_workspace_bytes(B, nA, N, C, k);
}

namespace dvm {
// dvm_dist_loss_fwd_f32 with the sum written at out[b * out_stride + out_off], the selected neighbours at idx_out (NULL: scratch) and, for
// a backward that does not read the feature rows again (C == 128), x_j / y_j at xsave [B][nA][k] x 2; fa_out [B][nA][C] = the anchors' rows
int launch_dist_loss_fwd(const float *feat, const float *dist, const int32_t *anchors, int B, int N, int C, int nA, int k, float *out, int out_stride,
                         int out_off, int32_t *idx_out, float *xsave, float *fa_out, void *ws, size_t ws_bytes, hipStream_t s) {
    Arena ar(ws, ws_bytes);
    float *fa = ar.take<float>((size_t)B * nA * C);
    int32_t *idx = ar.take<int32_t>((size_t)B * nA * k);
    int nblk = (nA + 3) / 4;
    double *partial = ar.take<double>((size_t)B * nblk);
    float *na = ar.take<float>((size_t)B * nA);
    float *nb = ar.take<float>((size_t)B * N);
    float *S = ar.take<float>((size_t)B * nA * N);
    if (!ar.ok()) {
        set_error("dvm_dist_loss_fwd_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    if (idx_out) idx = idx_out;
    if (fa_out) fa = fa_out;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(((long)nA * C + 255) / 256), B), dim3(256), 0, s, feat, anchors, N, C, nA,
                       fa);
    launch_knn_neg(fa, feat, B, nA, N, C, k, idx, na, nb, S, s);
    if (C == 128 && k <= 512)
        hipLaunchKernelGGL(dist_loss_kernel, dim3(nblk, B), dim3(256), 0, s, feat, dist, anchors, idx, N, nA, k, partial, xsave);
    else
        hipLaunchKernelGGL(dist_loss_generic_kernel, dim3(nblk, B), dim3(256), 0, s, feat, dist, anchors, idx, N, C, nA, k, partial);
    launch_reduce_partials(partial, B, nblk, 1.f, out, out_stride, out_off, s);
    return DVM_OK;
}
// W [B][nA][N] (zeroed here) and its row sums from the forward's xsave; gterm read at gterm[b * gstride]
void launch_dist_loss_bwd_weights_saved(const float *xsave, const int32_t *idx, const float *gterm, int gstride, int B, int N, int nA, int k, float *W,
                                        float *rs, hipStream_t s) {
    (void)hipMemsetAsync(W, 0, (size_t)B * nA * N * sizeof(float), s);
    hipLaunchKernelGGL(dist_loss_bwd_weights_saved_kernel, dim3((nA + 3) / 4, B), dim3(256), 0, s, xsave, idx, gterm, gstride, N, nA, k, W, rs);
}
}  // namespace dvm

DVM_EXPORT int dvm_dist_loss_fwd_f32(const float *feat, const float *dist, const int32_t *anchors, int B, int N, int C, int nA,
                                     int k, float *out, int32_t *idx_out, void *ws, size_t ws_bytes, void *stream) {
    DVM_REQUIRE(feat && dist && anchors && out, "dvm_dist_loss_fwd_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1 && nA >= 1 && C % 4 == 0, "dvm_dist_loss_fwd_f32: bad sizes");
    DVM_REQUIRE(k >= 1 && k <= 512 && k <= N, "dvm_dist_loss_fwd_f32: k=%d unsupported", k);
    const int rc = launch_dist_loss_fwd(feat, dist, anchors, B, N, C, nA, k, out, 1, 0, idx_out, nullptr, nullptr, ws, ws_bytes, (hipStream_t)stream);
    if (rc != DVM_OK) return rc;
    DVM_CHECK_LAUNCH("dist_loss");
    return DVM_OK;
}
