// dvm_n2p_bwd.hip — training twins of the N2P attention core (reference models/model.py:339-350, 375-386;
// SURVEY §8b "backward twins ... n2p").
//
// Per point i with feature-space neighbours j = idx[i, 0..K) and 4 heads of D = C/4 channels:
//     e_hj = q_h . (kp_j - kp_i)_h / sqrt(D),   a_h = softmax_j(e_h),   out_h = sum_j a_hj (vp_j - vp_i)_h
// The forward here also writes a [B,N,K,4] (10 MB at B=8, N=2048, K=40 — instead of the five (B,N,K,C)
// tensors the unfused formulation keeps for autograd).  Backward, with g = dL/d out:
//     da_hj = g_h . vp_j          (the - g_h . vp_i part is constant over j and cancels in the softmax backward)
//     de_hj = a_hj (da_hj - sum_j' a_hj' da_hj')
//     dq_h  = sum_j de_hj kp_j / sqrt(D)                      (sum_j de_hj = 0 removes kp_i)
//     dkp_j += de_hj q_h / sqrt(D) ,  dvp_j += a_hj g_h ,  dvp_i -= g_h
// One wave per point, a head = 16 consecutive lanes (C/64 channels per lane); neighbour rows are gathered
// with coalesced 256/512 B reads; the backward's scatter is a second gather over the reversed neighbour lists.
#include "dvm_common.h"

namespace dvm {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NP_H = 4;
constexpr int NP_KMAX = 64;

// Common layout of the three per-point kernels: one wave per point; a lane owns 4 consecutive channels (16-byte loads),
// so a row takes LPR = C/4 lanes and the wave works on EPW = 64 / LPR neighbours per instruction (4 at C = 64, 2 at
// C = 128); a head is LPH = LPR / 4 neighbouring lanes.  (One float per lane, one neighbour per instruction and a 16-lane
// butterfly per neighbour before: 72 / 93 us forward, 79 / 92 us backward per layer at 8 x 2048 x 40.)
template <int C>
struct NpLayout {
    static constexpr int LPR = C / 4, EPW = 64 / LPR, D = C / NP_H, LPH = LPR / NP_H, LD = 3 * C;
};
// sum over the LPH lanes of a head (LPH = 4 or 8: xor 1, 2[, 4] stay inside the head's lanes)
template <int LPH>
__device__ __forceinline__ float head_sum(float v) {   // the xor-1, 2[, 4] butterfly on the DPP network (dvm_common.h)
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    if (LPH == 8) v += dpp_move<0x141>(v);
    return v;
}
// sum over the edge slots of a wave: lanes l, l + LPR, ... (LPR = 16: xor 16 and 32; LPR = 32: xor 32)
template <int LPR>
__device__ __forceinline__ float slot_sum(float v) {
    if (LPR == 16) v += lane_xor16(v);
    return v + lane_xor32(v);
}

template <int C>
__global__ __launch_bounds__(256) void n2p_core_fwd_kernel(const float *__restrict__ qkv, const int32_t *__restrict__ idx, int N,
                                                           int K, float *__restrict__ out, float *__restrict__ attn) {
    using L = NpLayout<C>;
    __shared__ float se[4][NP_KMAX * NP_H];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long pt0 = (long)blockIdx.x * 4 + wave;
    const bool valid = pt0 < N;
    const long pt = valid ? pt0 : N - 1;
    const size_t base = (size_t)blockIdx.y * N;
    const int sub = lane / L::LPR, l = lane % L::LPR, hd = l / L::LPH;
    const float scale = sqrtf((float)L::D);
    const float *self = qkv + (base + pt) * L::LD + 4 * l;
    const f32x4 qv = *(const f32x4 *)self, ki = *(const f32x4 *)(self + C), vi = *(const f32x4 *)(self + 2 * C);
    const int32_t *nb = idx + (base + pt) * K;
    // logits e_hj = q_h . (kp_j - kp_i)_h / sqrt(D)
    for (int j = sub; j < K; j += L::EPW) {
        const f32x4 kj = *(const f32x4 *)(qkv + (base + nb[j]) * L::LD + C + 4 * l);
        float part = qv.x * (kj.x - ki.x);
        part = fmaf(qv.y, kj.y - ki.y, part);
        part = fmaf(qv.z, kj.z - ki.z, part);
        part = fmaf(qv.w, kj.w - ki.w, part);
        part = head_sum<L::LPH>(part);
        if (l % L::LPH == 0) se[wave][j * NP_H + hd] = part / scale;
    }
    __syncthreads();
    // softmax over the K neighbours of each head: lane t handles entries t, t + 64, ... whose head is t % 4
    float m = -INFINITY;
    for (int t = lane; t < K * NP_H; t += 64) m = fmaxf(m, se[wave][t]);
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
    for (int t = lane; t < K * NP_H; t += 64) {
        const float w = __expf(se[wave][t] - m);
        se[wave][t] = w;
        sum += w;
    }
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.0f / sum;
    for (int t = lane; t < K * NP_H; t += 64) {
        const float a = se[wave][t] * inv;
        se[wave][t] = a;
        if (valid) attn[(base + pt) * K * NP_H + t] = a;
    }
    __syncthreads();
    // out_h = sum_j a_hj (vp_j - vp_i)_h
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j = sub; j < K; j += L::EPW) {
        const f32x4 vj = *(const f32x4 *)(qkv + (base + nb[j]) * L::LD + 2 * C + 4 * l);
        acc += se[wave][j * NP_H + hd] * (vj - vi);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = slot_sum<L::LPR>(acc[c]);
    if (valid && sub == 0) *(f32x4 *)(out + (base + pt) * C + 4 * l) = acc;
}

// ---- backward.  Point pass: de (scaled by 1/sqrt(D)) for every (point, neighbour, head) and dq by gathering
// kp rows.  The scatter side (dkp_j += de q_i, dvp_j += a g_i) is turned into a gather as well: a counting
// sort of idx gives every target row its list of (point, slot) references (CSR), and one wave per target row
// sums its in-edges in registers — 40 int atomics per point instead of 80 x C float atomics.
template <int C>
__global__ __launch_bounds__(256) void n2p_bwd_point_kernel(const float *__restrict__ qkv, const int32_t *__restrict__ idx,
                                                            const float *__restrict__ attn, const float *__restrict__ gout, int N,
                                                            int K, float *__restrict__ de_out, float *__restrict__ dqkv) {
    using L = NpLayout<C>;
    __shared__ float sa[4][NP_KMAX * NP_H], sd[4][NP_KMAX * NP_H];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long pt0 = (long)blockIdx.x * 4 + wave;
    const bool valid = pt0 < N;
    const long pt = valid ? pt0 : N - 1;
    const size_t base = (size_t)blockIdx.y * N;
    const int sub = lane / L::LPR, l = lane % L::LPR, hd = l / L::LPH;
    const float inv_scale = 1.0f / sqrtf((float)L::D);
    f32x4 gv = {0.f, 0.f, 0.f, 0.f};
    if (valid) gv = *(const f32x4 *)(gout + (base + pt) * C + 4 * l);
    const int32_t *nb = idx + (base + pt) * K;
    for (int j = sub; j < K; j += L::EPW) {  // da_hj = g_h . vp_j
        const f32x4 vj = *(const f32x4 *)(qkv + (base + nb[j]) * L::LD + 2 * C + 4 * l);
        float part = gv.x * vj.x;
        part = fmaf(gv.y, vj.y, part);
        part = fmaf(gv.z, vj.z, part);
        part = fmaf(gv.w, vj.w, part);
        part = head_sum<L::LPH>(part);
        if (l % L::LPH == 0) sd[wave][j * NP_H + hd] = part;
    }
    __syncthreads();
    float dot = 0.f;  // sum_j a_hj da_hj for head lane % 4
    for (int t = lane; t < K * NP_H; t += 64) {
        const float a = attn[(base + pt) * K * NP_H + t];
        sa[wave][t] = a;
        dot = fmaf(a, sd[wave][t], dot);
    }
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) dot += __shfl_xor(dot, o, 64);
    for (int t = lane; t < K * NP_H; t += 64) {
        const float de = sa[wave][t] * (sd[wave][t] - dot) * inv_scale;
        sd[wave][t] = de;
        if (valid) de_out[(base + pt) * K * NP_H + t] = de;
    }
    __syncthreads();
    f32x4 dq = {0.f, 0.f, 0.f, 0.f};
    for (int j = sub; j < K; j += L::EPW) {
        const f32x4 kj = *(const f32x4 *)(qkv + (base + nb[j]) * L::LD + C + 4 * l);
        dq += sd[wave][j * NP_H + hd] * kj;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) dq[c] = slot_sum<L::LPR>(dq[c]);
    if (valid && sub == 0) *(f32x4 *)(dqkv + (base + pt) * L::LD + 4 * l) = dq;
}

__global__ void csr_count_kernel(const int32_t *__restrict__ idx, int N, int K, int32_t *__restrict__ cnt) {
    const int b = blockIdx.y;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)N * K) return;
    atomicAdd(cnt + (size_t)b * (N + 1) + idx[(size_t)b * N * K + e], 1);
}

// exclusive scan of cnt[b][0..N) in place -> offs (cnt[b][N] = total); cursor = copy of the offsets
__global__ __launch_bounds__(1024) void csr_scan_kernel(int32_t *__restrict__ cnt, int N, int32_t *__restrict__ cursor) {
    __shared__ int part[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    int32_t *c = cnt + (size_t)b * (N + 1);
    const int per = (N + 1023) / 1024;
    const int lo = min(N, tid * per), hi = min(N, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += c[i];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;  // exclusive prefix of this thread's chunk
    for (int i = lo; i < hi; ++i) {
        const int v = c[i];
        c[i] = run;
        cursor[(size_t)b * N + i] = run;
        run += v;
    }
    if (tid == 1023) c[N] = part[1023];
}

// edges[pos] = (e, source point) with e = point * K + slot (the point is stored, not derived: an integer division by the
// run-time K per edge was a fifth of the gather kernel's instructions)
__global__ void csr_fill_kernel(const int32_t *__restrict__ idx, int N, int K, int32_t *__restrict__ cursor,
                                int2 *__restrict__ edges) {
    const int b = blockIdx.y;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)N * K) return;
    const int pos = atomicAdd(cursor + (size_t)b * N + idx[(size_t)b * N * K + e], 1);
    edges[(size_t)b * N * K + pos] = make_int2((int)e, (int)(e / K));
}

// dkp_r = sum over the in-edges (i -> r) of de_i,slot * q_i (per head), dvp_r = sum a_i,slot * g_i - g_r.
// One wave per target point; a lane owns 4 consecutive channels (16-byte loads), so a row takes C/4 lanes and the wave
// walks 64 / (C/4) in-edges per instruction (4 at C = 64, 2 at C = 128), two such groups in flight; the partial sums of
// the edge slots are combined at the end.  (One float per lane and one edge per instruction before: 213 / 192 us.)
// The three kernels above in one, for clouds whose counters fit in LDS (2 (N + 1) ints): ONE workgroup per cloud counts,
// scans and fills with LDS atomics — 8 workgroups instead of 3 launches of global atomics (35 + 5 + 46 us per layer at
// 8 x 2048 x 40); it runs beside the other stream's kernels, and the order inside a list carries no contract (the gather
// sums fp32 either way).
__global__ __launch_bounds__(1024) void csr_build_lds_kernel(const int32_t *__restrict__ idx, int N, int K, int32_t *__restrict__ offs,
                                                             int2 *__restrict__ edges) {
    extern __shared__ int csr_lds[];          // cnt[N + 1] | cursor[N]
    __shared__ int part[1024];
    int *cnt = csr_lds, *cursor = csr_lds + N + 1;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int32_t *ib = idx + (size_t)b * N * K;
    const int E = N * K;
    for (int i = tid; i <= N; i += 1024) cnt[i] = 0;
    __syncthreads();
    // (8 index loads in flight per thread: one workgroup per cloud has only its own 16 waves to cover the load latency)
    for (int e0 = tid; e0 < E; e0 += 8 * 1024) {
        int t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = e0 + u * 1024 < E ? ib[e0 + u * 1024] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (t[u] >= 0) atomicAdd(&cnt[t[u]], 1);
    }
    __syncthreads();
    const int per = (N + 1023) / 1024;
    const int lo = min(N, tid * per), hi = min(N, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += cnt[i];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;  // exclusive prefix of this thread's chunk
    int32_t *ob = offs + (size_t)b * (N + 1);
    for (int i = lo; i < hi; ++i) {
        const int v = cnt[i];
        cursor[i] = run;
        ob[i] = run;
        run += v;
    }
    if (tid == 1023) ob[N] = part[1023];
    __syncthreads();
    int2 *eb = edges + (size_t)b * E;
    for (int e0 = tid; e0 < E; e0 += 8 * 1024) {
        int t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = e0 + u * 1024 < E ? ib[e0 + u * 1024] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (t[u] < 0) continue;
            const int e = e0 + u * 1024;
            const int pos = atomicAdd(&cursor[t[u]], 1);
            eb[pos] = make_int2(e, e / K);
        }
    }
}

// Deterministic mode (dvm_set_deterministic): the reversed lists in ASCENDING EDGE ORDER without atomics between threads.
// The edges of a cloud are cut into DET_CHUNKS contiguous chunks: (1) a histogram of targets per chunk (LDS counters, integer
// atomics inside one workgroup: order-free), (2) per target an exclusive scan over the chunks on top of the exclusive scan
// over the targets (= offs), (3) every chunk placed by ONE thread walking its edges in order from those bases.
constexpr int DET_CHUNKS = 64;
__global__ __launch_bounds__(256) void csr_det_hist_kernel(const int32_t *__restrict__ idx, int N, int K, int32_t *__restrict__ hist) {
    extern __shared__ int csr_lds[];   // cnt[N]
    const int b = blockIdx.y, ch = blockIdx.x;
    const int E = N * K, per = (E + DET_CHUNKS - 1) / DET_CHUNKS, e0 = ch * per, e1 = min(E, e0 + per);
    for (int i = threadIdx.x; i < N; i += blockDim.x) csr_lds[i] = 0;
    __syncthreads();
    const int32_t *ib = idx + (size_t)b * E;
    for (int e = e0 + threadIdx.x; e < e1; e += blockDim.x) atomicAdd(&csr_lds[ib[e]], 1);
    __syncthreads();
    int32_t *h = hist + ((size_t)b * DET_CHUNKS + ch) * N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) h[i] = csr_lds[i];
}
// hist[b][ch][t] -> base position of chunk ch's first edge into target t; offs[b][0..N]
__global__ __launch_bounds__(1024) void csr_det_scan_kernel(int32_t *__restrict__ hist, int N, int32_t *__restrict__ offs) {
    __shared__ int part[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    int32_t *h = hist + (size_t)b * DET_CHUNKS * N;
    const int per = (N + 1023) / 1024, lo = min(N, tid * per), hi = min(N, lo + per);
    int s = 0;
    for (int t = lo; t < hi; ++t)
        for (int ch = 0; ch < DET_CHUNKS; ++ch) s += h[(size_t)ch * N + t];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    int32_t *ob = offs + (size_t)b * (N + 1);
    for (int t = lo; t < hi; ++t) {
        ob[t] = run;
        for (int ch = 0; ch < DET_CHUNKS; ++ch) {
            const int v = h[(size_t)ch * N + t];
            h[(size_t)ch * N + t] = run;
            run += v;
        }
    }
    if (tid == 1023) ob[N] = part[1023];
}
__global__ __launch_bounds__(256) void csr_det_fill_kernel(const int32_t *__restrict__ idx, int N, int K, const int32_t *__restrict__ hist,
                                                           int2 *__restrict__ edges) {
    extern __shared__ int csr_lds[];   // cursor[N]
    const int b = blockIdx.y, ch = blockIdx.x;
    const int E = N * K, per = (E + DET_CHUNKS - 1) / DET_CHUNKS, e0 = ch * per, e1 = min(E, e0 + per);
    const int32_t *h = hist + ((size_t)b * DET_CHUNKS + ch) * N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) csr_lds[i] = h[i];
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int32_t *ib = idx + (size_t)b * E;
    int2 *eb = edges + (size_t)b * E;
    for (int e = e0; e < e1; ++e) {
        const int pos = csr_lds[ib[e]]++;
        eb[pos] = make_int2(e, e / K);
    }
}

template <int C>
__global__ __launch_bounds__(256) void n2p_bwd_gather_kernel(const float *__restrict__ qkv, const float *__restrict__ attn,
                                                             const float *__restrict__ de_buf, const float *__restrict__ gout,
                                                             const int32_t *__restrict__ offs, const int2 *__restrict__ edges,
                                                             int N, int K, float *__restrict__ dqkv) {
    constexpr int LPR = C / 4, EPW = 64 / LPR, LD = 3 * C, D = C / NP_H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r = (long)blockIdx.x * 4 + wave;
    if (r >= N) return;
    const int b = blockIdx.y;
    const size_t base = (size_t)b * N;
    const int sub = lane / LPR, l = lane % LPR, hd = (4 * l) / D;
    const int beg = offs[(size_t)b * (N + 1) + r], end = offs[(size_t)b * (N + 1) + r + 1];
    const int2 *ed = edges + base * K;
    const float *ab = attn + base * K * NP_H, *db = de_buf + base * K * NP_H;
    f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
    auto edge = [&](int e, f32x4 &k_acc, f32x4 &v_acc) {
        const int2 en = ed[e];
        const float d = db[(size_t)en.x * NP_H + hd], a = ab[(size_t)en.x * NP_H + hd];
        const f32x4 q = *(const f32x4 *)(qkv + (base + en.y) * LD + 4 * l);
        const f32x4 g = *(const f32x4 *)(gout + (base + en.y) * C + 4 * l);
        k_acc += d * q;
        v_acc += a * g;
    };
    int e = beg + sub;
    f32x4 ak2 = ak, av2 = av;
    for (; e + EPW < end; e += 2 * EPW) {
        edge(e, ak, av);
        edge(e + EPW, ak2, av2);
    }
    if (e < end) edge(e, ak, av);
    ak += ak2, av += av2;
    // combine the EPW edge slots (lanes l, l + LPR, ...)
#pragma unroll
    for (int c = 0; c < 4; ++c) ak[c] = slot_sum<LPR>(ak[c]), av[c] = slot_sum<LPR>(av[c]);
    if (sub == 0) {
        float *dst = dqkv + (base + r) * LD + 4 * l;
        const f32x4 gr = *(const f32x4 *)(gout + (base + r) * C + 4 * l);
        *(f32x4 *)(dst + C) = ak;
        *(f32x4 *)(dst + 2 * C) = av - gr;
    }
}

}  // namespace
}  // namespace dvm

using namespace dvm;

DVM_EXPORT int dvm_n2p_core_fwd_f32(const float *qkv, const int32_t *idx, int B, int N, int C, int K, int heads, float *out,
                                    float *attn, void *stream) {
    DVM_REQUIRE(qkv && idx && out && attn, "dvm_n2p_core_fwd_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1, "dvm_n2p_core_fwd_f32: empty input");
    DVM_REQUIRE((C == 64 || C == 128) && heads == NP_H, "dvm_n2p_core_fwd_f32: need C in {64,128}, heads == 4 (C=%d heads=%d)", C,
                heads);
    DVM_REQUIRE(K >= 1 && K <= NP_KMAX, "dvm_n2p_core_fwd_f32: K=%d unsupported (1..64)", K);
    dim3 grid((N + 3) / 4, B);
    hipStream_t s = (hipStream_t)stream;
    if (C == 64)
        hipLaunchKernelGGL(n2p_core_fwd_kernel<64>, grid, dim3(256), 0, s, qkv, idx, N, K, out, attn);
    else
        hipLaunchKernelGGL(n2p_core_fwd_kernel<128>, grid, dim3(256), 0, s, qkv, idx, N, K, out, attn);
    DVM_CHECK_LAUNCH("n2p_core_fwd");
    return DVM_OK;
}

DVM_EXPORT size_t dvm_n2p_core_bwd_workspace_bytes(int B, int N, int K) {
    return align_up((size_t)B * N * K * NP_H * sizeof(float)) + align_up((size_t)B * (N + 1) * sizeof(int32_t)) +
           align_up((size_t)B * N * sizeof(int32_t)) + align_up((size_t)B * N * K * sizeof(int2)) +
           align_up((size_t)B * DET_CHUNKS * N * sizeof(int32_t));   // (last: per-chunk histograms of the deterministic list build)
}

DVM_EXPORT int dvm_n2p_core_bwd_f32(const float *qkv, const int32_t *idx, const float *attn, const float *g_out, int B, int N, int C,
                                    int K, int heads, float *d_qkv, void *ws, size_t ws_bytes, void *stream) {
    DVM_REQUIRE(qkv && idx && attn && g_out && d_qkv, "dvm_n2p_core_bwd_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1, "dvm_n2p_core_bwd_f32: empty input");
    DVM_REQUIRE((C == 64 || C == 128) && heads == NP_H, "dvm_n2p_core_bwd_f32: need C in {64,128}, heads == 4 (C=%d heads=%d)", C,
                heads);
    DVM_REQUIRE(K >= 1 && K <= NP_KMAX, "dvm_n2p_core_bwd_f32: K=%d unsupported (1..64)", K);
    Arena ar(ws, ws_bytes);
    float *de = ar.take<float>((size_t)B * N * K * NP_H);
    int32_t *offs = ar.take<int32_t>((size_t)B * (N + 1));
    int32_t *cursor = ar.take<int32_t>((size_t)B * N);
    int2 *edges = ar.take<int2>((size_t)B * N * K);
    int32_t *hist = ar.take<int32_t>((size_t)B * DET_CHUNKS * N);
    if (!ar.ok()) {
        set_error("dvm_n2p_core_bwd_f32: workspace too small (%zu < %zu)", ws_bytes, ar.off);
        return DVM_ENOSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((N + 3) / 4, B), egrid((unsigned)(((long)N * K + 255) / 256), B);
    const size_t csr_lds = (size_t)(2 * N + 1) * sizeof(int);
    // (one workgroup per cloud: from 4 clouds on; fewer, larger clouds keep the chip busier with the three-kernel form)
    if (deterministic() && (size_t)N * sizeof(int) <= 96 * 1024) {
        ensure_dyn_lds((const void *)csr_det_hist_kernel, (int)(N * sizeof(int)));
        ensure_dyn_lds((const void *)csr_det_fill_kernel, (int)(N * sizeof(int)));
        hipLaunchKernelGGL(csr_det_hist_kernel, dim3(DET_CHUNKS, B), dim3(256), (size_t)N * sizeof(int), s, idx, N, K, hist);
        hipLaunchKernelGGL(csr_det_scan_kernel, dim3(B), dim3(1024), 0, s, hist, N, offs);
        hipLaunchKernelGGL(csr_det_fill_kernel, dim3(DET_CHUNKS, B), dim3(256), (size_t)N * sizeof(int), s, idx, N, K, hist, edges);
    } else if (B >= 4 && csr_lds <= 96 * 1024 && (long)N * K < (1L << 31)) {
        ensure_dyn_lds((const void *)csr_build_lds_kernel, (int)csr_lds);
        hipLaunchKernelGGL(csr_build_lds_kernel, dim3(B), dim3(1024), csr_lds, s, idx, N, K, offs, edges);
    } else {
        (void)hipMemsetAsync(offs, 0, (size_t)B * (N + 1) * sizeof(int32_t), s);
        hipLaunchKernelGGL(csr_count_kernel, egrid, dim3(256), 0, s, idx, N, K, offs);
        hipLaunchKernelGGL(csr_scan_kernel, dim3(B), dim3(1024), 0, s, offs, N, cursor);
        hipLaunchKernelGGL(csr_fill_kernel, egrid, dim3(256), 0, s, idx, N, K, cursor, edges);
    }
    if (C == 64) {
        hipLaunchKernelGGL(n2p_bwd_point_kernel<64>, grid, dim3(256), 0, s, qkv, idx, attn, g_out, N, K, de, d_qkv);
        hipLaunchKernelGGL(n2p_bwd_gather_kernel<64>, grid, dim3(256), 0, s, qkv, attn, de, g_out, offs, edges, N, K, d_qkv);
    } else {
        hipLaunchKernelGGL(n2p_bwd_point_kernel<128>, grid, dim3(256), 0, s, qkv, idx, attn, g_out, N, K, de, d_qkv);
        hipLaunchKernelGGL(n2p_bwd_gather_kernel<128>, grid, dim3(256), 0, s, qkv, attn, de, g_out, offs, edges, N, K, d_qkv);
    }
    DVM_CHECK_LAUNCH("n2p_core_bwd");
    return DVM_OK;
}
