// dvm_softcorr_sweep2.hip — pass A of the soft-correspondence kernel (K1), second form.
// Built with -fno-honor-nans (Makefile): no NaN is ever formed here (keys are bit patterns compared as ints or as the high
// words of finite doubles; +inf only passes through additions and one multiplication by a positive constant), and it lets
// fmin / fmax on the packed (key, column) doubles lower to bare v_min_f64 / v_max_f64 that the scheduler can place.
// (reference: models/loss.py:110-114, 1339-1347, 1404-1407)
#include <stdlib.h>

#define DVM_K1_BUILTIN_MINMAX 1
#include "dvm_softcorr_f16.h"

namespace dvm {
namespace k1 {
namespace {

// ---------------------------------------------------------------- pass A, second form: no per-entry work
// What the first form spends its time on is not the matrix work (24 matrix instructions per 32 x 32 sub-tile) but the ~300
// vector instructions behind it: per entry {fma, add, compare, mask, LDS staging} and a wave-level insertion loop that
// runs max-over-lanes(#flagged) times for a handful of active lanes.  This form removes the per-entry arithmetic and the
// data-dependent loop from the common path:
//  * the norms ride on the matrix pipe: one more matrix instruction per sub-tile whose 16 k-slots carry |k|^2 and |q|^2,
//    each as three exact fp16 pieces times power-of-two constants, so that the accumulator IS the (scaled, non-negative)
//    squared distance  acc = (|q|^2 (1 + 2^-13) + |k|^2 - 2 q.k) s^2 / 2  (one scale s for both sides; the row-constant bias
//    keeps every accumulator >= 0, so keys order identically as floats, as ints and as the high words of doubles);
//  * a lane's 16 entries go through a fixed selection network on their bit patterns (the register number embedded in the
//    low 4 mantissa bits: keys are unique, the column is recovered from the key): sorted three smallest in 46 three-input
//    integer min / med / max instructions, no comparison against a threshold per entry;
//  * the two smallest are inserted into the sorted list of 12 unconditionally (the packed (key, column) doubles of the
//    first form); the third only decides whether the lane may hold MORE than two entries below its threshold, in which case
//    the wave repeats the selection on the remaining entries (a few percent of the sub-tiles once the lists have filled).
// Everything in the common path is straight-line register code, which the compiler can place between the matrix
// instructions of the NEXT sub-tile (PIPE): the chain of sub-tile i+1 is issued before the epilogue of sub-tile i.
// Same outputs as the first form (candidate columns, approximate squared distances, partial softmax sums), same
// guarantees: a column that belongs to a row's 12 smallest is never lost (every entry at or below the row's running bound
// is inserted), the softmax sum covers every column outside the final list that lies within the cut.
constexpr int H2_LDS_BYTES = 2 * HB_KT * HB_ROWB + 2 * HB_KT * 32;   // key tiles + norm fragments (32 B per key)
constexpr int H2_REMOVED = 0x7f800000;                               // bit pattern of +inf: larger than every finite key

struct H2Group {
    const char *qp, *kp;      // planes of the query / key side [B][rows][512]
    const char *knf;          // key-side norm fragments [B][Mpad][32 B]: fp16 {a1, a2, a3, 2^15, 2^4, 2^-7, 0, 0 | 0 x 8}
    const float *nq;          // |q|^2 (ATen order)
    int N, M, Mpad, tiles;
    int32_t *cidx;
    float *cd2, *lsum;
};
struct H2Args {
    H2Group g[2];
    const int *amax;          // bit pattern of max |x| over BOTH sides (the common scale)
    int blocks0;
    float neg_alpha, cutw;
    unsigned long long *stamps;   // diagnostic build only (DVM_K1_STAMPS): [block][wave][8] cycle totals per phase
};

// three-input integer min / max / median, written so that instruction selection forms v_min3_i32 / v_max3_i32 / v_med3_i32
// (plain expressions, not asm statements: the instruction scheduler has to see them as vector instructions)
__device__ __forceinline__ int imin3(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int imax3(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ int imed3(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }
// the accumulator's bit pattern with the register number r in its low 4 bits: (x & ~15) | r, one v_and_or_b32
__device__ __forceinline__ int embed4(float x, int r) { return (__float_as_int(x) & ~15) | r; }

// three fp16 pieces of a non-negative fp32 value x < 2^31:  x = p1 2^15 + p2 2^4 + p3 2^-7  (exact: 33 >= 24 bits)
__device__ __forceinline__ void norm_pieces(float x, _Float16 &p1, _Float16 &p2, _Float16 &p3) {
    p1 = (_Float16)(x * 0x1p-15f);
    const float r1 = x - (float)p1 * 0x1p+15f;
    p2 = (_Float16)(r1 * 0x1p-4f);
    const float r2 = r1 - (float)p2 * 0x1p+4f;
    p3 = (_Float16)(r2 * 0x1p+7f);
}
// |x|^2 in accumulator units: n s^2 / 2, formed as (n s) (s / 2) so that no intermediate leaves the fp32 range
__device__ __forceinline__ float norm_scaled(float n, int se) { return (n * pow2i(se)) * pow2i(se - 1); }

// key-side norm fragments, padded to whole key tiles with +inf: out [B][Mpad][16 fp16]
__global__ void norm_frags_kernel(const float *__restrict__ nrm, int M, int Mpad, const int *__restrict__ amax, char *__restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Mpad) return;
    f16x8 lo = {0, 0, 0, (_Float16)0x1p+15f, (_Float16)0x1p+4f, (_Float16)0x1p-7f, 0, 0};
    const f16x8 hi = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i < M) {
        _Float16 p1, p2, p3;
        norm_pieces(norm_scaled(nrm[(size_t)b * M + i], scale_exp(*amax)), p1, p2, p3);
        lo[0] = p1, lo[1] = p2, lo[2] = p3;
    } else {
        lo[0] = (_Float16)INFINITY;
    }
    char *p = out + ((size_t)b * Mpad + i) * 32;
    *(f16x8 *)p = lo;
    *(f16x8 *)(p + 16) = hi;
}

// sorted (s0 <= s1 <= s2) three smallest of 16 distinct ints
struct Top3 {
    int s0, s1, s2;
};
__device__ __forceinline__ Top3 sort3(int a, int b, int c) { return Top3{imin3(a, b, c), imed3(a, b, c), imax3(a, b, c)}; }
__device__ __forceinline__ Top3 merge3(const Top3 &a, const Top3 &b) {
    Top3 c;
    const int m00 = max(a.s0, b.s0);
    c.s0 = min(a.s0, b.s0);
    c.s1 = imin3(m00, a.s1, b.s1);
    c.s2 = min(imin3(a.s2, b.s2, max(a.s1, b.s0)), max(a.s0, b.s1));
    return c;
}
__device__ __forceinline__ Top3 top3_of_16(const int (&v)[16]) {
    Top3 t = merge3(merge3(sort3(v[0], v[1], v[2]), sort3(v[3], v[4], v[5])),
                    merge3(merge3(sort3(v[6], v[7], v[8]), sort3(v[9], v[10], v[11])), sort3(v[12], v[13], v[14])));
    const int x = v[15];
    return Top3{min(t.s0, x), imed3(t.s0, t.s1, x), imed3(t.s1, t.s2, x)};
}

// STAMP: diagnostic build — every wave adds up the shader cycles (s_memtime) it spends per phase and writes the totals to
// args.stamps; no output value depends on them.  Phases: 0 LDS-DMA issue, 1 matrix chain (fragment reads, waits, 25 matrix
// instructions, until the accumulator is readable), 2 straight-line epilogue, 3 slow path, 4 bound update, 5 barrier
// (incl. the wait for the wave's own DMA pieces), 6 whole sweep, 7 sub-tiles.
template <int PIPE, bool STAMP = false>
__global__ __launch_bounds__(HB_THREADS) void softcorr_sweep2_kernel(const H2Args args) {
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    unsigned long long T[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0, tstart = 0;
    if (STAMP) tstart = tlast = __builtin_amdgcn_s_memtime();
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if (STAMP) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            T[slot] += now - tlast;
            tlast = now;
        }
    };
    auto stamp_after = [&](int slot, int vgpr_value) __attribute__((always_inline)) {   // after `vgpr_value` has been produced
        if (STAMP) {
            const int x = __builtin_amdgcn_readfirstlane(vgpr_value);
            asm volatile("" ::"s"(x));
            stamp(slot);
        }
    };
    char *const ktile0 = smem_b;                                         // [2][HB_KT][512], 16-B chunks XOR-swizzled
    char *const knf0 = smem_b + (size_t)2 * HB_KT * HB_ROWB;             // [2][HB_KT][32]

    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int grp = lid >= args.blocks0 ? 1 : 0;
    lid -= grp ? args.blocks0 : 0;
    const H2Group &G = args.g[grp];
    const int N = G.N, M = G.M;
    const int b = lid / G.tiles, qt = lid % G.tiles;
    const float neg_alpha = args.neg_alpha;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;

    const char *kbase = G.kp + (size_t)b * M * HB_ROWB;
    const char *nfbase = G.knf + (size_t)b * G.Mpad * 32;
    const int qrow = qt * HB_QB + wave * 32 + r32;
    const int qrc = qrow < N ? qrow : N - 1;
    const char *qptr = G.qp + ((size_t)b * N + qrc) * HB_ROWB + 16 * h;
    f16x8 qh[8], qm[8];  // B-operand fragments, NEGATED (the accumulator carries + |q|^2 + |k|^2 - 2 q.k): k = 16 s + 8 h + j
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        qh[s] = -*(const f16x8 *)(qptr + 32 * s);
        qm[s] = -*(const f16x8 *)(qptr + 256 + 32 * s);
    }
    const int se = scale_exp(*args.amax);
    const float cf = pow2i(1 - 2 * se);             // 2 / s^2: accumulator units -> squared distance
    const float icf = pow2i(2 * se - 1);
    // the query's norm, biased by 2^-13 of itself (accumulators stay >= 0 whatever the rounding: a negative value needs
    // |q| ~ |k|, where the bias is twice the error bound), as the B operand of the norm instruction
    const float nas = norm_scaled(G.nq[(size_t)b * N + qrc], se);
    const float nasb = nas + nas * 0x1p-13f;
    const float bias = nasb - nas;                  // exact
    f16x8 qn = {0, 0, 0, 0, 0, 0, 0, 0};
    if (h == 0) {
        _Float16 p1, p2, p3;
        norm_pieces(nasb, p1, p2, p3);
        qn[0] = (_Float16)0x1p+15f, qn[1] = (_Float16)0x1p+4f, qn[2] = (_Float16)0x1p-7f;
        qn[3] = p1, qn[4] = p2, qn[5] = p3;
    }

    PackedBest<HB_KC> kb;  // (key bits, sub-tile base column): the key's low 4 bits name the accumulator register
    kb.init();
    float cref = -INFINITY, l = 0.f;
    int lim = 0x7fffffff, cut_i = 0x7fffffff;
    const float cutw = args.cutw;

    const int ntiles = (M + HB_KT - 1) / HB_KT;
    auto stage_tile = [&](int t, int buf) __attribute__((always_inline)) {
        const int j0 = t * HB_KT;
        char *kt = ktile0 + (size_t)buf * HB_KT * HB_ROWB;
#pragma unroll
        for (int e = 0; e < HB_GLDS_PER_WAVE; ++e) {
            const int piece = wave * HB_GLDS_PER_WAVE + e;       // 2 rows
            const int r = 2 * piece + h;
            const int jr = j0 + r < M ? j0 + r : M - 1;           // padding keys re-read the last row (their norm is +inf)
            const char *src = kbase + (size_t)jr * HB_ROWB + ((r32 ^ (r & 15)) << 4);   // 16-B chunk c of the LDS row holds chunk c ^ (row & 15)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(kt + piece * 1024), 16, 0, 0);
        }
        if (wave < 2)   // 64 keys x 32 B of norm fragments = two 1-KiB pieces
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nfbase + (size_t)j0 * 32 + wave * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(knf0 + buf * HB_KT * 32 + wave * 1024), 16, 0, 0);
    };

    auto key_d2 = [&](int key) __attribute__((always_inline)) { return fmaxf(__int_as_float(key) - bias, 0.f) * cf; };   // key -> squared distance
    auto add_term = [&](int key) __attribute__((always_inline)) {  // l += exp(s - cref) for a finite key (otherwise nothing)
        const bool live = key < H2_REMOVED;
        const float s = __builtin_amdgcn_sqrtf(key_d2(key)) * neg_alpha;
        const float cnew = live ? fmaxf(cref, s) : cref;
        const float sc = (cnew == cref) ? 1.f : __builtin_amdgcn_exp2f((cref - cnew) * LOG2E);
        const float term = live ? __builtin_amdgcn_exp2f((s - cnew) * LOG2E) : 0.f;
        l = l * sc + term;
        cref = cnew;
    };

    // The matrix work of one sub-tile: 24 product instructions + the norm instruction.  A fragment (row r32, chunk 2s + h of
    // plane p) sits at row * 512 + ((2s ^ h ^ (row & 15)) << 4) + 256 p: the swizzle touches the low four chunk bits only (a
    // ds_read_b128 is served in groups of 16 lanes, which then hit 16 different 16-B bank slots), so the plane, the sub-tile
    // and the buffer are immediate offsets of eight loop-invariant address registers — no address arithmetic in the loop.
    unsigned fadr[8];
    {
        const unsigned rowb = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)ktile0 + r32 * HB_ROWB;
        const int tq = h ^ (r32 & 15);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            fadr[s] = rowb + (((2 * s) ^ tq) << 4);
            asm volatile("" : "+v"(fadr[s]));   // keep them: recomputing costs two vector instructions per fragment
        }
    }
    unsigned nadr = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)knf0 + r32 * 32 + 16 * h;
    asm volatile("" : "+v"(nadr));
    auto lds16 = [](unsigned adr, int off) __attribute__((always_inline)) {
        return *(const f16x8 *)(const __attribute__((address_space(3))) char *)(size_t)(adr + off);
    };
    auto chain = [&](int buf, int sub) __attribute__((always_inline)) -> f32x16 {   // buf, sub: literals after inlining
        const int toff = buf * (HB_KT * HB_ROWB) + sub * (32 * HB_ROWB);
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const f16x8 an = lds16(nadr, buf * (HB_KT * 32) + sub * (32 * 32));
        // fragments in two batches of 4 k-steps (32 VGPRs each): the 16 of a whole sub-tile at once do not fit next to the
        // previous sub-tile's epilogue
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f16x8 ah[4], am[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ah[u] = lds16(fadr[4 * half + u], toff);
                am[u] = lds16(fadr[4 * half + u], toff + 256);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = 4 * half + u;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[u], qh[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u], qm[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u], qh[s], acc, 0, 0, 0);
            }
        }
        // the norms last: every partial sum before it has the magnitude of q.k
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(an, qn, acc, 0, 0, 0);
    };

    // the two smallest of the lane's 16 keys go into the list; an entry that leaves the list (or fails to enter it) contributes
    // its softmax term if it lies within the cut: nearly never at the alphas this sweep is used for — the whole wave skips
    // the exponentials unless one lane needs them
    auto pace = [&]() __attribute__((always_inline)) {
        // PIPE == 2: the block that ends here holds the matrix chain of the NEXT sub-tile (25 instructions, 17 LDS reads) and
        // this sub-tile's selection + insertions (~115 vector instructions): ask the scheduler for one matrix instruction
        // per five vector instructions, the first batch of fragment reads up front and the second a third of the way in
        __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
#pragma unroll
        for (int i = 0; i < 25; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i == 3) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
    };
    // an entry that leaves the list (or fails to enter it) owes its softmax term if it lies within the cut
    auto terms2 = [&](int k0, int k1) __attribute__((always_inline)) {
        if (__builtin_amdgcn_ballot_w64(min(k0, k1) <= cut_i) != 0) {
            add_term(k0 <= cut_i ? k0 : H2_REMOVED);
            add_term(k1 <= cut_i ? k1 : H2_REMOVED);
        }
    };

    // One sub-tile's 16 keys per lane.  Straight-line part: register numbers into the keys, sorted three smallest, the two
    // smallest into the list.  ONE wave-uniform branch behind it covers everything that is rare once the lists have filled:
    // a third entry at or below the bound in some lane (the wave then takes the next two of every lane, until none is left),
    // or an entry within the cut that left a list (its softmax term).  The slow path uses the list, so nothing of the
    // straight-line part can be sunk below the branch, away from the matrix instructions it is meant to run beside.
    auto epilogue = [&](const f32x16 &acc, int jb, bool paced) __attribute__((always_inline)) {
        int v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = embed4(acc[r], r);
        Top3 w = top3_of_16(v);
        const int k0 = __double2hiint(kb.insert(__hiloint2double(w.s0, jb)));
        const int k1 = __double2hiint(kb.insert(__hiloint2double(w.s1, jb)));
        if (paced) pace();
        stamp_after(2, k0 ^ k1 ^ __double2hiint(kb.e[0]));
        bool more = w.s2 <= lim && w.s2 < H2_REMOVED;
        if (__builtin_amdgcn_ballot_w64(more || min(k0, k1) <= cut_i) != 0) {
            terms2(k0, k1);
            while (__builtin_amdgcn_ballot_w64(more) != 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = v[r] <= w.s1 ? (H2_REMOVED | r) : v[r];
                w = top3_of_16(v);
                const int q0 = __double2hiint(kb.insert(__hiloint2double(w.s0, jb)));
                const int q1 = __double2hiint(kb.insert(__hiloint2double(w.s1, jb)));
                terms2(q0, q1);
                more = w.s2 <= lim && w.s2 < H2_REMOVED;
            }
        }
        stamp_after(3, __double2hiint(kb.e[0]));
    };
    // bound for the following sub-tiles: the row's KC-th best is at most min(a_K, b_K, max(a_m, b_m)), m = KC/2, over the two
    // half-lanes (a, b) that share the row; everything within the cut is processed as well (it owes a softmax term).  Any
    // earlier bound stays valid (it only admits more entries): refreshed after every sub-tile while the lists fill, once per
    // tile afterwards.
    auto update_bound = [&]() __attribute__((always_inline)) {
        const int wk = __double2hiint(kb.e[HB_KC - 1]), wm = __double2hiint(kb.e[HB_KC / 2 - 1]), w0 = __double2hiint(kb.e[0]);
        const auto sk = __builtin_amdgcn_permlane32_swap((unsigned)wk, (unsigned)wk, false, false);
        const auto sm = __builtin_amdgcn_permlane32_swap((unsigned)wm, (unsigned)wm, false, false);
        const auto s0 = __builtin_amdgcn_permlane32_swap((unsigned)w0, (unsigned)w0, false, false);
        const int pk = (int)(h ? sk[0] : sk[1]), pm = (int)(h ? sm[0] : sm[1]), p0 = (int)(h ? s0[0] : s0[1]);
        const int thr = min(min(wk, pk), max(wm, pm));
        const int kmin = min(w0, p0);
        const float dmin = kmin < H2_REMOVED ? __builtin_amdgcn_sqrtf(key_d2(kmin)) : INFINITY;
        const float cut = dmin + cutw;                      // beyond this the softmax term is < e^-20 of the largest
        const float ck = fmaf((cut * cut) * 1.000001f, icf, bias);
        cut_i = ck < INFINITY ? __float_as_int(ck) + 32 : 0x7fffffff;   // (+32: the embedded register number, rounding of ck)
        lim = max(thr, cut_i);
        stamp_after(4, lim);
    };

    stage_tile(0, 0);
    __syncthreads();  // (drains the DMA: vmcnt(0))
    if (PIPE == 0) {
        auto tile = [&](int t, int buf) __attribute__((always_inline)) {
            if (t + 1 < ntiles) stage_tile(t + 1, buf ^ 1);  // the other buffer was last read before the previous barrier
            stamp(0);
            const f32x16 a0 = chain(buf, 0);
            stamp_after(1, __float_as_int(a0[0]));
            epilogue(a0, t * HB_KT + 4 * h, false);
            if (t < 8) update_bound();
            const f32x16 a1 = chain(buf, 1);
            stamp_after(1, __float_as_int(a1[0]));
            epilogue(a1, t * HB_KT + 32 + 4 * h, false);
            update_bound();
            __syncthreads();
            stamp(5);
            T[7] += 2;
        };
        for (int t = 0; t < ntiles; t += 2) {
            tile(t, 0);
            if (t + 1 < ntiles) tile(t + 1, 1);
        }
    } else {
        // software pipeline: the matrix chain of the next sub-tile is issued ahead of the epilogue of the current one
        f32x16 a0 = chain(0, 0);
        auto tile = [&](int t, int buf, bool last) __attribute__((always_inline)) {   // buf, last: literals after inlining
            if (!last) stage_tile(t + 1, buf ^ 1);  // last read (second sub-tile of tile t - 1) before the previous barrier
            stamp(0);
            const f32x16 a1 = chain(buf, 1);
            epilogue(a0, t * HB_KT + 4 * h, PIPE == 2);
            if (t < 8) update_bound();
            __syncthreads();
            stamp(5);
            if (!last) a0 = chain(buf ^ 1, 0);
            epilogue(a1, t * HB_KT + 32 + 4 * h, PIPE == 2 && !last);
            update_bound();
            T[7] += 2;
        };
        // (the last tile is peeled off so that inside the loop the next chain is unconditional: a branch between it and
        // the epilogue would put them into different scheduling regions)
        int t = 0;
        for (; t + 2 < ntiles; t += 2) {
            tile(t, 0, false);
            tile(t + 1, 1, false);
        }
        if (t + 1 < ntiles) {
            tile(t, 0, false);
            tile(t + 1, 1, true);
        } else {
            tile(t, 0, true);
        }
    }

    // merge the two half-lanes that share a query (lane, lane^32)
    {
        const float co = __shfl_xor(cref, 32, 64), lo = __shfl_xor(l, 32, 64);
        const float cm = fmaxf(cref, co);
        const float a = (cref == -INFINITY) ? 0.f : l * exp2f((cref - cm) * LOG2E);
        const float bb = (co == -INFINITY) ? 0.f : lo * exp2f((co - cm) * LOG2E);
        l = a + bb;
        cref = cm;
        double other[HB_KC];
#pragma unroll
        for (int t = 0; t < HB_KC; ++t)
            other[t] = __hiloint2double(__shfl_xor(__double2hiint(kb.e[t]), 32, 64), __shfl_xor(__double2loint(kb.e[t]), 32, 64));
#pragma unroll
        for (int t = 0; t < HB_KC; ++t) add_term(__double2hiint(kb.insert(other[t])));  // dropped from the union
    }
    if (h == 0 && qrow < N) {
        const size_t row = (size_t)b * N + qrow;
#pragma unroll
        for (int t = 0; t < HB_KC; ++t) {
            const int key = __double2hiint(kb.e[t]), r = key & 15;
            const bool live = key < H2_REMOVED;
            G.cidx[row * HB_KC + t] = live ? __double2loint(kb.e[t]) + (r & 3) + 8 * (r >> 2) : 0x7fffffff;
            G.cd2[row * HB_KC + t] = live ? key_d2(key) : INFINITY;
        }
        G.lsum[row * 2] = l;
        G.lsum[row * 2 + 1] = cref;
    }
    if (STAMP) {
        T[6] = __builtin_amdgcn_s_memtime() - tstart;
        if (lane == 0 && args.stamps)
            for (int i = 0; i < 8; ++i) args.stamps[((size_t)blockIdx.x * HB_WAVES + wave) * 8 + i] = T[i];
    }
}

}  // namespace

void launch_norm_frags(const float *nrm, int B, int M, int Mpad, const int *amax, char *out, hipStream_t s) {
    hipLaunchKernelGGL(norm_frags_kernel, dim3((Mpad + 255) / 256, B), dim3(256), 0, s, nrm, M, Mpad, amax, out);
}

// pass A for the groups in `a` (lean semantics), second form; knf = key-side norm fragments of either group
void launch_sweep2(const HBArgs &a, const char *knf0, const char *knf1, const int *amaxc, int blocks, int form, hipStream_t s) {
    H2Args b;
    for (int g = 0; g < 2; ++g) {
        const HBGroup &G = a.g[g];
        b.g[g] = H2Group{G.qp, G.kp, g == 0 ? knf0 : knf1, G.nq, G.N, G.M, G.Mpad, G.tiles, G.cidx, G.cd2, G.lsum};
    }
    b.amax = amaxc;
    b.blocks0 = a.blocks0;
    b.neg_alpha = a.neg_alpha;
    b.cutw = a.cutw;
    b.stamps = nullptr;
    static const bool stamps_on = getenv("DVM_K1_STAMPS") != nullptr;
    if (stamps_on) {   // diagnostic: synchronous, allocates — never taken in production
        unsigned long long *dbuf = nullptr;
        const size_t n = (size_t)blocks * HB_WAVES * 8;
        if (hipMalloc(&dbuf, n * sizeof(unsigned long long)) != hipSuccess) return;
        b.stamps = dbuf;
        if (form == 1) {
            ensure_dyn_lds((const void *)softcorr_sweep2_kernel<0, true>, H2_LDS_BYTES);
            hipLaunchKernelGGL((softcorr_sweep2_kernel<0, true>), dim3(blocks), dim3(HB_THREADS), H2_LDS_BYTES, s, b);
        } else {
            ensure_dyn_lds((const void *)softcorr_sweep2_kernel<1, true>, H2_LDS_BYTES);
            hipLaunchKernelGGL((softcorr_sweep2_kernel<1, true>), dim3(blocks), dim3(HB_THREADS), H2_LDS_BYTES, s, b);
        }
        (void)hipStreamSynchronize(s);
        unsigned long long *hbuf = (unsigned long long *)malloc(n * sizeof(unsigned long long));
        (void)hipMemcpy(hbuf, dbuf, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t w = 0; w < (size_t)blocks * HB_WAVES; ++w)
            for (int i = 0; i < 8; ++i) tot[i] += (double)hbuf[w * 8 + i];
        const double nw = (double)blocks * HB_WAVES, st = tot[7] / nw;
        fprintf(stderr, "K1 stamps (form %d, %d blocks): per wave and sub-tile, cycles: dma %.0f  chain %.0f  epilogue %.0f  slow %.0f  bound %.0f  "
                        "barrier %.0f  | whole sweep %.0f per sub-tile (%.0f sub-tiles per wave)\n", form, blocks, tot[0] / nw / st, tot[1] / nw / st,
                tot[2] / nw / st, tot[3] / nw / st, tot[4] / nw / st, tot[5] / nw / st, tot[6] / nw / st, st);
        free(hbuf);
        (void)hipFree(dbuf);
        return;
    }
    if (form == 1) {
        ensure_dyn_lds((const void *)softcorr_sweep2_kernel<0>, H2_LDS_BYTES);
        hipLaunchKernelGGL((softcorr_sweep2_kernel<0>), dim3(blocks), dim3(HB_THREADS), H2_LDS_BYTES, s, b);
    } else if (form == 3) {
        ensure_dyn_lds((const void *)softcorr_sweep2_kernel<2>, H2_LDS_BYTES);
        hipLaunchKernelGGL((softcorr_sweep2_kernel<2>), dim3(blocks), dim3(HB_THREADS), H2_LDS_BYTES, s, b);
    } else {
        ensure_dyn_lds((const void *)softcorr_sweep2_kernel<1>, H2_LDS_BYTES);
        hipLaunchKernelGGL((softcorr_sweep2_kernel<1>), dim3(blocks), dim3(HB_THREADS), H2_LDS_BYTES, s, b);
    }
}

}  // namespace k1
}  // namespace dvm
