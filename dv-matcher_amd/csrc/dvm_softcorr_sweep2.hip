// dvm_softcorr_sweep2.hip — pass A of the soft-correspondence kernel (K1), second form.
// Built with -fno-honor-nans (Makefile): no NaN is ever formed here (keys are unsigned bit patterns, list entries finite
// positive doubles), and it lets fmin / fmax on the packed (key, column) doubles lower to bare v_min_f64 / v_max_f64 that
// the instruction scheduler can place (an asm statement is opaque to it).
// (reference: models/loss.py:110-114, 1339-1347, 1404-1407)
#include <stdlib.h>

#define DVM_K1_BUILTIN_MINMAX 1
#include "dvm_softcorr_f16.h"

namespace dvm {
namespace k1 {
namespace {

// ---------------------------------------------------------------- pass A, second form
// Measured on the first form (s_memtime stamps per phase, profiles/r3_k1_stamps.txt): of a wave's ~4 700 cycles per
// 32 x 32 sub-tile only ~800 are the 24 matrix instructions; the rest is the per-entry arithmetic, a wave-level insertion
// loop that runs max-over-lanes(#flagged) times for a handful of active lanes, the barrier at which the 8 waves of the
// workgroup wait for the one with the most trips, and the issue of the LDS-DMA pieces.  This form:
//  * norms on the matrix pipe: one more matrix instruction per sub-tile whose 16 k-slots carry |k|^2 and |q|^2, each as
//    three exact fp16 pieces times power-of-two constants, so that the accumulator IS the scaled squared distance plus
//    row constants,  acc = (|q|^2 (1 + 2^-13) + |k|^2 - 2 q.k) s^2 / 2 + 4.5  (one scale s for both sides).  The row bias
//    keeps acc >= 4 whatever the rounding, and s is such that acc < 2^33: every accumulator lies in the 30 binades above 4;
//  * exact keys: key = ((bits(acc) - bits(4.0f)) << 4) + r — the float's 5 significant exponent bits, its whole mantissa and
//    the accumulator register number r in 32 bits, ONE v_lshl_add_u32 per entry; keys are unique, order like the
//    accumulators, and give the accumulator back bit for bit (nothing of the approximate distance is lost);
//  * a fixed selection network per lane: sorted three smallest of its 16 keys in 46 three-input unsigned min / med / max
//    instructions — no comparison against a threshold per entry, no mask, no LDS staging;
//  * the two smallest are inserted into the sorted list of 12 unconditionally (packed (key, column) doubles as in the first
//    form); the THIRD is only recorded (16 bits, rounded down, one LDS slot per lane and sub-tile);
//  * no data-dependent loop in the sweep: after the last tile, a sub-tile is RE-DONE (fragments straight from global
//    memory, the same matrix chain, then a loop over the entries at or below the bound) only if some lane's recorded third
//    key lies at or below that lane's FINAL bound — 1-2 % of the sub-tiles on unstructured data, because the final bound is
//    the tightest one.  Every wave of a workgroup does the same work per tile: nothing waits at the barrier for a straggler.
// Same outputs as the first form (candidate columns, approximate squared distances, partial softmax sums), same
// guarantees: a column that belongs to a row's 12 smallest is never lost (an entry at or below the row's final bound is one
// of the two smallest of its sub-tile, or its sub-tile is re-done), the softmax sum covers every column outside the final
// list that lies within the cut (the cut never exceeds the bound).
// WAVES = 8: one workgroup of 256 query rows per compute unit (133 KB of LDS).  WAVES = 4: 128 query rows, 16 records per lane
// (one per FOUR sub-tiles at M = 2048), 77 KB — TWO workgroups per compute unit, so that the two waves of a SIMD belong to
// different workgroups with their own barriers: one's LDS-DMA issue and barrier wait fall into the other's matrix chain instead
// of both waves of a SIMD doing the same thing at the same time (DESIGN §8.1 b).  Price: every key tile serves half the rows —
// twice the LDS-DMA pieces per wave — and a flagged record re-does four sub-tiles.
template <int WAVES> constexpr int h2_nrec() { return WAVES == 8 ? 64 : 16; }                             // third-key records per lane
template <int WAVES> constexpr int h2_lds_bytes() { return 2 * HB_KT * HB_ROWB + 2 * HB_KT * 32 + 1024 + 64 * WAVES * h2_nrec<WAVES>() * 2; }   // key tiles + norm fragments + DMA dump + records
constexpr unsigned H2_REMOVED = 0xffc00000u;   // keys >= this: removed / invalid (as a list entry: hi word 0x7fe00000, a finite double)
constexpr unsigned H2_KBASE = 129u << 23;      // bits(4.0f): bottom of the key window
constexpr float H2_FLOOR = 4.5f;               // added to every accumulator through the norm instruction

struct H2Group {
    const char *qp, *kp;      // planes of the query / key side [B][rows][512]
    const char *knf;          // key-side norm fragments [B][Mpad][32 B]: fp16 {a1, a2, a3, 2^15, 2^4, 2^-7, 0, 0 | 0 x 8}
    const float *nq;          // |q|^2 (ATen order)
    int N, M, Mpad, tiles;
    int32_t *cidx;
    float *cd2, *lsum;
    int kslices, Ms;          // see HBGroup
};
struct H2Args {
    H2Group g[2];
    const int *amax;          // bit pattern of max |x| over BOTH sides (the common scale)
    int blocks0;
    float neg_alpha, cutw;
    const int *route;         // see HBArgs
    int nb;
    unsigned long long *stamps;   // diagnostic build only (DVM_K1_STAMPS): [block][wave][8] cycle totals per phase
};

// three-input unsigned min / max / median, written so that instruction selection forms v_min3_u32 / v_max3_u32 /
// v_med3_u32 (plain expressions, not asm statements: the instruction scheduler has to see them as vector instructions)
__device__ __forceinline__ unsigned umin3(unsigned a, unsigned b, unsigned c) { return min(min(a, b), c); }
__device__ __forceinline__ unsigned umax3(unsigned a, unsigned b, unsigned c) { return max(max(a, b), c); }
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) { return max(min(a, b), min(max(a, b), c)); }

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

// key-side norm fragments, padded to whole key tiles (padding keys: zero norm; their entries are masked by column in the
// sweep): out [B][Mpad][16 fp16]
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
    }
    char *p = out + ((size_t)b * Mpad + i) * 32;
    *(f16x8 *)p = lo;
    *(f16x8 *)(p + 16) = hi;
}

// sorted (s0 <= s1 <= s2) three smallest of 16 distinct keys
struct Top3 {
    unsigned s0, s1, s2;
};
__device__ __forceinline__ Top3 sort3(unsigned a, unsigned b, unsigned c) { return Top3{umin3(a, b, c), umed3(a, b, c), umax3(a, b, c)}; }
__device__ __forceinline__ Top3 merge3(const Top3 &a, const Top3 &b) {
    Top3 c;
    const unsigned m00 = max(a.s0, b.s0);
    c.s0 = min(a.s0, b.s0);
    c.s1 = umin3(m00, a.s1, b.s1);
    c.s2 = min(umin3(a.s2, b.s2, max(a.s1, b.s0)), max(a.s0, b.s1));
    return c;
}
__device__ __forceinline__ Top3 top3_of_16(const unsigned (&v)[16]) {
    Top3 t = merge3(merge3(sort3(v[0], v[1], v[2]), sort3(v[3], v[4], v[5])),
                    merge3(merge3(sort3(v[6], v[7], v[8]), sort3(v[9], v[10], v[11])), sort3(v[12], v[13], v[14])));
    const unsigned x = v[15];
    return Top3{min(t.s0, x), umed3(t.s0, t.s1, x), umed3(t.s1, t.s2, x)};
}

// list entry of a key and its sub-tile's base column: hi = key >> 1, lo = (key's low bit << 31) | jb — a positive finite
// double whose order is (key, jb)
__device__ __forceinline__ double pack_entry(unsigned key, int jb) { return __hiloint2double((int)(key >> 1), (int)((key << 31) | (unsigned)jb)); }
__device__ __forceinline__ unsigned entry_key(double e) { return ((unsigned)__double2hiint(e) << 1) | ((unsigned)__double2loint(e) >> 31); }
__device__ __forceinline__ int entry_jb(double e) { return __double2loint(e) & 0x7fffffff; }

// STAMP: diagnostic build — every wave adds up the shader cycles (s_memtime) it spends per phase and writes the totals to
// args.stamps; no output value depends on them.  Phases: 0 LDS-DMA issue, 1 matrix chain (fragment reads, waits, 25 matrix
// instructions, until the accumulator is readable), 2 epilogue (selection + insertions), 3 softmax terms + re-done
// sub-tiles, 4 bound update, 5 barrier (incl. the wait for the wave's own DMA pieces), 6 whole sweep, 7 sub-tiles.
template <int PIPE, bool STAMP = false, int WAVES = 8>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 2 : 1) void softcorr_sweep2_kernel(const H2Args args) {
    constexpr int H2_NREC = h2_nrec<WAVES>(), HB_QB = 32 * WAVES, HB_GLDS_PER_WAVE = HB_KT * HB_ROWB / 1024 / WAVES, HB_WAVES = WAVES;   // (shadow the 8-wave constants)
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
    char *const dump0 = knf0 + 2 * HB_KT * 32;                            // 1 KiB nobody reads (see stage_tile)
    unsigned short *const rec0 = (unsigned short *)(dump0 + 1024);        // [wave][H2_NREC][64 lanes]

    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int grp = lid >= args.blocks0 ? 1 : 0;
    lid -= grp ? args.blocks0 : 0;
    const H2Group &G = args.g[grp];
    const int N = G.N;
    const int bs = lid / G.tiles, qt = lid % G.tiles;                 // launch entry = (batch entry, key slice)
    const int b = bs / G.kslices, sl = bs - b * G.kslices;            // (one slice: bs == b, sl == 0)
    const int M = G.kslices > 1 ? min(G.Ms, G.M - sl * G.Ms) : G.M;   // keys swept by this workgroup
    if (args.route && args.route[grp * args.nb + b] != K1_ROUTE_SECOND) return;   // this pair goes through the first form
    const float neg_alpha = args.neg_alpha;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave: a scalar register)
    const int r32 = lane & 31, h = lane >> 5;

    const char *kbase = G.kp + ((size_t)b * G.M + (size_t)sl * G.Ms) * HB_ROWB;
    const char *nfbase = G.knf + ((size_t)b * G.Mpad + (size_t)sl * G.Ms) * 32;
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
    // the query's norm, biased by 2^-13 of itself (a rounding error can only pull an accumulator below its row constants
    // where |q| ~ |k|, and there the bias is twice the error bound), plus the floor: the B operand of the norm instruction
    const float nas = norm_scaled(G.nq[(size_t)b * N + qrc], se);
    const float nasb = (nas + nas * 0x1p-13f) + H2_FLOOR;
    const float rowc = nasb - nas;                  // what a key's accumulator carries on top of the scaled squared distance (exact)
    f16x8 qn = {0, 0, 0, 0, 0, 0, 0, 0};
    if (h == 0) {
        _Float16 p1, p2, p3;
        norm_pieces(nasb, p1, p2, p3);
        qn[0] = (_Float16)0x1p+15f, qn[1] = (_Float16)0x1p+4f, qn[2] = (_Float16)0x1p-7f;
        qn[3] = p1, qn[4] = p2, qn[5] = p3;
    }

    PackedBest<HB_KC> kb;
#pragma unroll
    for (int t = 0; t < HB_KC; ++t) kb.e[t] = __hiloint2double((int)(H2_REMOVED >> 1), 0x7fffffff);
    float cref = -INFINITY, l = 0.f;
    unsigned lim = 0xffffffffu, cut_k = 0xffffffffu;   // bound / cut in key space
    const float cutw = args.cutw;

    const int ntiles = (M + HB_KT - 1) / HB_KT, nsub = 2 * ntiles;
    const int rgrp = (nsub + H2_NREC - 1) / H2_NREC;    // sub-tiles per record (1 up to M = 2048)
    unsigned short *const rec = rec0 + (size_t)wave * H2_NREC * 64 + lane;
    unsigned urec = 0xffffffffu;                        // smallest third key of the current record's sub-tiles

    // Key tiles go global -> LDS by LDS-DMA (1 KiB = 2 rows per wave instruction, 4 per wave and tile).  The source address is
    // a wave-uniform tile base (scalar registers) plus a per-lane byte offset that does not change from tile to tile — row
    // r = 2 piece + h of the tile, 16-B chunk r32 ^ (r & 15): chunk c of an LDS row holds chunk c ^ (row & 15) of the key — so
    // staging a tile costs no vector arithmetic.  Only a ragged last tile clamps its rows (padding keys re-read the last row;
    // their entries are masked by column).
    // (The per-lane offsets are re-formed for every tile from the lane number, itself re-read from the execution mask
    // (v_mbcnt): a handful of vector instructions per piece.  Anything kept in a vector register across the tile for this is
    // spilled — the register file is full of fragments and lists — and a scratch reload next to the DMA issue waits,
    // through vmcnt, for every DMA piece still in flight: measured 1 300 - 2 100 cycles per sub-tile.)
    auto stage_tile = [&](int t, int buf, bool clamp) __attribute__((always_inline)) {
        int lane;   // (a volatile statement: otherwise the offsets are hoisted out of the loop as invariants — and spilled)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        const int r32 = lane & 31, h = lane >> 5;
        const int j0 = t * HB_KT;
        const char *tb = kbase + (size_t)j0 * HB_ROWB;   // wave-uniform
        char *kt = ktile0 + (size_t)buf * HB_KT * HB_ROWB;
#pragma unroll
        for (int e = 0; e < HB_GLDS_PER_WAVE; ++e) {
            const int piece = wave * HB_GLDS_PER_WAVE + e;
            const int r = 2 * piece + h, rc = clamp ? min(r, M - 1 - j0) : r;
            const unsigned off = rc * HB_ROWB + ((r32 ^ (r & 15)) << 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tb + off),
                                             (__attribute__((address_space(3))) void *)(kt + piece * 1024), 16, 0, 0);
        }
        // 64 keys x 32 B of norm fragments = two 1-KiB pieces, brought by waves 0 and 1; the other six waves issue the same
        // instruction into a 1-KiB dump area (destination chosen by a scalar select): every wave issues five pieces per tile, and
        // there is no branch here that would cut the scheduling region (an execution-masked DMA is branched around as well).
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nfbase + (size_t)j0 * 32 + (wave & 1) * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(wave < 2 ? knf0 + buf * HB_KT * 32 + wave * 1024 : dump0), 16, 0, 0);
    };

    // Workgroup barrier for LDS-DMA data: every wave first waits for ITS OWN pieces (vmcnt), then joins the barrier — the
    // compiler places its own vmcnt wait only in front of the wave's next LDS read, which orders nothing for the rows the
    // OTHER waves were to deliver (seen as a timing-dependent failure of the M = 5 edge case when it ran first in a process).
    auto dma_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    auto key_d2 = [&](unsigned key) __attribute__((always_inline)) {   // key -> squared distance
        return fmaxf(__uint_as_float((key >> 4) + H2_KBASE) - rowc, 0.f) * cf;
    };
    auto add_term = [&](unsigned key) __attribute__((always_inline)) {  // l += exp(s - cref) for a live key (otherwise nothing)
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
        // Fragments in four batches of 2 k-steps (16 VGPRs each), two batches in flight: the scheduler, left alone, requests
        // all 17 reads of a sub-tile at once (68 VGPRs) next to the previous sub-tile's epilogue — and what does not fit is
        // spilled.  The barriers below let vector, scalar and matrix instructions cross, LDS reads not.
        f16x8 ah[4][2], am[4][2];
        auto reads = [&](int q) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                ah[q][u] = lds16(fadr[2 * q + u], toff);
                am[q][u] = lds16(fadr[2 * q + u], toff + 256);
            }
        };
        auto products = [&](int q) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int s = 2 * q + u;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[q][u], qh[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][u], qm[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][u], qh[s], acc, 0, 0, 0);
            }
        };
        reads(0);
        reads(1);
        const f16x8 an = lds16(nadr, buf * (HB_KT * 32) + sub * (32 * 32));
        __builtin_amdgcn_sched_barrier(0x00E);
        products(0);
        reads(2);
        __builtin_amdgcn_sched_barrier(0x00E);
        products(1);
        reads(3);
        __builtin_amdgcn_sched_barrier(0x00E);
        products(2);
        products(3);
        // the norms last: every partial sum before it has the magnitude of q.k
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(an, qn, acc, 0, 0, 0);
    };

    // keys of the lane's 16 accumulators (register r holds local key (r & 3) + 8 (r >> 2) of the lane's half)
    auto make_keys = [&](const f32x16 &acc, unsigned (&v)[16], int jb, bool mask_pads) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[r] = (__float_as_uint(acc[r]) << 4) + ((unsigned)r - (H2_KBASE << 4));   // one v_lshl_add_u32 (constant in an SGPR)
            if (mask_pads) v[r] = jb + (r & 3) + 8 * (r >> 2) < M ? v[r] : (H2_REMOVED | r);
        }
    };
    // an entry that leaves the list (or fails to enter it) owes its softmax term if it lies within the cut
    auto terms2 = [&](double o0, double o1) __attribute__((always_inline)) {
        const unsigned k0 = entry_key(o0), k1 = entry_key(o1);
        if (__builtin_amdgcn_ballot_w64(min(k0, k1) <= cut_k) != 0) {
            add_term(k0 <= cut_k ? k0 : H2_REMOVED);
            add_term(k1 <= cut_k ? k1 : H2_REMOVED);
#pragma unroll
            for (int t = 0; t < HB_KC; ++t) asm volatile("" : "+v"(kb.e[t]));   // (the list is "used" on this path: see epilogue)
        }
    };

    // One sub-tile's 16 keys per lane, straight-line: keys, sorted three smallest, the two smallest into the list, the third
    // into the record.  The one wave-uniform branch behind it (softmax terms of entries within the cut that left a list:
    // nearly never taken at the alphas this sweep is used for) touches the list, so nothing of the straight-line part
    // can be sunk below it, away from the matrix instructions it is meant to run beside.
    auto pace = [&]() __attribute__((always_inline)) {
        // PIPE == 2: the scheduling region that ends here holds the matrix chain of the NEXT sub-tile (25 instructions, 17 LDS
        // reads) and this sub-tile's selection + insertions (~125 vector instructions): ask for one matrix instruction per
        // five vector instructions, the first batch of fragment reads up front and the second a quarter of the way in
        __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
#pragma unroll
        for (int i = 0; i < 25; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i == 5) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
    };
    auto epilogue = [&](const f32x16 &acc, int s, bool mask_pads, bool paced = false) __attribute__((always_inline)) {
        const int jb = s * 32 + 4 * h;
        unsigned v[16];
        make_keys(acc, v, jb, mask_pads);
        const Top3 w = top3_of_16(v);
        const double o0 = kb.insert(pack_entry(w.s0, jb));
        const double o1 = kb.insert(pack_entry(w.s1, jb));
        // (branch-free: a record is rewritten by every sub-tile of its group with the running minimum — one sub-tile per
        // record up to M = 2048; a branch here would cut the scheduling region between the selection and the insertions)
        urec = (s % rgrp == 0) ? w.s2 : min(urec, w.s2);
        rec[(s / rgrp) * 64] = (unsigned short)(urec >> 16);
        if (paced) pace();
        stamp_after(2, (int)(w.s2 ^ (unsigned)__double2hiint(kb.e[0])));
        // (Finishing a sub-tile right here when a lane's third key lies within the cut — the clustered regime, where most
        // sub-tiles end up re-done — was measured: 14.6 instead of 16.0 ms per 256 pairs on the "trained-like" set at alpha 33,
        // but 4.26 instead of 3.61 ms per launch on random features: the loop's registers and code sit in the hot epilogue.
        // Such inputs are routed to the first form instead, per pair, by the probe: see k1_probe_kernel.)
        terms2(o0, o1);
        stamp_after(3, __double2hiint(kb.e[0]));
    };
    // bound for the final check: the row's KC-th best is at most min(a_K, b_K, max(a_m, b_m)), m = KC/2, over the two
    // half-lanes (a, b) that share the row; everything within the cut counts as well (it owes a softmax term).  In the sweep
    // only the cut is used (the bound itself is applied once, at the end).
    auto update_bound = [&]() __attribute__((always_inline)) {
        const unsigned wk = entry_key(kb.e[HB_KC - 1]), wm = entry_key(kb.e[HB_KC / 2 - 1]), w0 = entry_key(kb.e[0]);
        const auto sk = __builtin_amdgcn_permlane32_swap(wk, wk, false, false);
        const auto sm = __builtin_amdgcn_permlane32_swap(wm, wm, false, false);
        const auto s0 = __builtin_amdgcn_permlane32_swap(w0, w0, false, false);
        const unsigned pk = h ? sk[0] : sk[1], pm = h ? sm[0] : sm[1], p0 = h ? s0[0] : s0[1];
        const unsigned thr = min(min(wk, pk), max(wm, pm));
        const unsigned kmin = min(w0, p0);
        const float dmin = kmin < H2_REMOVED ? __builtin_amdgcn_sqrtf(key_d2(kmin)) : INFINITY;
        const float cut = dmin + cutw;                      // beyond this the softmax term is < e^-20 of the largest
        const float ca = fmaf((cut * cut) * 1.000001f, icf, rowc);   // the cut as an accumulator value
        cut_k = ca < 0x1p+33f ? ((__float_as_uint(fmaxf(ca, 4.f)) - H2_KBASE) << 4) + 15u : 0xffffffffu;
        lim = max(thr, cut_k);
        stamp_after(4, (int)lim);
    };

    const bool ragged = (M & (HB_KT - 1)) != 0;   // the last tile holds padding keys
    // Tile t + 1 is requested at the top of tile t, into the buffer whose last reads lie before the previous barrier.
    // (Requesting tile t + 2 right after the barrier that frees its buffer — a tile and a half ahead instead of half a
    // tile — was measured slower, 4.10 vs 3.77 ms per launch: the barrier time does not come from waiting for the DMA.)
    auto stage = [&](int t, int buf) __attribute__((always_inline)) {
        if (t < ntiles) {
            if (ragged && t + 1 == ntiles) stage_tile(t, buf, true); else stage_tile(t, buf, false);
        }
    };
    if (PIPE != 3) {
        stage(0, 0);
        dma_barrier();  // (drains the DMA: vmcnt(0))
    }
    if (PIPE == 0) {
        auto tile = [&](int t, int buf) __attribute__((always_inline)) {
            stage(t + 1, buf ^ 1);
            stamp(0);
            const bool pads = ragged && t + 1 == ntiles;
            const f32x16 a0 = chain(buf, 0);
            stamp_after(1, __float_as_int(a0[0]));
            if (pads) epilogue(a0, 2 * t, true); else epilogue(a0, 2 * t, false);
            const f32x16 a1 = chain(buf, 1);
            stamp_after(1, __float_as_int(a1[0]));
            if (pads) epilogue(a1, 2 * t + 1, true); else epilogue(a1, 2 * t + 1, false);
            if ((t & 3) == 3 || t < 8) update_bound();   // (the cut moves slowly once the lists have filled)
            dma_barrier();
            stamp(5);
            T[7] += 2;
        };
        for (int t = 0; t < ntiles; t += 2) {
            tile(t, 0);
            if (t + 1 < ntiles) tile(t + 1, 1);
        }
    } else if (PIPE == 3) {
        // As PIPE 1, but tile t + 2 is requested in the SECOND half of tile t, behind the matrix chain of that half and in front
        // of the epilogue's vector tail (the barriers around it let vector and scalar instructions cross, matrix instructions
        // and LDS reads not): the DMA instructions then issue while no fragment reads compete for the LDS path, and a tile has
        // a tile and a half to land.  Branch-free: the tile index and the rows are clamped (a tile past the end re-stages the
        // last tile into a buffer nobody reads any more).
        stage(0, 0);
        stage(1, 1);
        dma_barrier();
        f32x16 a0 = chain(0, 0);
        auto tile = [&](int t, int buf, bool last) __attribute__((always_inline)) {   // buf, last: literals after inlining
            const f32x16 a1 = chain(buf, 1);
            if (last && ragged) epilogue(a0, 2 * t, true); else epilogue(a0, 2 * t, false);
            dma_barrier();
            stamp(5);
            if (!last) {
                a0 = chain(buf ^ 1, 0);
                __builtin_amdgcn_sched_barrier(0x006);
                stage_tile(min(t + 2, ntiles - 1), buf, true);
                __builtin_amdgcn_sched_barrier(0x006);
            }
            stamp(0);
            if (last && ragged) epilogue(a1, 2 * t + 1, true); else epilogue(a1, 2 * t + 1, false);
            if ((t & 3) == 3 || t < 8) update_bound();
            T[7] += 2;
        };
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
    } else {
        // software pipeline: the matrix chain of the next sub-tile is issued ahead of the epilogue of the current one
        f32x16 a0 = chain(0, 0);
        auto tile = [&](int t, int buf, bool last) __attribute__((always_inline)) {   // buf, last: literals after inlining
            if (!last) stage(t + 1, buf ^ 1);   // (last read — second sub-tile of tile t - 1 — before the previous barrier)
            stamp(0);
            const f32x16 a1 = chain(buf, 1);
            if (last && ragged) epilogue(a0, 2 * t, true); else epilogue(a0, 2 * t, false, PIPE == 2);
            dma_barrier();
            stamp(5);
            if (!last) a0 = chain(buf ^ 1, 0);
            if (last && ragged) epilogue(a1, 2 * t + 1, true); else epilogue(a1, 2 * t + 1, false, PIPE == 2 && !last);
            if ((t & 3) == 3 || t < 8) update_bound();   // (the cut moves slowly once the lists have filled)
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

    // ---- re-do the sub-tiles in which some lane may hold MORE than two entries at or below its final bound: the same
    // fragments (straight from global memory), the same chain, hence the same keys; the two smallest are in the list
    // already, the rest goes through a loop that takes two per trip while any lane has one left at or below the bound
    update_bound();
    {
        const unsigned limr = lim >> 16;
        const int nrec = (nsub + rgrp - 1) / rgrp;
        unsigned long long todo = 0;   // wave-uniform: records with a lane at or below its bound (64 independent LDS reads)
#pragma unroll
        for (int g = 0; g < H2_NREC; ++g)
            if (g < nrec && __builtin_amdgcn_ballot_w64((unsigned)rec[g * 64] <= limr) != 0) todo |= 1ull << g;
        while (todo != 0) {
            const int g = __builtin_ctzll(todo);
            todo &= todo - 1;
            if (STAMP) T[7] += 1u << 20;   // (re-done records, counted in the high bits)
            for (int s = g * rgrp; s < (g + 1) * rgrp && s < nsub; ++s) {
                const int j = s * 32 + r32, jc = j < M ? j : M - 1;
                const char *arow = kbase + (size_t)jc * HB_ROWB + 16 * h;
                const f16x8 an = *(const f16x8 *)(nfbase + (size_t)j * 32 + 16 * h);
                f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u) {   // the sweep's instruction order: (m, h), (h, m), (h, h) per k-step, the norms last
                    const f16x8 ah = *(const f16x8 *)(arow + 32 * u), am = *(const f16x8 *)(arow + 256 + 32 * u);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, qh[u], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qm[u], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[u], acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, qn, acc, 0, 0, 0);
                const int jb = s * 32 + 4 * h;
                unsigned v[16];
                make_keys(acc, v, jb, true);
                Top3 w = top3_of_16(v);   // w.s0, w.s1: inserted by the sweep
                bool more = w.s2 <= lim && w.s2 < H2_REMOVED;
                while (__builtin_amdgcn_ballot_w64(more) != 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = v[r] <= w.s1 ? (H2_REMOVED | r) : v[r];
                    w = top3_of_16(v);
                    const double o0 = kb.insert(pack_entry(w.s0, jb));
                    const double o1 = kb.insert(pack_entry(w.s1, jb));
                    terms2(o0, o1);
                    more = w.s2 <= lim && w.s2 < H2_REMOVED;
                }
            }
        }
        stamp(3);
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
        for (int t = 0; t < HB_KC; ++t) add_term(entry_key(kb.insert(other[t])));  // dropped from the union
    }
    if (h == 0 && qrow < N) {
        const size_t row = (size_t)bs * N + qrow;
#pragma unroll
        for (int t = 0; t < HB_KC; ++t) {
            const unsigned key = entry_key(kb.e[t]), r = key & 15;
            const bool live = key < H2_REMOVED;
            G.cidx[row * HB_KC + t] = live ? entry_jb(kb.e[t]) + (int)((r & 3) + 8 * (r >> 2)) : 0x7fffffff;
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

// ---------------------------------------------------------------- pass A, third form: split roles (round 4, DVM_K1_SWEEP=5)
// Same tiles, keys, lists, records and guarantees as the second form; what changes is WHO does what.  Measured on the second form
// (profiles/r4_k1_waves.txt): a wave alone on its SIMD needs 2 301 cycles per 32 x 32 sub-tile, two symmetric waves 3 140 each —
// every phase of a wave stretches when its partner issues anything, and the LDS-DMA instructions (~200 cycles of the issuing
// wave each) and the list bookkeeping stand in the same instruction stream as the matrix chain.  Here a SIMD holds
//  * two PRODUCER waves (waves 0-7, 32 query rows each as before): fragments -> matrix chain -> keys -> sorted three smallest of
//    16 -> 3 keys per lane and sub-tile into an LDS hand-off buffer, and nothing else;
//  * one CONSUMER wave (waves 8-11, two row blocks each): ALL LDS-DMA pieces of the workgroup, the lists, third-key records,
//    softmax terms and bounds, working one TILE behind the producers (hand-off buffers alternate with the key tiles; the one
//    barrier per tile covers both), and, after the last tile, the re-done sub-tiles and the outputs.
// 12 waves x <= 168 registers (a producer carries no lists, a consumer no query fragments until the end).  (64 rows per producer with
// two chains per key fragment was tried first: 128 registers of query fragments + two accumulators + fragments in flight spill
// into the matrix loop.)
// STAMP (DVM_K1_STAMPS): per wave, cycles waiting at the barriers (slot 5), in the matrix chains / in the hand-off processing (1),
// in the selection / the DMA issue (2), whole kernel (6), tiles (7): tells which role the workgroup waits for.
constexpr int H3_THREADS = 768;
constexpr int H3_HAND_WORDS = 2 * 2 * 8 * 3 * 64;                          // [buf][sub][row block][key][lane]
constexpr int H3_LDS_BYTES = 2 * HB_KT * HB_ROWB + 2 * HB_KT * 32 + 1024 + H3_HAND_WORDS * 4 + 4 * 2 * 64 * 64 * 2;   // 160 768 B
static_assert(H3_LDS_BYTES <= 160 * 1024, "third sweep form: LDS");

struct H3Row {                       // a consumer lane's state for one of its two row blocks
    PackedBest<HB_KC> kb;
    float cref, l, rowc;
    unsigned lim, cut_k, urec;
};

template <bool STAMP = false>
__global__ __launch_bounds__(H3_THREADS) void softcorr_sweep3_kernel(const H2Args args) {
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
    char *const ktile0 = smem_b;                                         // [2][HB_KT][512], 16-B chunks XOR-swizzled
    char *const knf0 = smem_b + (size_t)2 * HB_KT * HB_ROWB;             // [2][HB_KT][32]
    char *const dump0 = knf0 + 2 * HB_KT * 32;                            // 1 KiB nobody reads
    unsigned *const hand0 = (unsigned *)(dump0 + 1024);                   // hand-off: [buf][sub][row block][key][lane]
    unsigned short *const rec0 = (unsigned short *)(hand0 + H3_HAND_WORDS);   // [consumer][rb][64 records][64 lanes]

    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int grp = lid >= args.blocks0 ? 1 : 0;
    lid -= grp ? args.blocks0 : 0;
    const H2Group &G = args.g[grp];
    const int N = G.N;
    const int bs = lid / G.tiles, qt = lid % G.tiles;
    const int b = bs / G.kslices, sl = bs - b * G.kslices;
    const int M = G.kslices > 1 ? min(G.Ms, G.M - sl * G.Ms) : G.M;
    if (args.route && args.route[grp * args.nb + b] != K1_ROUTE_SECOND) return;
    const float neg_alpha = args.neg_alpha;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    const bool producer = wave < 8;
    const int pw = wave & 7;      // producer: its row block; consumer (waves 8-11 -> 0-3): row blocks 2 pw and 2 pw + 1
    const char *kbase = G.kp + ((size_t)b * G.M + (size_t)sl * G.Ms) * HB_ROWB;
    const char *nfbase = G.knf + ((size_t)b * G.Mpad + (size_t)sl * G.Ms) * 32;
    const int se = scale_exp(*args.amax);
    const float cf = pow2i(1 - 2 * se), icf = pow2i(2 * se - 1);
    const float cutw = args.cutw;
    const int ntiles = (M + HB_KT - 1) / HB_KT, nsub = 2 * ntiles;
    const int rgrp = (nsub + 63) / 64;
    const bool ragged = (M & (HB_KT - 1)) != 0;
    auto dma_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // the hand-off slot of (buffer, sub-tile, row block, key) for this lane
    auto hand = [&](int buf, int sub, int blk, int key) __attribute__((always_inline)) {
        return hand0 + (((buf * 2 + sub) * 8 + blk) * 3 + key) * 64 + lane;
    };
    auto finish_stamps = [&]() __attribute__((always_inline)) {
        if (STAMP) {
            T[6] = __builtin_amdgcn_s_memtime() - tstart;
            if (lane == 0 && args.stamps)
                for (int i = 0; i < 8; ++i) args.stamps[((size_t)blockIdx.x * 12 + wave) * 8 + i] = T[i];
        }
    };

    if (producer) {
        // ---------------------------------------------------------------- producer
        const int qrow = qt * HB_QB + pw * 32 + r32;
        const int qrc = qrow < N ? qrow : N - 1;
        const char *qptr = G.qp + ((size_t)b * N + qrc) * HB_ROWB + 16 * h;
        f16x8 qh[8], qm[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            qh[s] = -*(const f16x8 *)(qptr + 32 * s);
            qm[s] = -*(const f16x8 *)(qptr + 256 + 32 * s);
        }
        f16x8 qn = {0, 0, 0, 0, 0, 0, 0, 0};
        if (h == 0) {
            const float nas = norm_scaled(G.nq[(size_t)b * N + qrc], se);
            const float nasb = (nas + nas * 0x1p-13f) + H2_FLOOR;
            _Float16 p1, p2, p3;
            norm_pieces(nasb, p1, p2, p3);
            qn[0] = (_Float16)0x1p+15f, qn[1] = (_Float16)0x1p+4f, qn[2] = (_Float16)0x1p-7f;
            qn[3] = p1, qn[4] = p2, qn[5] = p3;
        }
        unsigned fadr[8];
        {
            const unsigned rowb = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)ktile0 + r32 * HB_ROWB;
            const int tq = h ^ (r32 & 15);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                fadr[s] = rowb + (((2 * s) ^ tq) << 4);
                asm volatile("" : "+v"(fadr[s]));
            }
        }
        unsigned nadr = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)knf0 + r32 * 32 + 16 * h;
        asm volatile("" : "+v"(nadr));
        auto lds16 = [](unsigned adr, int off) __attribute__((always_inline)) {
            return *(const f16x8 *)(const __attribute__((address_space(3))) char *)(size_t)(adr + off);
        };
        auto chain = [&](int buf, int sub) __attribute__((always_inline)) -> f32x16 {   // as the second form's
            const int toff = buf * (HB_KT * HB_ROWB) + sub * (32 * HB_ROWB);
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f16x8 ah[4][2], am[4][2];
            auto reads = [&](int q) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    ah[q][u] = lds16(fadr[2 * q + u], toff);
                    am[q][u] = lds16(fadr[2 * q + u], toff + 256);
                }
            };
            auto products = [&](int q) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int s = 2 * q + u;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[q][u], qh[s], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][u], qm[s], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q][u], qh[s], acc, 0, 0, 0);
                }
            };
            reads(0);
            reads(1);
            const f16x8 an = lds16(nadr, buf * (HB_KT * 32) + sub * (32 * 32));
            __builtin_amdgcn_sched_barrier(0x00E);
            products(0);
            reads(2);
            __builtin_amdgcn_sched_barrier(0x00E);
            products(1);
            reads(3);
            __builtin_amdgcn_sched_barrier(0x00E);
            products(2);
            products(3);
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(an, qn, acc, 0, 0, 0);
        };
        // keys of an accumulator, their sorted three smallest into the hand-off buffer
        auto emit = [&](const f32x16 &acc, int t, int buf, int sub, bool pads) __attribute__((always_inline)) {
            const int s = 2 * t + sub, jb = s * 32 + 4 * h;
            unsigned v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = (__float_as_uint(acc[r]) << 4) + ((unsigned)r - (H2_KBASE << 4));
                if (pads) v[r] = jb + (r & 3) + 8 * (r >> 2) < M ? v[r] : (H2_REMOVED | r);
            }
            const Top3 w = top3_of_16(v);
            *hand(buf, sub, pw, 0) = w.s0;
            *hand(buf, sub, pw, 1) = w.s1;
            *hand(buf, sub, pw, 2) = w.s2;
        };
        // (no software pipeline inside the wave: the selection is ~65 vector instructions, and the SIMD's other producer keeps the
        // matrix pipe busy meanwhile; a second accumulator in flight does not fit 168 registers)
        dma_barrier();                      // tile 0 has landed (staged by the consumers)
        stamp(5);
        auto tile = [&](int t, int buf) __attribute__((always_inline)) {   // buf: literal after inlining
            const bool pads = ragged && t + 1 == ntiles;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const f32x16 a = chain(buf, sub);
                if (STAMP) {
                    const int x = __builtin_amdgcn_readfirstlane(__float_as_int(a[0]));
                    asm volatile("" ::"s"(x));
                    stamp(1);
                }
                if (pads) emit(a, t, buf, sub, true); else emit(a, t, buf, sub, false);
                stamp(2);
            }
            dma_barrier();
            stamp(5);
            T[7] += 1;
        };
        for (int t = 0; t < ntiles; t += 2) {
            tile(t, 0);
            if (t + 1 < ntiles) tile(t + 1, 1);
        }
        finish_stamps();
        return;
    }

    // ---------------------------------------------------------------- consumer
    H3Row R[2];
    int qrowv[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int qrow = qt * HB_QB + (2 * pw + rb) * 32 + r32;
        const int qrc = qrow < N ? qrow : N - 1;
        qrowv[rb] = qrow;
        const float nas = norm_scaled(G.nq[(size_t)b * N + qrc], se);
        const float nasb = (nas + nas * 0x1p-13f) + H2_FLOOR;
        R[rb].rowc = nasb - nas;
#pragma unroll
        for (int t = 0; t < HB_KC; ++t) R[rb].kb.e[t] = __hiloint2double((int)(H2_REMOVED >> 1), 0x7fffffff);
        R[rb].cref = -INFINITY, R[rb].l = 0.f, R[rb].lim = 0xffffffffu, R[rb].cut_k = 0xffffffffu, R[rb].urec = 0xffffffffu;
    }
    auto key_d2 = [&](const H3Row &S, unsigned key) __attribute__((always_inline)) {
        return fmaxf(__uint_as_float((key >> 4) + H2_KBASE) - S.rowc, 0.f) * cf;
    };
    auto add_term = [&](H3Row &S, unsigned key) __attribute__((always_inline)) {
        const bool live = key < H2_REMOVED;
        const float s = __builtin_amdgcn_sqrtf(key_d2(S, key)) * neg_alpha;
        const float cnew = live ? fmaxf(S.cref, s) : S.cref;
        const float sc = (cnew == S.cref) ? 1.f : __builtin_amdgcn_exp2f((S.cref - cnew) * LOG2E);
        const float term = live ? __builtin_amdgcn_exp2f((s - cnew) * LOG2E) : 0.f;
        S.l = S.l * sc + term;
        S.cref = cnew;
    };
    auto terms2 = [&](H3Row &S, double o0, double o1) __attribute__((always_inline)) {
        const unsigned k0 = entry_key(o0), k1 = entry_key(o1);
        if (__builtin_amdgcn_ballot_w64(min(k0, k1) <= S.cut_k) != 0) {
            add_term(S, k0 <= S.cut_k ? k0 : H2_REMOVED);
            add_term(S, k1 <= S.cut_k ? k1 : H2_REMOVED);
        }
    };
    auto update_bound = [&](H3Row &S) __attribute__((always_inline)) {
        const unsigned wk = entry_key(S.kb.e[HB_KC - 1]), wm = entry_key(S.kb.e[HB_KC / 2 - 1]), w0 = entry_key(S.kb.e[0]);
        const auto sk = __builtin_amdgcn_permlane32_swap(wk, wk, false, false);
        const auto sm = __builtin_amdgcn_permlane32_swap(wm, wm, false, false);
        const auto s0 = __builtin_amdgcn_permlane32_swap(w0, w0, false, false);
        const unsigned pk = h ? sk[0] : sk[1], pm = h ? sm[0] : sm[1], p0 = h ? s0[0] : s0[1];
        const unsigned thr = min(min(wk, pk), max(wm, pm));
        const unsigned kmin = min(w0, p0);
        const float dmin = kmin < H2_REMOVED ? __builtin_amdgcn_sqrtf(key_d2(S, kmin)) : INFINITY;
        const float cut = dmin + cutw;
        const float ca = fmaf((cut * cut) * 1.000001f, icf, S.rowc);
        S.cut_k = ca < 0x1p+33f ? ((__float_as_uint(fmaxf(ca, 4.f)) - H2_KBASE) << 4) + 15u : 0xffffffffu;
        S.lim = max(thr, S.cut_k);
    };
    unsigned short *const recw = rec0 + (size_t)pw * 2 * 64 * 64 + lane;     // + (rb * 64 + record) * 64
    // all LDS-DMA pieces of a tile: 8 key pieces per consumer + the two norm-fragment pieces (consumers 0 and 1; 2 and 3 into the dump)
    auto stage_tile = [&](int t, int buf, bool clamp) __attribute__((always_inline)) {
        const int j0 = t * HB_KT;
        const char *tb = kbase + (size_t)j0 * HB_ROWB;
        char *kt = ktile0 + (size_t)buf * HB_KT * HB_ROWB;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int piece = pw * 8 + e;
            const int r = 2 * piece + h, rc = clamp ? min(r, M - 1 - j0) : r;
            const unsigned off = rc * HB_ROWB + ((r32 ^ (r & 15)) << 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tb + off),
                                             (__attribute__((address_space(3))) void *)(kt + piece * 1024), 16, 0, 0);
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nfbase + (size_t)j0 * 32 + (pw & 1) * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(pw < 2 ? knf0 + buf * HB_KT * 32 + pw * 1024 : dump0), 16, 0, 0);
    };
    auto stage = [&](int t, int buf) __attribute__((always_inline)) {
        if (t < ntiles) {
            if (ragged && t + 1 == ntiles) stage_tile(t, buf, true); else stage_tile(t, buf, false);
        }
    };
    // the hand-off of tile tt: two smallest keys of every (sub-tile, row block) into the lists, the third into the record
    auto process = [&](int tt, int buf) __attribute__((always_inline)) {   // buf: literal after inlining
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int s = 2 * tt + sub, jb = s * 32 + 4 * h;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const unsigned s0 = *hand(buf, sub, 2 * pw + rb, 0), s1 = *hand(buf, sub, 2 * pw + rb, 1), s2 = *hand(buf, sub, 2 * pw + rb, 2);
                const double o0 = R[rb].kb.insert(pack_entry(s0, jb));
                const double o1 = R[rb].kb.insert(pack_entry(s1, jb));
                R[rb].urec = (s % rgrp == 0) ? s2 : min(R[rb].urec, s2);
                recw[(rb * 64 + s / rgrp) * 64] = (unsigned short)(R[rb].urec >> 16);
                terms2(R[rb], o0, o1);
            }
        }
        if ((tt & 3) == 3 || tt < 8) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) update_bound(R[rb]);
        }
    };
    stage(0, 0);
    dma_barrier();
    stamp(5);
    {
        auto iter = [&](int t, int buf) __attribute__((always_inline)) {   // producers compute tile t from buffer buf meanwhile
            stage(t + 1, buf ^ 1);
            stamp(2);
            if (t >= 1) process(t - 1, buf ^ 1);
            if (STAMP) {
                const int x = __builtin_amdgcn_readfirstlane(__double2hiint(R[0].kb.e[0]) ^ __double2hiint(R[1].kb.e[0]));
                asm volatile("" ::"s"(x));
                stamp(1);
            }
            dma_barrier();
            stamp(5);
            T[7] += 1;
        };
        for (int t = 0; t < ntiles; t += 2) {
            iter(t, 0);
            if (t + 1 < ntiles) iter(t + 1, 1);
        }
        if ((ntiles - 1) & 1) process(ntiles - 1, 1); else process(ntiles - 1, 0);
    }

    // ---- per row block: final bound, re-done sub-tiles, merge of the two half-lanes, outputs (as in the second form)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        H3Row &S = R[rb];
        update_bound(S);
        const int qrow = qrowv[rb], qrc = qrow < N ? qrow : N - 1;
        const char *qptr = G.qp + ((size_t)b * N + qrc) * HB_ROWB + 16 * h;
        f16x8 qh[8], qm[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            qh[s] = -*(const f16x8 *)(qptr + 32 * s);
            qm[s] = -*(const f16x8 *)(qptr + 256 + 32 * s);
        }
        f16x8 qn = {0, 0, 0, 0, 0, 0, 0, 0};
        if (h == 0) {
            const float nas = norm_scaled(G.nq[(size_t)b * N + qrc], se);
            const float nasb = (nas + nas * 0x1p-13f) + H2_FLOOR;
            _Float16 p1, p2, p3;
            norm_pieces(nasb, p1, p2, p3);
            qn[0] = (_Float16)0x1p+15f, qn[1] = (_Float16)0x1p+4f, qn[2] = (_Float16)0x1p-7f;
            qn[3] = p1, qn[4] = p2, qn[5] = p3;
        }
        const unsigned limr = S.lim >> 16;
        const int nrec = (nsub + rgrp - 1) / rgrp;
        unsigned long long todo = 0;
#pragma unroll
        for (int g = 0; g < 64; ++g)
            if (g < nrec && __builtin_amdgcn_ballot_w64((unsigned)recw[(rb * 64 + g) * 64] <= limr) != 0) todo |= 1ull << g;
        while (todo != 0) {
            const int g = __builtin_ctzll(todo);
            todo &= todo - 1;
            for (int s = g * rgrp; s < (g + 1) * rgrp && s < nsub; ++s) {
                const int j = s * 32 + r32, jc = j < M ? j : M - 1;
                const char *arow = kbase + (size_t)jc * HB_ROWB + 16 * h;
                const f16x8 an = *(const f16x8 *)(nfbase + (size_t)j * 32 + 16 * h);
                f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const f16x8 ah = *(const f16x8 *)(arow + 32 * u), am = *(const f16x8 *)(arow + 256 + 32 * u);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, qh[u], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qm[u], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[u], acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, qn, acc, 0, 0, 0);
                const int jb = s * 32 + 4 * h;
                unsigned v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v[r] = (__float_as_uint(acc[r]) << 4) + ((unsigned)r - (H2_KBASE << 4));
                    v[r] = jb + (r & 3) + 8 * (r >> 2) < M ? v[r] : (H2_REMOVED | r);
                }
                Top3 w = top3_of_16(v);
                bool more = w.s2 <= S.lim && w.s2 < H2_REMOVED;
                while (__builtin_amdgcn_ballot_w64(more) != 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = v[r] <= w.s1 ? (H2_REMOVED | r) : v[r];
                    w = top3_of_16(v);
                    const double o0 = S.kb.insert(pack_entry(w.s0, jb));
                    const double o1 = S.kb.insert(pack_entry(w.s1, jb));
                    terms2(S, o0, o1);
                    more = w.s2 <= S.lim && w.s2 < H2_REMOVED;
                }
            }
        }
        {   // merge the two half-lanes that share a query (lane, lane^32)
            const float co = __shfl_xor(S.cref, 32, 64), lo = __shfl_xor(S.l, 32, 64);
            const float cm = fmaxf(S.cref, co);
            const float a = (S.cref == -INFINITY) ? 0.f : S.l * exp2f((S.cref - cm) * LOG2E);
            const float bb = (co == -INFINITY) ? 0.f : lo * exp2f((co - cm) * LOG2E);
            S.l = a + bb;
            S.cref = cm;
            double other[HB_KC];
#pragma unroll
            for (int t = 0; t < HB_KC; ++t)
                other[t] = __hiloint2double(__shfl_xor(__double2hiint(S.kb.e[t]), 32, 64), __shfl_xor(__double2loint(S.kb.e[t]), 32, 64));
#pragma unroll
            for (int t = 0; t < HB_KC; ++t) add_term(S, entry_key(S.kb.insert(other[t])));
        }
        if (h == 0 && qrow < N) {
            const size_t row = (size_t)bs * N + qrow;
#pragma unroll
            for (int t = 0; t < HB_KC; ++t) {
                const unsigned key = entry_key(S.kb.e[t]), r = key & 15;
                const bool live = key < H2_REMOVED;
                G.cidx[row * HB_KC + t] = live ? entry_jb(S.kb.e[t]) + (int)((r & 3) + 8 * (r >> 2)) : 0x7fffffff;
                G.cd2[row * HB_KC + t] = live ? key_d2(S, key) : INFINITY;
            }
            G.lsum[row * 2] = S.l;
            G.lsum[row * 2 + 1] = S.cref;
        }
    }
    finish_stamps();
}

}  // namespace

void launch_norm_frags(const float *nrm, int B, int M, int Mpad, const int *amax, char *out, hipStream_t s) {
    hipLaunchKernelGGL(norm_frags_kernel, dim3((Mpad + 255) / 256, B), dim3(256), 0, s, nrm, M, Mpad, amax, out);
}

template <int PIPE, bool STAMP, int WAVES>
static void launch_form(const H2Args &b, int blocks, hipStream_t s) {
    // DVM_K1_LDS_PAD (diagnostic): extra dynamic LDS per workgroup — 8192 keeps a second 4-wave workgroup off the compute unit, so
    // that every wave has its SIMD to itself (what a wave costs when nothing runs beside it)
    static const int pad = [] { const char *e = getenv("DVM_K1_LDS_PAD"); return e ? atoi(e) : 0; }();
    const int lds = h2_lds_bytes<WAVES>() + pad;
    ensure_dyn_lds((const void *)softcorr_sweep2_kernel<PIPE, STAMP, WAVES>, lds);
    hipLaunchKernelGGL((softcorr_sweep2_kernel<PIPE, STAMP, WAVES>), dim3(blocks), dim3(64 * WAVES), lds, s, b);
}

// workgroup size of the second form: 8 waves (256 query rows, one workgroup per compute unit) or 4 (128 rows, two per compute unit)
int sweep2_waves() {
    static const int w = [] {
        const char *e = getenv("DVM_K1_WAVES");
        return e && atoi(e) == 4 ? 4 : 8;
    }();
    return w;
}

// pass A for the groups in `a` (lean semantics), second form; knf = key-side norm fragments of either group.  `a` and `blocks` are
// laid out for 256-row workgroups (HB_QB); the 4-wave form re-derives its own tiling from them.
void launch_sweep2(const HBArgs &a, const char *knf0, const char *knf1, const int *amaxc, int blocks, int form, hipStream_t s) {
    H2Args b;
    const int waves = sweep2_waves();
    for (int g = 0; g < 2; ++g) {
        const HBGroup &G = a.g[g];
        b.g[g] = H2Group{G.qp, G.kp, g == 0 ? knf0 : knf1, G.nq, G.N, G.M, G.Mpad, G.tiles, G.cidx, G.cd2, G.lsum, G.kslices, G.Ms};
    }
    b.blocks0 = a.blocks0;
    if (waves == 4) {
        const int e0 = a.blocks0 / a.g[0].tiles, e1 = blocks > a.blocks0 ? (blocks - a.blocks0) / a.g[1].tiles : 0;   // launch entries per group
        b.g[0].tiles = (a.g[0].N + 127) / 128;
        b.g[1].tiles = (a.g[1].N + 127) / 128;
        b.blocks0 = e0 * b.g[0].tiles;
        blocks = b.blocks0 + e1 * b.g[1].tiles;
    }
    b.amax = amaxc;
    b.neg_alpha = a.neg_alpha;
    b.cutw = a.cutw;
    b.route = a.route;
    b.nb = a.nb;
    b.stamps = nullptr;
    static const bool stamps_on = getenv("DVM_K1_STAMPS") != nullptr;
    if (form == 5 && waves == 8) {   // third form: split roles
        if (stamps_on) {   // diagnostic: synchronous, allocates
            unsigned long long *dbuf = nullptr;
            const size_t n = (size_t)blocks * 12 * 8;
            if (hipMalloc(&dbuf, n * sizeof(unsigned long long)) != hipSuccess) return;
            (void)hipMemset(dbuf, 0, n * sizeof(unsigned long long));
            b.stamps = dbuf;
            ensure_dyn_lds((const void *)softcorr_sweep3_kernel<true>, H3_LDS_BYTES);
            hipLaunchKernelGGL(softcorr_sweep3_kernel<true>, dim3(blocks), dim3(H3_THREADS), H3_LDS_BYTES, s, b);
            (void)hipStreamSynchronize(s);
            unsigned long long *hbuf = (unsigned long long *)malloc(n * sizeof(unsigned long long));
            (void)hipMemcpy(hbuf, dbuf, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double tot[2][8] = {{0}}, cnt[2] = {0, 0};
            for (size_t w = 0; w < (size_t)blocks * 12; ++w) {
                if (hbuf[w * 8 + 6] == 0) continue;   // (workgroup routed elsewhere)
                const int role = (w % 12) < 8 ? 0 : 1;
                cnt[role] += 1;
                for (int i = 0; i < 8; ++i) tot[role][i] += (double)hbuf[w * 8 + i];
            }
            for (int role = 0; role < 2; ++role) {
                const double tiles = tot[role][7] > 0 ? tot[role][7] : 1;
                fprintf(stderr, "K1 stamps (third form, %s): cycles per wave and TILE: %s %.0f  %s %.0f  barrier %.0f  | whole %.0f per tile\n",
                        role ? "consumer" : "producer", role ? "hand-off processing" : "matrix chains", tot[role][1] / tiles,
                        role ? "DMA issue" : "selection", tot[role][2] / tiles, tot[role][5] / tiles, tot[role][6] / tiles);
            }
            free(hbuf);
            (void)hipFree(dbuf);
            return;
        }
        ensure_dyn_lds((const void *)softcorr_sweep3_kernel<false>, H3_LDS_BYTES);
        hipLaunchKernelGGL(softcorr_sweep3_kernel<false>, dim3(blocks), dim3(H3_THREADS), H3_LDS_BYTES, s, b);
        return;
    }
    if (stamps_on) {   // diagnostic: synchronous, allocates — never taken in production
        unsigned long long *dbuf = nullptr;
        const size_t n = (size_t)blocks * waves * 8;
        if (hipMalloc(&dbuf, n * sizeof(unsigned long long)) != hipSuccess) return;
        b.stamps = dbuf;
        if (waves == 4) launch_form<1, true, 4>(b, blocks, s);
        else if (form == 1) launch_form<0, true, 8>(b, blocks, s);
        else if (form == 4) launch_form<3, true, 8>(b, blocks, s);
        else launch_form<1, true, 8>(b, blocks, s);
        (void)hipStreamSynchronize(s);
        unsigned long long *hbuf = (unsigned long long *)malloc(n * sizeof(unsigned long long));
        (void)hipMemcpy(hbuf, dbuf, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t w = 0; w < (size_t)blocks * waves; ++w)
            for (int i = 0; i < 8; ++i) tot[i] += (double)hbuf[w * 8 + i];
        double redo = 0;
        for (size_t w = 0; w < (size_t)blocks * waves; ++w) {
            redo += (double)(hbuf[w * 8 + 7] >> 20);
            tot[7] -= (double)(hbuf[w * 8 + 7] >> 20 << 20);
        }
        const double nw = (double)blocks * waves, st = tot[7] / nw;
        fprintf(stderr, "K1 stamps: %.2f re-done records per wave\n", redo / nw);
        fprintf(stderr, "K1 stamps (form %d, %d waves, %d blocks): per wave and sub-tile, cycles: dma %.0f  chain %.0f  epilogue %.0f  terms+redo %.0f  "
                        "bound %.0f  barrier %.0f  | whole sweep %.0f per sub-tile (%.0f sub-tiles per wave)\n", form, waves, blocks, tot[0] / nw / st,
                tot[1] / nw / st, tot[2] / nw / st, tot[3] / nw / st, tot[4] / nw / st, tot[5] / nw / st, tot[6] / nw / st, st);
        free(hbuf);
        (void)hipFree(dbuf);
        return;
    }
    if (waves == 4) launch_form<1, false, 4>(b, blocks, s);
    else if (form == 1) launch_form<0, false, 8>(b, blocks, s);
    else if (form == 3) launch_form<2, false, 8>(b, blocks, s);
    else if (form == 4) launch_form<3, false, 8>(b, blocks, s);
    else launch_form<1, false, 8>(b, blocks, s);
}

}  // namespace k1
}  // namespace dvm
