// dvm_softcorr_coarse.hip — pass A of the soft-correspondence kernel (K1), coarse screen: ONE fp16 plane.
// Built with -fno-honor-nans (Makefile): no NaN is formed here (keys and list entries are unsigned bit patterns).
// (reference: models/loss.py:110-114, 1339-1347, 1404-1407; the hard map: models/loss.py:91-95, test.py:19-23)
//
// The other pass-A kernels evaluate three partial products (hh + hm + mh) of the exact 2-way fp16 split so that their
// approximate distances carry fp32-class errors — only to RANK columns that pass B then re-evaluates with the reference's own
// arithmetic anyway.  This form screens with the hh product alone: 8 + 1 matrix instructions per 32 x 32 sub-tile instead of
// 24 + 1, half the LDS-DMA pieces and half the LDS bytes (the m plane is never staged).  What it gives up is accuracy,
//     |d2_coarse - d2_chain| <= HC_ERR (|q|^2 + max |k|^2),  HC_ERR = 1.12e-3   (derivation at HC_ERR in dvm_softcorr_f16.h)
// — 44 x the three-product bound — and pass B pays for it with a wider certification band around a LONGER candidate list
// (16 per row instead of 12).  The screen owes no softmax terms: it serves rows whose softmax cut lies inside the certified list
// (the probe routes to it: large alpha on spread-out features, the benchmark's regime) and the hard map; pass B certifies, per
// row, both the top-10 and that no column outside the list can lie within the cut, and sends a row that fails either test
// through the exact-rows kernel.  Integer outputs stay bit-exact by construction.
//
// Structure (the tile staging, the norm instruction, the exact keys and the selection network come from the three-product
// "second form" of rounds 3 - 4, which this kernel replaced: profiles/notes_k1.md):
//  * a wave owns 2 blocks of 32 query rows (64 VGPRs of query fragments, as the two-plane form needs for one block): every
//    key fragment read from LDS feeds two matrix instructions, and a workgroup of 4 waves covers 256 query rows (a pair's key plane is
//    staged 8 times; two such workgroups share a compute unit — HC_WAVES below);
//  * key tiles of 64 keys x 256 B by LDS-DMA (buffer_load ... lds: two 1-KiB pieces per wave and tile + the norm fragments),
//    16-B chunks XOR-swizzled with the row number: a 256-byte row is one full bank row, the swizzle spreads the 16 rows of a
//    ds_read_b128 lane group over the 16 chunk positions; the two halves of a workgroup's waves issue their pieces half a tile apart;
//  * norms on the matrix pipe (a 9th instruction per block), exact 32-bit keys (accumulator bits + register number), a fixed
//    selection network for the sorted three smallest of a lane's 16 keys: 46 three-input min / med / max instructions;
//  * 32-bit list entries [key: 19 bits | sub-tile: 8 | half: 1 | register: 4]: a sorted insertion is ONE v_med3_u32 per slot
//    (the packed doubles of the first form take a v_min_f64 + v_max_f64 pair).  19 key bits = 5 exponent bits + 14 of the
//    mantissa, rounded DOWN: 2^-14 of the accumulator, part of HC_ERR.  Lists of 12 per half-lane; the two smallest of a
//    sub-tile go in unconditionally, the third is recorded (16 bits, one LDS slot per lane and record), and after the last
//    tile a record at or below the row's bound has its sub-tiles re-done from global memory;
//  * no softmax terms, no cut, no bound updates in the sweep;
//  * the next sub-tile's 18 matrix instructions and this sub-tile's ~190 vector instructions are interleaved BY HAND, one
//    matrix instruction per slice of ~10 vector instructions, fenced (`fused`): what bounds the kernel is the SIMD's vector
//    issue (4 cycles per instruction, shared by its two waves — 60 % busy over the kernel, the matrix pipe 27 %);
//  * at the end the two half-lanes of a row merge into 16 entries; entries above lim = min of the two 12th entries (and of the
//    larger of the two 8th) are dropped — each half-lane is complete only up to its own 12th —, so the largest entry written
//    IS the completeness bound pass B certifies against.
#include <stdlib.h>
#include <type_traits>

#include "dvm_softcorr_f16.h"

namespace dvm {
namespace k1 {
namespace {

constexpr int HC_ROWB = 256;                   // bytes of a key row in LDS (h plane)
constexpr int HC_KL = 12;                      // list entries per half-lane
constexpr unsigned HC_REMOVED = 0xffc00000u;   // keys / entries >= this: removed / invalid
constexpr unsigned HC_KBASE = 129u << 23;      // bits(4.0f): bottom of the key window
constexpr float HC_FLOOR = 4.5f;               // added to every accumulator through the norm instruction
constexpr unsigned HC_EMASK = 0xffffe00fu;     // bits of a key that survive in a list entry (19 key bits + register number)
constexpr int HC_MAX_M = 256 * 32;             // 8 bits of sub-tile number

constexpr int HC_QB = 2;                       // blocks of 32 query rows per wave
constexpr int HC_NREC = 32;                    // third-key records per lane and block (8 KB of LDS per wave)
// Waves per workgroup: 4, i.e. 256 query rows and 71 KB of LDS — TWO independent workgroups per compute unit, one wave of each per SIMD
// (round 6).  With one 8-wave workgroup (512 rows, 103 KB; rounds 4 - 5) both waves of a SIMD stand at the same barrier and the same
// LDS-DMA issue; two workgroups in different phases fill each other's dead time: 1.74 -> 1.63 ms per launch of 512 pairs alone,
// 2.30 -> 2.20 ms inside the step — for twice the L2 -> LDS staging of the key plane (8 instead of 4 times per pair: L2 hits).
constexpr int HC_WAVES = 4;
// W = waves per workgroup: 64 W query rows
template <int KT, int W> constexpr int hc_lds_bytes() { return 2 * KT * HC_ROWB + 2 * KT * 32 + 1024 + W * HC_QB * HC_NREC * 64 * 2; }

struct HCGroup {
    const char *qp, *kp;      // planes of the query / key side [B][rows][512]: h plane = the first 256 B of a row
    const char *knf;          // key-side norm fragments [B][Mpad][32 B] (launch_norm_prep)
    const float *nq;          // |q|^2 (ATen order)
    int N, M, Mpad, tiles;    // tiles: workgroups per batch entry (512 query rows each)
    int32_t *cidx;            // [B][N][K1_KC_COARSE]
    float *cd2, *lsum;
};
struct HCArgs {
    HCGroup g[2];
    const int *amax;          // bit pattern of max |x| over BOTH sides (the common scale)
    int blocks0;
    const int *route;         // per (group, batch entry), or nullptr = this launch takes all
    int nb;
    unsigned long long *stamps;   // diagnostic build only (DVM_K1_STAMPS): [block][wave][8] cycle totals per phase
};

// three-input unsigned min / max / median as the instructions themselves: written as min / max expressions, instruction selection
// forms them only where the inner two-input node has a single use, and the selection network shares those nodes (55 instead of
// 46 instructions per block; the kernel is bound by its vector instructions: 4 cycles each on a SIMD, whichever wave issues)
__device__ __forceinline__ unsigned umin3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned umax3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

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
// sweep): [B][Mpad][16 fp16] = {a1, a2, a3, 2^15, 2^4, 2^-7, 0, 0 | 0 x 8} — the A operand of the norm instruction; written by
// norm_prep_kernel below

// sorted (s0 <= s1 <= s2) three smallest of 16 distinct keys: 46 three-input unsigned min / med / max instructions
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

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// sorted list of K unsigned entries; insertion = one three-input median per slot
template <int K>
struct List32 {
    unsigned e[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int t = 0; t < K; ++t) e[t] = 0xffffffffu;
    }
    __device__ __forceinline__ void insert(unsigned x) {
#pragma unroll
        for (int p = K - 1; p > 0; --p) e[p] = umed3(e[p - 1], e[p], x);   // (descending p: e[p - 1] is still the old one)
        e[0] = min(e[0], x);
    }
};

// list entry of a key: its 19 top bits and its register number kept, (sub-tile, half) in between
__device__ __forceinline__ unsigned make_entry(unsigned key, unsigned sh) { return (key & HC_EMASK) | sh; }   // one v_bfi_b32
__device__ __forceinline__ int entry_col(unsigned e) {
    const unsigned r = e & 15u;
    return (int)(((e >> 5) & 0xffu) * 32u + ((e >> 4) & 1u) * 4u + (r & 3u) + 8u * (r >> 2));
}

// STAMP: diagnostic build — every wave adds up the shader cycles (s_memtime) it spends per phase; no output depends on them.
// Phases: 0 LDS-DMA issue, 1 matrix chain + epilogue (until the accumulators are readable), 3 re-done sub-tiles, 5 barrier
// (incl. the wait for the wave's own DMA pieces), 6 whole kernel, 7 sub-tiles (re-done records counted in the high bits).
// KT: keys per LDS tile (a multiple of 64: an even number of 32-key sub-tiles between two barriers).
template <int KT, int W, bool STAMP = false>
__global__ __launch_bounds__(64 * W, 8 / W) void softcorr_coarse_kernel(const HCArgs args) {
    constexpr int QB = HC_QB, NREC = HC_NREC, ROWS = 32 * QB * W, SUBS = KT / 32;
    static_assert(SUBS % 2 == 0 && KT * HC_ROWB / 1024 % W == 0 && KT * 32 / 1024 <= W, "coarse screen: tile size");
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
    char *const ktile0 = smem_b;                                         // [2][KT][256], 16-B chunks XOR-swizzled
    char *const knf0 = smem_b + (size_t)2 * KT * HC_ROWB;                // [2][KT][32]
    char *const dump0 = knf0 + 2 * KT * 32;                              // 1 KiB nobody reads (see stage_tile)
    unsigned short *const rec0 = (unsigned short *)(dump0 + 1024);       // [wave][block][NREC][64 lanes]

    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int grp = lid >= args.blocks0 ? 1 : 0;
    lid -= grp ? args.blocks0 : 0;
    const HCGroup &G = args.g[grp];
    const int N = G.N, M = G.M;
    const int b = lid / G.tiles, qt = lid % G.tiles;
    if (args.route && args.route[grp * args.nb + b] != K1_ROUTE_COARSE) return;   // this pair goes through another form
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave: a scalar register)
    const int r32 = lane & 31, h = lane >> 5;

    const char *kbase = G.kp + (size_t)b * M * HB_ROWB;
    const char *nfbase = G.knf + (size_t)b * G.Mpad * 32;
    const int se = scale_exp(*args.amax);
    const float cf = pow2i(1 - 2 * se);             // 2 / s^2: accumulator units -> squared distance
    int qrow[QB];
    float rowc[QB];                                 // what a key's accumulator carries on top of the scaled squared distance
    f16x8 qh[QB][8], qn[QB];  // B-operand fragments, NEGATED (the accumulator carries + |q|^2 + |k|^2 - 2 q.k): k = 16 s + 8 h + j
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        qrow[qb] = qt * ROWS + (wave * QB + qb) * 32 + r32;
        const int qrc = qrow[qb] < N ? qrow[qb] : N - 1;
        const char *qptr = G.qp + ((size_t)b * N + qrc) * HB_ROWB + 16 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) qh[qb][s] = -*(const f16x8 *)(qptr + 32 * s);
        // the query's norm, biased by 2^-8 of itself plus the floor, so that the accumulator stays >= 4 whatever the rounding: in
        // accumulator units the one-plane product is off by at most e |q||k| s^2, e = 2^-10, and with x = |k| / |q| the exact part
        // ((x - 1)^2 + bias) |q|^2 s^2 / 2 exceeds it for every x once bias >= 2 e + e^2; the rest of 2^-8 covers the fp32
        // accumulation (2^-16 |q||k| s^2).
        const float nas = norm_scaled(G.nq[(size_t)b * N + qrc], se);
        const float nasb = (nas + nas * 0x1p-8f) + HC_FLOOR;
        rowc[qb] = nasb - nas;
        qn[qb] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (h == 0) {
            _Float16 p1, p2, p3;
            norm_pieces(nasb, p1, p2, p3);
            qn[qb][0] = (_Float16)0x1p+15f, qn[qb][1] = (_Float16)0x1p+4f, qn[qb][2] = (_Float16)0x1p-7f;
            qn[qb][3] = p1, qn[qb][4] = p2, qn[qb][5] = p3;
        }
    }

    List32<HC_KL> kb[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) kb[qb].init();

    const int ntiles = (M + KT - 1) / KT, nsub = SUBS * ntiles;
    const int rgrp = (nsub + NREC - 1) / NREC;          // sub-tiles per record
    unsigned short *const rec = rec0 + (size_t)wave * QB * NREC * 64 + lane;
    unsigned urec[QB];                                  // smallest third key of the current record's sub-tiles
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) urec[qb] = 0xffffffffu;

    // Key tiles go global -> LDS by LDS-DMA (buffer_load ... lds: a scalar resource descriptor per source array, a 32-bit
    // per-lane offset, a scalar tile offset), 1 KiB = 4 rows of 256 B per wave instruction.  Lane L of piece p delivers position
    // L & 15 of LDS row 4 p + (L >> 4), which holds chunk (L & 15) ^ (row & 15) of the key.  The per-lane offsets are re-formed
    // for every tile from the lane number (v_mbcnt): anything kept in a vector register across the tile for this is spilled,
    // and a scratch reload next to the DMA issue waits, through vmcnt, for every piece in flight.
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void *)kbase, 0, M * HB_ROWB, 0x00020000);
    const __amdgpu_buffer_rsrc_t nrs = __builtin_amdgcn_make_buffer_rsrc((void *)nfbase, 0, G.Mpad * 32, 0x00020000);
    auto stage_tile = [&](int t, int buf, bool clamp) __attribute__((always_inline)) {
        int lane;   // (a volatile statement: otherwise the offsets are hoisted out of the loop as invariants — and spilled)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        const int j0 = t * KT;
        char *kt = ktile0 + (size_t)buf * KT * HC_ROWB;
        constexpr int PIECES = KT * HC_ROWB / 1024 / W, NP = KT * 32 / 1024;
#pragma unroll
        for (int e = 0; e < PIECES; ++e) {
            const int piece = wave * PIECES + e;
            const int r = 4 * piece + (lane >> 4), rc = clamp ? min(r, M - 1 - j0) : r;
            const unsigned off = rc * HB_ROWB + (((lane & 15) ^ (r & 15)) << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void *)(kt + piece * 1024), 16, off, j0 * HB_ROWB, 0, 0);
        }
        // KT keys x 32 B of norm fragments = NP 1-KiB pieces, brought by the first NP waves; the others issue the same instruction
        // into a dump area (destination chosen by a scalar select): no branch here
        __builtin_amdgcn_raw_ptr_buffer_load_lds(nrs, (__attribute__((address_space(3))) void *)(wave < NP ? knf0 + buf * KT * 32 + wave * 1024 : dump0), 16,
                                                 (wave < NP ? wave : 0) * 1024 + lane * 16, j0 * 32, 0, 0);
    };
    // every wave first waits for ITS OWN pieces (vmcnt), then joins the barrier: the compiler places its own vmcnt wait only in
    // front of the wave's next LDS read, which orders nothing for the rows the OTHER waves were to deliver
    auto dma_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    // A fragment (row r32 of the sub-tile, chunk 2 s + h) sits at row * 256 + (((2 s) ^ h ^ (row & 15)) << 4): the sub-tile and
    // the buffer are immediate offsets of eight loop-invariant address registers
    unsigned fadr[8];
    {
        const unsigned rowb = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)ktile0 + r32 * HC_ROWB;
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
    struct Acc {
        f32x16 a[QB];
    };
    // the matrix work of one sub-tile: per block 8 product instructions + the norm instruction (last: every partial sum before it
    // has the magnitude of q.k), every key fragment used QB times.  (Plain form: the first sub-tile and a ragged last tile.)
    auto chain = [&](int buf, int sub) __attribute__((always_inline)) -> Acc {   // buf, sub: literals after inlining
        const int toff = buf * (KT * HC_ROWB) + sub * (32 * HC_ROWB);
        Acc A;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) A.a[qb] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const f16x8 fr = lds16(fadr[s], toff);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) A.a[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr, qh[qb][s], A.a[qb], 0, 0, 0);
        }
        const f16x8 an = lds16(nadr, buf * (KT * 32) + sub * (32 * 32));
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) A.a[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, qn[qb], A.a[qb], 0, 0, 0);
        return A;
    };

    // keys of the lane's 16 accumulators (register r holds local key (r & 3) + 8 (r >> 2) of the lane's half)
    auto make_keys = [&](const f32x16 &acc, unsigned (&v)[16], int jb, bool mask_pads) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[r] = (__float_as_uint(acc[r]) << 4) + ((unsigned)r - (HC_KBASE << 4));   // one v_lshl_add_u32
            if (mask_pads) v[r] = jb + (r & 3) + 8 * (r >> 2) < M ? v[r] : (HC_REMOVED | r);
        }
    };
    const unsigned h4 = (unsigned)h << 4;
    // one sub-tile: keys, sorted three smallest, the two smallest into the list, the third into the record (plain form)
    auto epilogue = [&](const Acc &A, int s, bool mask_pads) __attribute__((always_inline)) {
        const unsigned sh = ((unsigned)s << 5) | h4;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            unsigned v[16];
            make_keys(A.a[qb], v, s * 32 + 4 * h, mask_pads);
            const Top3 w = top3_of_16(v);
            kb[qb].insert(make_entry(w.s0, sh));
            kb[qb].insert(make_entry(w.s1, sh));
            // (branch-free: a record is rewritten by every sub-tile of its group with the running minimum)
            urec[qb] = (s % rgrp == 0) ? w.s2 : min(urec[qb], w.s2);
            rec[(qb * NREC + s / rgrp) * 64] = (unsigned short)(urec[qb] >> 16);
        }
    };

    // The hot form of the same two things — the matrix chain of the NEXT sub-tile, the epilogue of this one — written out as 18
    // groups of ONE matrix instruction + one slice of the epilogue (~10 vector instructions), fenced so that the order stands.  A
    // wave issues in order: behind a run of matrix instructions it is parked at the matrix pipe (32 cycles each) while its vector
    // instructions wait, and what bounds this kernel is the SIMD's vector issue — 4 cycles per instruction whichever of its two
    // waves issues it (measured: a wave's 18 + ~190 instructions of a sub-tile take ~1 800 cycles beside its partner's, ~900 being
    // its own issue time).  Fragment reads are requested three k-steps ahead of their use.  Block qb's epilogue rides on the
    // matrix instructions of the groups [9 qb, 9 qb + 9).
    struct Epi {
        unsigned v[16];
        Top3 A, B, C, D, E, AB, CD, CDE, T, w;
        unsigned m00, e0, e1;
    };
    auto epi_slice = [&](auto sc, Epi &S, const f32x16 &acc, int qb, int s) __attribute__((always_inline)) {
        constexpr int SL = decltype(sc)::value;
        auto key = [&](int r) __attribute__((always_inline)) { return (__float_as_uint(acc[r]) << 4) + ((unsigned)r - (HC_KBASE << 4)); };
        if constexpr (SL == 0) {
#pragma unroll
            for (int r = 0; r < 11; ++r) S.v[r] = key(r);
        } else if constexpr (SL == 1) {
#pragma unroll
            for (int r = 11; r < 16; ++r) S.v[r] = key(r);
            S.A = sort3(S.v[0], S.v[1], S.v[2]);
            S.B = sort3(S.v[3], S.v[4], S.v[5]);
        } else if constexpr (SL == 2) {
            S.C = sort3(S.v[6], S.v[7], S.v[8]);
            S.D = sort3(S.v[9], S.v[10], S.v[11]);
            S.E = sort3(S.v[12], S.v[13], S.v[14]);
        } else if constexpr (SL == 3) {
            S.AB = merge3(S.A, S.B);
            S.m00 = max(S.C.s0, S.D.s0);
            S.CD.s0 = min(S.C.s0, S.D.s0);
            S.CD.s1 = umin3(S.m00, S.C.s1, S.D.s1);
        } else if constexpr (SL == 4) {
            S.CD.s2 = min(umin3(S.C.s2, S.D.s2, max(S.C.s1, S.D.s0)), max(S.C.s0, S.D.s1));
            S.CDE = merge3(S.CD, S.E);
        } else if constexpr (SL == 5) {
            S.T = merge3(S.AB, S.CDE);
            const unsigned x = S.v[15];
            S.w = Top3{(unsigned)min(S.T.s0, x), umed3(S.T.s0, S.T.s1, x), umed3(S.T.s1, S.T.s2, x)};
            const unsigned sh = ((unsigned)s << 5) | h4;
            S.e0 = make_entry(S.w.s0, sh);
            S.e1 = make_entry(S.w.s1, sh);
        } else if constexpr (SL == 6) {
            kb[qb].insert(S.e0);
        } else if constexpr (SL == 7) {
            kb[qb].insert(S.e1);
        } else {
            urec[qb] = (s % rgrp == 0) ? S.w.s2 : min(urec[qb], S.w.s2);
            rec[(qb * NREC + s / rgrp) * 64] = (unsigned short)(urec[qb] >> 16);
        }
    };
    // -> accumulators of sub-tile (buf, sub); `cur`: accumulators of sub-tile s (complete), whose epilogue runs meanwhile
    auto fused = [&](int buf, int sub, const Acc &cur, int s) __attribute__((always_inline)) -> Acc {   // buf, sub: literals after inlining
        const int toff = buf * (KT * HC_ROWB) + sub * (32 * HC_ROWB);
        Acc A;
        f16x8 fr[8], an;
        Epi S[QB];
        fr[0] = lds16(fadr[0], toff);
        fr[1] = lds16(fadr[1], toff);
        fr[2] = lds16(fadr[2], toff);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 9 * QB>([&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value, ks = g / QB, qb = g % QB;   // matrix instruction: k-step ks (8 = the norms) of block qb
            constexpr int eb = g / 9, sl = g % 9;                             // epilogue slice sl of block eb
            if constexpr (qb == 0 && ks + 3 < 8) fr[ks + 3] = lds16(fadr[ks + 3], toff);
            if constexpr (qb == 0 && ks == 5) an = lds16(nadr, buf * (KT * 32) + sub * (32 * 32));
            if constexpr (ks == 0) A.a[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[0], qh[qb][0], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else if constexpr (ks < 8) A.a[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[ks], qh[qb][ks], A.a[qb], 0, 0, 0);
            else A.a[qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, qn[qb], A.a[qb], 0, 0, 0);
            epi_slice(std::integral_constant<int, sl>{}, S[eb], cur.a[eb], eb, s);
            __builtin_amdgcn_sched_barrier(0);
        });
        return A;
    };

    const bool ragged = M % KT != 0;   // the last tile holds padding keys
    auto stage = [&](int t, int buf) __attribute__((always_inline)) {
        if (t < ntiles) {
            if (ragged && t + 1 == ntiles) stage_tile(t, buf, true); else stage_tile(t, buf, false);
        }
    };
    // Tile t + 1 is requested at the top of tile t, into the buffer whose last reads lie before the previous barrier — by the
    // first half of the waves.  The second half (their SIMD partners) brings ITS pieces half a tile earlier: tile t + 2 behind the
    // barrier of tile t, into the buffer whose last reads lie in front of that barrier — so that the two waves of a SIMD do not
    // both stand at the LDS-DMA issue at the same time.
    const bool late = wave >= W / 2;
    stage(0, 0);
    if (late) stage(1, 1);
    dma_barrier();
    {
        // software pipeline across the sub-tiles of the whole sweep: accumulator sets alternate (even sub-tiles a[0], odd a[1])
        Acc a[2];
        a[0] = chain(0, 0);
        auto tile = [&](int t, int buf, bool last) __attribute__((always_inline)) {   // buf, last: literals after inlining
            if (!last && !late) stage(t + 1, buf ^ 1);
            stamp(0);
            const bool plain = last && ragged;
#pragma unroll
            for (int sub = 1; sub < SUBS; ++sub) {
                if (!plain) {
                    a[sub & 1] = fused(buf, sub, a[(sub & 1) ^ 1], t * SUBS + sub - 1);
                } else {
                    a[sub & 1] = chain(buf, sub);
                    epilogue(a[(sub & 1) ^ 1], t * SUBS + sub - 1, true);
                }
                stamp_after(1, __float_as_int(a[sub & 1].a[QB - 1][0]));
            }
            dma_barrier();
            stamp(5);
            if (late && !last) {
                stage(t + 2, buf);
                stamp(0);
            }
            if (!last) {
                a[0] = fused(buf ^ 1, 0, a[1], t * SUBS + SUBS - 1);
                stamp_after(1, __float_as_int(a[0].a[QB - 1][0]));
            } else if (plain) {
                epilogue(a[1], t * SUBS + SUBS - 1, true);
            } else {
                epilogue(a[1], t * SUBS + SUBS - 1, false);
            }
            T[7] += SUBS;
        };
        // (the last tile is peeled off so that inside the loop the next chain is unconditional)
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

    // ---- the row's bound and the re-done sub-tiles.  Each half-lane is complete up to ITS 12th entry; the row up to the
    // smaller of the two: lim.  A sub-tile in which a lane may hold MORE than two entries at or below lim (its recorded third
    // key says so) is re-done: the same fragments (straight from global memory), the same chain, hence the same keys; the two
    // smallest are in the list already, the rest goes through a loop that takes two per trip.
    // (Also capped by max(a_8, b_8), a and b the two half-lanes: 16 entries lie at or below it, so the row's 16 best do — nothing
    // beyond them is handed to pass B anyway, and the lower bound re-does a third of the sub-tiles min(a_12, b_12) alone would.)
    unsigned lim[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const unsigned wk = kb[qb].e[HC_KL - 1], wm = kb[qb].e[K1_KC_COARSE / 2 - 1];
        const auto sk = __builtin_amdgcn_permlane32_swap(wk, wk, false, false);
        const auto sm = __builtin_amdgcn_permlane32_swap(wm, wm, false, false);
        lim[qb] = min(min(wk, h ? sk[0] : sk[1]), max(wm, h ? sm[0] : sm[1]));
    }
    const int nrec = (nsub + rgrp - 1) / rgrp;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const unsigned limr = lim[qb] >> 16, limk = lim[qb] | ~HC_EMASK;   // (as a key: every key that shares the bound's 19 bits counts)
        unsigned long long todo = 0;   // wave-uniform: records with a lane at or below its bound
        unsigned short rv[NREC];       // (all reads requested before the first is used: one LDS round trip, not NREC)
#pragma unroll
        for (int g = 0; g < NREC; ++g) rv[g] = rec[(qb * NREC + g) * 64];
#pragma unroll
        for (int g = 0; g < NREC; ++g)
            if (g < nrec && __builtin_amdgcn_ballot_w64((unsigned)rv[g] <= limr) != 0) todo |= 1ull << g;
        while (todo != 0) {
            const int g = __builtin_ctzll(todo);
            todo &= todo - 1;
            if (STAMP) T[7] += 1u << 20;
            for (int s = g * rgrp; s < (g + 1) * rgrp && s < nsub; ++s) {
                const int j = s * 32 + r32, jc = j < M ? j : M - 1;
                const char *arow = kbase + (size_t)jc * HB_ROWB + 16 * h;
                const f16x8 an = *(const f16x8 *)(nfbase + (size_t)jc * 32 + 16 * h);   // (padding keys: masked below)
                f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const f16x8 *)(arow + 32 * u), qh[qb][u], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, qn[qb], acc, 0, 0, 0);
                const unsigned sh = ((unsigned)s << 5) | h4;
                unsigned v[16];
                make_keys(acc, v, s * 32 + 4 * h, true);
                Top3 w = top3_of_16(v);   // w.s0, w.s1: inserted by the sweep
                bool more = w.s2 <= limk && w.s2 < HC_REMOVED;
                while (__builtin_amdgcn_ballot_w64(more) != 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = v[r] <= w.s1 ? (HC_REMOVED | r) : v[r];
                    w = top3_of_16(v);
                    kb[qb].insert(make_entry(w.s0, sh));
                    kb[qb].insert(make_entry(w.s1, sh));
                    more = w.s2 <= limk && w.s2 < HC_REMOVED;
                }
            }
        }
    }
    stamp(3);

    // ---- merge the two half-lanes that share a query (lane, lane ^ 32) into 16 entries and write the row
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        List32<K1_KC_COARSE> out;
#pragma unroll
        for (int t = 0; t < K1_KC_COARSE; ++t) out.e[t] = t < HC_KL ? kb[qb].e[t] : 0xffffffffu;
#pragma unroll
        for (int t = 0; t < HC_KL; ++t) out.insert((unsigned)__shfl_xor((int)kb[qb].e[t], 32, 64));
        if (h == 0 && qrow[qb] < N) {
            const size_t row = (size_t)b * N + qrow[qb];
            // (16-byte stores: a row's 16 columns / distances are 64 contiguous bytes.  Round 6: transposing the block's 32 rows through LDS so
            // that every store instruction covers 1 KiB of consecutive addresses changed neither the launch time nor WRITE_SIZE
            // (0.55 -> 0.59 GB for 0.27 GB of lists) and was removed: the excess is the kernel's scratch — 36 spilled registers,
            // 148 B per lane, written once per wave = 0.31 GB per launch — not partial lines)
            typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int q = 0; q < K1_KC_COARSE / 4; ++q) {
                i32x4 ci;
                f32x4 cd;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned e = out.e[4 * q + u];
                    const bool live = e < HC_REMOVED && e <= lim[qb];
                    // the entry's accumulator, rounded down to the 19 key bits it kept -> squared distance (a lower bound of the
                    // coarse value within 2^-14 of the accumulator: part of HC_ERR)
                    const float d2 = fmaxf(__uint_as_float(((e & 0xffffe000u) >> 4) + HC_KBASE) - rowc[qb], 0.f) * cf;
                    ci[u] = live ? entry_col(e) : 0x7fffffff;
                    cd[u] = live ? d2 : INFINITY;
                }
                *(i32x4 *)(G.cidx + row * K1_KC_COARSE + 4 * q) = ci;
                *(f32x4 *)(G.cd2 + row * K1_KC_COARSE + 4 * q) = cd;
            }
            G.lsum[row * 2] = 0.f;            // no softmax terms from this screen: pass B certifies that none is owed
            G.lsum[row * 2 + 1] = -INFINITY;
        }
    }
    if (STAMP) {
        T[6] = __builtin_amdgcn_s_memtime() - tstart;
        if (lane == 0 && args.stamps)
            for (int i = 0; i < 8; ++i) args.stamps[((size_t)blockIdx.x * W + wave) * 8 + i] = T[i];
    }
}

template <int KT, int W, bool STAMP>
static void launch_form(const HCArgs &a, int blocks, hipStream_t s) {
    const int lds = hc_lds_bytes<KT, W>();
    ensure_dyn_lds((const void *)softcorr_coarse_kernel<KT, W, STAMP>, lds);
    hipLaunchKernelGGL((softcorr_coarse_kernel<KT, W, STAMP>), dim3(blocks), dim3(64 * W), lds, s, a);
}

}  // namespace

bool coarse_supports(int N, int M) { return M <= HC_MAX_M && N <= HC_MAX_M; }

// Everything pass A and pass B want of the row norms of both sides in ONE launch (round 6; six launches before: norm_max x 2,
// pad_norms x 2, norm_frags x 2 — at a strong-scaling rank's batch the prologue of the sweep is a chain of ~20 launches of a few
// microseconds each): per side the maximum per batch element (atomic: positive floats order like their bit patterns), the norms padded
// to whole key tiles with +inf, the coarse screen's norm fragments; block (0, 0, 0) also zeroes the flagged-row counters.
__global__ __launch_bounds__(256) void norm_prep_kernel(const NormPrep a) {
    const int sd = blockIdx.z, b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = a.rows[sd], Mp = a.pad[sd];
    if (blockIdx.x == 0 && b == 0 && sd == 0 && threadIdx.x < 2 && a.zero[threadIdx.x]) *a.zero[threadIdx.x] = 0;
    if (blockIdx.x * blockDim.x >= Mp) return;
    const float nv = i < M ? a.nrm[sd][(size_t)b * M + i] : 0.f;
    float v = nv;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0 && blockIdx.x * blockDim.x + (threadIdx.x & ~63) < M) atomicMax((int *)a.nmax[sd] + b, __float_as_int(v));
    if (i >= Mp) return;
    if (a.npad[sd]) a.npad[sd][(size_t)b * Mp + i] = i < M ? nv : INFINITY;
    if (a.nfrag[sd]) {
        f16x8 lo = {0, 0, 0, (_Float16)0x1p+15f, (_Float16)0x1p+4f, (_Float16)0x1p-7f, 0, 0};
        const f16x8 hi = {0, 0, 0, 0, 0, 0, 0, 0};
        if (i < M) {
            _Float16 p1, p2, p3;
            norm_pieces(norm_scaled(nv, scale_exp(*a.amax)), p1, p2, p3);
            lo[0] = p1, lo[1] = p2, lo[2] = p3;
        }
        char *p = a.nfrag[sd] + ((size_t)b * Mp + i) * 32;
        *(f16x8 *)p = lo;
        *(f16x8 *)(p + 16) = hi;
    }
}
void launch_norm_prep(const NormPrep &a, int B, hipStream_t s) {
    const int mp = a.pad[0] > a.pad[1] ? a.pad[0] : a.pad[1];
    hipLaunchKernelGGL(norm_prep_kernel, dim3((mp + 255) / 256, B, 2), dim3(256), 0, s, a);
}

// pass A for the groups in `a`, coarse screen; knf = key-side norm fragments of either group (launch_norm_prep).  `a` is laid
// out for the 256-row workgroups of the other forms; this form re-derives its own tiling.
void launch_coarse(const HBArgs &a, const char *knf0, const char *knf1, const int *amaxc, int blocks, hipStream_t s) {
    constexpr int kt = 64;   // keys per LDS tile (128 — half the barriers, 137 KB of LDS — measured 5 % slower: profiles/notes_k1.md)
    HCArgs c;
    const int rows = 32 * HC_QB * HC_WAVES;
    const int e0 = a.blocks0 / a.g[0].tiles, e1 = blocks > a.blocks0 ? (blocks - a.blocks0) / a.g[1].tiles : 0;   // batch entries per group
    for (int g = 0; g < 2; ++g) {
        const HBGroup &G = a.g[g];
        c.g[g] = HCGroup{G.qp, G.kp, g == 0 ? knf0 : knf1, G.nq, G.N, G.M, G.Mpad, (G.N + rows - 1) / rows, G.cidx, G.cd2, G.lsum};
    }
    c.blocks0 = e0 * c.g[0].tiles;
    const int nblocks = c.blocks0 + e1 * c.g[1].tiles;
    c.amax = amaxc;
    c.route = a.route;
    c.nb = a.nb;
    c.stamps = nullptr;
    if (options().debug & DVM_DEBUG_K1_STAMPS) {   // diagnostic: synchronous, allocates — never taken in production
        unsigned long long *dbuf = nullptr;
        const size_t n = (size_t)nblocks * HC_WAVES * 8;
        if (hipMalloc(&dbuf, n * sizeof(unsigned long long)) != hipSuccess) return;
        (void)hipMemset(dbuf, 0, n * sizeof(unsigned long long));
        c.stamps = dbuf;
        launch_form<kt, HC_WAVES, true>(c, nblocks, s);
        (void)hipStreamSynchronize(s);
        unsigned long long *hbuf = (unsigned long long *)malloc(n * sizeof(unsigned long long));
        (void)hipMemcpy(hbuf, dbuf, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double tot[8] = {0, 0, 0, 0, 0, 0, 0, 0}, redo = 0, nw = 0;
        for (size_t w = 0; w < (size_t)nblocks * HC_WAVES; ++w) {
            if (hbuf[w * 8 + 6] == 0) continue;   // (workgroup routed elsewhere)
            nw += 1;
            for (int i = 0; i < 7; ++i) tot[i] += (double)hbuf[w * 8 + i];
            redo += (double)(hbuf[w * 8 + 7] >> 20);
            tot[7] += (double)(hbuf[w * 8 + 7] & 0xfffff);
        }
        if (nw > 0) {
            const double st = tot[7] / nw * HC_QB;   // 32 x 32 blocks per wave
            fprintf(stderr, "K1 stamps (coarse screen, %d keys per tile, %d workgroups): %.2f re-done records per wave; cycles per wave and 32 x 32 block: "
                            "dma %.0f  chain + epilogue %.0f  redo %.0f  barrier %.0f  | whole kernel %.0f per block (%.0f blocks per wave)\n",
                    kt, nblocks, redo / nw, tot[0] / nw / st, tot[1] / nw / st, tot[3] / nw / st, tot[5] / nw / st, tot[6] / nw / st, st);
        }
        free(hbuf);
        (void)hipFree(dbuf);
        return;
    }
    launch_form<kt, HC_WAVES, false>(c, nblocks, s);
}

}  // namespace k1
}  // namespace dvm
