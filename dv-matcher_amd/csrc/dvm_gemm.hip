// dvm_gemm.hip — the 1x1 convolutions / linear layers of LG-Net as a k-ordered fp32 chain on
// v_mfma_f32_32x32x2_f32, with the layer's bias, residual, eval-mode BatchNorm affine and (Leaky)ReLU
// fused into the epilogue.
//
// Why not the BLAS library: the reference's nn.Conv1d(kernel_size=1) (models/model.py:506-529, the ff /
// q / k / v projections of 325-395 and SA_Layer 97-123) evaluated by one CPU thread is, bit for bit,
// acc = fma(w[co][k], x[k][n], acc) for k = 0..K-1 (verified against torch.matmul, which is that chain at
// every thread count; oneDNN's multi-threaded conv splits K across threads and is NOT reproducible
// against itself).  v_mfma_f32_32x32x2_f32 computes exactly this chain, so the activations that feed the
// feature-space kNNs carry no reduction-order noise relative to the reference's deterministic form, and
// results do not depend on batch size, tile choice or launch geometry.
//
// Mapping: D[i][j] = sum_k P[i][k] Q[k][j].  P is always K-contiguous ([row][k]).
//   point-major   (inference):        P = x [B*N][K],  Q = w [Co][K] (also K-contiguous), y [B*N][Co]
//   channel-major (training forward): P = w [Co][K],   Q = x[b] [K][N] (N-contiguous),    y [b][Co][N]
// A workgroup = 4 waves, each wave TM x TN accumulator tiles of 32x32 (16 VGPRs each); k-step 32 staged
// through two alternating LDS buffers with the even / odd k of a row de-interleaved so that a lane's half-wave (k parity) reads its
// four next operands with one ds_read_b128; the next k-step's global loads are in flight during the MFMAs.
// The fp32 matrix instruction issues at the vector rate (64 cycles per 32x32x2), so LDS and global traffic
// are far below their limits: the kernel is bound by MFMA issue.
#include <stdint.h>

#include "dvm_common.h"

#include <stdlib.h>

namespace dvm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GK = 32;       // k-step
constexpr int GLD = GK + 4;  // LDS row stride of a [row][evens | odds] tile
constexpr int GEMM_MAX_KB = 24;

struct LinArgs {
    const float *P, *Q;
    float *Y;
    const float *bias, *res, *alpha, *beta;  // per output channel (bias, alpha, beta); res laid out like Y
    int I, J, K;                             // rows of P, columns of Q, reduction length
    long q_bs, y_bs;                         // batch strides (channel-major); 0 otherwise
    long p_bs;                               // channel-major: batch stride of P (0: one weight matrix for every batch element)
    int ldy;
    float slope;                             // 1: no activation; 0: ReLU; < 0: ELU (alpha = 1); else LeakyReLU(slope)
    const float *post_res;                   // NULL, or laid out like Y: y = post_res + post_scale * (what the epilogue made)
    float post_scale;
    int tiles_i, tiles_j;
    int qvec;                                // channel-major: rows of Q are 16-byte aligned (N % 4 == 0)
    int yvec;                                // rows of Y / res / post_res and the per-channel vectors allow 16-byte accesses
    int kvec;                                // every K-block starts and ends on a multiple of 4
    int nkb, kb[GEMM_MAX_KB + 1];            // K-blocks [kb[b], kb[b+1]) of the reference's CPU sgemm (gemm_kblocks)
    const float *G;                          // point-major: per-shape row prefix [B][Cg] (NULL: none): row i of the
    int Cg, Nrow, ldp;                       // operand is [G[i / Nrow] (Cg) | P[i] (K - Cg)]; ldp = row stride of P
};

// K-blocking of the CPU sgemm behind torch.matmul / nn.Conv1d(k=1) (MKL / oneDNN, one thread; probed K = 5..3000, see
// oracle/dvm_oracle.c dvo_gemm_kblocks): blocks of 384 while more than 768 remain, then one block or two halves.  Every
// block is a k-ordered fma chain started at 0; block results are added in order.  LG-Net: K = 1152 -> 3 x 384,
// 768 -> 2 x 384, 512 -> 2 x 256, K <= 384 -> one chain.
static int gemm_kblocks(int K, int *starts) {
    int n = 0, k = 0;
    starts[0] = 0;
    while (K - k > 768 && n < GEMM_MAX_KB - 2) starts[++n] = (k += 384);
    if (K - k > 384 && n < GEMM_MAX_KB - 1) starts[++n] = (k += (K - k + 1) / 2);
    starts[++n] = K;
    return n;
}

// DMA (point-major only, every K-block a multiple of the k-step, no row prefix): both operand tiles go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write, the next k-step lands while this one's matrix instructions
// run), rows in their natural [row][32 k] layout, the 16-byte chunks of a row XOR-swizzled with the row number on the SOURCE
// address; a lane reads its two operands of a chunk (k = 4 c + h and 4 c + 2 + h) with one ds_read2_b32 — the inner loop holds
// matrix instructions and LDS reads only.
template <bool CM, int WM, int WN, int TM, int TN, bool DMA = false>
__global__ __launch_bounds__(256, 2) void linear_mfma_kernel(const LinArgs a) {
    static_assert(!DMA || !CM, "the DMA staging is for K-contiguous operands");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int QSZ = CM ? GK * BN : BN * GLD;
    constexpr int DMA_BUF = (BM + BN) * GK;          // floats of one buffer: [P rows | Q rows][32]
    constexpr int DMA_NI = (BM + BN) / 8 / 4;          // 1-KiB pieces (8 rows) per wave and k-step
    constexpr int PL = BM / 32;                     // f32x4 chunks of P per thread and k-step
    constexpr int QL = CM ? (GK * BN / 4) / 256 : BN / 32;
    // two LDS buffers per operand: the next k-step is written while the current one is read, one barrier per step
    extern __shared__ __attribute__((aligned(16))) float lds_gemm[];
    float *const Ps0 = lds_gemm, *const Qs0 = lds_gemm + 2 * BM * GLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: the workgroups of one XCD (blockIdx % 8) walk consecutive tiles, column tiles of a row
    // tile first, so the P rows a row tile streams are shared in that XCD's L2 by all its column tiles
    int t = blockIdx.x;
    const int total = a.tiles_i * a.tiles_j;
    if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);
    const int ti = t / a.tiles_j, tj = t % a.tiles_j;
    const int i0 = ti * BM, j0 = tj * BN, b = blockIdx.y;
    const float *__restrict__ P = a.P + (size_t)b * a.p_bs;
    const float *__restrict__ Q = a.Q + (size_t)b * a.q_bs;
    const int K = a.K, I = a.I, J = a.J;

    f32x4 pp[PL], pq[QL];
    // 4 consecutive k of one K-contiguous row, zero beyond the block's end (fma(0, 0, acc) == acc)
    auto ldk = [&](const float *row, int k, int kend) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < kend) {
            if (a.kvec) {
                v = *(const f32x4 *)(row + k);
            } else {
                v.x = row[k];
                if (k + 1 < kend) v.y = row[k + 1];
                if (k + 2 < kend) v.z = row[k + 2];
                if (k + 3 < kend) v.w = row[k + 3];
            }
        }
        return v;
    };
    auto fetch = [&](int k0, int kend) {
#pragma unroll
        for (int u = 0; u < PL; ++u) {
            const int r = (tid >> 3) + 32 * u, c = tid & 7;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const int i = i0 + r, k = k0 + 4 * c;
            if (i < I) {
                if (!CM && a.G && k < a.Cg)  // broadcast prefix (Cg and every block edge are multiples of 4 here)
                    v = ldk(a.G + (size_t)(i / a.Nrow) * a.Cg, k, kend);
                else
                    v = ldk(P + (size_t)i * a.ldp - (CM ? 0 : a.Cg), k, kend);
            }
            pp[u] = v;
        }
        if (CM) {
#pragma unroll
            for (int u = 0; u < QL; ++u) {
                const int e = tid + 256 * u, kk = e / (BN / 4), c = e % (BN / 4);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                const int j = j0 + 4 * c;
                if (k0 + kk < kend && j < J) {
                    const float *src = Q + (size_t)(k0 + kk) * J + j;
                    if (a.qvec && j + 3 < J) {
                        v = *(const f32x4 *)src;
                    } else {
                        v.x = src[0];
                        if (j + 1 < J) v.y = src[1];
                        if (j + 2 < J) v.z = src[2];
                        if (j + 3 < J) v.w = src[3];
                    }
                }
                pq[u] = v;
            }
        } else {
#pragma unroll
            for (int u = 0; u < QL; ++u) {
                const int r = (tid >> 3) + 32 * u, c = tid & 7;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (j0 + r < J) v = ldk(Q + (size_t)(j0 + r) * K, k0 + 4 * c, kend);
                pq[u] = v;
            }
        }
    };
    // Fast path of the fetch: per-thread row pointers made once; a k-step then costs one 16-byte load per chunk and no
    // address arithmetic or bounds test (the guarded lambda above compiles to ~30 VALU + 20 SALU and a branch per chunk —
    // as much issue time as the step's 16 matrix instructions, which share the vector issue slot on gfx950).  Rows beyond
    // the matrix are clamped to its last row: their products land in accumulator rows the epilogue never stores.
    const float *pfast[PL], *gfast[PL], *qfast[QL];
#pragma unroll
    for (int u = 0; u < PL; ++u) {
        const int r = (tid >> 3) + 32 * u, c = tid & 7;
        const int i = i0 + r < I ? i0 + r : I - 1;
        pfast[u] = P + (size_t)i * a.ldp - (CM ? 0 : a.Cg) + 4 * c;
        gfast[u] = (!CM && a.G) ? a.G + (size_t)(i / a.Nrow) * a.Cg + 4 * c : pfast[u];
    }
#pragma unroll
    for (int u = 0; u < QL; ++u) {
        if (CM) {
            const int e = tid + 256 * u, kk = e / (BN / 4), c = e % (BN / 4);
            qfast[u] = Q + (size_t)kk * J + j0 + 4 * c;
        } else {
            const int r = (tid >> 3) + 32 * u, c = tid & 7;
            const int j = j0 + r < J ? j0 + r : J - 1;
            qfast[u] = Q + (size_t)j * K + 4 * c;
        }
    }
    const float *dsrc[DMA ? DMA_NI : 1];   // DMA: this lane's source of piece e (row 8 (wave NI + e) + lane / 8 of [P rows | Q rows])
    int roff[DMA ? 8 : 1];                 // DMA: byte offset of chunk c in this lane's operand rows: r32 * 128 + ((c ^ (r32 & 7)) << 4) + 4 h
    if (DMA) {
#pragma unroll
        for (int e = 0; e < DMA_NI; ++e) {
            const int gr = 8 * (wave * DMA_NI + e) + (lane >> 3), c = (lane & 7) ^ (lane >> 3);
            if (gr < BM) {
                const int i = i0 + gr < I ? i0 + gr : I - 1;
                dsrc[e] = P + (size_t)i * a.ldp + 4 * c;
            } else {
                const int j = j0 + gr - BM < J ? j0 + gr - BM : J - 1;
                dsrc[e] = Q + (size_t)j * K + 4 * c;
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) roff[c] = r32 * 128 + ((c ^ (r32 & 7)) << 4) + 4 * h;
    }
    auto dma_issue = [&](int k0, int buf) {
        char *dst = (char *)lds_gemm + (size_t)buf * DMA_BUF * 4 + wave * DMA_NI * 1024;
#pragma unroll
        for (int e = 0; e < (DMA ? DMA_NI : 0); ++e)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(dsrc[e] + k0),
                                             (__attribute__((address_space(3))) void *)(dst + e * 1024), 16, 0, 0);
    };
    const bool q_interior = !CM || (a.qvec && j0 + BN <= J);
    auto fetch_step = [&](int k0, int kend) {
        const int Cg = CM ? 0 : a.Cg;
        if (a.kvec && q_interior && k0 + GK <= kend && (k0 >= Cg || k0 + GK <= Cg)) {
            const bool in_prefix = k0 < Cg;
#pragma unroll
            for (int u = 0; u < PL; ++u) pp[u] = *(const f32x4 *)((in_prefix ? gfast[u] : pfast[u]) + k0);
#pragma unroll
            for (int u = 0; u < QL; ++u) pq[u] = *(const f32x4 *)(qfast[u] + (CM ? (size_t)k0 * J : (size_t)k0));
        } else {
            fetch(k0, kend);
        }
    };
    auto stage = [&](int buf) {
        float *const Ps = Ps0 + buf * BM * GLD, *const Qs = Qs0 + buf * QSZ;
#pragma unroll
        for (int u = 0; u < PL; ++u) {
            const int r = (tid >> 3) + 32 * u, c = tid & 7;
            float2 ev = {pp[u].x, pp[u].z}, od = {pp[u].y, pp[u].w};
            *(float2 *)(Ps + r * GLD + 2 * c) = ev;
            *(float2 *)(Ps + r * GLD + GK / 2 + 2 * c) = od;
        }
        if (CM) {
#pragma unroll
            for (int u = 0; u < QL; ++u) {
                const int e = tid + 256 * u, kk = e / (BN / 4), c = e % (BN / 4);
                *(f32x4 *)(Qs + kk * BN + 4 * c) = pq[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < QL; ++u) {
                const int r = (tid >> 3) + 32 * u, c = tid & 7;
                float2 ev = {pq[u].x, pq[u].z}, od = {pq[u].y, pq[u].w};
                *(float2 *)(Qs + r * GLD + 2 * c) = ev;
                *(float2 *)(Qs + r * GLD + GK / 2 + 2 * c) = od;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    float *__restrict__ Y = a.Y + (size_t)b * a.y_bs;
    f32x16 tot[DMA ? TM : 1][DMA ? TN : 1];   // DMA: running total over the closed K-blocks

    // Flat walk over the k-steps of all K-blocks.  Step s+1's global loads are issued before step s's MFMAs and written to
    // the other LDS buffer after them; the single barrier at the end of a step both publishes that buffer and retires the
    // reads of the current one.
    // (the block edges live in the kernel-argument segment: a dynamically indexed a.kb[..] is a scalar memory load with its
    // wait — two per k-step at the head of the loop in the first version; the current block's end and the next one's are
    // carried in registers and re-read only when a block closes)
    int blk = 0, k0 = a.kb[0], kend = a.kb[1], cur = 0;
    if (DMA) {
        dma_issue(k0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave waits for ITS pieces, then the barrier publishes them
    } else {
        fetch_step(k0, kend);
        stage(0);
    }
    __syncthreads();
    while (blk < a.nkb) {
        int nblk = blk, nk = k0 + GK, nkend = kend;
        if (nk >= kend) {
            ++nblk;
            nk = kend;           // blocks are contiguous: kb[nblk] is the end of the block that closes
            if (nblk < a.nkb) nkend = a.kb[nblk + 1];
        }
        const bool has_next = nblk < a.nkb;
        if (has_next) {
            if (DMA) dma_issue(nk, cur ^ 1); else fetch_step(nk, nkend);   // (the other buffer was last read before the previous barrier)
        }
        if (DMA) {
            const char *const Pl = (const char *)lds_gemm + (size_t)cur * DMA_BUF * 4 + wm * TM * 32 * 128;
            const char *const Ql = (const char *)lds_gemm + (size_t)cur * DMA_BUF * 4 + (BM + wn * TN * 32) * 128;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float av[TM][2], bv[TN][2];
#pragma unroll
                for (int x = 0; x < TM; ++x) {
                    const float *ap = (const float *)(Pl + x * 32 * 128 + roff[c]);
                    av[x][0] = ap[0], av[x][1] = ap[2];
                }
#pragma unroll
                for (int y = 0; y < TN; ++y) {
                    const float *bp = (const float *)(Ql + y * 32 * 128 + roff[c]);
                    bv[y][0] = bp[0], bv[y][1] = bp[2];
                }
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int x = 0; x < TM; ++x)
#pragma unroll
                        for (int y = 0; y < TN; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[y][e], av[x][e], acc[x][y], 0, 0, 0);
            }
        } else {
            const float *const Ps = Ps0 + cur * BM * GLD, *const Qs = Qs0 + cur * QSZ;
            // the inner loop is MFMA + ds_read only: on gfx950 the fp32 matrix instruction and the vector ALU share the
            // issue slot, so every VALU instruction in here would cost matrix time.  All fragments of the k-step are
            // requested before the first matrix instruction (one-accumulator tiles: 32 VGPRs), so that the reads of the
            // later groups return under the earlier groups' matrix instructions instead of stalling between them.
            constexpr bool FRONT = (TM * TN == 1);
            f32x4 avf[FRONT ? GK / 8 : 1][TM], bvf[FRONT ? GK / 8 : 1][TN];
            if (FRONT) {
#pragma unroll
                for (int cc = 0; cc < GK / 8; ++cc) {
#pragma unroll
                    for (int x = 0; x < TM; ++x) avf[cc][x] = *(const f32x4 *)(Ps + ((wm * TM + x) * 32 + r32) * GLD + h * (GK / 2) + 4 * cc);
                    if (!CM) {
#pragma unroll
                        for (int y = 0; y < TN; ++y) bvf[cc][y] = *(const f32x4 *)(Qs + ((wn * TN + y) * 32 + r32) * GLD + h * (GK / 2) + 4 * cc);
                    }
                    // (keeps the reads in (A, B) pairs and above the matrix instructions: the scheduler otherwise sinks half
                    // of them below the first 8 MFMAs; in pair order the first MFMAs wait for the first pair only)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int cc = 0; cc < GK / 8; ++cc) {
                f32x4 av[TM], bv[TN];
#pragma unroll
                for (int x = 0; x < TM; ++x)
                    av[x] = FRONT ? avf[FRONT ? cc : 0][x] : *(const f32x4 *)(Ps + ((wm * TM + x) * 32 + r32) * GLD + h * (GK / 2) + 4 * cc);
                if (!CM) {
#pragma unroll
                    for (int y = 0; y < TN; ++y)
                        bv[y] = FRONT ? bvf[FRONT ? cc : 0][y] : *(const f32x4 *)(Qs + ((wn * TN + y) * 32 + r32) * GLD + h * (GK / 2) + 4 * cc);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float bs[TN];
#pragma unroll
                    for (int y = 0; y < TN; ++y) bs[y] = CM ? Qs[(2 * (4 * cc + e) + h) * BN + (wn * TN + y) * 32 + r32] : bv[y][e];
#pragma unroll
                    for (int x = 0; x < TM; ++x)
#pragma unroll
                        for (int y = 0; y < TN; ++y)   // D[column-tile index][row-tile index]: see the epilogue
                            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(bs[y], av[x][e], acc[x][y], 0, 0, 0);
                }
            }
        }
        if (DMA && a.nkb > 1 && nblk != blk) {
            // close the block: total += block, restart the chain at 0 — the DMA form has the registers for a second accumulator set
            const bool last = blk == a.nkb - 1;
#pragma unroll
            for (int x = 0; x < TM; ++x)
#pragma unroll
                for (int y = 0; y < TN; ++y)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float t = blk ? tot[DMA ? x : 0][DMA ? y : 0][r] + acc[x][y][r] : acc[x][y][r];
                        tot[DMA ? x : 0][DMA ? y : 0][r] = t;
                        acc[x][y][r] = last ? t : 0.f;
                    }
        } else if (a.nkb > 1 && nblk != blk) {
            // close the block: total += block, restart the chain at 0.  The running total lives in the output buffer
            // between blocks (every lane re-reads exactly the words it wrote) instead of a second accumulator set — 16
            // registers per tile that the large tiles do not have; 2 extra passes over the output at K = 1152.
            const bool last = blk == a.nkb - 1;
#pragma unroll
            for (int x = 0; x < TM; ++x)
#pragma unroll
                for (int y = 0; y < TN; ++y) {
                    const int i = i0 + (wm * TM + x) * 32 + r32;
                    if (i >= I) continue;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int j = j0 + (wn * TN + y) * 32 + 8 * g4 + 4 * h;
                        if (j >= J) continue;
                        const size_t off = (size_t)i * a.ldy + j;
                        f32x4 t = {acc[x][y][4 * g4], acc[x][y][4 * g4 + 1], acc[x][y][4 * g4 + 2], acc[x][y][4 * g4 + 3]};
                        if (a.yvec && j + 3 < J) {
                            if (blk) t = *(const f32x4 *)(Y + off) + t;
                            if (!last) *(f32x4 *)(Y + off) = t;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (j + e >= J) continue;
                                if (blk) t[e] = Y[off + e] + t[e];
                                if (!last) Y[off + e] = t[e];
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[x][y][4 * g4 + e] = last ? t[e] : 0.f;
                    }
                }
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else if (has_next) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
        blk = nblk;
        k0 = nk;
        kend = nkend;
    }

    // epilogue: conv bias (added after the chain, as oneDNN does), residual, y = fma(y, alpha, beta) (ATen's eval-mode
    // BatchNorm is exactly this fma with alpha = w / sqrt(var + eps), beta = fma(-mean, alpha, b)), activation.
    // The matrix instructions were issued with the operands' roles swapped (products commute, the chain over k is the same):
    // a lane owns ONE output row i and its registers 4 x 4 consecutive columns j, so the tile leaves as four 16-byte stores
    // per lane instead of sixteen 4-byte ones (the wide-output layers were bound by store issue: 33 MB in 44 us).
    const float slope = a.slope;
    const float *__restrict__ R = a.res ? a.res + (size_t)b * a.y_bs : nullptr;
    const float *__restrict__ R2 = a.post_res ? a.post_res + (size_t)b * a.y_bs : nullptr;
    const bool vec_ok = a.yvec != 0;
#pragma unroll
    for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y) {
            const int i = i0 + (wm * TM + x) * 32 + r32;
            if (i >= I) continue;
            float rb = 0.f, ra = 1.f, rt = 0.f;   // channel-major: per output row
            if (CM) {
                if (a.bias) rb = a.bias[i];
                if (a.alpha) ra = a.alpha[i], rt = a.beta[i];
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int j = j0 + (wn * TN + y) * 32 + 8 * g4 + 4 * h;
                if (j >= J) continue;
                const size_t off = (size_t)i * a.ldy + j;
                const bool full = vec_ok && j + 3 < J;
                f32x4 v = {acc[x][y][4 * g4], acc[x][y][4 * g4 + 1], acc[x][y][4 * g4 + 2], acc[x][y][4 * g4 + 3]};
                f32x4 cb = {rb, rb, rb, rb}, ca = {ra, ra, ra, ra}, ct = {rt, rt, rt, rt}, rr = {0.f, 0.f, 0.f, 0.f}, r2 = rr;
                if (full) {
                    if (!CM) {
                        if (a.bias) cb = *(const f32x4 *)(a.bias + j);
                        if (a.alpha) ca = *(const f32x4 *)(a.alpha + j), ct = *(const f32x4 *)(a.beta + j);
                    }
                    if (R) rr = *(const f32x4 *)(R + off);
                    if (R2) r2 = *(const f32x4 *)(R2 + off);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (j + e >= J) continue;
                        if (!CM) {
                            if (a.bias) cb[e] = a.bias[j + e];
                            if (a.alpha) ca[e] = a.alpha[j + e], ct[e] = a.beta[j + e];
                        }
                        if (R) rr[e] = R[off + e];
                        if (R2) r2[e] = R2[off + e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = v[e];
                    if (a.bias) t = t + cb[e];
                    if (R) t = t + rr[e];
                    if (a.alpha) t = fmaf(t, ca[e], ct[e]);
                    if (slope != 1.f)   // (ELU as torch evaluates it on fp32: exp(x) - 1, see dvm_deformer.hip::elu1)
                        t = t > 0.f ? t : (slope == 0.f ? 0.f : slope < 0.f ? __builtin_amdgcn_exp2f(t * 1.4426950408889634f) - 1.f : t * slope);
                    if (R2) t = __fadd_rn(__fmul_rn(t, a.post_scale), r2[e]);   // two roundings, like `conv(x) * s + r` in torch
                    v[e] = t;
                }
                if (full) {
                    *(f32x4 *)(Y + off) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (j + e < J) Y[off + e] = v[e];
                }
            }
        }
}

template <bool CM, int WM, int WN, int TM, int TN, bool DMA = false>
static void launch_cfg(LinArgs &a, int B, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    a.tiles_i = (a.I + BM - 1) / BM;
    a.tiles_j = (a.J + BN - 1) / BN;
    const uintptr_t bits = (uintptr_t)a.Y | (uintptr_t)a.res | (uintptr_t)a.post_res | (uintptr_t)a.bias | (uintptr_t)a.alpha |
                           (uintptr_t)a.beta | (uintptr_t)((size_t)a.y_bs * sizeof(float));
    a.yvec = (a.ldy % 4 == 0 && bits % 16 == 0) ? 1 : 0;
    constexpr int QSZ = CM ? GK * BN : BN * GLD;
    constexpr size_t lds = DMA ? (size_t)2 * (BM + BN) * GK * sizeof(float) : (size_t)2 * (BM * GLD + QSZ) * sizeof(float);
    ensure_dyn_lds((const void *)linear_mfma_kernel<CM, WM, WN, TM, TN, DMA>, (int)lds);
    hipLaunchKernelGGL((linear_mfma_kernel<CM, WM, WN, TM, TN, DMA>), dim3((unsigned)(a.tiles_i * a.tiles_j), (unsigned)B), dim3(256), lds, s, a);
}

// Tile choice (measured on MI355X at 16384 points, tools/bench_linear_cfg.py; us for conv 1152->384 / conv0 384->64 /
// conv6 512->128): 64x64 workgroup tiles — one 32x32 accumulator per wave, 3-4 workgroups per CU — win everywhere
// (196 / 20 / 35) over 64x128 (219 / 31 / 42) and 128x128 (340 / 50 / 94): this kernel stages through a single LDS
// buffer with a one-step register prefetch, so it relies on co-resident workgroups to cover each other's staging phases,
// and small tiles keep more of them resident.  At 64x64 the big layers are bound by L2 -> LDS operand traffic
// (0.9 GB for conv 1152->384 = 4.6 TB/s); the larger tiles that would cut it need LDS double buffering first.
// DVM_LINEAR_CFG=<index> forces an entry of the table (tuning knob).
struct TileCfg {
    int wm, wn, tm, tn;
};
// Point-major with LDS-DMA staging (round 3, profiles/r3_linear_cfg.txt; same shapes, us, register staging 64x64 -> DMA 64x64 /
// 128x64 / 64x128 / 128x128): conv 1152->384: 161 -> 133 / 131 / 129 / 163; conv1 256->512: 52 -> 47 / 53 / 51 / 46; qkv128: 27 -> 24;
// conv6 512->128: 30 -> 28; the 64-wide layers unchanged (10 - 16 us: launch- and epilogue-bound).  What the DMA form removes: the
// staging registers and their ds_write instructions (102 -> 61 VGPRs at 64x64), and the K-block totals' round trips through the
// output buffer (they live in a second accumulator set).  Default: DMA 64x64, DMA 64x128 from K = 768 on.
static int pick_cfg(int n, long rows_out_narrow, int K = 0) {
    const int c = options().linear_cfg;   // DVM_LINEAR_CFG: every configuration gives the same bits (tests/test_gpu_linear.py runs them all)
    if (c >= 0 && c < n) return c;
    if (rows_out_narrow <= 32) return 0;
    return n > 4 ? (K >= 768 ? 6 : 4) : 1;
}

static void launch_linear_impl(const float *x, const float *w, int B, int N, int K, int Co, int channel_major, const float *bias,
                               const float *res, const float *alpha, const float *beta, float slope, float *y, hipStream_t s,
                               const float *xg, int Cg, const float *post_res, float post_scale, long w_bs) {
    LinArgs a;
    a.p_bs = channel_major ? w_bs : 0;   // w_bs: a weight matrix PER batch element (channel-major only): y[b] = w[b] x[b]
    a.G = xg, a.Cg = xg ? Cg : 0, a.Nrow = N, a.ldp = K - a.Cg;
    a.post_res = post_res, a.post_scale = post_scale;
    a.bias = bias, a.res = res, a.alpha = alpha, a.beta = beta, a.slope = slope, a.K = K, a.Y = y;
    a.nkb = gemm_kblocks(K, a.kb);
    a.kvec = 1;
    for (int i = 0; i <= a.nkb; ++i)
        if (a.kb[i] % 4) a.kvec = 0;
    if (!channel_major) {
        a.P = x, a.Q = w, a.I = B * N, a.J = Co, a.q_bs = 0, a.y_bs = 0, a.ldy = Co, a.qvec = 1;
        bool dma_ok = a.kvec && !a.G && ((uintptr_t)x | (uintptr_t)w) % 16 == 0;
        for (int i = 0; i <= a.nkb; ++i) dma_ok = dma_ok && a.kb[i] % GK == 0;
        int cfg = pick_cfg(8, Co, K);
        if (cfg >= 4 && !dma_ok) {   // register-staged twin of the DMA tile: 64x64 -> 64x64, 128x64 -> 64x64, 64x128 -> 64x128, 128x128 -> 128x128
            static const int twin[4] = {1, 1, 2, 3};
            cfg = twin[cfg - 4];
        }
        switch (cfg) {   // {4,1,1,1} 128x32, {2,2,1,1} 64x64, {2,2,1,2} 64x128, {2,2,2,2} 128x128; 4..7: LDS-DMA staging, 64x64 / 128x64 / 64x128 / 128x128
            case 0: launch_cfg<false, 4, 1, 1, 1>(a, 1, s); break;
            case 1: launch_cfg<false, 2, 2, 1, 1>(a, 1, s); break;
            case 2: launch_cfg<false, 2, 2, 1, 2>(a, 1, s); break;
            case 3: launch_cfg<false, 2, 2, 2, 2>(a, 1, s); break;
            case 4: launch_cfg<false, 2, 2, 1, 1, true>(a, 1, s); break;
            case 5: launch_cfg<false, 2, 2, 2, 1, true>(a, 1, s); break;
            case 6: launch_cfg<false, 2, 2, 1, 2, true>(a, 1, s); break;
            default: launch_cfg<false, 2, 2, 2, 2, true>(a, 1, s); break;
        }
    } else {
        a.P = w, a.Q = x, a.I = Co, a.J = N, a.q_bs = (long)K * N, a.y_bs = (long)Co * N, a.ldy = N, a.qvec = (N % 4 == 0);
        switch (pick_cfg(4, Co)) {   // {1,4,1,1} 32x128, {2,2,1,1} 64x64, {2,2,2,1} 128x64, {2,2,2,2} 128x128
            case 0: launch_cfg<true, 1, 4, 1, 1>(a, B, s); break;
            case 1: launch_cfg<true, 2, 2, 1, 1>(a, B, s); break;
            case 2: launch_cfg<true, 2, 2, 2, 1>(a, B, s); break;
            default: launch_cfg<true, 2, 2, 2, 2>(a, B, s); break;
        }
    }
}

void launch_linear(const float *x, const float *w, int B, int N, int K, int Co, int channel_major, const float *bias,
                   const float *res, const float *alpha, const float *beta, float slope, float *y, hipStream_t s,
                   const float *xg = nullptr, int Cg = 0, const float *post_res = nullptr, float post_scale = 1.f) {
    launch_linear_impl(x, w, B, N, K, Co, channel_major, bias, res, alpha, beta, slope, y, s, xg, Cg, post_res, post_scale, 0);
}
// y[b] [Co][N] = w[b] [Co][K] x[b] [K][N]: the channel-major form with a weight matrix per batch element (a batched matrix product with
// both operands K-major / N-major as they lie; the k-chain order of dvm_linear_f32)
void launch_linear_bmm(const float *x, const float *w, int B, int N, int K, int Co, float *y, hipStream_t s) {
    launch_linear_impl(x, w, B, N, K, Co, 1, nullptr, nullptr, nullptr, nullptr, 1.f, y, s, nullptr, 0, nullptr, 1.f, (long)Co * K);
}

// ---------------------------------------------------------------- weight gradient of a point-major linear layer
// dW[co][k] = sum_r gy[r][co] * x[r][k]  (r over the B*N rows): a "TN" product whose reduction runs over the ROWS, so both
// operands are read exactly as they lie in memory — the A fragment of v_mfma_f32_32x32x2_f32 wants, per k-step, 32
// consecutive `co` of one row (lanes 0-31) and of the next row (lanes 32-63): contiguous 128-byte reads of gy; B likewise
// from x.  (rocBLAS runs this shape — reduction length 16384, outputs 64..1152 wide, K-strided operands — at ~1 TFLOP/s.)
// A workgroup = 4 waves = a 64 x 64 tile of dW over one chunk of rows; chunks are combined with fp32 atomics (dW zeroed
// by the caller).  Gradient sums carry no ordering contract.
// DET (dvm_set_deterministic): every row chunk writes its tile to its own slice of a scratch buffer and a second kernel adds
// the slices to dW in chunk order — the same sums, one fixed order, bit-reproducible from run to run.
constexpr int WG_ROWS = 32;   // rows staged per step
template <bool DET>
__global__ __launch_bounds__(256, 2) void linear_wgrad_kernel(const float *__restrict__ gy, const float *__restrict__ x, long R, int Co, int K,
                                                              long rchunk, float *__restrict__ dW) {
    __shared__ __attribute__((aligned(16))) float gs[2][WG_ROWS * 64], xs[2][WG_ROWS * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r32 = lane & 31, h = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    const int tiles_k = (K + 63) / 64;
    const int co0 = (blockIdx.x / tiles_k) * 64, k0 = (blockIdx.x % tiles_k) * 64;
    const long rbeg = (long)blockIdx.y * rchunk, rend = rbeg + rchunk < R ? rbeg + rchunk : R;
    if (!DET) {   // blockIdx.z: independent products of the same shape, operands and results back to back (launch_wgrad_batched)
        gy += (size_t)blockIdx.z * R * Co, x += (size_t)blockIdx.z * R * K, dW += (size_t)blockIdx.z * Co * K;
    }
    // staging: 32 rows x 64 floats per operand = 512 float4: 2 per thread
    f32x4 pg[2], px[2];
    auto fetch = [&](long r0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 256 * u, rr = e >> 4, c = (e & 15) * 4;
            const long r = r0 + rr;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            if (r < rend) {
                const float *ga = gy + (size_t)r * Co + co0 + c, *xa = x + (size_t)r * K + k0 + c;
                if (co0 + c + 3 < Co && (Co & 3) == 0) a = *(const f32x4 *)ga;
                else
                    for (int q = 0; q < 4; ++q) if (co0 + c + q < Co) a[q] = ga[q];
                if (k0 + c + 3 < K && (K & 3) == 0) b = *(const f32x4 *)xa;
                else
                    for (int q = 0; q < 4; ++q) if (k0 + c + q < K) b[q] = xa[q];
            }
            pg[u] = a, px[u] = b;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 256 * u;
            *(f32x4 *)(gs[buf] + 4 * e) = pg[u];
            *(f32x4 *)(xs[buf] + 4 * e) = px[u];
        }
    };
    // fast path of the fetch (interior tiles, whole row blocks): two 16-byte loads per operand through pointers made once —
    // the guarded lambda above costs ~35 branches and as many moves per step, beside 16 matrix instructions
    const bool interior = co0 + 64 <= Co && k0 + 64 <= K && (Co & 3) == 0 && (K & 3) == 0;
    const float *gfast[2], *xfast[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = tid + 256 * u, rr = e >> 4, c = (e & 15) * 4;
        gfast[u] = gy + (size_t)(rbeg + rr) * Co + co0 + c;
        xfast[u] = x + (size_t)(rbeg + rr) * K + k0 + c;
    }
    auto fetch_step = [&](long r0) {
        if (interior && r0 + WG_ROWS <= rend) {
            const size_t d = (size_t)(r0 - rbeg);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pg[u] = *(const f32x4 *)(gfast[u] + d * Co);
                px[u] = *(const f32x4 *)(xfast[u] + d * K);
            }
        } else {
            fetch(r0);
        }
    };
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int cur = 0;
    fetch_step(rbeg);
    stage(0);
    __syncthreads();
    for (long r0 = rbeg; r0 < rend; r0 += WG_ROWS) {
        const bool has_next = r0 + WG_ROWS < rend;
        if (has_next) fetch_step(r0 + WG_ROWS);
        const float *ga = gs[cur] + wi * 32 + r32, *xa = xs[cur] + wj * 32 + r32;
#pragma unroll
        for (int t = 0; t < WG_ROWS / 2; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[(2 * t + h) * 64], xa[(2 * t + h) * 64], acc, 0, 0, 0);
        if (has_next) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    const int k = k0 + wj * 32 + r32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < Co && k < K) {
            if (DET)
                dW[((size_t)blockIdx.y * Co + co) * K + k] = acc[r];   // dW = the scratch: [chunk][Co][K]
            else
                atomicAdd(dW + (size_t)co * K + k, acc[r]);
        }
    }
}
// dW[i] += part[0][i] + part[1][i] + ... in chunk order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, int chunks, long n, float *__restrict__ dW) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float t = part[i];
    for (int c = 1; c < chunks; ++c) t += part[(size_t)c * n + i];
    dW[i] += t;
}

}  // namespace dvm

using namespace dvm;

DVM_EXPORT int dvm_linear_prefix_f32(const float *xg, int Cg, const float *x, const float *w, int B, int N, int K, int Co,
                                     const float *bias, const float *res, const float *bn_alpha, const float *bn_beta, float slope,
                                     float *y, void *stream) {
    DVM_REQUIRE(xg && x && w && y, "dvm_linear_prefix_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1 && Co >= 1 && Cg >= 4 && Cg < K, "dvm_linear_prefix_f32: bad sizes (B=%d N=%d K=%d Cg=%d Co=%d)", B, N,
                K, Cg, Co);
    DVM_REQUIRE((bn_alpha == nullptr) == (bn_beta == nullptr), "dvm_linear_prefix_f32: bn_alpha and bn_beta go together");
    DVM_REQUIRE(K <= 384 * (GEMM_MAX_KB - 2), "dvm_linear_prefix_f32: K=%d exceeds %d", K, 384 * (GEMM_MAX_KB - 2));
    int kb[GEMM_MAX_KB + 1];
    const int nkb = gemm_kblocks(K, kb);
    bool vec = Cg % 4 == 0;
    for (int i = 0; i <= nkb; ++i) vec = vec && kb[i] % 4 == 0;
    DVM_REQUIRE(vec, "dvm_linear_prefix_f32: Cg=%d and the K-block edges of K=%d must be multiples of 4", Cg, K);
    launch_linear(x, w, B, N, K, Co, 0, bias, res, bn_alpha, bn_beta, slope, y, (hipStream_t)stream, xg, Cg);
    DVM_CHECK_LAUNCH("linear_prefix");
    return DVM_OK;
}

DVM_EXPORT int dvm_linear_f32(const float *x, const float *w, int B, int N, int K, int Co, int channel_major, const float *bias,
                              const float *res, const float *bn_alpha, const float *bn_beta, float slope, float *y, void *stream) {
    DVM_REQUIRE(x && w && y, "dvm_linear_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1 && K >= 1 && Co >= 1, "dvm_linear_f32: empty input (B=%d N=%d K=%d Co=%d)", B, N, K, Co);
    DVM_REQUIRE(K <= 384 * (GEMM_MAX_KB - 2), "dvm_linear_f32: K=%d exceeds %d", K, 384 * (GEMM_MAX_KB - 2));
    DVM_REQUIRE((bn_alpha == nullptr) == (bn_beta == nullptr), "dvm_linear_f32: bn_alpha and bn_beta go together");
    DVM_REQUIRE((long)B * N < (1L << 31) && (long)B * N * (Co > K ? Co : K) < (1L << 40), "dvm_linear_f32: too large");
    launch_linear(x, w, B, N, K, Co, channel_major, bias, res, bn_alpha, bn_beta, slope, y, (hipStream_t)stream);
    DVM_CHECK_LAUNCH("linear");
    return DVM_OK;
}

DVM_EXPORT int dvm_linear_scaled_residual_f32(const float *x, const float *w, int B, int N, int K, int Co, int channel_major, const float *bias,
                                              float scale, const float *res, float *y, void *stream) {
    DVM_REQUIRE(x && w && y && res, "dvm_linear_scaled_residual_f32: null pointer");
    DVM_REQUIRE(B >= 1 && N >= 1 && K >= 1 && Co >= 1, "dvm_linear_scaled_residual_f32: empty input (B=%d N=%d K=%d Co=%d)", B, N, K, Co);
    DVM_REQUIRE(K <= 384 * (GEMM_MAX_KB - 2), "dvm_linear_scaled_residual_f32: K=%d exceeds %d", K, 384 * (GEMM_MAX_KB - 2));
    DVM_REQUIRE((long)B * N < (1L << 31) && (long)B * N * (Co > K ? Co : K) < (1L << 40), "dvm_linear_scaled_residual_f32: too large");
    launch_linear(x, w, B, N, K, Co, channel_major, bias, nullptr, nullptr, nullptr, 1.f, y, (hipStream_t)stream, nullptr, 0, res, scale);
    DVM_CHECK_LAUNCH("linear_scaled_residual");
    return DVM_OK;
}

namespace dvm {
static long wgrad_chunks(long R, int Co, int K, long &rchunk) {
    const int tiles = ((Co + 63) / 64) * ((K + 63) / 64);
    // row chunks until ~1024 workgroups, each at least 256 rows (measured at R = 16384: 64 chunks of 256 rows beat 256
    // chunks of 64 on the one-tile layers, 10.7 vs 13.6 us — four times the atomics — and 16 chunks beat 8 on the
    // 108-tile conv, 152 vs 172 us)
    long chunks = 1;
    while (tiles * chunks < 1024 && R / (chunks * 2) >= 256) chunks *= 2;
    rchunk = (R + chunks - 1) / chunks;
    rchunk = (rchunk + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
    return (R + rchunk - 1) / rchunk;
}
}  // namespace dvm

namespace dvm {
// dW[b] [Co][K] += gy[b]^T x[b] for nb products of one shape, operands and results back to back; row chunks combined with fp32 atomics
// (dW zeroed by the caller)
void launch_wgrad_batched(const float *gy, const float *x, int nb, long R, int Co, int K, float *dW, hipStream_t s) {
    const int tiles = ((Co + 63) / 64) * ((K + 63) / 64);
    long chunks = 1;
    while ((long)tiles * chunks * nb < 1024 && R / (chunks * 2) >= 256) chunks *= 2;
    long rchunk = (R + chunks - 1) / chunks;
    rchunk = (rchunk + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
    chunks = (R + rchunk - 1) / rchunk;
    hipLaunchKernelGGL(linear_wgrad_kernel<false>, dim3((unsigned)tiles, (unsigned)chunks, (unsigned)nb), dim3(256), 0, s, gy, x, R, Co, K, rchunk, dW);
}
}  // namespace dvm

DVM_EXPORT size_t dvm_linear_wgrad_workspace_bytes(long R, int Co, int K) {
    if (R < 1 || Co < 1 || K < 1) return 0;
    long rchunk;
    return align_up((size_t)wgrad_chunks(R, Co, K, rchunk) * Co * K * sizeof(float));
}

// ws == NULL or the deterministic mode off: row chunks combined with fp32 atomics; with a workspace AND dvm_set_deterministic(1):
// per-chunk partial tiles added in chunk order
DVM_EXPORT int dvm_linear_wgrad_ws_f32(const float *gy, const float *x, long R, int Co, int K, float *dW, void *ws, size_t ws_bytes, void *stream) {
    DVM_REQUIRE(gy && x && dW, "dvm_linear_wgrad_f32: null pointer");
    DVM_REQUIRE(R >= 1 && Co >= 1 && K >= 1, "dvm_linear_wgrad_f32: empty input (R=%ld Co=%d K=%d)", R, Co, K);
    const int tiles = ((Co + 63) / 64) * ((K + 63) / 64);
    long rchunk;
    const long chunks = wgrad_chunks(R, Co, K, rchunk);
    DVM_REQUIRE(chunks <= 65535, "dvm_linear_wgrad_f32: too many row chunks");
    hipStream_t s = (hipStream_t)stream;
    if (ws && deterministic()) {
        const size_t need = (size_t)chunks * Co * K * sizeof(float);
        if (ws_bytes < need) {
            set_error("dvm_linear_wgrad_ws_f32: workspace too small (%zu < %zu)", ws_bytes, need);
            return DVM_ENOSPACE;
        }
        hipLaunchKernelGGL(linear_wgrad_kernel<true>, dim3((unsigned)tiles, (unsigned)chunks), dim3(256), 0, s, gy, x, R, Co, K, rchunk, (float *)ws);
        const long n = (long)Co * K;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)ws, (int)chunks, n, dW);
    } else {
        hipLaunchKernelGGL(linear_wgrad_kernel<false>, dim3((unsigned)tiles, (unsigned)chunks), dim3(256), 0, s, gy, x, R, Co, K, rchunk, dW);
    }
    DVM_CHECK_LAUNCH("linear_wgrad");
    return DVM_OK;
}

DVM_EXPORT int dvm_linear_wgrad_f32(const float *gy, const float *x, long R, int Co, int K, float *dW, void *stream) {
    return dvm_linear_wgrad_ws_f32(gy, x, R, Co, K, dW, nullptr, 0, stream);
}
