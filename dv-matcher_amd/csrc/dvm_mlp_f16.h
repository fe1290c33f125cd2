// dvm_mlp_f16.h — what the Deformer MLP on the 16-bit matrix cores (dvm_mlp_f16.hip) shares with the kernels that produce its
// input rows (dvm_deformer.hip): the two-plane fp16 row layout of z and the split itself.
// (reference models/model.py:433-452, 464-478)
#pragma once
#include "dvm_common.h"

namespace dvm {

constexpr int MH_K0 = 272;                       // z's 262 columns padded to a multiple of 16
constexpr int MH_SZ = 2 * MH_K0 * 2 + 16;        // 1104 B: one z row as two fp16 planes (h at 0, m at 2 * MH_K0 bytes) + 16 B
constexpr float MH_SA = 32.f, MH_SW = 256.f;     // activation / weight scales (powers of two)
constexpr int MH_NODES = 64;                     // rows per workgroup pass (the plane buffer is padded to a multiple of it)

// The PLANE form of z (input of the persistent kernel): row n = MH_SZ bytes in exactly the layout the kernel keeps in LDS, so
// that a block of 64 rows goes global -> LDS as one flat LDS-DMA stream.  Column order (the contraction index of layer 0 — any
// order, as long as the packed weights use the same): [g_src 0..127 | g_corr 128..255 | v_src 256..258 | v_corr 259..261 | 0 x 10],
// i.e. the two 128-wide blocks first: a thread's four features are one aligned 8-byte store per plane.
__host__ __device__ constexpr int mh_zcol_of_plane_col(int c) {   // z column (reference order) held by plane column c; -1: padding
    return c < 128 ? 3 + c : c < 256 ? 134 + (c - 128) : c < 259 ? c - 256 : c < 262 ? 131 + (c - 259) : -1;
}

// Two values -> their packed fp16 planes: h = rn16(a), m = rn16(a - h) (a - h is exact in fp32), three instructions per pair:
// the packed round-to-nearest conversion (gfx950) and one v_fma_mix per value, which reads the fp16 h directly and writes its
// half of m.  (The compiler's form of the split is convert, convert back, subtract, convert, pack: 4.5 per value.)
__device__ __forceinline__ void split2x2(float a0, float a1, unsigned &h, unsigned &m) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a0), "v"(a1));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(m) : "v"(a0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(m) : "v"(a1), "v"(h));
}

size_t mlp_zplane_bytes(int rows);   // bytes of the plane form of `rows` z rows (padded to whole 64-row blocks)
size_t mlp_zplane_row_bytes();       // MH_SZ

}  // namespace dvm
