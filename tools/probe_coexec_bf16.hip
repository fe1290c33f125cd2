// probe: does a bf16 MFMA chain in one wave co-execute with VALU work in the co-resident wave?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, float a, float b) {
    int wave = threadIdx.x >> 6;
    bool do_mfma = MODE == 0 || ((MODE == 2 || MODE == 3) && wave < 4);
    bool do_valu = MODE == 1 || ((MODE == 2 || MODE == 4) && wave >= 4);
    f32x16 acc = {0};
    bf16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (__bf16)a; bv[i] = (__bf16)b; }
    float v0 = a, v1 = b, v2 = a + b, v3 = a - b;
    if (do_mfma) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
        }
    } else if (do_valu) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) { v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b); }
        }
    }
    float r = v0 + v1 + v2 + v3;
    for (int i = 0; i < 16; ++i) r += acc[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int MODE> float run(float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    int iters = 2000;
    printf("all-bf16-MFMA (2 waves/SIMD): %.3f ms\n", run<0>(out, iters));
    printf("all-VALU (2 waves/SIMD): %.3f ms\n", run<1>(out, iters));
    printf("bf16-MFMA waves + VALU waves: %.3f ms\n", run<2>(out, iters));
    printf("bf16-MFMA waves only (1/SIMD): %.3f ms\n", run<3>(out, iters));
    printf("VALU waves only (1/SIMD): %.3f ms\n", run<4>(out, iters));
    return 0;
}
