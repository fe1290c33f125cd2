// probe: within ONE wave's stream, how many independent VALU instructions hide in the gap of a
// dependent MFMA chain?  (f32 32x32x2: 64-cycle gap; bf16 32x32x16: 32-cycle gap)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int KV, bool BF16>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float a, float b) {
    f32x16 acc = {0};
    bf16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (__bf16)a; bv[i] = (__bf16)b; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (BF16) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < KV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(a), "v"(b));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += v[i];
    for (int i = 0; i < 16; ++i) r += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int KV, bool BF16> float run(float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KV, BF16>), dim3(256), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KV, BF16>), dim3(256), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    int iters = 1000;  // 16000 MFMAs per wave, one wave per SIMD
    printf("f32 MFMA + k VALU per gap:  k=0 %.3f  k=4 %.3f  k=8 %.3f  k=12 %.3f  k=16 %.3f  k=24 %.3f  k=32 %.3f ms\n",
           run<0, false>(out, iters), run<4, false>(out, iters), run<8, false>(out, iters), run<12, false>(out, iters),
           run<16, false>(out, iters), run<24, false>(out, iters), run<32, false>(out, iters));
    printf("bf16 MFMA + k VALU per gap: k=0 %.3f  k=2 %.3f  k=4 %.3f  k=6 %.3f  k=8 %.3f  k=12 %.3f  k=16 %.3f ms\n",
           run<0, true>(out, iters), run<2, true>(out, iters), run<4, true>(out, iters), run<6, true>(out, iters),
           run<8, true>(out, iters), run<12, true>(out, iters), run<16, true>(out, iters));
    return 0;
}
