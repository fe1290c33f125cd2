// ubench: how many vector instructions fit into the gaps of a dependent v_mfma_f32_32x32x16_f16 chain for free?
// One instruction stream per wave, every instruction an asm volatile statement (strict program order):
//   INTERLEAVED: { 1 matrix instruction, K vector instructions } x 8 per iteration
//   CLUMPED    : { 8 matrix instructions, 8 K vector instructions } per iteration (what hipcc's scheduler emits for the K1 sweep)
// with 1 or 2 waves per SIMD (256 / 512 threads per workgroup, one workgroup per CU), vector instruction = v_min_i32 on
// independent registers or v_min_f64 / v_max_f64 pairs (the sorted-list compare-swap).
// Output: ns per matrix instruction per wave for every (form, K, waves, type) -> tools/gpu/ubench_gap.sh prints a table.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int K, int FORM, int TYPE>
__global__ void gap_kernel(float *out, int iters, float seed) {
    f32x16 acc = {0};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(seed + i), b[i] = (_Float16)(seed * 0.5f);
    int v[8];
    double d[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 7 + i, d[i] = (double)(threadIdx.x + i);
    const int c = (int)seed + 3;
    const double dc = (double)seed + 3.0;
    for (int it = 0; it < iters; ++it) {
        if (FORM == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    if (TYPE == 0) asm volatile("v_min_i32 %0, %0, %1" : "+v"(v[j % 8]) : "v"(c));
                    else if (j & 1) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[j % 8]) : "v"(dc));
                    else asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[j % 8]) : "v"(dc));
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < 8 * K; ++j) {
                if (TYPE == 0) asm volatile("v_min_i32 %0, %0, %1" : "+v"(v[j % 8]) : "v"(c));
                else if (j & 1) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[j % 8]) : "v"(dc));
                else asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[j % 8]) : "v"(dc));
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc[i];
    for (int i = 0; i < 8; ++i) r += (float)v[i] + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int K, int FORM, int TYPE>
void run(float *out, int threads) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL((gap_kernel<K, FORM, TYPE>), dim3(256), dim3(threads), 0, 0, out, 100, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((gap_kernel<K, FORM, TYPE>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("form %s  type %s  waves/SIMD %d  K %2d : %7.2f ns per matrix instruction per wave (%.3f ms)\n", FORM ? "clumped    " : "interleaved",
           TYPE ? "f64minmax" : "i32min   ", threads / 256, K, ms * 1e6 / (iters * 8.0), ms);
}

template <int FORM, int TYPE>
void sweep(float *out, int threads) {
    run<0, FORM, TYPE>(out, threads);
    run<2, FORM, TYPE>(out, threads);
    run<4, FORM, TYPE>(out, threads);
    run<5, FORM, TYPE>(out, threads);
    run<6, FORM, TYPE>(out, threads);
    run<8, FORM, TYPE>(out, threads);
    run<12, FORM, TYPE>(out, threads);
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    for (int threads : {256, 512}) {
        sweep<0, 0>(out, threads);
        sweep<1, 0>(out, threads);
        sweep<0, 1>(out, threads);
        sweep<1, 1>(out, threads);
    }
    return 0;
}
