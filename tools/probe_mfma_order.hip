// probe: what exactly does v_mfma_f32_32x32x2_f32 compute? (run on the GPU box)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, const float* C, float* D, int steps) {
    // A[32][2*steps], B[2*steps][32], C/D [32][32]
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) { int row = (i & 3) + 8 * (i >> 2) + 4 * h; acc[i] = C[row * 32 + r]; }
    for (int s = 0; s < steps; ++s) {
        float a = A[r * 2 * steps + 2 * s + h], b = B[(2 * s + h) * 32 + r];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) { int row = (i & 3) + 8 * (i >> 2) + 4 * h; D[row * 32 + r] = acc[i]; }
}
int main() {
    for (int steps : {1, 64}) {
        int K = 2 * steps;
        float *A, *B, *C, *D;
        hipMallocManaged(&A, 32 * K * 4); hipMallocManaged(&B, 32 * K * 4); hipMallocManaged(&C, 4096); hipMallocManaged(&D, 4096);
        srand(1);
        for (int i = 0; i < 32 * K; ++i) { A[i] = (float)rand() / RAND_MAX * 2 - 1; B[i] = (float)rand() / RAND_MAX * 2 - 1; }
        for (int i = 0; i < 1024; ++i) C[i] = steps == 1 ? ((float)rand() / RAND_MAX * 200 - 100) : 0.f;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, C, D, steps);
        hipDeviceSynchronize();
        int m01 = 0, m10 = 0, mex = 0, mpair = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            float c01 = C[i * 32 + j], c10 = c01, cex = c01, cp = c01;
            for (int s = 0; s < steps; ++s) {
                float a0 = A[i * K + 2 * s], a1 = A[i * K + 2 * s + 1], b0 = B[(2 * s) * 32 + j], b1 = B[(2 * s + 1) * 32 + j];
                c01 = fmaf(a1, b1, fmaf(a0, b0, c01));
                c10 = fmaf(a0, b0, fmaf(a1, b1, c10));
                cex = (float)((double)a0 * b0 + (double)a1 * b1 + (double)cex);  // single rounding (approx: double is enough)
                cp = fmaf(a0, b0, cp); cp = cp + a1 * b1;                         // product rounded separately
            }
            float d = D[i * 32 + j];
            m01 += d != c01; m10 += d != c10; mex += d != cex; mpair += d != cp;
        }
        printf("steps=%d mismatches: k0-then-k1 %d, k1-then-k0 %d, single-rounding %d, unfused-second %d (of 1024)\n", steps, m01, m10, mex, mpair);
    }
    return 0;
}
