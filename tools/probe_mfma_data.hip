#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// C[key][query] = sum_k key[k] * (-2 q[k]),  A = keys, B = -2*queries
__global__ void k(const float* Q, const float* Kf, float* D, int scaleq) {
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 acc = {0};
    for (int s = 0; s < 64; ++s) {
        float a = Kf[r * 128 + 2 * s + h];
        float b = Q[r * 128 + 2 * s + h]; if (scaleq) b = -2.f * b;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) { int row = (i & 3) + 8 * (i >> 2) + 4 * h; D[row * 32 + r] = acc[i]; }
}
int main() {
    float *Q, *Kf, *D;
    hipMallocManaged(&Q, 32 * 128 * 4); hipMallocManaged(&Kf, 32 * 128 * 4); hipMallocManaged(&D, 4096);
    // inputs generated here (no binary fixtures in the tree): a 32-bit LCG mapped to roughly N(0,1)-scaled values
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) / 16777216.0f - 0.5f) * 3.4f; };
    for (int i = 0; i < 32 * 128; ++i) Q[i] = rnd();
    for (int i = 0; i < 32 * 128; ++i) Kf[i] = rnd();
    for (int scaleq = 0; scaleq < 2; ++scaleq) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, Q, Kf, D, scaleq);
        hipDeviceSynchronize();
        int mm = 0, mm2 = 0; double maxrel = 0;
        for (int key = 0; key < 32; ++key) for (int q = 0; q < 32; ++q) {
            float c = 0.f, c2 = 0.f;
            for (int kk = 0; kk < 128; ++kk) {
                float b = Q[q * 128 + kk]; if (scaleq) b = -2.f * b;
                c = fmaf(Kf[key * 128 + kk], b, c);
                float p = Kf[key * 128 + kk] * b; c2 = c2 + p;
            }
            float d = D[key * 32 + q];
            mm += d != c; mm2 += d != c2;
            if (d != c) { double rel = fabs((double)d - c) / fabs(c); if (rel > maxrel) maxrel = rel; }
        }
        printf("scaleq=%d: mismatches vs fmaf chain %d / 1024 (vs unfused chain %d), max rel %.3g\n", scaleq, mm, mm2, maxrel);
    }
    return 0;
}
