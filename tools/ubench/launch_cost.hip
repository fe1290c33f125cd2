// Host cost of the runtime calls the native nodes are made of (round 6): back-to-back kernel launches with a 200-byte argument
// struct on one stream / round-robin on three, an event record + stream wait pair, a small hipMemsetAsync.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/launch_cost.hip -o tools/ubench/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Args { float x[50]; };
__global__ void empty_kernel(Args a, float *out) { if (a.x[0] == 12345.f) out[0] = a.x[1]; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s[3];
    for (auto &q : s) hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    float *buf;
    hipMalloc(&buf, 1 << 20);
    Args a{};
    const int n = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[0], a, buf);
        double t1 = now();
        hipDeviceSynchronize();
        double t2 = now();
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[i % 3], a, buf);
        double t3 = now();
        hipDeviceSynchronize();
        double t4 = now();
        for (int i = 0; i < n; ++i) { hipEventRecord(ev, s[0]); hipStreamWaitEvent(s[1], ev, 0); }
        double t5 = now();
        hipDeviceSynchronize();
        double t6 = now();
        for (int i = 0; i < n; ++i) hipMemsetAsync(buf, 0, 1024, s[0]);
        double t7 = now();
        hipDeviceSynchronize();
        double t8 = now();
        for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[0], a, buf); hipEventRecord(ev, s[0]); hipStreamWaitEvent(s[1], ev, 0); hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[1], a, buf); }
        double t9 = now();
        hipDeviceSynchronize();
        double t10 = now();
        if (rep == 1)
            printf("host us per call: launch (1 stream) %.2f [drain %.0f us]; launch (3 streams) %.2f [drain %.0f]; record+wait %.2f [drain %.0f]; memsetAsync 1 KB %.2f [drain %.0f]; "
                   "launch, record, wait, launch on the other stream %.2f per group [drain %.0f]\n",
                   (t1 - t0) / n, t2 - t1, (t3 - t2) / n, t4 - t3, (t5 - t4) / n, t6 - t5, (t7 - t6) / n, t8 - t7, (t9 - t8) / n, t10 - t9);
    }
    return 0;
}
