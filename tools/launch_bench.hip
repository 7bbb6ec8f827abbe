// Launch-rate microbenchmark: N dependent tiny kernels on one stream; host enqueue time vs completion.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { float k[600]; };
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void tiny_big(float* p, Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += b.k[3]; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float* d; hipMalloc(&d, 256); hipMemset(d, 0, 256);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    Big b; for (int i = 0; i < 600; ++i) b.k[i] = 1.f;
    const int N = 32;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 5; ++rep) {
            hipStreamSynchronize(s); hipStreamSynchronize(s2);
            double t0 = now();
            for (int i = 0; i < N; ++i) {
                if (mode == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, s, d);
                if (mode == 1) hipLaunchKernelGGL(tiny_big, dim3(1), dim3(256), 0, s, d, b);
                if (mode == 2) hipLaunchKernelGGL(tiny, dim3(600), dim3(256), 0, s, d);
                if (mode == 3) {
                    hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, s, d);
                    if (i % 4 == 3) { hipEventRecord(ev, s); hipStreamWaitEvent(s2, ev, 0); hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, s2, d + 8); }
                }
            }
            double t1 = now();
            hipStreamSynchronize(s); hipStreamSynchronize(s2);
            double t2 = now();
            if (rep == 4) printf("mode %d: enqueue %.1f us (%.2f/launch), complete %.1f us (%.2f/launch)\n", mode, t1 - t0, (t1 - t0) / N, t2 - t0, (t2 - t0) / N);
        }
    }
    return 0;
}
