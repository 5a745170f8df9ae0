// What does a fork of a side stream cost the main stream?  Chain of N ~10 us kernels on the main stream; after each one a side
// stream is told "that kernel is done" and runs one short kernel.  Variants: no fork, device-only events (what the engine uses),
// hipStreamWriteValue32 / hipStreamWaitValue32 on signal memory, and (round 5) the event carried by the kernel's OWN dispatch packet
// (hipExtLaunchKernelGGL stop event: no separate marker packet on the main stream).  Prints the main chain's wall time per kernel.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.000001f + 0.5f;
    p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
int main() {
    const int N = 64, REP = 20;
    float *a, *b;
    CK(hipMalloc(&a, 1 << 24)); CK(hipMalloc(&b, 1 << 24));
    hipStream_t m, s;
    CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ev[N];
    for (int i = 0; i < N; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming | hipEventDisableSystemFence));
    unsigned* sig = nullptr;
    hipError_t se = hipExtMallocWithFlags((void**)&sig, 4096, hipMallocSignalMemory);
    printf("signal memory: %s\n", hipGetErrorString(se));
    if (se != hipSuccess) { (void)hipGetLastError(); se = hipMalloc((void**)&sig, 4096); printf("plain device memory instead: %s\n", hipGetErrorString(se)); }
    if (se == hipSuccess) CK(hipMemset(sig, 0, 4096));
    unsigned counter = 0;
    for (int mode = 0; mode < 4; ++mode) {
        if (mode == 2 && se != hipSuccess) continue;
        double best = 1e9;
        for (int r = 0; r < REP; ++r) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int i = 0; i < N; ++i) {
                if (mode == 3) { hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, m, nullptr, ev[i], 0, a, 2000); CK(hipStreamWaitEvent(s, ev[i], 0)); }
                else spin<<<256, 256, 0, m>>>(a, 2000);
                if (mode == 1) { CK(hipEventRecord(ev[i], m)); CK(hipStreamWaitEvent(s, ev[i], 0)); }
                if (mode == 2) {
                    ++counter;
                    hipError_t e1 = hipStreamWriteValue32(m, sig, counter, 0);
                    hipError_t e2 = hipStreamWaitValue32(s, sig, counter, hipStreamWaitValueGte, 0xffffffffu);
                    if (e1 != hipSuccess || e2 != hipSuccess) { printf("write/wait value: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2)); return 0; }
                }
                if (mode) spin<<<64, 256, 0, s>>>(b, 500);
            }
            CK(hipStreamSynchronize(m));
            auto t1 = std::chrono::high_resolution_clock::now();
            CK(hipStreamSynchronize(s));
            double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
            if (us < best) best = us;
        }
        printf("mode %d (%s): %.2f us per main-stream kernel\n", mode, mode == 0 ? "no fork" : (mode == 1 ? "events" : (mode == 2 ? "write/wait value" : "stop event of the launch itself")), best);
    }
    return 0;
}
