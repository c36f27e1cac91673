// What does an event record between two dependent kernels cost the recording stream, and does attaching the event to the
// kernel's own dispatch packet (hipExtLaunchKernelGGL stopEvent) avoid it?   hipcc --offload-arch=gfx950 -O2 tools/archive/probe_event.hip -o tools/probe_event
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
__global__ void work(float* p, int n, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = p[i % n];
    for (int k = 0; k < iters; k++) v = v * 1.0001f + 0.5f;
    p[i % n] = v;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    float* p; const int n = 1 << 22;
    CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4));
    hipStream_t s, side; CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, -1)); CK(hipStreamCreate(&side));
    const int R = 2000;
    hipEvent_t evs[R];
    for (int i = 0; i < R; i++) CK(hipEventCreateWithFlags(&evs[i], hipEventDisableTiming));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const dim3 g(2048), b(256);
    for (int mode = 0; mode < 5; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, s));
            for (int i = 0; i < R; i++) {
                if (mode == 2 || mode == 4) hipExtLaunchKernelGGL(work, g, b, 0, s, nullptr, evs[i], 0, p, n, 64);
                else hipLaunchKernelGGL(work, g, b, 0, s, p, n, 64);
                if (mode == 1 || mode == 3) CK(hipEventRecord(evs[i], s));
                if (mode == 3 || mode == 4) { CK(hipStreamWaitEvent(side, evs[i], 0)); hipLaunchKernelGGL(work, dim3(64), b, 0, side, p + (n >> 1), n >> 1, 16); }
                hipLaunchKernelGGL(work, g, b, 0, s, p, n, 64);
            }
            CK(hipEventRecord(t1, s));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            if (rep) printf("mode %d (%s): %.2f us per kernel pair\n", mode,
                            mode == 0 ? "A;B" : mode == 1 ? "A;record;B" : mode == 2 ? "A(stopEvent);B" : mode == 3 ? "A;record;side waits+runs;B" : "A(stopEvent);side waits+runs;B",
                            ms * 1e3 / R);
        }
    }
    return 0;
}
