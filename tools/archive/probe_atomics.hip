// Probe (test infrastructure): cost of device-scope int64 atomics used as a deterministic cross-block reduction.
//   hipcc --offload-arch=gfx950 -O3 tools/archive/probe_atomics.hip -o tools/probe_atomics && tools/probe_atomics
// 8192 blocks (the tile count of a full-resolution conv) x 128 no-return 64-bit adds each, into S slices of 128 addresses.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_atomic(long long* acc, int slices) {
    const int t = threadIdx.x;
    if (t < 128) {
        const long long v = (long long)(blockIdx.x * 131 + t);
        atomicAdd(reinterpret_cast<unsigned long long*>(acc + (size_t)(blockIdx.x % slices) * 128 + t), (unsigned long long)v);
    }
}
__global__ void k_plain(float* out) {
    const int t = threadIdx.x;
    if (t < 128) out[(size_t)blockIdx.x * 128 + t] = (float)(blockIdx.x * 131 + t);
}
int main() {
    long long* acc; float* out;
    hipMalloc(&acc, 8192 * 128 * 8); hipMalloc(&out, 8192 * 128 * 4);
    hipMemset(acc, 0, 8192 * 128 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int slices : {1, 8, 64, 512, 8192}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_atomic, dim3(8192), dim3(256), 0, 0, acc, slices);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("int64 atomics, %5d slices: %7.2f us per launch (1M atomics)\n", slices, ms * 50);
        }
    }
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_plain, dim3(8192), dim3(256), 0, 0, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("plain coalesced stores of the same count: %7.2f us per launch\n", ms * 50);
    return 0;
}
