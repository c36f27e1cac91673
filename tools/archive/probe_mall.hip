// Does the order in which a consumer walks a tensor matter for the 256 MB Infinity Cache?  A producer writes N MB front to
// back; a consumer then reads it front to back (the oldest lines first: gone if N > cache) or back to front (newest first).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void writer(uint4* p, size_t units_per_block) {
    uint4* q = p + (size_t)blockIdx.x * units_per_block;
    for (size_t i = threadIdx.x; i < units_per_block; i += blockDim.x) q[i] = make_uint4(i, blockIdx.x, 3, 4);
}
__global__ void reader(const uint4* p, size_t units_per_block, int reverse, unsigned* out) {
    const size_t b = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const uint4* q = p + b * units_per_block;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i < units_per_block; i += blockDim.x) { uint4 v = q[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    unsigned* out; hipMalloc(&out, 4);
    uint4* buf; hipMalloc(&buf, (size_t)1200 << 20);
    uint4* other; hipMalloc(&other, (size_t)600 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int sizes[] = {64, 134, 200, 268, 400, 536};
    for (int s : sizes) {
        const size_t bytes = (size_t)s << 20, upb = 4096;            // 64 KB per block
        const int blocks = (int)(bytes / 16 / upb);
        for (int rev = 0; rev < 2; rev++) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                hipLaunchKernelGGL(writer, dim3((600 << 20) / 16 / upb), dim3(256), 0, 0, other, upb);   // flush the cache with other data
                hipLaunchKernelGGL(writer, dim3(blocks), dim3(256), 0, 0, buf, upb);
                hipEventRecord(e0);
                hipLaunchKernelGGL(reader, dim3(blocks), dim3(256), 0, 0, buf, upb, rev, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("%4d MB written front->back, read %s: %7.1f us  %6.2f TB/s\n", s, rev ? "back->front" : "front->back", best * 1e3, bytes / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
