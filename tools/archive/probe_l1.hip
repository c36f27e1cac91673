// L1 / L2 vector-load bandwidth probe: every wave streams 16 B/lane loads over a working set of `ws` bytes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void __launch_bounds__(256) rd(const uint4* __restrict__ buf, unsigned mask_units, int iters, uint4* out, int per_block_region) {
    const unsigned lane = threadIdx.x;
    unsigned base = per_block_region ? (blockIdx.x * (mask_units + 1)) : 0;
    uint4 acc = make_uint4(0, 0, 0, 0);
    unsigned idx = lane;
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            uint4 v = buf[base + ((idx + u * 256) & mask_units)];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
        idx += 8 * 256;
    }
    if (acc.x == 0x12345678) out[blockIdx.x * 256 + lane] = acc;
}
int main() {
    const size_t total = 512u << 20;
    uint4* buf; uint4* out;
    hipMalloc(&buf, total); hipMalloc(&out, 1 << 24);
    hipMemset(buf, 1, total);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int nblk = 256 * 4, iters = 2000;
    for (int mode = 0; mode < 2; mode++)
    for (size_t ws = 4096; ws <= (mode ? (256u << 10) : (256u << 20)); ws *= 4) {
        unsigned mask = (unsigned)(ws / 16 - 1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(rd, dim3(nblk), dim3(256), 0, 0, buf, mask, iters, out, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double bytes = (double)nblk * 256 * 16 * 8 * iters;
        printf("%s ws=%8zu KB  %.2f TB/s  = %.1f B/clk/CU @2.4GHz\n", mode ? "per-block region" : "shared region   ", ws >> 10, bytes / ms / 1e9, bytes / (ms * 1e-3) / 2.4e9 / 256);
    }
    return 0;
}
