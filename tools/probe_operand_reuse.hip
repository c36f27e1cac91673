// Does the ORDER in which a wave's MFMAs meet their operands move the power wall?  (test infrastructure, follow-up of probe_power_wall)
// The bare loop of probe_power_wall changes BOTH operands between consecutive MFMAs.  A convolution's wave tile does not: in the 1x4
// layout four consecutive MFMAs share their filter fragment, in a 2x2 layout the order of the four (pixel, channel) pairs is free.
// If the matrix core's input toggling matters, operand-stationary orders should clock higher on random data.
// Answer (profiles/r5_operand_reuse.txt): barely -- +0 % for the 1x4 / 4x1 orders, +1.3 % for 2x2, +2.8 % when nothing changes.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_operand_reuse.hip -o tools/probe_operand_reuse && tools/probe_operand_reuse
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// PAT 0: both operands change every MFMA (probe_power_wall)      PAT 1: B shared by four consecutive MFMAs (1x4 wave tile)
// PAT 2: one A, one B for every MFMA (only the accumulators differ) PAT 3: 2x2 tile, Gray order (one operand changes per MFMA)
// PAT 4: 2x2 tile, row-major order (the third MFMA changes both)   PAT 5: A shared by four consecutive MFMAs (4x1 wave tile)
template <int PAT>
__global__ void __launch_bounds__(256, 1) k32(const uint4* __restrict__ ops, float* out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    uint4 f[8];
    for (int i = 0; i < 8; i++) f[i] = ops[(blockIdx.x * 8 + i) * 256 + threadIdx.x];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                int ia, ib;
                if (PAT == 0) { ia = (t + u) & 3; ib = (t + 2 * u) & 3; }
                else if (PAT == 1) { ia = t; ib = u; }
                else if (PAT == 2) { ia = 0; ib = 0; }
                else if (PAT == 3) { ia = (t >> 1) + 2 * (u & 1); ib = ((t >> 1) ^ (t & 1)) + 2 * (u >> 1); }
                else if (PAT == 4) { ia = (t >> 1) + 2 * (u & 1); ib = (t & 1) + 2 * (u >> 1); }
                else { ia = u; ib = t; }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[ia]), __builtin_bit_cast(bf16x8, f[4 + ib]), acc[t], 0, 0, 0);
            }
    }
    float r = 0.f;
    for (int t = 0; t < 4; t++) for (int q = 0; q < 16; q++) r += acc[t][q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// the 16x16x32 shape (same FLOP per instruction pair): eight chains, both operands change
__global__ void __launch_bounds__(256, 1) k16(const uint4* __restrict__ ops, float* out, int iters) {
    f32x4 acc[8];
    for (int t = 0; t < 8; t++) for (int r = 0; r < 4; r++) acc[t][r] = 0.f;
    uint4 f[8];
    for (int i = 0; i < 8; i++) f[i] = ops[(blockIdx.x * 8 + i) * 256 + threadIdx.x];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int t = 0; t < 8; t++)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f[(t + u) & 3]), __builtin_bit_cast(bf16x8, f[4 + ((t + 2 * u) & 3)]), acc[t], 0, 0, 0);
    }
    float r = 0.f;
    for (int t = 0; t < 8; t++) for (int q = 0; q < 4; q++) r += acc[t][q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

static uint16_t bf16_of(float v) { uint32_t u; memcpy(&u, &v, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

template <typename K>
static void run(const char* name, K kern, const uint4* d, float* out, int blocks, double flop_per_inst, int inst_per_iter) {
    const int iters = 30000;
    // warm-up = the whole timed launch twice: the first tens of ms after a host-side pause run 5-12 % slower (clock ramp), which a short
    // warm-up does not cover -- it made the first row of every data block look like an order effect in the first version of this probe
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, out, iters);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, out, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)iters * inst_per_iter * 4 * blocks;
    printf("  %-64s %7.2f ms  %7.1f TFLOP/s\n", name, ms, inst * flop_per_inst / (ms * 1e-3) / 1e12);
}

int main() {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus, n = blocks * 8 * 256 * 8;
    uint16_t* h = (uint16_t*)malloc(n * 2);
    uint4* d; float* out;
    hipMalloc(&d, n * 2); hipMalloc(&out, blocks * 256 * 4);
    const char* dn[] = {"N(0,1) random", "A relu(N(0,1)), B N(0,1)", "A and B relu-like (half zero each)", "all zero"};
    for (int mode = 0; mode < 4; mode++) {
        uint64_t s = 88172645463325252ull;
        for (int i = 0; i < n; i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            float u1 = ((s >> 11) & 0xffffff) / 16777216.f + 1e-7f, u2 = ((s >> 35) & 0xffffff) / 16777216.f;
            float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
            const bool isA = ((i / 8 / 256) % 8) < 4;
            float v = g;
            if (mode == 1 && isA && g < 0.f) v = 0.f;
            if (mode == 2 && g < 0.f) v = 0.f;
            if (mode == 3) v = 0.f;
            h[i] = bf16_of(v);
        }
        hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
        printf("data: %s\n", dn[mode]);
        const double f32 = 2.0 * 32 * 32 * 16, f16 = 2.0 * 16 * 16 * 32;
        run("32x32x16, both operands change every MFMA", k32<0>, d, out, blocks, f32, 16);
        run("32x32x16, B shared by 4 consecutive MFMAs (1x4 tile)", k32<1>, d, out, blocks, f32, 16);
        run("32x32x16, A shared by 4 consecutive MFMAs (4x1 tile)", k32<5>, d, out, blocks, f32, 16);
        run("32x32x16, 2x2 tile in Gray order", k32<3>, d, out, blocks, f32, 16);
        run("32x32x16, 2x2 tile row-major", k32<4>, d, out, blocks, f32, 16);
        run("32x32x16, one A and one B throughout", k32<2>, d, out, blocks, f32, 16);
        run("16x16x32, both operands change", k16, d, out, blocks, f16, 32);
        run("32x32x16, both operands change every MFMA (again, last)", k32<0>, d, out, blocks, f32, 16);
    }
    return 0;
}
