// What do the operand FEEDS cost the matrix pipe's clock?  (test infrastructure; follow-up of probe_power_wall / probe_operand_reuse)
// A bare v_mfma_f32_32x32x16_bf16 loop in the convolutions' 1x4 order (four pixel fragments under one filter fragment), one or two waves per
// SIMD, N(0,1) bf16 data, and on top of it, per four MFMAs:
//   LDS = 0 / 4 / 2 / 1   fresh pixel fragments read from LDS with ds_read_b128 (4 = the 1x4 wave tile: 1 KB per MFMA; 2 = what a 2x2
//                         64x64 tile or a 128x64 tile would read; the rest of the fragments stay in registers)
//   L1  = 0 / 1           one fresh filter fragment (1 KB per wave) from a 144 KB L2-resident image through the vector L1 (0.25 KB per MFMA)
// Everything is requested one group ahead, so latency is hidden as far as one or two waves per SIMD can: what remains is the clock the
// power budget allows with that much LDS / L1 / register traffic beside the MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_feed_power.hip -o tools/probe_feed_power && tools/probe_feed_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int LDS, int L1, int WAVES>
__global__ void __launch_bounds__(256 * WAVES, 1) k(const uint4* __restrict__ ops, const uint4* __restrict__ filt, float* out, int iters) {
    extern __shared__ uint4 sm[];                                   // 64 KB of random fragments
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256 * WAVES) sm[i] = ops[(blockIdx.x % 8) * 4096 + i];
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    uint4 a[4], an[4], b, bn;
    for (int i = 0; i < 4; i++) { a[i] = sm[(i * 64 + lane) & 4095]; an[i] = a[i]; }
    b = filt[lane]; bn = b;
    unsigned off = (tid >> 6) * 1031u;                               // wave-uniform: the reads stay lane-linear (conflict-free ds_read_b128)
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            // request the next group's operands, then the four MFMAs of this group
            off += 64 * 4;
#pragma unroll
            for (int i = 0; i < LDS; i++) an[i] = sm[(off + i * 64 + lane) & 4095];
            if (L1) bn = filt[((it * 4 + u) * 64 + lane) % 9216];   // 9216 x 16 B = 144 KB image
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; t++)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t]), __builtin_bit_cast(bf16x8, b), acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < LDS; i++) a[i] = an[i];
            if (L1) b = bn;
        }
    }
    float r = 0.f;
    for (int t = 0; t < 4; t++) for (int q = 0; q < 16; q++) r += acc[t][q];
    out[blockIdx.x * 256 * WAVES + tid] = r;
}


// the same feed with DEEP prefetch: filter fragments three groups ahead (ring of four), pixel fragments two groups ahead (three sets) --
// if the rows above were latency-bound this one is faster; if it is not, what limits them is not latency
template <int LDS, int WAVES, bool BLDS = false>
__global__ void __launch_bounds__(256 * WAVES, 1) kdeep(const uint4* __restrict__ ops, const uint4* __restrict__ filt, float* out, int iters) {
    extern __shared__ uint4 sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256 * WAVES) sm[i] = ops[(blockIdx.x % 8) * 4096 + i];
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    uint4 a0[4], a1[4], a2[4], b0, b1, b2, b3;
    for (int i = 0; i < 4; i++) { a0[i] = sm[(i * 64 + lane) & 4095]; a1[i] = a0[i]; a2[i] = a0[i]; }
    b0 = filt[lane]; b1 = filt[64 + lane]; b2 = filt[128 + lane]; b3 = filt[192 + lane];
    unsigned off = (tid >> 6) * 1031u, fo = 256;
#define GROUP(AC, AN, BC, BN)                                                                            \
    {                                                                                                   \
        off += 256; fo = fo + 64 >= 9216 ? 0 : fo + 64;                                                 \
        _Pragma("unroll") for (int i = 0; i < LDS; i++) AN[i] = sm[(off + i * 64 + lane) & 4095];        \
        if (BLDS) BN = sm[(off + 2048 + lane) & 4095]; else BN = filt[fo + lane];                       \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        _Pragma("unroll") for (int t = 0; t < 4; t++)                                                    \
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, AC[t]), __builtin_bit_cast(bf16x8, BC), acc[t], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    }
    for (int it = 0; it < iters; it += 3) {
        // 12 groups per trip: pixel sets rotate with period 3 (compute s, request s+2), filter ring with period 4 (compute r, request r+3)
        GROUP(a0, a2, b0, b3) GROUP(a1, a0, b1, b0) GROUP(a2, a1, b2, b1) GROUP(a0, a2, b3, b2)
        GROUP(a1, a0, b0, b3) GROUP(a2, a1, b1, b0) GROUP(a0, a2, b2, b1) GROUP(a1, a0, b3, b2)
        GROUP(a2, a1, b0, b3) GROUP(a0, a2, b1, b0) GROUP(a1, a0, b2, b1) GROUP(a2, a1, b3, b2)
    }
#undef GROUP
    float r = 0.f;
    for (int t = 0; t < 4; t++) for (int q = 0; q < 16; q++) r += acc[t][q];
    out[blockIdx.x * 256 * WAVES + tid] = r;
}

static uint16_t bf16_of(float v) { uint32_t u; memcpy(&u, &v, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

template <typename K>
static void run(const char* name, K kern, int waves, const uint4* d, const uint4* f, float* out, int blocks) {
    const int iters = 20000 / waves;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256 * waves), 65536, 0, d, f, out, iters);   // whole-launch warm-ups (clock ramp)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256 * waves), 65536, 0, d, f, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)iters * 16 * 4 * waves * blocks;
    printf("  %-72s %7.2f ms  %7.1f TFLOP/s\n", name, ms, inst * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12);
}

int main() {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus, n = 8 * 4096 * 8 + 9216 * 8;
    uint16_t* h = (uint16_t*)malloc(n * 2);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        float u1 = ((s >> 11) & 0xffffff) / 16777216.f + 1e-7f, u2 = ((s >> 35) & 0xffffff) / 16777216.f;
        float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
        if (i < 8 * 4096 * 8 && g < 0.f) g = 0.f;                   // pixel fragments: ReLU-like
        h[i] = bf16_of(g);
    }
    uint4* d; float* out;
    hipMalloc(&d, n * 2); hipMalloc(&out, blocks * 512 * 4);
    hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
    const uint4* f = d + 8 * 4096;
    printf("pixel fragments relu(N(0,1)) from LDS, filter fragments N(0,1) from L2 through the vector L1; per four MFMAs:\n");
    run("1 wave/SIMD: operands stay in registers", k<0, 0, 1>, 1, d, f, out, blocks);
    run("1 wave/SIMD: + 1 filter fragment from L1", k<0, 1, 1>, 1, d, f, out, blocks);
    run("1 wave/SIMD: + 4 pixel fragments from LDS (1x4 tile)", k<4, 0, 1>, 1, d, f, out, blocks);
    run("1 wave/SIMD: + 4 from LDS + 1 from L1 (the shipped wide kernel's feed)", k<4, 1, 1>, 1, d, f, out, blocks);
    run("1 wave/SIMD: + 2 from LDS + 1 from L1", k<2, 1, 1>, 1, d, f, out, blocks);
    run("1 wave/SIMD: + 1 from LDS + 1 from L1", k<1, 1, 1>, 1, d, f, out, blocks);
    run("2 waves/SIMD: operands stay in registers", k<0, 0, 2>, 2, d, f, out, blocks);
    run("2 waves/SIMD: + 4 from LDS + 1 from L1 (the shipped wide kernel's feed)", k<4, 1, 2>, 2, d, f, out, blocks);
    run("2 waves/SIMD: + 2 from LDS + 1 from L1", k<2, 1, 2>, 2, d, f, out, blocks);
    run("1 wave/SIMD, deep prefetch: + 4 from LDS + 1 from L1", kdeep<4, 1>, 1, d, f, out, blocks);
    run("2 waves/SIMD, deep prefetch: + 4 from LDS + 1 from L1", kdeep<4, 2>, 2, d, f, out, blocks);
    run("2 waves/SIMD, deep prefetch: + 2 from LDS + 1 from L1", kdeep<2, 2>, 2, d, f, out, blocks);
    run("2 waves/SIMD, deep prefetch: + 1 from LDS + 1 from L1", kdeep<1, 2>, 2, d, f, out, blocks);
    run("2 waves/SIMD, deep prefetch: + 4 from LDS + the filter fragment from LDS too", kdeep<4, 2, true>, 2, d, f, out, blocks);
    run("2 waves/SIMD, deep prefetch: + 2 from LDS + the filter fragment from LDS too", kdeep<2, 2, true>, 2, d, f, out, blocks);
    run("1 wave/SIMD, deep prefetch: + 4 from LDS + the filter fragment from LDS too", kdeep<4, 1, true>, 1, d, f, out, blocks);
    run("1 wave/SIMD: operands stay in registers (again)", k<0, 0, 1>, 1, d, f, out, blocks);
    return 0;
}
