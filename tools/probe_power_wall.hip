// How fast does a bare v_mfma_f32_32x32x16_bf16 loop run on MI355X as a function of its OPERAND DATA?  (test infrastructure)
// One wave per SIMD on every CU, four independent accumulator chains, operands in registers: nothing but the matrix pipe.  The chip
// clocks to its power budget (MI355X_MICROARCH.md, DVFS give-back), and a matrix core fed with random bf16 values toggles far more
// than one fed with zeros -- this is the ceiling a real convolution's data allows, as opposed to the 2.5 PFLOP/s of the data sheet.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_power_wall.hip -o tools/probe_power_wall && tools/probe_power_wall
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// SHARED = false: both source registers change between consecutive MFMAs.  SHARED = true: srcB is shared by four consecutive MFMAs and srcA
// changes -- the order of the convolutions' 1x4 wave tile (tools/probe_operand_reuse.hip has the other orders: a change of srcA costs ~4x
// a change of srcB, so the ceiling depends on the walk)
template <bool SHARED>
__global__ void __launch_bounds__(256, 1) k(const uint4* __restrict__ ops, float* out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    // eight operand fragments per lane (4 A, 4 B), rotated through the chains so that consecutive MFMAs see different data
    uint4 f[8];
    for (int i = 0; i < 8; i++) f[i] = ops[(blockIdx.x * 8 + i) * 256 + threadIdx.x];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int t = 0; t < 4; t++)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[SHARED ? t : (t + u) & 3]),
                                                                 __builtin_bit_cast(bf16x8, f[4 + (SHARED ? u : (t + 2 * u) & 3)]), acc[t], 0, 0, 0);
    }
    float r = 0.f;
    for (int t = 0; t < 4; t++) for (int q = 0; q < 16; q++) r += acc[t][q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

static uint16_t bf16_of(float v) { uint32_t u; memcpy(&u, &v, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main() {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus, n = blocks * 8 * 256 * 8;              // bf16 elements
    uint16_t* h = (uint16_t*)malloc(n * 2);
    uint4* d; float* out;
    hipMalloc(&d, n * 2); hipMalloc(&out, blocks * 256 * 4);
    const char* names[] = {"all zero", "all 1.0", "N(0,1) random", "N(0,1), half of the A values zero (ReLU-like)", "N(0,1) random, srcB shared by 4 MFMAs (1x4 tile)"};
    const int iters = 40000;
    for (int mode = 0; mode < 5; mode++) {
        uint64_t s = 88172645463325252ull;
        for (int i = 0; i < n; i++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            float u1 = ((s >> 11) & 0xffffff) / 16777216.f + 1e-7f, u2 = ((s >> 35) & 0xffffff) / 16777216.f;
            float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
            float v = mode == 0 ? 0.f : mode == 1 ? 1.f : g;                            // mode 4: random data, the other kernel
            if (mode == 3 && ((i / 8 / 256) % 8) < 4 && g < 0.f) v = 0.f;        // the four A fragments: relu
            h[i] = bf16_of(v);
        }
        hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
        auto kern = mode == 4 ? k<true> : k<false>;
        // warm-up = the whole timed launch three times: after the host-side fill above the first tens of ms run 5-12 % slower (clock ramp);
        // the 2 000-iteration warm-up of rounds 3-4 did not cover it: their 1.53-1.61 PFLOP/s on random data is 1.77-1.81 sustained
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, out, iters);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, out, iters);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, out, iters);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (double)iters * 16 * 4 * blocks;                        // MFMA instructions (4 waves per block)
        const double tf = mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
        const double clk = mfma / (4.0 * blocks) * 32 / (ms * 1e-3) / 1e9;          // 32 cycles per MFMA per SIMD when the pipe is full
        printf("%-48s %7.2f ms  %7.1f TFLOP/s  (%.2f of 2500)  implied clock %.2f GHz\n", names[mode], ms, tf, tf / 2500.0, clk);
    }
    return 0;
}
