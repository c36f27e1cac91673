// How many independent VALU / LDS-read / SALU instructions issue in the shadow of one 32x32x16 bf16 MFMA -- with one wave per SIMD
// (256 threads) and with two (512 threads)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int K, int MODE>
__global__ void __launch_bounds__(512, 1) k(float* out, int iters) {
    __shared__ float lds[4096];
    f32x16 acc[4];
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)1.0f; }
    float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
    const float m = 1.0001f, c = 0.5f;
    lds[threadIdx.x] = x0; __syncthreads();
    const float* lp = lds + (threadIdx.x & 63);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
            if (MODE == 0) {
                if (K > 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(m), "v"(c));
                if (K > 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(m), "v"(c));
                if (K > 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(m), "v"(c));
                if (K > 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(m), "v"(c));
                if (K > 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x4) : "v"(m), "v"(c));
                if (K > 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x5) : "v"(m), "v"(c));
                if (K > 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x6) : "v"(m), "v"(c));
                if (K > 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x7) : "v"(m), "v"(c));
                if (K > 8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(m), "v"(c));
                if (K > 9) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(m), "v"(c));
                if (K > 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(m), "v"(c));
                if (K > 11) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(m), "v"(c));
            } else if (MODE == 1) {   // LDS reads (b64), results unused until the end
                float2 v;
                if (K > 0) { asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)lp)); }
                if (K > 1) { asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(v) : "v"((unsigned)(size_t)lp)); }
                if (K > 2) { asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(v) : "v"((unsigned)(size_t)lp)); }
                if (K > 3) { asm volatile("ds_read_b64 %0, %1 offset:1536" : "=v"(v) : "v"((unsigned)(size_t)lp)); }
                if (K > 0) { asm volatile("s_waitcnt lgkmcnt(0)"); x1 += 0.f; }
            } else {                  // SALU
                int s;
                for (int q = 0; q < K; q++) asm volatile("s_add_u32 %0, %1, 1" : "=s"(s) : "s"(it));
            }
        }
    }
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    for (int t = 0; t < 4; t++) r += acc[t][0];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int K, int MODE> void run(float* out, const char* what, int threads = 256) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL((k<K, MODE>), dim3(256), dim3(threads), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<K, MODE>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double per = ms * 1e-3 / (iters * 4.0 * (threads / 256));
    printf("[%d thr] %s K=%2d : %.1f ns per MFMA (+K)  = %.1f cycles @2.4GHz\n", threads, what, K, per * 1e9, per * 2.4e9);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<0, 0>(out, "valu"); run<2, 0>(out, "valu"); run<4, 0>(out, "valu"); run<6, 0>(out, "valu"); run<7, 0>(out, "valu");
    run<8, 0>(out, "valu"); run<10, 0>(out, "valu"); run<12, 0>(out, "valu");
    run<1, 1>(out, "lds "); run<2, 1>(out, "lds "); run<4, 1>(out, "lds ");
    run<4, 2>(out, "salu"); run<8, 2>(out, "salu"); run<16, 2>(out, "salu");
    printf("-- two waves per SIMD (time per MFMA per SIMD)\n");
    run<0, 0>(out, "valu", 512); run<4, 0>(out, "valu", 512); run<8, 0>(out, "valu", 512); run<12, 0>(out, "valu", 512);
    run<2, 1>(out, "lds ", 512); run<4, 1>(out, "lds ", 512);
    run<8, 2>(out, "salu", 512); run<16, 2>(out, "salu", 512);
    return 0;
}
