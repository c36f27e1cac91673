// Probe (test infrastructure, round 6): can the BatchNorm statistics leave the convolution epilogues through 64-bit INTEGER (fixed-point)
// device-scope atomics into XCD-striped slots -- integer addition is associative, so a step would stay bit-reproducible -- with the CONSUMER
// deriving scale / shift in its prologue, so that the 36 dependent reduce_rows / bn_finalize launches of a forward disappear?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_atomics.hip -o tools/probe_atomics && tools/probe_atomics
// Part A  producer side: a grid shaped like a layer's convolution (blocks x channels per block); every block spins for `busy` us (the MFMA loop's
//         stand-in, so the epilogues arrive spread out as in the real kernel; busy = 0 is the worst case: all at once), then its 256 threads
//         add 2 x CH fixed-point values to slot[blockIdx % 8][2][C] with no-return agent-scope atomics -- against the same kernel storing its
//         2 x CH partials as plain floats (today's epilogue) and against no epilogue at all.
// Part B  consumer side: a grid of 2048 blocks whose prologue reads the 8 slots x 2 x C int64 sums and forms scale / shift for its C channels
//         (what every consuming convolution block would do) against reading a finished [2][C] float table (today).
// Part C  today's dependent pair alone: reduce_rows-like + finalize-like launches behind a producer (what the atomics would remove).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__device__ __forceinline__ void spin_us(float us) {
    if (us <= 0.f) return;
    const long long t0 = wall_clock64();                       // 100 MHz
    const long long dt = (long long)(us * 100.f);
    while (wall_clock64() - t0 < dt) { }
}

// mode 0: no epilogue   1: plain float partial rows   2: int64 atomics, slot = blockIdx % slots
__global__ __launch_bounds__(256) void producer(int mode, int C, int CH, int n_ntiles, float busy, float* part, unsigned long long* slots, int nslots) {
    const int t = threadIdx.x;
    spin_us(busy * (0.75f + 0.5f * (float)((blockIdx.x * 2654435761u) >> 24) / 256.f));      // +-25 % spread
    const int ntile = blockIdx.x % n_ntiles, mtile = blockIdx.x / n_ntiles;
    const int col0 = ntile * CH;
    if (t < 2 * CH) {
        const int k = t / CH, c = col0 + t % CH;
        const float v = (float)((blockIdx.x * 131 + t) & 1023) * 0.37f;
        if (mode == 1) part[((size_t)mtile * 2 + k) * C + c] = v;
        else if (mode == 2) {
            const long long q = (long long)(v * 1048576.f);                 // 2^20 fixed point
            __hip_atomic_fetch_add(slots + ((size_t)(blockIdx.x % nslots) * 2 + k) * C + c, (unsigned long long)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// consumer prologue.  mode 0: finished float table [2][C] -> LDS;  mode 1: 8 slots of int64 sums -> mean / invstd -> scale / shift -> LDS
__global__ __launch_bounds__(256) void consumer(int mode, int C, const float* tab, const unsigned long long* slots, int nslots, const float* gamma, const float* beta,
                                                float invM, float* sink) {
    extern __shared__ float sm[];
    const int t = threadIdx.x;
    if (mode == 0) {
        for (int i = t; i < 2 * C; i += 256) sm[i] = tab[i];
    } else {
        for (int c = t; c < C; c += 256) {
            long long s = 0, q = 0;
            for (int x = 0; x < nslots; x++) { s += (long long)slots[((size_t)x * 2 + 0) * C + c]; q += (long long)slots[((size_t)x * 2 + 1) * C + c]; }
            const double mean = (double)s * (1.0 / 1048576.0) * invM, ex2 = (double)q * (1.0 / 1048576.0) * invM;
            const float var = (float)(ex2 - mean * mean), inv = rsqrtf(fmaxf(var, 0.f) + 1e-5f);
            const float sc = gamma[c] * inv;
            sm[c] = sc; sm[C + c] = beta[c] - (float)mean * sc;
        }
    }
    __syncthreads();
    float a = 0.f;
    for (int i = t; i < 2 * C; i += 256) a += sm[i];
    if (a == 123.456f) sink[blockIdx.x] = a;                    // keep the work alive
}

// today's dependent pair: rows -> [2][C] in double (reduce_rows_kernel's shape), then a one-block finalize
__global__ void reduce_rows_like(const float* part, int rows, int C, double* out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= 2 * C) return;
    const int k = c / C, cc = c % C;
    const int r0 = blockIdx.y * (rows / gridDim.y), r1 = r0 + rows / gridDim.y;
    double s = 0;
    for (int r = r0; r < r1; r++) s += part[((size_t)r * 2 + k) * C + cc];
    out[(size_t)blockIdx.y * 2 * C + c] = s;
}
__global__ void finalize_like(const double* in, int parts, int C, const float* gamma, const float* beta, float invM, float* tab) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double s = 0, q = 0;
        for (int p = 0; p < parts; p++) { s += in[(size_t)p * 2 * C + c]; q += in[(size_t)p * 2 * C + C + c]; }
        const double mean = s * invM;
        const float inv = rsqrtf(fmaxf((float)(q * invM - mean * mean), 0.f) + 1e-5f), sc = gamma[c] * inv;
        tab[c] = sc; tab[C + c] = beta[c] - (float)mean * sc;
    }
}

template <typename F> static float time_us(F f, int iters = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; i++) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

int main() {
    float *part, *tab, *gamma, *beta, *sink; unsigned long long* slots; double* dbl;
    hipMalloc(&part, (size_t)8192 * 2 * 512 * 4); hipMalloc(&slots, 8 * 2 * 512 * 8); hipMalloc(&tab, 2 * 512 * 4);
    hipMalloc(&gamma, 512 * 4); hipMalloc(&beta, 512 * 4); hipMalloc(&sink, 8192 * 4); hipMalloc(&dbl, 64 * 2 * 512 * 8);
    hipMemset(slots, 0, 8 * 2 * 512 * 8); hipMemset(gamma, 0, 512 * 4); hipMemset(beta, 0, 512 * 4); hipMemset(tab, 0, 2 * 512 * 4);
    struct L { const char* name; int mtiles, ntiles, C, CH; float busy; };
    // blocks = mtiles x ntiles; CH channels per block; busy ~ the layer's per-block MFMA time (us), 0 = everything at once
    const L layers[] = {{"e1b 8192 x 1, 64 ch", 8192, 1, 64, 64, 6.f}, {"e2b 4096 x 1, 128 ch", 4096, 1, 128, 128, 12.f},
                        {"e3b 1024 x 2, 256 ch", 1024, 2, 256, 128, 22.f}, {"e4b 256 x 4, 512 ch", 256, 4, 512, 128, 40.f},
                        {"e5b 64 x 8, 512 ch", 64, 8, 512, 64, 20.f}};
    printf("Part A  producer epilogue (us per launch: no epilogue / plain partial rows / int64 atomics into 8 slots / into 1 slot)\n");
    for (const L& l : layers) {
        for (float busy : {0.f, l.busy}) {
            float t[4];
            for (int m = 0; m < 4; m++) {
                const int mode = m < 3 ? m : 2, ns = m == 3 ? 1 : 8;
                t[m] = time_us([&] { hipLaunchKernelGGL(producer, dim3(l.mtiles * l.ntiles), dim3(256), 0, 0, mode, l.C, l.CH, l.ntiles, busy, part, slots, ns); });
            }
            printf("  %-24s busy %5.1f us/block: %8.2f  %8.2f  %8.2f  %8.2f   atomics - plain = %+6.2f us per launch, %+.4f us per block\n", l.name, busy, t[0], t[1], t[2], t[3],
                   t[2] - t[1], (t[2] - t[1]) / (l.mtiles * l.ntiles) * 512.f);     // per block: 512 block slots run at once
        }
    }
    printf("Part B  consumer prologue, 2048 blocks (us per launch: finished float table / 8 int64 slots -> scale, shift)\n");
    for (int C : {64, 128, 256, 512}) {
        const float t0 = time_us([&] { hipLaunchKernelGGL(consumer, dim3(2048), dim3(256), 2 * C * 4, 0, 0, C, tab, slots, 8, gamma, beta, 1e-6f, sink); });
        const float t1 = time_us([&] { hipLaunchKernelGGL(consumer, dim3(2048), dim3(256), 2 * C * 4, 0, 1, C, tab, slots, 8, gamma, beta, 1e-6f, sink); });
        printf("  C = %3d: %7.2f  %7.2f   (+%.2f us per launch = +%.3f us per block round of 512)\n", C, t0, t1, t1 - t0, (t1 - t0) / 4.f);
    }
    printf("Part C  today's dependent pair behind the producer (us: producer alone / producer + reduce_rows-like + finalize-like, same stream)\n");
    for (const L& l : layers) {
        const int rows = l.mtiles, parts = rows >= 512 ? 16 : 1;
        const float t0 = time_us([&] { hipLaunchKernelGGL(producer, dim3(l.mtiles * l.ntiles), dim3(256), 0, 0, 1, l.C, l.CH, l.ntiles, l.busy, part, slots, 8); });
        const float t1 = time_us([&] {
            hipLaunchKernelGGL(producer, dim3(l.mtiles * l.ntiles), dim3(256), 0, 0, 1, l.C, l.CH, l.ntiles, l.busy, part, slots, 8);
            hipLaunchKernelGGL(reduce_rows_like, dim3((2 * l.C + 63) / 64, parts), dim3(64), 0, 0, part, rows, l.C, dbl);
            hipLaunchKernelGGL(finalize_like, dim3(1), dim3(256), 0, 0, dbl, parts, l.C, gamma, beta, 1e-6f, tab);
        });
        const float t2 = time_us([&] {
            hipLaunchKernelGGL(producer, dim3(l.mtiles * l.ntiles), dim3(256), 0, 0, 2, l.C, l.CH, l.ntiles, l.busy, part, slots, 8);
            hipLaunchKernelGGL(consumer, dim3(2048), dim3(256), 2 * l.C * 4, 0, 1, l.C, tab, slots, 8, gamma, beta, 1e-6f, sink);
        });
        const float t3 = time_us([&] {
            hipLaunchKernelGGL(producer, dim3(l.mtiles * l.ntiles), dim3(256), 0, 0, 1, l.C, l.CH, l.ntiles, l.busy, part, slots, 8);
            hipLaunchKernelGGL(reduce_rows_like, dim3((2 * l.C + 63) / 64, parts), dim3(64), 0, 0, part, rows, l.C, dbl);
            hipLaunchKernelGGL(finalize_like, dim3(1), dim3(256), 0, 0, dbl, parts, l.C, gamma, beta, 1e-6f, tab);
            hipLaunchKernelGGL(consumer, dim3(2048), dim3(256), 2 * l.C * 4, 0, 0, l.C, tab, slots, 8, gamma, beta, 1e-6f, sink);
        });
        printf("  %-24s producer %8.2f   + reduce + finalize %8.2f (+%.2f)   |  today: producer, reduce, finalize, consumer %8.2f   atomics: producer, consumer %8.2f  (%+.2f us)\n",
               l.name, t0, t1, t1 - t0, t3, t2, t2 - t3);
    }
    return 0;
}
