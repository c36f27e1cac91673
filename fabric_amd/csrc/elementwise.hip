// HBM-bound kernels around the convolutions: layout converters, BatchNorm finalize / backward,
// max-pool, date-fusion product, bilinear upsample (+ their backward gathers), the 1x1
// classifier, the Tversky loss and the SGD update.  All activations NHWC, 16-byte vector access
// along channels; per-channel reductions go through per-block partials (deterministic, no atomics).
#include "common.hpp"

static inline unsigned grid_for(size_t n, int block = 256) { return (unsigned)((n + block - 1) / block); }

// ============================================================ pack_input
// reference boundary: BiDateNet.forward(x_d1, x_d2), models/bidate_model.py:22 (NCHW f32)
template <typename T>
__global__ void pack_input_kernel(const float* __restrict__ x1, const float* __restrict__ x2, T* __restrict__ out,
                                  int B, int C, int H, int W, int Cpad) {
    const size_t npix = (size_t)2 * B * H * W;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const size_t hw = (size_t)H * W;
    const int n = i / hw; const size_t p = i % hw;
    const float* src = (n < B ? x1 + (size_t)n * C * hw : x2 + (size_t)(n - B) * C * hw) + p;
    T* dst = out + i * Cpad;
    for (int c = 0; c < Cpad; c++) dst[c] = from_f<T>(c < C ? src[(size_t)c * hw] : 0.f);
}

extern "C" int bdn_pack_input(int dtype, const float* x_d1, const float* x_d2, void* out,
                              int B, int C, int H, int W, int Cpad, void* stream) {
    if (!x_d1 || !x_d2 || !out) BDN_FAIL(BDN_E_ARG, "pack_input: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C || Cpad % 16) BDN_FAIL(BDN_E_SHAPE, "pack_input: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)2 * B * H * W;
    if (dtype == BDN_BF16) hipLaunchKernelGGL(pack_input_kernel<bf16s>, dim3(grid_for(npix)), dim3(256), 0, st, x_d1, x_d2, (bf16s*)out, B, C, H, W, Cpad);
    else if (dtype == BDN_F32) hipLaunchKernelGGL(pack_input_kernel<float>, dim3(grid_for(npix)), dim3(256), 0, st, x_d1, x_d2, (float*)out, B, C, H, W, Cpad);
    else BDN_FAIL(BDN_E_ARG, "pack_input: bad dtype");
    BDN_CHECK_LAUNCH("pack_input");
    return BDN_OK;
}

// ============================================================ pack_weights
// wf[co][tap][ci] = w[co][ci][r][c];  wd[ci][tap][co] = w[co][ci][2-r][2-c]   (tap = 3r+c)
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd,
                                    int Cout, int Cin, int Cinp) {
    const size_t total = (size_t)Cout * 9 * Cinp;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ci = i % Cinp; const size_t t = i / Cinp; const int tap = t % 9; const int co = t / 9;
    const float v = ci < Cin ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.f;
    if (wf) wf[i] = from_f<T>(v);
    if (wd) wd[((size_t)ci * 9 + (8 - tap)) * Cout + co] = from_f<T>(v);
}

extern "C" int bdn_pack_weights(int dtype, const float* w_oihw, void* wf, void* wd,
                                int Cout, int Cin, int Cin_pad, void* stream) {
    if (!w_oihw || (!wf && !wd)) BDN_FAIL(BDN_E_ARG, "pack_weights: null pointer");
    if (Cout <= 0 || Cin <= 0 || Cin_pad < Cin) BDN_FAIL(BDN_E_SHAPE, "pack_weights: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)Cout * 9 * Cin_pad;
    if (dtype == BDN_BF16) hipLaunchKernelGGL(pack_weights_kernel<bf16s>, dim3(grid_for(total)), dim3(256), 0, st, w_oihw, (bf16s*)wf, (bf16s*)wd, Cout, Cin, Cin_pad);
    else if (dtype == BDN_F32) hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, w_oihw, (float*)wf, (float*)wd, Cout, Cin, Cin_pad);
    else BDN_FAIL(BDN_E_ARG, "pack_weights: bad dtype");
    BDN_CHECK_LAUNCH("pack_weights");
    return BDN_OK;
}

// ============================================================ partial reduction helper
// partial: [n_rows][2][C] f32, rows of group g are [g*rows_per_group, (g+1)*rows_per_group).
// block = 256 threads = 16 row lanes x 16 channels; returns (sum0,sum1) in double for thread rl==0.
__device__ __forceinline__ void reduce_rows(const float* __restrict__ partial, int row0, int nrows, int C, int c,
                                            double& s0, double& s1, double (*sm)[16][2]) {
    const int rl = threadIdx.x >> 4, cl = threadIdx.x & 15;
    double a0 = 0.0, a1 = 0.0;
    if (c < C)
        for (int r = rl; r < nrows; r += 16) {
            a0 += partial[((size_t)(row0 + r) * 2 + 0) * C + c];
            a1 += partial[((size_t)(row0 + r) * 2 + 1) * C + c];
        }
    sm[rl][cl][0] = a0; sm[rl][cl][1] = a1;
    __syncthreads();
    s0 = 0.0; s1 = 0.0;
    if (rl == 0) for (int r = 0; r < 16; r++) { s0 += sm[r][cl][0]; s1 += sm[r][cl][1]; }
    __syncthreads();
}

// ============================================================ bn_finalize (nn.BatchNorm2d training, unet_parts.py:14,17)
__global__ void bn_finalize_kernel(const float* __restrict__ partial, int rows_per_group, int G, int C, float count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                   float* running_mean, float* running_var, int64_t* nbt, float* __restrict__ bn) {
    __shared__ double sm[16][16][2];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    for (int g = 0; g < G; g++) {                       // sequential: running stats see date 1 then date 2
        double s0, s1;
        reduce_rows(partial, g * rows_per_group, rows_per_group, C, c, s0, s1, sm);
        if ((threadIdx.x >> 4) == 0 && c < C) {
            const double mean = s0 / count;
            double var = s1 / count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float inv = (float)(1.0 / sqrt(var + (double)eps));
            const float scale = gamma[c] * inv;
            bn[((size_t)g * 4 + 0) * C + c] = (float)mean;
            bn[((size_t)g * 4 + 1) * C + c] = inv;
            bn[((size_t)g * 4 + 2) * C + c] = scale;
            bn[((size_t)g * 4 + 3) * C + c] = beta[c] - (float)mean * scale;
            if (running_mean) {
                const double unb = count > 1.f ? var * (count / (count - 1.0)) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
            }
        }
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += G;
}

extern "C" int bdn_bn_finalize(const float* stats_partial, int n_mtiles, int G, int C, int count_per_group,
                               const float* gamma, const float* beta, float eps, float momentum,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked,
                               float* bn, void* stream) {
    if (!stats_partial || !gamma || !beta || !bn) BDN_FAIL(BDN_E_ARG, "bn_finalize: null pointer");
    if (G <= 0 || C <= 0 || n_mtiles <= 0 || n_mtiles % G || count_per_group <= 0) BDN_FAIL(BDN_E_SHAPE, "bn_finalize: bad shape");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                       stats_partial, n_mtiles / G, G, C, (float)count_per_group, gamma, beta, eps, momentum,
                       running_mean, running_var, num_batches_tracked, bn);
    BDN_CHECK_LAUNCH("bn_finalize");
    return BDN_OK;
}

__global__ void bn_eval_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                               int G, int C, float* bn) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float inv = 1.0f / sqrtf(rv[c] + eps);
    const float scale = gamma[c] * inv;
    for (int g = 0; g < G; g++) {
        bn[((size_t)g * 4 + 0) * C + c] = rm[c];
        bn[((size_t)g * 4 + 1) * C + c] = inv;
        bn[((size_t)g * 4 + 2) * C + c] = scale;
        bn[((size_t)g * 4 + 3) * C + c] = beta[c] - rm[c] * scale;
    }
}

extern "C" int bdn_bn_eval(const float* gamma, const float* beta, const float* running_mean,
                           const float* running_var, float eps, int G, int C, float* bn, void* stream) {
    if (!gamma || !beta || !running_mean || !running_var || !bn) BDN_FAIL(BDN_E_ARG, "bn_eval: null pointer");
    hipLaunchKernelGGL(bn_eval_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       gamma, beta, running_mean, running_var, eps, G, C, bn);
    BDN_CHECK_LAUNCH("bn_eval");
    return BDN_OK;
}

// ============================================================ bn backward (BatchNorm2d + ReLU)
// pass 1: per-block partial sums over a pixel range.  block = 256 threads; thread t owns channel unit
// (t % CU) and walks pixels (t / CU), stepping by 256/CU.  Rows of one block stay inside one group.
constexpr int BNB_PIX = 512;                                  // pixels per block in pass 1
template <typename T>
__global__ void bn_bwd_reduce_kernel(const T* __restrict__ dA, int ldA, const T* __restrict__ z, const float* __restrict__ bn,
                                     int pix_per_group, int blocks_per_group, int C, float* __restrict__ partial) {
    constexpr int EPU = ET<T>::EPU;
    extern __shared__ float sred[];                           // [256][EPU][2]
    const int CU = C / EPU;                                   // host guarantees 256 % CU == 0
    const int rows = 256 / CU;
    const int g = blockIdx.x / blocks_per_group, bg = blockIdx.x % blocks_per_group;
    const int p_begin = bg * BNB_PIX, p_end = min(pix_per_group, p_begin + BNB_PIX);
    const int tid = threadIdx.x, cu = tid % CU, row = tid / CU, c = cu * EPU;
    float s0[EPU], s1[EPU], mean[EPU], inv[EPU], sc[EPU], sh[EPU];
#pragma unroll
    for (int i = 0; i < EPU; i++) {
        s0[i] = 0.f; s1[i] = 0.f;
        mean[i] = bn_row(bn, g, 0, C)[c + i]; inv[i] = bn_row(bn, g, 1, C)[c + i];
        sc[i] = bn_row(bn, g, 2, C)[c + i]; sh[i] = bn_row(bn, g, 3, C)[c + i];
    }
    for (int p = p_begin + row; p < p_end; p += rows) {
        const size_t pix = (size_t)g * pix_per_group + p;
        float fz[EPU], fg[EPU];
        Unit<T>::unpack(*reinterpret_cast<const uint4*>(z + pix * C + c), fz);
        Unit<T>::unpack(*reinterpret_cast<const uint4*>(dA + pix * ldA + c), fg);
#pragma unroll
        for (int i = 0; i < EPU; i++) {
            const float gm = fmaf(fz[i], sc[i], sh[i]) > 0.f ? fg[i] : 0.f;
            s0[i] += gm; s1[i] += gm * ((fz[i] - mean[i]) * inv[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < EPU; i++) { sred[(tid * EPU + i) * 2] = s0[i]; sred[(tid * EPU + i) * 2 + 1] = s1[i]; }
    __syncthreads();
    for (int o = tid; o < C * 2; o += 256) {
        const int k = o & 1, cc = o >> 1, ccu = cc / EPU, i = cc % EPU;
        float s = 0.f;
        for (int r = 0; r < rows; r++) s += sred[((r * CU + ccu) * EPU + i) * 2 + k];
        partial[((size_t)blockIdx.x * 2 + k) * C + cc] = s;
    }
}

// pass 2: reduce partials -> sums[g][2][C], dgamma, dbeta (summed over groups)
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, int blocks_per_group, int G, int C,
                                       float* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double sm[16][16][2];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    double t0 = 0.0, t1 = 0.0;
    for (int g = 0; g < G; g++) {
        double s0, s1;
        reduce_rows(partial, g * blocks_per_group, blocks_per_group, C, c, s0, s1, sm);
        if ((threadIdx.x >> 4) == 0 && c < C) {
            sums[((size_t)g * 2 + 0) * C + c] = (float)s0;
            sums[((size_t)g * 2 + 1) * C + c] = (float)s1;
            t0 += s0; t1 += s1;
        }
    }
    if ((threadIdx.x >> 4) == 0 && c < C) {
        if (dbeta) dbeta[c] = (float)t0;
        if (dgamma) dgamma[c] = (float)t1;
    }
}

// pass 3: dz = scale * (g - s0/M - xhat * s1/M)
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dA, int ldA, const T* __restrict__ z, const float* __restrict__ bn,
                                    const float* __restrict__ sums, int pix_per_group, int C, size_t total_units,
                                    T* __restrict__ dz) {
    constexpr int EPU = ET<T>::EPU;
    const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= total_units) return;
    const int CU = C / EPU;
    const size_t pix = u / CU; const int c = (u % CU) * EPU;
    const int g = pix / pix_per_group;
    const float invM = 1.f / (float)pix_per_group;
    float fz[EPU], fg[EPU], o[EPU];
    Unit<T>::unpack(*reinterpret_cast<const uint4*>(z + pix * C + c), fz);
    Unit<T>::unpack(*reinterpret_cast<const uint4*>(dA + pix * ldA + c), fg);
#pragma unroll
    for (int i = 0; i < EPU; i++) {
        const float mean = bn_row(bn, g, 0, C)[c + i], inv = bn_row(bn, g, 1, C)[c + i];
        const float sc = bn_row(bn, g, 2, C)[c + i], sh = bn_row(bn, g, 3, C)[c + i];
        const float gm = fmaf(fz[i], sc, sh) > 0.f ? fg[i] : 0.f;
        const float xhat = (fz[i] - mean) * inv;
        o[i] = sc * (gm - sums[((size_t)g * 2 + 0) * C + c + i] * invM - xhat * sums[((size_t)g * 2 + 1) * C + c + i] * invM);
    }
    *reinterpret_cast<uint4*>(dz + pix * C + c) = Unit<T>::pack(o);
}

static inline int bnb_blocks_per_group(int pix_per_group) { return (pix_per_group + BNB_PIX - 1) / BNB_PIX; }

extern "C" size_t bdn_bn_bwd_workspace_bytes(int N, int H, int W, int C) {
    // worst case one group per image set; blocks never straddle groups so bound with N groups of H*W
    size_t blocks = (size_t)N * bnb_blocks_per_group(H * W) + 2;
    return blocks * 2 * C * sizeof(float);
}

template <typename T>
static int bn_bwd_impl(const void* dA, int ldA, const void* z, const float* bn, int imgs_per_group,
                       int N, int H, int W, int C, float* ws, float* sums, float* dgamma, float* dbeta, void* dz, hipStream_t st) {
    constexpr int EPU = ET<T>::EPU;
    const int G = N / imgs_per_group;
    const int ppg = imgs_per_group * H * W;
    const int bpg = bnb_blocks_per_group(ppg);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<T>, dim3(G * bpg), dim3(256), 256 * EPU * 2 * sizeof(float), st,
                       (const T*)dA, ldA, (const T*)z, bn, ppg, bpg, C, ws);
    BDN_CHECK_LAUNCH("bn_bwd_reduce");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, ws, bpg, G, C, sums, dgamma, dbeta);
    BDN_CHECK_LAUNCH("bn_bwd_finalize");
    const size_t units = (size_t)N * H * W * (C / EPU);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(grid_for(units)), dim3(256), 0, st,
                       (const T*)dA, ldA, (const T*)z, bn, sums, ppg, C, units, (T*)dz);
    BDN_CHECK_LAUNCH("bn_bwd_apply");
    return BDN_OK;
}

extern "C" int bdn_bn_bwd(int dtype, const void* dA, int ldA, const void* z, const float* bn,
                          int imgs_per_group, int N, int H, int W, int C,
                          float* ws, float* sums, float* dgamma, float* dbeta, void* dz, void* stream) {
    if (!dA || !z || !bn || !ws || !sums || !dz) BDN_FAIL(BDN_E_ARG, "bn_bwd: null pointer");
    if (N <= 0 || imgs_per_group <= 0 || N % imgs_per_group || C % 16 || ldA < C || ldA % 16) BDN_FAIL(BDN_E_SHAPE, "bn_bwd: bad shape");
    if (C > 1024 || 1024 % C) BDN_FAIL(BDN_E_SHAPE, "bn_bwd: C=%d must divide 1024", C);
    if (dtype == BDN_BF16) return bn_bwd_impl<bf16s>(dA, ldA, z, bn, imgs_per_group, N, H, W, C, ws, sums, dgamma, dbeta, dz, (hipStream_t)stream);
    if (dtype == BDN_F32) return bn_bwd_impl<float>(dA, ldA, z, bn, imgs_per_group, N, H, W, C, ws, sums, dgamma, dbeta, dz, (hipStream_t)stream);
    BDN_FAIL(BDN_E_ARG, "bn_bwd: bad dtype");
}

// ============================================================ bnrelu + MaxPool2d(2) (unet_parts.py:40)
template <typename T>
__global__ void bnrelu_pool_kernel(const T* __restrict__ z, const float* __restrict__ bn, int imgs_per_group,
                                   T* __restrict__ out, int N, int H, int W, int C, size_t total_units) {
    constexpr int EPU = ET<T>::EPU;
    const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= total_units) return;
    const int Ho = H / 2, Wo = W / 2, CU = C / EPU;
    const int c = (u % CU) * EPU; size_t t = u / CU;
    const int xo = t % Wo; t /= Wo; const int yo = t % Ho; const int n = t / Ho;
    const int g = n / imgs_per_group;
    const float* sc = bn_row(bn, g, 2, C) + c; const float* sh = bn_row(bn, g, 3, C) + c;
    float m[EPU];
#pragma unroll
    for (int i = 0; i < EPU; i++) m[i] = 0.f;                 // post-ReLU values are >= 0
#pragma unroll
    for (int dy = 0; dy < 2; dy++)
#pragma unroll
        for (int dx = 0; dx < 2; dx++) {
            float f[EPU];
            Unit<T>::unpack(*reinterpret_cast<const uint4*>(z + ((size_t)(n * H + 2 * yo + dy) * W + 2 * xo + dx) * C + c), f);
#pragma unroll
            for (int i = 0; i < EPU; i++) m[i] = fmaxf(m[i], to_f(from_f<T>(fmaxf(fmaf(f[i], sc[i], sh[i]), 0.f))));
        }
    *reinterpret_cast<uint4*>(out + u * EPU) = Unit<T>::pack(m);
}

extern "C" int bdn_bnrelu_pool(int dtype, const void* z, const float* bn, int imgs_per_group,
                               void* out, int N, int H, int W, int C, void* stream) {
    if (!z || !bn || !out) BDN_FAIL(BDN_E_ARG, "bnrelu_pool: null pointer");
    if (H < 2 || W < 2 || C % 16 || imgs_per_group <= 0) BDN_FAIL(BDN_E_SHAPE, "bnrelu_pool: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BDN_BF16) { size_t units = (size_t)N * (H / 2) * (W / 2) * (C / 8);
        hipLaunchKernelGGL(bnrelu_pool_kernel<bf16s>, dim3(grid_for(units)), dim3(256), 0, st, (const bf16s*)z, bn, imgs_per_group, (bf16s*)out, N, H, W, C, units); }
    else if (dtype == BDN_F32) { size_t units = (size_t)N * (H / 2) * (W / 2) * (C / 4);
        hipLaunchKernelGGL(bnrelu_pool_kernel<float>, dim3(grid_for(units)), dim3(256), 0, st, (const float*)z, bn, imgs_per_group, (float*)out, N, H, W, C, units); }
    else BDN_FAIL(BDN_E_ARG, "bnrelu_pool: bad dtype");
    BDN_CHECK_LAUNCH("bnrelu_pool");
    return BDN_OK;
}

// ============================================================ date fusion relu(a_d2 * a_d1) (bidate_model.py:35-38)
template <typename T>
__global__ void fuse_product_kernel(const T* __restrict__ z, const float* __restrict__ bn, T* __restrict__ f,
                                    size_t units_per_date, int C) {
    constexpr int EPU = ET<T>::EPU;
    const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= units_per_date) return;
    const int c = (u % (C / EPU)) * EPU;
    float a[EPU], b[EPU];
    Unit<T>::unpack(*reinterpret_cast<const uint4*>(z + u * EPU), a);
    Unit<T>::unpack(*reinterpret_cast<const uint4*>(z + (units_per_date + u) * EPU), b);
#pragma unroll
    for (int i = 0; i < EPU; i++) {
        // each date's activation is rounded to the storage type first, exactly as every other consumer sees it
        const float a1 = to_f(from_f<T>(fmaxf(fmaf(a[i], bn_row(bn, 0, 2, C)[c + i], bn_row(bn, 0, 3, C)[c + i]), 0.f)));
        const float a2 = to_f(from_f<T>(fmaxf(fmaf(b[i], bn_row(bn, 1, 2, C)[c + i], bn_row(bn, 1, 3, C)[c + i]), 0.f)));
        a[i] = a1 * a2;                                       // >= 0: the reference's relu is a no-op
    }
    *reinterpret_cast<uint4*>(f + u * EPU) = Unit<T>::pack(a);
}

extern "C" int bdn_fuse_product(int dtype, const void* z, const float* bn, void* f,
                                int B, int H, int W, int C, void* stream) {
    if (!z || !bn || !f) BDN_FAIL(BDN_E_ARG, "fuse_product: null pointer");
    if (C % 16) BDN_FAIL(BDN_E_SHAPE, "fuse_product: bad C");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BDN_BF16) { size_t units = (size_t)B * H * W * (C / 8);
        hipLaunchKernelGGL(fuse_product_kernel<bf16s>, dim3(grid_for(units)), dim3(256), 0, st, (const bf16s*)z, bn, (bf16s*)f, units, C); }
    else if (dtype == BDN_F32) { size_t units = (size_t)B * H * W * (C / 4);
        hipLaunchKernelGGL(fuse_product_kernel<float>, dim3(grid_for(units)), dim3(256), 0, st, (const float*)z, bn, (float*)f, units, C); }
    else BDN_FAIL(BDN_E_ARG, "fuse_product: bad dtype");
    BDN_CHECK_LAUNCH("fuse_product");
    return BDN_OK;
}

// ============================================================ bilinear x2, align_corners=True, + F.pad (unet_parts.py:56-58,68-72)
// src index/weight of destination index d (ATen area_pixel_compute_source_index, align_corners)
__device__ __forceinline__ void up_tap(int d, int n_in, int n_out, int& i0, int& i1, float& lam) {
    const float scale = n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f;
    const float s = scale * (float)d;
    i0 = (int)s; if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    lam = s - (float)i0;
}

template <typename T>
__global__ void upsample2x_kernel(const T* __restrict__ src, const float* __restrict__ bn, T* __restrict__ out,
                                  int B, int h, int w, int H, int W, int C, size_t total_units) {
    constexpr int EPU = ET<T>::EPU;
    const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= total_units) return;
    const int CU = C / EPU;
    const int c = (u % CU) * EPU; size_t t = u / CU;
    const int X = t % W; t /= W; const int Y = t % H; const int n = t / H;
    const int top = (H - 2 * h) / 2, left = (W - 2 * w) / 2;
    const int yy = Y - top, xx = X - left;
    float o[EPU];
#pragma unroll
    for (int i = 0; i < EPU; i++) o[i] = 0.f;
    if (yy >= 0 && yy < 2 * h && xx >= 0 && xx < 2 * w) {
        int y0, y1, x0, x1; float ly, lx;
        up_tap(yy, h, 2 * h, y0, y1, ly); up_tap(xx, w, 2 * w, x0, x1, lx);
        const int ys[2] = {y0, y1}, xs[2] = {x0, x1};
        const float wy[2] = {1.f - ly, ly}, wx[2] = {1.f - lx, lx};
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
                float f[EPU];
                Unit<T>::unpack(*reinterpret_cast<const uint4*>(src + ((size_t)(n * h + ys[a]) * w + xs[b]) * C + c), f);
#pragma unroll
                for (int i = 0; i < EPU; i++) {
                    float v = f[i];
                    if (bn) v = to_f(from_f<T>(fmaxf(fmaf(v, bn_row(bn, 0, 2, C)[c + i], bn_row(bn, 0, 3, C)[c + i]), 0.f)));
                    o[i] += wy[a] * wx[b] * v;
                }
            }
    }
    *reinterpret_cast<uint4*>(out + u * EPU) = Unit<T>::pack(o);
}

extern "C" int bdn_upsample2x(int dtype, const void* src, int in_mode, const float* bn,
                              void* out, int B, int h, int w, int H, int W, int C, void* stream) {
    if (!src || !out) BDN_FAIL(BDN_E_ARG, "upsample2x: null pointer");
    if (in_mode == BDN_IN_BNRELU && !bn) BDN_FAIL(BDN_E_ARG, "upsample2x: BNRELU needs bn");
    if (H < 2 * h || W < 2 * w || C % 16) BDN_FAIL(BDN_E_SHAPE, "upsample2x: bad shape");
    const float* b = in_mode == BDN_IN_BNRELU ? bn : nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BDN_BF16) { size_t units = (size_t)B * H * W * (C / 8);
        hipLaunchKernelGGL(upsample2x_kernel<bf16s>, dim3(grid_for(units)), dim3(256), 0, st, (const bf16s*)src, b, (bf16s*)out, B, h, w, H, W, C, units); }
    else if (dtype == BDN_F32) { size_t units = (size_t)B * H * W * (C / 4);
        hipLaunchKernelGGL(upsample2x_kernel<float>, dim3(grid_for(units)), dim3(256), 0, st, (const float*)src, b, (float*)out, B, h, w, H, W, C, units); }
    else BDN_FAIL(BDN_E_ARG, "upsample2x: bad dtype");
    BDN_CHECK_LAUNCH("upsample2x");
    return BDN_OK;
}

// transpose: every source pixel gathers from the destination rows/cols that read it
template <typename T>
__global__ void upsample2x_bwd_kernel(const T* __restrict__ dU, int ldU, T* __restrict__ dsrc,
                                      int B, int h, int w, int H, int W, int C, size_t total_units) {
    constexpr int EPU = ET<T>::EPU;
    const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= total_units) return;
    const int CU = C / EPU;
    const int c = (u % CU) * EPU; size_t t = u / CU;
    const int x = t % w; t /= w; const int y = t % h; const int n = t / h;
    const int top = (H - 2 * h) / 2, left = (W - 2 * w) / 2;
    float o[EPU];
#pragma unroll
    for (int i = 0; i < EPU; i++) o[i] = 0.f;
    const int ylo = max(0, 2 * y - 3), yhi = min(2 * h - 1, 2 * y + 5);
    const int xlo = max(0, 2 * x - 3), xhi = min(2 * w - 1, 2 * x + 5);
    for (int dy = ylo; dy <= yhi; dy++) {
        int a0, a1; float ly; up_tap(dy, h, 2 * h, a0, a1, ly);
        const float wy = (a0 == y ? 1.f - ly : 0.f) + (a1 == y ? ly : 0.f);
        if (wy == 0.f) continue;
        for (int dx = xlo; dx <= xhi; dx++) {
            int b0, b1; float lx; up_tap(dx, w, 2 * w, b0, b1, lx);
            const float wx = (b0 == x ? 1.f - lx : 0.f) + (b1 == x ? lx : 0.f);
            if (wx == 0.f) continue;
            float f[EPU];
            Unit<T>::unpack(*reinterpret_cast<const uint4*>(dU + ((size_t)(n * H + dy + top) * W + dx + left) * ldU + c), f);
#pragma unroll
            for (int i = 0; i < EPU; i++) o[i] += wy * wx * f[i];
        }
    }
    *reinterpret_cast<uint4*>(dsrc + u * EPU) = Unit<T>::pack(o);
}

extern "C" int bdn_upsample2x_bwd(int dtype, const void* dU, int ldU, void* dsrc,
                                  int B, int h, int w, int H, int W, int C, void* stream) {
    if (!dU || !dsrc) BDN_FAIL(BDN_E_ARG, "upsample2x_bwd: null pointer");
    if (H < 2 * h || W < 2 * w || C % 16 || ldU < C || ldU % 16) BDN_FAIL(BDN_E_SHAPE, "upsample2x_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BDN_BF16) { size_t units = (size_t)B * h * w * (C / 8);
        hipLaunchKernelGGL(upsample2x_bwd_kernel<bf16s>, dim3(grid_for(units)), dim3(256), 0, st, (const bf16s*)dU, ldU, (bf16s*)dsrc, B, h, w, H, W, C, units); }
    else if (dtype == BDN_F32) { size_t units = (size_t)B * h * w * (C / 4);
        hipLaunchKernelGGL(upsample2x_bwd_kernel<float>, dim3(grid_for(units)), dim3(256), 0, st, (const float*)dU, ldU, (float*)dsrc, B, h, w, H, W, C, units); }
    else BDN_FAIL(BDN_E_ARG, "upsample2x_bwd: bad dtype");
    BDN_CHECK_LAUNCH("upsample2x_bwd");
    return BDN_OK;
}

// ============================================================ backward of product fusion + max-pool into encoder outputs
// one thread = one 2x2 window x EPU channels x both dates
template <typename T>
__global__ void enc_skip_bwd_kernel(const T* __restrict__ dF, int ldF, const T* __restrict__ z, const float* __restrict__ bn,
                                    const T* __restrict__ dP, T* __restrict__ dA, int B, int H, int W, int C, size_t total_units) {
    constexpr int EPU = ET<T>::EPU;
    const size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= total_units) return;
    const int CU = C / EPU, Hc = (H + 1) / 2, Wc = (W + 1) / 2, Ho = H / 2, Wo = W / 2;
    const int c = (u % CU) * EPU; size_t t = u / CU;
    const int xc = t % Wc; t /= Wc; const int yc = t % Hc; const int b = t / Hc;
    float act[2][4][EPU];                                     // [date][window pos][channel]
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int y = 2 * yc + (k >> 1), x = 2 * xc + (k & 1);
        ok[k] = y < H && x < W;
#pragma unroll
        for (int d = 0; d < 2; d++) {
            float f[EPU];
            if (ok[k]) Unit<T>::unpack(*reinterpret_cast<const uint4*>(z + ((size_t)((d * B + b) * H + y) * W + x) * C + c), f);
#pragma unroll
            for (int i = 0; i < EPU; i++)
                act[d][k][i] = ok[k] ? to_f(from_f<T>(fmaxf(fmaf(f[i], bn_row(bn, d, 2, C)[c + i], bn_row(bn, d, 3, C)[c + i]), 0.f))) : 0.f;
        }
    }
    const bool pooled = dP != nullptr && yc < Ho && xc < Wo;  // floor-mode pooling: a trailing odd row/col is unpooled
    float gp[2][EPU];
    int arg[2][EPU];
#pragma unroll
    for (int d = 0; d < 2; d++) {
        if (pooled) Unit<T>::unpack(*reinterpret_cast<const uint4*>(dP + ((size_t)((d * B + b) * Ho + yc) * Wo + xc) * C + c), gp[d]);
#pragma unroll
        for (int i = 0; i < EPU; i++) {
            int am = 0; float m = act[d][0][i];
#pragma unroll
            for (int k = 1; k < 4; k++) if (act[d][k][i] > m) { m = act[d][k][i]; am = k; }   // first maximum wins
            arg[d][i] = am;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (!ok[k]) continue;
        const int y = 2 * yc + (k >> 1), x = 2 * xc + (k & 1);
        float g[EPU];
        Unit<T>::unpack(*reinterpret_cast<const uint4*>(dF + ((size_t)(b * H + y) * W + x) * ldF + c), g);
#pragma unroll
        for (int d = 0; d < 2; d++) {
            float o[EPU];
#pragma unroll
            for (int i = 0; i < EPU; i++) {
                o[i] = g[i] * act[1 - d][k][i];
                if (pooled && arg[d][i] == k) o[i] += gp[d][i];
            }
            *reinterpret_cast<uint4*>(dA + ((size_t)((d * B + b) * H + y) * W + x) * C + c) = Unit<T>::pack(o);
        }
    }
}

extern "C" int bdn_enc_skip_bwd(int dtype, const void* dF, int ldF, const void* z, const float* bn,
                                const void* dP, void* dA, int B, int H, int W, int C, void* stream) {
    if (!dF || !z || !bn || !dA) BDN_FAIL(BDN_E_ARG, "enc_skip_bwd: null pointer");
    if (C % 16 || ldF < C || ldF % 16) BDN_FAIL(BDN_E_SHAPE, "enc_skip_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const size_t cells = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2);
    if (dtype == BDN_BF16) { size_t units = cells * (C / 8);
        hipLaunchKernelGGL(enc_skip_bwd_kernel<bf16s>, dim3(grid_for(units)), dim3(256), 0, st, (const bf16s*)dF, ldF, (const bf16s*)z, bn, (const bf16s*)dP, (bf16s*)dA, B, H, W, C, units); }
    else if (dtype == BDN_F32) { size_t units = cells * (C / 4);
        hipLaunchKernelGGL(enc_skip_bwd_kernel<float>, dim3(grid_for(units)), dim3(256), 0, st, (const float*)dF, ldF, (const float*)z, bn, (const float*)dP, (float*)dA, B, H, W, C, units); }
    else BDN_FAIL(BDN_E_ARG, "enc_skip_bwd: bad dtype");
    BDN_CHECK_LAUNCH("enc_skip_bwd");
    return BDN_OK;
}

// ============================================================ outconv 1x1 (unet_parts.py:86)
constexpr int OUTC_MAXCLS = 8;
template <typename T>
__global__ void outc_fwd_kernel(const T* __restrict__ z, const float* __restrict__ bn, const float* __restrict__ w,
                                const float* __restrict__ bias, float* __restrict__ logits, int B, int H, int W, int C, int ncls) {
    constexpr int EPU = ET<T>::EPU;
    const size_t npix = (size_t)B * H * W;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float acc[OUTC_MAXCLS];
#pragma unroll
    for (int k = 0; k < OUTC_MAXCLS; k++) acc[k] = k < ncls ? bias[k] : 0.f;
    for (int c = 0; c < C; c += EPU) {
        float f[EPU];
        Unit<T>::unpack(*reinterpret_cast<const uint4*>(z + p * C + c), f);
#pragma unroll
        for (int i = 0; i < EPU; i++) {
            const float a = to_f(from_f<T>(fmaxf(fmaf(f[i], bn_row(bn, 0, 2, C)[c + i], bn_row(bn, 0, 3, C)[c + i]), 0.f)));
#pragma unroll
            for (int k = 0; k < OUTC_MAXCLS; k++) if (k < ncls) acc[k] = fmaf(a, w[k * C + c + i], acc[k]);
        }
    }
    const size_t hw = (size_t)H * W; const size_t b = p / hw, q = p % hw;
    for (int k = 0; k < ncls; k++) logits[(b * ncls + k) * hw + q] = acc[k];
}

extern "C" int bdn_outc_fwd(int dtype, const void* z, const float* bn, const float* w, const float* b,
                            float* logits, int B, int H, int W, int C, int ncls, void* stream) {
    if (!z || !bn || !w || !b || !logits) BDN_FAIL(BDN_E_ARG, "outc_fwd: null pointer");
    if (ncls < 1 || ncls > OUTC_MAXCLS || C % 16) BDN_FAIL(BDN_E_SHAPE, "outc_fwd: ncls=%d (max %d), C=%d", ncls, OUTC_MAXCLS, C);
    hipStream_t st = (hipStream_t)stream; const size_t npix = (size_t)B * H * W;
    if (dtype == BDN_BF16) hipLaunchKernelGGL(outc_fwd_kernel<bf16s>, dim3(grid_for(npix)), dim3(256), 0, st, (const bf16s*)z, bn, w, b, logits, B, H, W, C, ncls);
    else if (dtype == BDN_F32) hipLaunchKernelGGL(outc_fwd_kernel<float>, dim3(grid_for(npix)), dim3(256), 0, st, (const float*)z, bn, w, b, logits, B, H, W, C, ncls);
    else BDN_FAIL(BDN_E_ARG, "outc_fwd: bad dtype");
    BDN_CHECK_LAUNCH("outc_fwd");
    return BDN_OK;
}

// backward: dA[p][c] = sum_k dl[k][p] w[k][c];  dw[k][c] = sum_p dl[k][p] a[p][c];  db[k] = sum_p dl[k][p]
// block = 256 pixels; dw/db block partials are combined with f32 atomics on a zeroed buffer (130 addresses).
template <typename T>
__global__ void outc_bwd_kernel(const float* __restrict__ dl, const T* __restrict__ z, const float* __restrict__ bn,
                                const float* __restrict__ w, T* __restrict__ dA, float* __restrict__ dw, float* __restrict__ db,
                                int B, int H, int W, int C, int ncls) {
    constexpr int EPU = ET<T>::EPU;
    extern __shared__ float sm[];                             // [ncls][C+1] block accumulators
    const size_t npix = (size_t)B * H * W, hw = (size_t)H * W;
    for (int i = threadIdx.x; i < ncls * (C + 1); i += blockDim.x) sm[i] = 0.f;
    __syncthreads();
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float g[OUTC_MAXCLS];
    const bool live = p < npix;
    if (live) { const size_t b = p / hw, q = p % hw;
        for (int k = 0; k < ncls; k++) g[k] = dl[(b * ncls + k) * hw + q]; }
    else for (int k = 0; k < ncls; k++) g[k] = 0.f;
    for (int c = 0; c < C; c += EPU) {
        float f[EPU], o[EPU];
        if (live) Unit<T>::unpack(*reinterpret_cast<const uint4*>(z + p * C + c), f);
#pragma unroll
        for (int i = 0; i < EPU; i++) {
            const float a = live ? to_f(from_f<T>(fmaxf(fmaf(f[i], bn_row(bn, 0, 2, C)[c + i], bn_row(bn, 0, 3, C)[c + i]), 0.f))) : 0.f;
            float s = 0.f;
            for (int k = 0; k < ncls; k++) {
                s = fmaf(g[k], w[k * C + c + i], s);
                float v = g[k] * a;                           // wave-reduce before touching LDS
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                if ((threadIdx.x & 63) == 0) atomicAdd(&sm[k * (C + 1) + c + i], v);
            }
            o[i] = s;
        }
        if (live) *reinterpret_cast<uint4*>(dA + p * C + c) = Unit<T>::pack(o);
    }
    for (int k = 0; k < ncls; k++) {
        float v = g[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) atomicAdd(&sm[k * (C + 1) + C], v);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncls * (C + 1); i += blockDim.x) {
        const int k = i / (C + 1), c = i % (C + 1);
        if (c < C) atomicAdd(&dw[k * C + c], sm[i]); else atomicAdd(&db[k], sm[i]);
    }
}

extern "C" int bdn_outc_bwd(int dtype, const float* dlogits, const void* z, const float* bn, const float* w,
                            void* dA, float* dw, float* db, int B, int H, int W, int C, int ncls, void* stream) {
    if (!dlogits || !z || !bn || !w || !dA || !dw || !db) BDN_FAIL(BDN_E_ARG, "outc_bwd: null pointer");
    if (ncls < 1 || ncls > OUTC_MAXCLS || C % 16) BDN_FAIL(BDN_E_SHAPE, "outc_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream; const size_t npix = (size_t)B * H * W;
    hipMemsetAsync(dw, 0, sizeof(float) * ncls * C, st);
    hipMemsetAsync(db, 0, sizeof(float) * ncls, st);
    const size_t smem = sizeof(float) * ncls * (C + 1);
    if (dtype == BDN_BF16) hipLaunchKernelGGL(outc_bwd_kernel<bf16s>, dim3(grid_for(npix)), dim3(256), smem, st, dlogits, (const bf16s*)z, bn, w, (bf16s*)dA, dw, db, B, H, W, C, ncls);
    else if (dtype == BDN_F32) hipLaunchKernelGGL(outc_bwd_kernel<float>, dim3(grid_for(npix)), dim3(256), smem, st, dlogits, (const float*)z, bn, w, (float*)dA, dw, db, B, H, W, C, ncls);
    else BDN_FAIL(BDN_E_ARG, "outc_bwd: bad dtype");
    BDN_CHECK_LAUNCH("outc_bwd");
    return BDN_OK;
}

// ============================================================ Tversky loss (utils/metrics.py:130-171, dims == (0,2))
// sums[k][c][w], k = 0 TP, 1 FP, 2 FN, reduced over batch and H for every (class, column w).
// pass 1: grid (B*H rows) -> atomics on [3][ncls][W] (one add per row and address);  pass 2: single block
// loss + coefficient tables;  pass 3: dlogits.
__global__ void tversky_sums_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ labels,
                                    float* __restrict__ sums, int32_t* __restrict__ counts, int B, int ncls, int H, int W) {
    // block = 256 threads = columns; grid.x = column blocks, grid.y = row groups of RG rows
    constexpr int RG = 16;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t hw = (size_t)H * W;
    float tp[OUTC_MAXCLS], fp[OUTC_MAXCLS], fn[OUTC_MAXCLS];
#pragma unroll
    for (int k = 0; k < OUTC_MAXCLS; k++) { tp[k] = 0.f; fp[k] = 0.f; fn[k] = 0.f; }
    int c_tp = 0, c_fp = 0, c_fn = 0, c_ok = 0;
    const int rows = B * H;
    if (x < W)
        for (int r = blockIdx.y * RG; r < min(rows, (blockIdx.y + 1) * RG); r++) {
            const int b = r / H, y = r % H;
            const size_t q = (size_t)y * W + x;
            float l[OUTC_MAXCLS]; float m = -INFINITY; int am = 0;
#pragma unroll
            for (int k = 0; k < OUTC_MAXCLS; k++) if (k < ncls) { l[k] = logits[((size_t)b * ncls + k) * hw + q]; if (l[k] > m) { m = l[k]; am = k; } }
            float den = 0.f;
#pragma unroll
            for (int k = 0; k < OUTC_MAXCLS; k++) if (k < ncls) { l[k] = expf(l[k] - m); den += l[k]; }
            const int t = labels[(size_t)b * hw + q];
#pragma unroll
            for (int k = 0; k < OUTC_MAXCLS; k++) if (k < ncls) {
                const float p = l[k] / den;
                if (t == k) { tp[k] += p; fn[k] += 1.f - p; } else fp[k] += p;
            }
            c_tp += (am == 1 && t == 1); c_fp += (am == 1 && t != 1); c_fn += (am != 1 && t == 1); c_ok += (am == t);
        }
    if (x < W)
        for (int k = 0; k < ncls; k++) {
            atomicAdd(&sums[(0 * ncls + k) * W + x], tp[k]);
            atomicAdd(&sums[(1 * ncls + k) * W + x], fp[k]);
            atomicAdd(&sums[(2 * ncls + k) * W + x], fn[k]);
        }
    if (counts) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            c_tp += __shfl_xor(c_tp, off); c_fp += __shfl_xor(c_fp, off); c_fn += __shfl_xor(c_fn, off); c_ok += __shfl_xor(c_ok, off);
        }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&counts[0], c_tp); atomicAdd(&counts[1], c_fp); atomicAdd(&counts[2], c_fn); atomicAdd(&counts[3], c_ok); }
    }
}

// loss = 1 - mean_{c,w} TP/(TP + a FP + b FN + eps).  Overwrites sums[0] with 1/D and sums[1] with TP/D^2.
__global__ void tversky_finish_kernel(float* __restrict__ sums, float alpha, float beta, float eps, int ncls, int W, float* __restrict__ loss) {
    __shared__ double red[256];
    double acc = 0.0;
    const int n = ncls * W;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float tp = sums[i], fp = sums[n + i], fn = sums[2 * n + i];
        const float D = tp + alpha * fp + beta * fn + eps;
        acc += (double)(tp / D);
        sums[i] = 1.f / D; sums[n + i] = tp / (D * D);
    }
    red[threadIdx.x] = acc; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) *loss = (float)(1.0 - red[0] / n);
}

__global__ void tversky_bwd_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ labels,
                                   const float* __restrict__ coef, float alpha, float beta, float* __restrict__ dlogits,
                                   int B, int ncls, int H, int W) {
    const size_t hw = (size_t)H * W, npix = (size_t)B * hw;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const size_t b = p / hw, q = p % hw; const int x = q % W;
    const int n = ncls * W;
    float l[OUTC_MAXCLS], dp[OUTC_MAXCLS]; float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < OUTC_MAXCLS; k++) if (k < ncls) { l[k] = logits[(b * ncls + k) * hw + q]; m = fmaxf(m, l[k]); }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < OUTC_MAXCLS; k++) if (k < ncls) { l[k] = expf(l[k] - m); den += l[k]; }
    const int t = labels[p];
    const float norm = -1.f / (float)n;
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < OUTC_MAXCLS; k++) if (k < ncls) {
        l[k] /= den;
        const float invD = coef[k * W + x], tpD2 = coef[n + k * W + x];
        const float tk = t == k ? 1.f : 0.f;
        // d(TP/D)/dp = t/D - TP/D^2 * (t + alpha (1-t) - beta t)
        dp[k] = norm * (tk * invD - tpD2 * (tk + alpha * (1.f - tk) - beta * tk));
        dot += l[k] * dp[k];
    }
#pragma unroll
    for (int k = 0; k < OUTC_MAXCLS; k++) if (k < ncls) dlogits[(b * ncls + k) * hw + q] = l[k] * (dp[k] - dot);
}

extern "C" int bdn_tversky(const float* logits, const uint8_t* labels, float alpha, float beta, float eps,
                           float* ws, float* loss, int32_t* counts, float* dlogits,
                           int B, int ncls, int H, int W, void* stream) {
    if (!logits || !labels || !ws || !loss) BDN_FAIL(BDN_E_ARG, "tversky: null pointer");
    if (ncls < 2 || ncls > OUTC_MAXCLS) BDN_FAIL(BDN_E_SHAPE, "tversky: ncls=%d unsupported (2..%d)", ncls, OUTC_MAXCLS);
    hipStream_t st = (hipStream_t)stream;
    hipMemsetAsync(ws, 0, sizeof(float) * 3 * ncls * W, st);
    if (counts) hipMemsetAsync(counts, 0, sizeof(int32_t) * 4, st);
    dim3 grid((W + 255) / 256, (B * H + 15) / 16);
    hipLaunchKernelGGL(tversky_sums_kernel, grid, dim3(256), 0, st, logits, labels, ws, counts, B, ncls, H, W);
    BDN_CHECK_LAUNCH("tversky_sums");
    hipLaunchKernelGGL(tversky_finish_kernel, dim3(1), dim3(256), 0, st, ws, alpha, beta, eps, ncls, W, loss);
    BDN_CHECK_LAUNCH("tversky_finish");
    if (dlogits) {
        hipLaunchKernelGGL(tversky_bwd_kernel, dim3(grid_for((size_t)B * H * W)), dim3(256), 0, st, logits, labels, ws, alpha, beta, dlogits, B, ncls, H, W);
        BDN_CHECK_LAUNCH("tversky_bwd");
    }
    return BDN_OK;
}

// ============================================================ SGD (train.py:55,95)
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float step, size_t n4, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        float4 a = reinterpret_cast<float4*>(p)[i]; const float4 b = reinterpret_cast<const float4*>(g)[i];
        a.x -= step * b.x; a.y -= step * b.y; a.z -= step * b.z; a.w -= step * b.w;
        reinterpret_cast<float4*>(p)[i] = a;
    }
    if (i == 0) for (size_t k = n4 * 4; k < n; k++) p[k] -= step * g[k];
}

extern "C" int bdn_sgd_step(float* params, const float* grads, float lr, float grad_scale, size_t n, void* stream) {
    if (!params || !grads) BDN_FAIL(BDN_E_ARG, "sgd_step: null pointer");
    if (((uintptr_t)params | (uintptr_t)grads) & 15) BDN_FAIL(BDN_E_ARG, "sgd_step: buffers must be 16-byte aligned");
    if (n == 0) return BDN_OK;
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n4 > 0 ? n4 : 1)), dim3(256), 0, (hipStream_t)stream, params, grads, lr * grad_scale, n4, n);
    BDN_CHECK_LAUNCH("sgd_step");
    return BDN_OK;
}

// ============================================================ misc
static thread_local char g_err[512] = "";
void bdn_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* bdn_last_error(void) { return g_err; }
extern "C" int bdn_version(void) { return 1; }
