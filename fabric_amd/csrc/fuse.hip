// Stage-boundary kernels between the convolutions (all HBM-bound, NHWC, 16-byte channel units):
//   bnrelu_pool   nn.MaxPool2d(2) of relu(bn(z))                 reference models/unet_parts.py:40
//   fuse_product  torch.relu(x_d2 * x_d1)                        reference models/bidate_model.py:35-38
//   upsample2x    nn.Upsample(x2, bilinear, align_corners) + pad reference models/unet_parts.py:56-58,68-72
//   ..._bwd       their backward gathers (no atomics: every output element gathers its contributions)
// Every thread owns one channel unit (its BatchNorm scale/shift stay in registers) and walks pixels.
#include "common.hpp"

template <typename T>
__device__ __forceinline__ void load_consts(const float* __restrict__ p, float (&o)[ET<T>::EPU]) {
#pragma unroll
    for (int i = 0; i < ET<T>::EPU; i++) o[i] = p[i];
}
// a = storage-rounded relu(z*sc+sh), exactly what the conv kernels feed the MFMA
template <typename T>
__device__ __forceinline__ float act1(float z, float sc, float sh) {
    return to_f(from_f<T>(fmaxf(fmaf(z, sc, sh), 0.f)));
}

// a lane's channel unit of E elements: 16 bytes (Unit<T>) or, for bf16 kernels that are short of registers, 8 bytes
template <typename T, int E> struct UnitE;
template <> struct UnitE<bf16s, 8> : Unit<bf16s> { using V = uint4; __device__ __forceinline__ static V zero() { return make_uint4(0, 0, 0, 0); } };
template <> struct UnitE<float, 4> : Unit<float> { using V = uint4; __device__ __forceinline__ static V zero() { return make_uint4(0, 0, 0, 0); } };
template <> struct UnitE<bf16s, 4> {
    using V = uint2;
    __device__ __forceinline__ static V zero() { return make_uint2(0, 0); }
    __device__ __forceinline__ static void unpack(const uint2& u, float* f) {
        f[0] = bf2f(u.x & 0xffffu); f[1] = bf2f(u.x >> 16); f[2] = bf2f(u.y & 0xffffu); f[3] = bf2f(u.y >> 16);
    }
    __device__ __forceinline__ static uint2 pack(const float* f) { return make_uint2(f2bf2(f[0], f[1]), f2bf2(f[2], f[3])); }
};
template <int E>
__device__ __forceinline__ void load_consts_n(const float* __restrict__ p, float (&o)[E]) {
#pragma unroll
    for (int i = 0; i < E; i++) o[i] = p[i];
}

constexpr int ITERS = 8;        // pixels per thread

// ============================================================ bnrelu + MaxPool2d(2)
template <typename T>
__global__ void bnrelu_pool_kernel(const T* __restrict__ z, const float* __restrict__ bn, int imgs_per_group, int bpg,
                                   T* __restrict__ out, int H, int W, int C) {
    constexpr int EPU = ET<T>::EPU;
    const int CU = C / EPU, rows = 256 / CU, tid = threadIdx.x, cu = tid % CU, row = tid / CU, c = cu * EPU;
    const int Ho = H / 2, Wo = W / 2;
    const int g = blockIdx.x / bpg, bg = blockIdx.x % bpg;
    const int ppg = imgs_per_group * Ho * Wo;
    float sc[EPU], sh[EPU];
    load_consts<T>(bn_row(bn, g, 2, C) + c, sc); load_consts<T>(bn_row(bn, g, 3, C) + c, sh);
    const int p_end = min(ppg, (bg + 1) * rows * ITERS);
    for (int p = bg * rows * ITERS + row; p < p_end; p += rows) {
        const int xo = p % Wo, t = p / Wo, yo = t % Ho, n = g * imgs_per_group + t / Ho;
        float m[EPU];
#pragma unroll
        for (int i = 0; i < EPU; i++) m[i] = 0.f;                 // post-ReLU values are >= 0
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float f[EPU];
            Unit<T>::unpack(*reinterpret_cast<const uint4*>(z + ((size_t)(n * H + 2 * yo + (k >> 1)) * W + 2 * xo + (k & 1)) * C + c), f);
#pragma unroll
            for (int i = 0; i < EPU; i++) m[i] = fmaxf(m[i], act1<T>(f[i], sc[i], sh[i]));
        }
        *reinterpret_cast<uint4*>(out + ((size_t)(n * Ho + yo) * Wo + xo) * C + c) = Unit<T>::pack(m);
    }
}

extern "C" int bdn_bnrelu_pool(int dtype, const void* z, const float* bn, int imgs_per_group,
                               void* out, int N, int H, int W, int C, void* stream) {
    if (!z || !bn || !out) BDN_FAIL(BDN_E_ARG, "bnrelu_pool: null pointer");
    if (H < 2 || W < 2 || C % 16 || C > 1024 || 1024 % C || imgs_per_group <= 0 || N % imgs_per_group) BDN_FAIL(BDN_E_SHAPE, "bnrelu_pool: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int G = N / imgs_per_group, ppg = imgs_per_group * (H / 2) * (W / 2);
    if (dtype == BDN_BF16) { const int per = 256 / (C / 8) * ITERS, bpg = (ppg + per - 1) / per;
        hipLaunchKernelGGL(bnrelu_pool_kernel<bf16s>, dim3(G * bpg), dim3(256), 0, st, (const bf16s*)z, bn, imgs_per_group, bpg, (bf16s*)out, H, W, C); }
    else if (dtype == BDN_F32) { const int per = 256 / (C / 4) * ITERS, bpg = (ppg + per - 1) / per;
        hipLaunchKernelGGL(bnrelu_pool_kernel<float>, dim3(G * bpg), dim3(256), 0, st, (const float*)z, bn, imgs_per_group, bpg, (float*)out, H, W, C); }
    else BDN_FAIL(BDN_E_ARG, "bnrelu_pool: bad dtype");
    BDN_CHECK_LAUNCH("bnrelu_pool");
    return BDN_OK;
}

// ============================================================ a = relu(bn(z)) materialised (nn.BatchNorm2d + nn.ReLU, models/unet_parts.py:14-15)
// The 3x3 consumers normally apply this while staging; the LDS-DMA weight-gradient kernel cannot (its operands never pass
// through registers), so the training schedule writes the post-activation tensor once per layer, on the weight-gradient
// stream.  Same rounding as every on-load application (act1), so the GEMM operands are bit-identical.
template <typename T>
__global__ void bnrelu_kernel(const T* __restrict__ z, const float* __restrict__ bn, int ppg, int bpg, T* __restrict__ out, int C) {
    constexpr int EPU = ET<T>::EPU;
    const int CU = C / EPU, rows = 256 / CU, tid = threadIdx.x, cu = tid % CU, row = tid / CU, c = cu * EPU;
    const int g = blockIdx.x / bpg, bg = blockIdx.x % bpg;
    float sc[EPU], sh[EPU];
    load_consts<T>(bn_row(bn, g, 2, C) + c, sc); load_consts<T>(bn_row(bn, g, 3, C) + c, sh);
    const int p_end = min(ppg, (bg + 1) * rows * ITERS);
    const size_t base = (size_t)g * ppg;
    for (int p = bg * rows * ITERS + row; p < p_end; p += rows) {
        const uint4 v = *reinterpret_cast<const uint4*>(z + (base + p) * C + c);
        *reinterpret_cast<uint4*>(out + (base + p) * C + c) = bnrelu_unit<T>(v, sc, sh);
    }
}

extern "C" int bdn_bnrelu(int dtype, const void* z, const float* bn, int imgs_per_group, void* out,
                          int N, int H, int W, int C, void* stream) {
    if (!z || !bn || !out) BDN_FAIL(BDN_E_ARG, "bnrelu: null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || C % 16 || C > 1024 || 1024 % C || imgs_per_group <= 0 || N % imgs_per_group)
        BDN_FAIL(BDN_E_SHAPE, "bnrelu: bad shape N=%d H=%d W=%d C=%d imgs_per_group=%d", N, H, W, C, imgs_per_group);
    if ((size_t)imgs_per_group * H * W >= ((size_t)1 << 31)) BDN_FAIL(BDN_E_SHAPE, "bnrelu: group too large");
    hipStream_t st = (hipStream_t)stream;
    const int G = N / imgs_per_group, ppg = imgs_per_group * H * W;
    if (dtype == BDN_BF16) { const int per = 256 / (C / 8) * ITERS, bpg = (ppg + per - 1) / per;
        hipLaunchKernelGGL(bnrelu_kernel<bf16s>, dim3(G * bpg), dim3(256), 0, st, (const bf16s*)z, bn, ppg, bpg, (bf16s*)out, C); }
    else if (dtype == BDN_F32) { const int per = 256 / (C / 4) * ITERS, bpg = (ppg + per - 1) / per;
        hipLaunchKernelGGL(bnrelu_kernel<float>, dim3(G * bpg), dim3(256), 0, st, (const float*)z, bn, ppg, bpg, (float*)out, C); }
    else BDN_FAIL(BDN_E_ARG, "bnrelu: bad dtype");
    BDN_CHECK_LAUNCH("bnrelu");
    return BDN_OK;
}

// ============================================================ date fusion relu(a_d2 * a_d1)
template <typename T>
__global__ void fuse_product_kernel(const T* __restrict__ z, const float* __restrict__ bn, T* __restrict__ f, int npix, int C) {
    constexpr int EPU = ET<T>::EPU;
    const int CU = C / EPU, rows = 256 / CU, tid = threadIdx.x, cu = tid % CU, row = tid / CU, c = cu * EPU;
    float sc0[EPU], sh0[EPU], sc1[EPU], sh1[EPU];
    load_consts<T>(bn_row(bn, 0, 2, C) + c, sc0); load_consts<T>(bn_row(bn, 0, 3, C) + c, sh0);
    load_consts<T>(bn_row(bn, 1, 2, C) + c, sc1); load_consts<T>(bn_row(bn, 1, 3, C) + c, sh1);
    const int p_end = min(npix, (int)(blockIdx.x + 1) * rows * ITERS);
    for (int p = blockIdx.x * rows * ITERS + row; p < p_end; p += rows) {
        float a[EPU], b[EPU];
        Unit<T>::unpack(*reinterpret_cast<const uint4*>(z + (size_t)p * C + c), a);
        Unit<T>::unpack(*reinterpret_cast<const uint4*>(z + ((size_t)npix + p) * C + c), b);
#pragma unroll
        for (int i = 0; i < EPU; i++) a[i] = act1<T>(a[i], sc0[i], sh0[i]) * act1<T>(b[i], sc1[i], sh1[i]);   // >= 0: relu is a no-op
        *reinterpret_cast<uint4*>(f + (size_t)p * C + c) = Unit<T>::pack(a);
    }
}

extern "C" int bdn_fuse_product(int dtype, const void* z, const float* bn, void* f,
                                int B, int H, int W, int C, void* stream) {
    if (!z || !bn || !f) BDN_FAIL(BDN_E_ARG, "fuse_product: null pointer");
    if (C % 16 || C > 1024 || 1024 % C) BDN_FAIL(BDN_E_SHAPE, "fuse_product: bad C");
    hipStream_t st = (hipStream_t)stream;
    const int npix = B * H * W;
    if (dtype == BDN_BF16) { const int per = 256 / (C / 8) * ITERS;
        hipLaunchKernelGGL(fuse_product_kernel<bf16s>, dim3((npix + per - 1) / per), dim3(256), 0, st, (const bf16s*)z, bn, (bf16s*)f, npix, C); }
    else if (dtype == BDN_F32) { const int per = 256 / (C / 4) * ITERS;
        hipLaunchKernelGGL(fuse_product_kernel<float>, dim3((npix + per - 1) / per), dim3(256), 0, st, (const float*)z, bn, (float*)f, npix, C); }
    else BDN_FAIL(BDN_E_ARG, "fuse_product: bad dtype");
    BDN_CHECK_LAUNCH("fuse_product");
    return BDN_OK;
}

// ============================================================ product fusion + MaxPool2d(2) in one pass over z
// reference models/bidate_model.py:35-38 + models/unet_parts.py:40: the skip f = relu(a_d2 * a_d1) of an encoder level
// and the pooled input of the next level (both dates) both start from a = relu(bn(z)) of the same tensor; one
// iteration = one 2x2 window x EPU channels x both dates, so z is read from HBM once instead of twice.
template <typename T>
__global__ void product_pool_kernel(const T* __restrict__ z, const float* __restrict__ bn, T* __restrict__ f, T* __restrict__ pool,
                                    int B, int H, int W, int C, int ncell, SplitOut sf, SplitOut sp, FastDiv dWc, FastDiv dHc, int pool_dates) {
    constexpr int EPU = ET<T>::EPU;
    const int CU = C / EPU, rows = 256 / CU, tid = threadIdx.x, cu = tid % CU, row = tid / CU, c = cu * EPU;
    const int Hc = (H + 1) / 2, Wc = (W + 1) / 2, Ho = H / 2, Wo = W / 2;
    float sc0[EPU], sh0[EPU], sc1[EPU], sh1[EPU];
    load_consts<T>(bn_row(bn, 0, 2, C) + c, sc0); load_consts<T>(bn_row(bn, 0, 3, C) + c, sh0);
    load_consts<T>(bn_row(bn, 1, 2, C) + c, sc1); load_consts<T>(bn_row(bn, 1, 3, C) + c, sh1);
    constexpr int IT = 4;
    const int q_end = min(ncell, (int)(blockIdx.x + 1) * rows * IT);
    for (int q = blockIdx.x * rows * IT + row; q < q_end; q += rows) {
        int xc, t, yc, b; dWc.divmod(q, t, xc); dHc.divmod(t, b, yc);
        float m0[EPU], m1[EPU];
#pragma unroll
        for (int i = 0; i < EPU; i++) { m0[i] = 0.f; m1[i] = 0.f; }     // activations are >= 0
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int y = 2 * yc + (k >> 1), x = 2 * xc + (k & 1);
            if (y < H && x < W) {
                const size_t p0 = (size_t)(b * H + y) * W + x, p1 = (size_t)((B + b) * H + y) * W + x;
                float a0[EPU], a1[EPU];
                Unit<T>::unpack(*reinterpret_cast<const uint4*>(z + p0 * C + c), a0);
                Unit<T>::unpack(*reinterpret_cast<const uint4*>(z + p1 * C + c), a1);
#pragma unroll
                for (int i = 0; i < EPU; i++) {
                    a0[i] = act1<T>(a0[i], sc0[i], sh0[i]); a1[i] = act1<T>(a1[i], sc1[i], sh1[i]);
                    m0[i] = fmaxf(m0[i], a0[i]); m1[i] = fmaxf(m1[i], a1[i]);
                    a0[i] *= a1[i];                                         // >= 0: relu is a no-op
                }
                store_out<T>(f, p0 * C + c, sf, p0, c, a0);
            }
        }
        if (yc < Ho && xc < Wo) {                                           // floor-mode pooling drops a trailing odd row / column
            const size_t q0 = (size_t)(b * Ho + yc) * Wo + xc, q1 = (size_t)((B + b) * Ho + yc) * Wo + xc;
            if (pool_dates & 1) store_out<T>(pool, q0 * C + c, sp, q0, c, m0);      // pool_dates: which dates' pooled maps this launch writes
            if (pool_dates & 2) store_out<T>(pool, q1 * C + c, sp, q1, c, m1);
        }
    }
}

static int product_pool_impl(int dtype, const void* z, const float* bn, void* f, void* pool, SplitOut sf, SplitOut sp,
                             int B, int H, int W, int C, void* stream, int pool_dates = 3) {
    if (C % 16 || C > 1024 || 1024 % C || H < 2 || W < 2) BDN_FAIL(BDN_E_SHAPE, "product_pool: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int ncell = B * ((H + 1) / 2) * ((W + 1) / 2);
    if (dtype == BDN_BF16) { const int per = 256 / (C / 8) * 4;
        hipLaunchKernelGGL(product_pool_kernel<bf16s>, dim3((ncell + per - 1) / per), dim3(256), 0, st, (const bf16s*)z, bn, (bf16s*)f, (bf16s*)pool, B, H, W, C, ncell, sf, sp, FastDiv((W + 1) / 2), FastDiv((H + 1) / 2), pool_dates); }
    else if (dtype == BDN_F32) { const int per = 256 / (C / 4) * 4;
        hipLaunchKernelGGL(product_pool_kernel<float>, dim3((ncell + per - 1) / per), dim3(256), 0, st, (const float*)z, bn, (float*)f, (float*)pool, B, H, W, C, ncell, sf, sp, FastDiv((W + 1) / 2), FastDiv((H + 1) / 2), pool_dates); }
    else BDN_FAIL(BDN_E_ARG, "product_pool: bad dtype");
    BDN_CHECK_LAUNCH("product_pool");
    return BDN_OK;
}

extern "C" int bdn_product_pool(int dtype, const void* z, const float* bn, void* f, void* pool,
                                int B, int H, int W, int C, void* stream) {
    if (!z || !bn || !f || !pool) BDN_FAIL(BDN_E_ARG, "product_pool: null pointer");
    const SplitOut none = {nullptr, 0, 0, 0};
    return product_pool_impl(dtype, z, bn, f, pool, none, none, B, H, W, C, stream);
}

extern "C" int bdn_product_pool_dates(int dtype, const void* z, const float* bn, void* f, void* pool, int pool_dates,
                                      int B, int H, int W, int C, void* stream) {
    if (!z || !bn || !f || !pool) BDN_FAIL(BDN_E_ARG, "product_pool_dates: null pointer");
    if (pool_dates < 0 || pool_dates > 3) BDN_FAIL(BDN_E_ARG, "product_pool_dates: pool_dates must be a mask of bits 0 (date 1) and 1 (date 2)");
    const SplitOut none = {nullptr, 0, 0, 0};
    return product_pool_impl(dtype, z, bn, f, pool, none, none, B, H, W, C, stream, pool_dates);
}

// bf16x3 setting: both outputs leave as the [hi | lo] bf16 operands of the convolutions that consume them -- f into channels [0, C) of the
// decoder stage's two-source operand [B,H,W,f_ld] (lo half at f_half), pool as [2B,H/2,W/2,2C] -- instead of float32 tensors that a
// bdn_split_pack pass would read again.  z float32.
extern "C" int bdn_product_pool_split(const void* z, const float* bn, void* f_split, int f_ld, int f_half, void* pool_split,
                                      int B, int H, int W, int C, void* stream) {
    if (!z || !bn || !f_split || !pool_split) BDN_FAIL(BDN_E_ARG, "product_pool_split: null pointer");
    if (f_half < C || f_ld < f_half + C || f_ld % 8 || f_half % 8) BDN_FAIL(BDN_E_SHAPE, "product_pool_split: bad operand layout ld=%d half=%d", f_ld, f_half);
    const SplitOut sf = {(bf16s*)f_split, f_ld, 0, f_half}, sp = {(bf16s*)pool_split, 2 * C, 0, C};
    return product_pool_impl(BDN_F32, z, bn, nullptr, nullptr, sf, sp, B, H, W, C, stream);
}

// ============================================================ bilinear x2 (align_corners=True) + F.pad
// source index / weight of destination index d (ATen area_pixel_compute_source_index, align_corners)
__device__ __forceinline__ void up_tap(int d, int n_in, float scale, int& i0, int& i1, float& lam) {
    const float s = scale * (float)d;
    i0 = (int)s; if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    lam = s - (float)i0;
}
static inline float up_scale(int n_in) { return n_in > 1 ? (float)(n_in - 1) / (float)(2 * n_in - 1) : 0.f; }

template <typename T>
__global__ void upsample2x_kernel(const T* __restrict__ src, const float* __restrict__ bn, T* __restrict__ out,
                                  int npix, int h, int w, int H, int W, int C, float sy, float sx, SplitOut so) {
    constexpr int EPU = ET<T>::EPU;
    const int CU = C / EPU, rows = 256 / CU, tid = threadIdx.x, cu = tid % CU, row = tid / CU, c = cu * EPU;
    const int top = (H - 2 * h) / 2, left = (W - 2 * w) / 2;
    float sc[EPU], sh[EPU];
    if (bn) { load_consts<T>(bn_row(bn, 0, 2, C) + c, sc); load_consts<T>(bn_row(bn, 0, 3, C) + c, sh); }
    const int p_end = min(npix, (int)(blockIdx.x + 1) * rows * ITERS);
    for (int p = blockIdx.x * rows * ITERS + row; p < p_end; p += rows) {
        const int X = p % W, t = p / W, Y = t % H, n = t / H;
        const int yy = Y - top, xx = X - left;
        float o[EPU];
#pragma unroll
        for (int i = 0; i < EPU; i++) o[i] = 0.f;
        if (yy >= 0 && yy < 2 * h && xx >= 0 && xx < 2 * w) {
            int y0, y1, x0, x1; float ly, lx;
            up_tap(yy, h, sy, y0, y1, ly); up_tap(xx, w, sx, x0, x1, lx);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int ys = (k >> 1) ? y1 : y0, xs = (k & 1) ? x1 : x0;
                const float wgt = ((k >> 1) ? ly : 1.f - ly) * ((k & 1) ? lx : 1.f - lx);
                float f[EPU];
                Unit<T>::unpack(*reinterpret_cast<const uint4*>(src + ((size_t)(n * h + ys) * w + xs) * C + c), f);
#pragma unroll
                for (int i = 0; i < EPU; i++) o[i] += wgt * (bn ? act1<T>(f[i], sc[i], sh[i]) : f[i]);
            }
        }
        store_out<T>(out, (size_t)p * C + c, so, (size_t)p, c, o);
    }
}

// Tiled variant for maps of at least 8x8 sources: a block owns a 16x16 OUTPUT tile x 4 channel units.  The <= 10x10
// source pixels behind it are staged once in LDS with BatchNorm+ReLU already applied (the per-pixel gather above
// re-applies it for every one of the four taps of every output: ~16x per source element, which made the kernel
// VALU-bound at 1.8 TB/s), then each thread blends four outputs from LDS with the same tap order and weights.
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_tiled_kernel(const T* __restrict__ src, const float* __restrict__ bn, T* __restrict__ out,
                                                             int h, int w, int H, int W, int C, int tiles_x, int tiles_y, float sy, float sx, SplitOut so) {
    constexpr int EPU = ET<T>::EPU, TO = 16, RS = 10, UB = 4, PSTR = UB * 16 + 16;
    __shared__ __attribute__((aligned(16))) unsigned char sm[RS * RS * PSTR];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int c0 = blockIdx.y * UB * EPU;
    const int top = (H - 2 * h) / 2, left = (W - 2 * w) / 2;
    const int Y0 = ty * TO, X0 = tx * TO;                      // output tile origin (padded frame)
    // first source row / column any output of the tile can touch (outputs above / left of the upsampled area touch none)
    const int yyf = max(Y0 - top, 0), xxf = max(X0 - left, 0);
    int ys0, xs0, t1; float tl;
    up_tap(min(yyf, 2 * h - 1), h, sy, ys0, t1, tl);
    up_tap(min(xxf, 2 * w - 1), w, sx, xs0, t1, tl);
    // ---- stage the source window, activation applied once per element
    for (int i = tid; i < RS * RS * UB; i += 256) {
        const int uu = i % UB, pix = i / UB, rx = pix % RS, ry = pix / RS;
        const int ys = ys0 + ry, xs = xs0 + rx;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ys < h && xs < w) {
            v = *reinterpret_cast<const uint4*>(src + ((size_t)(n * h + ys) * w + xs) * C + c0 + uu * EPU);
            if (bn) v = bnrelu_unit<T>(v, bn_row(bn, 0, 2, C) + c0 + uu * EPU, bn_row(bn, 0, 3, C) + c0 + uu * EPU);
        }
        *reinterpret_cast<uint4*>(sm + pix * PSTR + uu * 16) = v;
    }
    __syncthreads();
    const int uu = tid % UB, op = tid / UB;                     // 64 pixel lanes x 4 units; 4 outputs per thread
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
        const int ly = (op >> 4) + 4 * rep, lx = op & 15;
        const int Y = Y0 + ly, X = X0 + lx;
        if (Y >= H || X >= W) continue;
        const int yy = Y - top, xx = X - left;
        float o[EPU];
#pragma unroll
        for (int i = 0; i < EPU; i++) o[i] = 0.f;
        if (yy >= 0 && yy < 2 * h && xx >= 0 && xx < 2 * w) {
            int y0, y1, x0, x1; float ly_, lx_;
            up_tap(yy, h, sy, y0, y1, ly_); up_tap(xx, w, sx, x0, x1, lx_);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int ys = ((k >> 1) ? y1 : y0) - ys0, xs = ((k & 1) ? x1 : x0) - xs0;
                const float wgt = ((k >> 1) ? ly_ : 1.f - ly_) * ((k & 1) ? lx_ : 1.f - lx_);
                float f[EPU];
                Unit<T>::unpack(*reinterpret_cast<const uint4*>(sm + (ys * RS + xs) * PSTR + uu * 16), f);
#pragma unroll
                for (int i = 0; i < EPU; i++) o[i] += wgt * f[i];
            }
        }
        { const size_t pp = (size_t)(n * H + Y) * W + X; store_out<T>(out, pp * C + c0 + uu * EPU, so, pp, c0 + uu * EPU, o); }
    }
}

static int upsample2x_impl(int dtype, const void* src, int in_mode, const float* bn,
                           void* out, SplitOut so, int B, int h, int w, int H, int W, int C, void* stream) {
    if (in_mode == BDN_IN_BNRELU && !bn) BDN_FAIL(BDN_E_ARG, "upsample2x: BNRELU needs bn");
    if (H < 2 * h || W < 2 * w || C % 16 || C > 1024 || 1024 % C) BDN_FAIL(BDN_E_SHAPE, "upsample2x: bad shape");
    const float* b = in_mode == BDN_IN_BNRELU ? bn : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const int npix = B * H * W;
    const float sy = up_scale(h), sx = up_scale(w);
    const int epu = dtype == BDN_BF16 ? 8 : 4;
    if (h >= 8 && w >= 8 && C % (4 * epu) == 0 && (dtype == BDN_BF16 || dtype == BDN_F32)) {
        const int tx = (W + 15) / 16, ty = (H + 15) / 16;
        const dim3 grid(tx * ty * B, C / (4 * epu));
        if (dtype == BDN_BF16) hipLaunchKernelGGL(upsample2x_tiled_kernel<bf16s>, grid, dim3(256), 0, st, (const bf16s*)src, b, (bf16s*)out, h, w, H, W, C, tx, ty, sy, sx, so);
        else hipLaunchKernelGGL(upsample2x_tiled_kernel<float>, grid, dim3(256), 0, st, (const float*)src, b, (float*)out, h, w, H, W, C, tx, ty, sy, sx, so);
        BDN_CHECK_LAUNCH("upsample2x_tiled");
        return BDN_OK;
    }
    if (dtype == BDN_BF16) { const int per = 256 / (C / 8) * ITERS;
        hipLaunchKernelGGL(upsample2x_kernel<bf16s>, dim3((npix + per - 1) / per), dim3(256), 0, st, (const bf16s*)src, b, (bf16s*)out, npix, h, w, H, W, C, sy, sx, so); }
    else if (dtype == BDN_F32) { const int per = 256 / (C / 4) * ITERS;
        hipLaunchKernelGGL(upsample2x_kernel<float>, dim3((npix + per - 1) / per), dim3(256), 0, st, (const float*)src, b, (float*)out, npix, h, w, H, W, C, sy, sx, so); }
    else BDN_FAIL(BDN_E_ARG, "upsample2x: bad dtype");
    BDN_CHECK_LAUNCH("upsample2x");
    return BDN_OK;
}

extern "C" int bdn_upsample2x(int dtype, const void* src, int in_mode, const float* bn,
                              void* out, int B, int h, int w, int H, int W, int C, void* stream) {
    if (!src || !out) BDN_FAIL(BDN_E_ARG, "upsample2x: null pointer");
    const SplitOut none = {nullptr, 0, 0, 0};
    return upsample2x_impl(dtype, src, in_mode, bn, out, none, B, h, w, H, W, C, stream);
}

// bf16x3 setting: the upsampled map leaves as channels [off, off + C) of the decoder stage's [hi | lo] two-source operand
// out_split [B,H,W,ld] bf16 (lo half at `half`) instead of a float32 tensor that bdn_split_pack would read again.  src float32.
extern "C" int bdn_upsample2x_split(const void* src, int in_mode, const float* bn, void* out_split, int ld, int off, int half,
                                    int B, int h, int w, int H, int W, int C, void* stream) {
    if (!src || !out_split) BDN_FAIL(BDN_E_ARG, "upsample2x_split: null pointer");
    if (off < 0 || off % 8 || half < off + C || ld < half + off + C || ld % 8 || half % 8)
        BDN_FAIL(BDN_E_SHAPE, "upsample2x_split: bad operand layout ld=%d off=%d half=%d", ld, off, half);
    const SplitOut so = {(bf16s*)out_split, ld, off, half};
    return upsample2x_impl(BDN_F32, src, in_mode, bn, nullptr, so, B, h, w, H, W, C, stream);
}

// transpose: every source pixel gathers from the destination rows / columns that read it
template <typename T>
__global__ void upsample2x_bwd_kernel(const T* __restrict__ dU, int ldU, T* __restrict__ dsrc,
                                      int npix, int h, int w, int H, int W, int C, float sy, float sx) {
    constexpr int EPU = ET<T>::EPU;
    const int CU = C / EPU, rows = 256 / CU, tid = threadIdx.x, cu = tid % CU, row = tid / CU, c = cu * EPU;
    const int top = (H - 2 * h) / 2, left = (W - 2 * w) / 2;
    const int p_end = min(npix, (int)(blockIdx.x + 1) * rows * ITERS);
    for (int p = blockIdx.x * rows * ITERS + row; p < p_end; p += rows) {
        const int x = p % w, t = p / w, y = t % h, n = t / h;
        float o[EPU];
#pragma unroll
        for (int i = 0; i < EPU; i++) o[i] = 0.f;
        // destination d reads sources floor(s)+{0,1} with s = d*(n-1)/(2n-1) in [d/2 - 1/2, d/2]: only
        // d in [2y-2, 2y+3] can touch source y.  Separable weights: six per axis, computed once per pixel.
        float wy[6], wx[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int dy = 2 * y - 2 + k, dx = 2 * x - 2 + k;
            int a0, a1; float l;
            wy[k] = 0.f; wx[k] = 0.f;
            if (dy >= 0 && dy < 2 * h) { up_tap(dy, h, sy, a0, a1, l); wy[k] = (a0 == y ? 1.f - l : 0.f) + (a1 == y ? l : 0.f); }
            if (dx >= 0 && dx < 2 * w) { up_tap(dx, w, sx, a0, a1, l); wx[k] = (a0 == x ? 1.f - l : 0.f) + (a1 == x ? l : 0.f); }
        }
#pragma unroll
        for (int ky = 0; ky < 6; ky++) {
            if (wy[ky] == 0.f) continue;
            const int dy = 2 * y - 2 + ky;
#pragma unroll
            for (int kx = 0; kx < 6; kx++) {
                if (wx[kx] == 0.f) continue;
                const int dx = 2 * x - 2 + kx;
                float f[EPU];
                Unit<T>::unpack(*reinterpret_cast<const uint4*>(dU + ((size_t)(n * H + dy + top) * W + dx + left) * ldU + c), f);
                const float wgt = wy[ky] * wx[kx];
#pragma unroll
                for (int i = 0; i < EPU; i++) o[i] += wgt * f[i];
            }
        }
        *reinterpret_cast<uint4*>(dsrc + (size_t)p * C + c) = Unit<T>::pack(o);
    }
}

// Tiled variant: a block owns 8x8 source pixels x UB channel units.  The 20x20 destination pixels that can reach them
// ([2y-2, 2y+3] per axis, see above) are staged ONCE in LDS (one coalesced pass over dU, 1.56x halo overhead instead of the ~3x
// re-reads of the per-pixel gather), then every thread gathers source pixels from LDS with the same separable weights.
// UB = 8 (round 4; 64 channels = the whole 128-byte channel row of a pixel of the decoder's gradient slices, 50 KB of LDS) where C
// allows it: with UB = 4 a block read 64-byte halves of 128-byte lines (level 1: 99 us for 168 MB).  UB = 4 (32 KB) otherwise.
// Optionally (z_prev != NULL) the block also leaves the BatchNorm-backward partial sums of the layer whose relu(bn(z_prev)) was
// upsampled -- sum g and sum g*z_prev over its 64 source pixels, g = (rounded) dsrc * [scale*z_prev + shift > 0] -- so that the
// three decoder layers behind an upsampling need no reduction pass over dsrc and z either (every other producer of a dA had them).
template <typename T, int UB>
__global__ __launch_bounds__(UB == 8 ? 512 : 256) void upsample2x_bwd_tiled_kernel(const T* __restrict__ dU, int ldU, T* __restrict__ dsrc,
                                                                 int h, int w, int H, int W, int C, int tiles_x, int tiles_y,
                                                                 float sy, float sx, const T* __restrict__ z_prev, const float* __restrict__ bn_prev,
                                                                 float* __restrict__ bs_partial) {
    // pixel stride WITHOUT padding (round 6): 51.2 KB instead of 57.6 KB per block = three blocks per CU instead of two; the kernel is bound by the
    // bytes its blocks keep in flight, not by the two-way bank conflicts of the gather (alone at B = 64: 134.7 -> 112 us, 2.8 -> 3.35 TB/s)
    constexpr int EPU = ET<T>::EPU, TS = 8, R = 2 * TS + 4, PSTR = UB * 16, NT = 64 * UB;      // one thread per (source pixel, unit)
    __shared__ __attribute__((aligned(16))) unsigned char sm[R * R * PSTR];
    const int tid = threadIdx.x;
    // neighbouring tiles share 4 of their 20 window rows / columns: consecutive tiles stay on one XCD so that the overlap is an L2 hit
    // (round-robin over the XCDs every tile fetched its whole window from memory: 1.5x the gradient's bytes in the PMC table)
    const int tile = xcd_remap(blockIdx.x, gridDim.x), tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int c0 = blockIdx.y * UB * EPU;
    const int ys0 = ty * TS, xs0 = tx * TS;
    const int top = (H - 2 * h) / 2, left = (W - 2 * w) / 2;
    // ---- stage the destination window (zero outside the upsampled image): all loads in flight before the first LDS store
    constexpr int NL = (R * R * UB + NT - 1) / NT;
    uint4 stg[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const int i = tid + j * NT;
        const int su = i % UB, pix = i / UB, rx = pix % R, ry = pix / R;
        const int dy = 2 * ys0 - 2 + ry, dx = 2 * xs0 - 2 + rx;
        stg[j] = make_uint4(0, 0, 0, 0);
        if (i < R * R * UB && dy >= 0 && dy < 2 * h && dx >= 0 && dx < 2 * w)
            stg[j] = *reinterpret_cast<const uint4*>(dU + ((size_t)(n * H + dy + top) * W + dx + left) * ldU + c0 + su * EPU);
    }
    const int uu = tid % UB, sp = tid / UB;                      // 64 source pixels x UB units
    const int ly = sp >> 3, lx = sp & 7;
    const int y = ys0 + ly, x = xs0 + lx;
    const bool live = y < h && x < w;
    const bool bs = z_prev != nullptr;
    uint4 zq = make_uint4(0, 0, 0, 0);
    if (bs && live) zq = *reinterpret_cast<const uint4*>(z_prev + ((size_t)(n * h + y) * w + x) * C + c0 + uu * EPU);
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const int i = tid + j * NT;
        if (i < R * R * UB) *reinterpret_cast<uint4*>(sm + (i / UB) * PSTR + (i % UB) * 16) = stg[j];
    }
    __syncthreads();
    float s0[EPU], s1[EPU];
#pragma unroll
    for (int i = 0; i < EPU; i++) { s0[i] = 0.f; s1[i] = 0.f; }
    if (live) {
        float wy[6], wx[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int dy = 2 * y - 2 + k, dx = 2 * x - 2 + k;
            int a0, a1; float l;
            wy[k] = 0.f; wx[k] = 0.f;
            if (dy >= 0 && dy < 2 * h) { up_tap(dy, h, sy, a0, a1, l); wy[k] = (a0 == y ? 1.f - l : 0.f) + (a1 == y ? l : 0.f); }
            if (dx >= 0 && dx < 2 * w) { up_tap(dx, w, sx, a0, a1, l); wx[k] = (a0 == x ? 1.f - l : 0.f) + (a1 == x ? l : 0.f); }
        }
        float o[EPU];
#pragma unroll
        for (int i = 0; i < EPU; i++) o[i] = 0.f;
        const unsigned char* base = sm + ((2 * ly) * R + 2 * lx) * PSTR + uu * 16;
#pragma unroll
        for (int ky = 0; ky < 6; ky++) {
            if (wy[ky] == 0.f) continue;
#pragma unroll
            for (int kx = 0; kx < 6; kx++) {
                if (wx[kx] == 0.f) continue;
                float f[EPU];
                Unit<T>::unpack(*reinterpret_cast<const uint4*>(base + (ky * R + kx) * PSTR), f);
                const float wgt = wy[ky] * wx[kx];
#pragma unroll
                for (int i = 0; i < EPU; i++) o[i] += wgt * f[i];
            }
        }
        uint4 ou = Unit<T>::pack(o);
        if (bs) {                                                  // on the (rounded) gradient, like every other producer -- and what is
            float g[EPU], fz[EPU];                                 // stored is g, MASKED (bidate_hip.h, bdn_conv3x3_dgrad_bs)
            Unit<T>::unpack(ou, g);
            Unit<T>::unpack(zq, fz);
            const float* ps = bn_row(bn_prev, 0, 2, C) + c0 + uu * EPU;
            const float* ph = bn_row(bn_prev, 0, 3, C) + c0 + uu * EPU;
#pragma unroll
            for (int i = 0; i < EPU; i++) {
                const float gg = fmaf(fz[i], ps[i], ph[i]) > 0.f ? g[i] : 0.f;
                s0[i] = gg; s1[i] = gg * fz[i];
                g[i] = gg;
            }
            ou = Unit<T>::pack(g);
        }
        *reinterpret_cast<uint4*>(dsrc + ((size_t)(n * h + y) * w + x) * C + c0 + uu * EPU) = ou;
    }
    if (bs) {
        // 64 source pixels per channel unit: lanes uu, uu + UB, ... of a wave hold the same channels -- butterfly over those lane bits,
        // then the waves meet in LDS (the staged window is dead) in a fixed order
#pragma unroll
        for (int i = 0; i < EPU; i++) {
#pragma unroll
            for (int m = UB; m < 64; m <<= 1) { s0[i] += __shfl_xor(s0[i], m); s1[i] += __shfl_xor(s1[i], m); }
        }
        constexpr int NW = NT / 64;
        __syncthreads();
        float* red = reinterpret_cast<float*>(sm);                 // [NW waves][UB][2 EPU]
        const int lane = tid & 63, wv = tid >> 6;
        if (lane < UB) {
#pragma unroll
            for (int i = 0; i < EPU; i++) { red[(wv * UB + lane) * 2 * EPU + i] = s0[i]; red[(wv * UB + lane) * 2 * EPU + EPU + i] = s1[i]; }
        }
        __syncthreads();
        if (tid < UB * 2 * EPU) {                                  // one thread per (unit, moment, channel)
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < NW; k++) t += red[k * UB * 2 * EPU + tid];
            const int u = tid / (2 * EPU), r = tid % (2 * EPU), mom = r / EPU, ch = r % EPU;
            bs_partial[((size_t)tile * 2 + mom) * C + c0 + u * EPU + ch] = t;
        }
    }
}

static bool ups_bwd_tiled_ok(int dtype, int h, int w, int C) {
    const int epu = dtype == BDN_BF16 ? 8 : 4;
    return h >= 8 && w >= 8 && C % (4 * epu) == 0 && (dtype == BDN_BF16 || dtype == BDN_F32);
}

extern "C" int bdn_upsample2x_bwd_rows(int dtype, int B, int h, int w, int C) {
    if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || !ups_bwd_tiled_ok(dtype, h, w, C)) return 0;
    return B * ((h + 7) / 8) * ((w + 7) / 8);
}

static int upsample2x_bwd_impl(int dtype, const void* dU, int ldU, void* dsrc, const void* z_prev, const float* bn_prev, float* bs_partial,
                               int B, int h, int w, int H, int W, int C, void* stream) {
    if (!dU || !dsrc) BDN_FAIL(BDN_E_ARG, "upsample2x_bwd: null pointer");
    if (H < 2 * h || W < 2 * w || C % 16 || C > 1024 || 1024 % C || ldU < C || ldU % 16) BDN_FAIL(BDN_E_SHAPE, "upsample2x_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int npix = B * h * w;
    const float sy = up_scale(h), sx = up_scale(w);
    const int epu = dtype == BDN_BF16 ? 8 : 4;
    if (ups_bwd_tiled_ok(dtype, h, w, C)) {
        const int tx = (w + 7) / 8, ty = (h + 7) / 8;
        const bool wide = C % (8 * epu) == 0;                      // whole 128-byte (bf16) channel rows per pixel
        const dim3 grid(tx * ty * B, C / ((wide ? 8 : 4) * epu));
#define UPS_BWD_LAUNCH(T_, UB_) hipLaunchKernelGGL((upsample2x_bwd_tiled_kernel<T_, UB_>), grid, dim3(64 * UB_), 0, st, (const T_*)dU, ldU, (T_*)dsrc, h, w, H, W, C, \
                                                   tx, ty, sy, sx, (const T_*)z_prev, bn_prev, bs_partial)
        if (dtype == BDN_BF16) { if (wide) UPS_BWD_LAUNCH(bf16s, 8); else UPS_BWD_LAUNCH(bf16s, 4); }
        else { if (wide) UPS_BWD_LAUNCH(float, 8); else UPS_BWD_LAUNCH(float, 4); }
#undef UPS_BWD_LAUNCH
        BDN_CHECK_LAUNCH("upsample2x_bwd_tiled");
        return BDN_OK;
    }
    if (z_prev) BDN_FAIL(BDN_E_SHAPE, "upsample2x_bwd_bs: shape outside the tiled kernel (bdn_upsample2x_bwd_rows returned 0)");
    if (dtype == BDN_BF16) { const int per = 256 / (C / 8) * ITERS;
        hipLaunchKernelGGL(upsample2x_bwd_kernel<bf16s>, dim3((npix + per - 1) / per), dim3(256), 0, st, (const bf16s*)dU, ldU, (bf16s*)dsrc, npix, h, w, H, W, C, sy, sx); }
    else if (dtype == BDN_F32) { const int per = 256 / (C / 4) * ITERS;
        hipLaunchKernelGGL(upsample2x_bwd_kernel<float>, dim3((npix + per - 1) / per), dim3(256), 0, st, (const float*)dU, ldU, (float*)dsrc, npix, h, w, H, W, C, sy, sx); }
    else BDN_FAIL(BDN_E_ARG, "upsample2x_bwd: bad dtype");
    BDN_CHECK_LAUNCH("upsample2x_bwd");
    return BDN_OK;
}

extern "C" int bdn_upsample2x_bwd(int dtype, const void* dU, int ldU, void* dsrc,
                                  int B, int h, int w, int H, int W, int C, void* stream) {
    return upsample2x_bwd_impl(dtype, dU, ldU, dsrc, nullptr, nullptr, nullptr, B, h, w, H, W, C, stream);
}

extern "C" int bdn_upsample2x_bwd_bs(int dtype, const void* dU, int ldU, void* dsrc, const void* z_prev, const float* bn_prev,
                                     float* bs_partial, int B, int h, int w, int H, int W, int C, void* stream) {
    if (!z_prev || !bn_prev || !bs_partial) BDN_FAIL(BDN_E_ARG, "upsample2x_bwd_bs: null pointer");
    return upsample2x_bwd_impl(dtype, dU, ldU, dsrc, z_prev, bn_prev, bs_partial, B, h, w, H, W, C, stream);
}

// ============================================================ backward of product fusion + max-pool into encoder outputs
// one loop iteration = one 2x2 window x EPU channels x both dates.  Two passes over the window: the first finds,
// per channel and date, WHICH position holds the (first) maximum; the second re-reads z (cache hits), forms the
// gradients and -- when bs_partial is given -- also the BatchNorm-backward partial sums of the layer (sum g,
// sum g*z with g = dA * [relu(bn(z)) > 0], on the STORED, rounded dA), so no separate reduction pass reads dA and z.
// The two-pass variant that never writes dA (sums pass + fused apply pass, 22 % fewer bytes) measured +1.9 % step time in
// round 2 -- its argmax / product work runs twice -- and lives in tools/experimental/enc_skip_two_pass.hip.inc.
template <typename T, int EPU>
__global__ __launch_bounds__(256, 1) void enc_skip_bwd_kernel(const T* __restrict__ dF, int ldF, const T* __restrict__ z, const float* __restrict__ bn,
                                    const T* __restrict__ dP, T* __restrict__ dA, float* __restrict__ bs_partial,
                                    int B, int H, int W, int C, int ncell, int IT, FastDiv dWc, FastDiv dHc) {
    using U = UnitE<T, EPU>;
    using V = typename U::V;
    extern __shared__ float sred[];                            // [256][EPU][4] when bs_partial
    const int CU = C / EPU, rows = 256 / CU, tid = threadIdx.x, cu = tid % CU, row = tid / CU, c = cu * EPU;
    const int Hc = (H + 1) / 2, Wc = (W + 1) / 2, Ho = H / 2, Wo = W / 2;
    float sc0[EPU], sh0[EPU], sc1[EPU], sh1[EPU];
    load_consts_n<EPU>(bn_row(bn, 0, 2, C) + c, sc0); load_consts_n<EPU>(bn_row(bn, 0, 3, C) + c, sh0);
    load_consts_n<EPU>(bn_row(bn, 1, 2, C) + c, sc1); load_consts_n<EPU>(bn_row(bn, 1, 3, C) + c, sh1);
    float t00[EPU], t01[EPU], t10[EPU], t11[EPU];              // [date][sum g | sum g*z]
#pragma unroll
    for (int i = 0; i < EPU; i++) { t00[i] = 0.f; t01[i] = 0.f; t10[i] = 0.f; t11[i] = 0.f; }
    const bool bs = bs_partial != nullptr;
    const int q_end = min(ncell, (int)(blockIdx.x + 1) * rows * IT);
    for (int q = blockIdx.x * rows * IT + row; q < q_end; q += rows) {
        int xc, t, yc, b; dWc.divmod(q, t, xc); dHc.divmod(t, b, yc);
        const bool pooled = dP != nullptr && yc < Ho && xc < Wo;   // floor-mode pooling leaves a trailing odd row/col unpooled
        // every input of the cell is requested up front (z of both dates: 8 units, dF: 4, dP: 2) and z is kept in registers for
        // both passes: the first version re-loaded z in pass 2 behind the wait of pass 1 -- two dependent round trips per cell
        V zq0[4], zq1[4], dfq[4], gq0 = U::zero(), gq1 = U::zero();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int y = 2 * yc + (k >> 1), x = 2 * xc + (k & 1);
            const bool in = y < H && x < W;
            const size_t p0 = ((size_t)(b * H + (in ? y : 2 * yc)) * W + (in ? x : 2 * xc)), p1 = p0 + (size_t)B * H * W;
            zq0[k] = *reinterpret_cast<const V*>(z + p0 * C + c);
            zq1[k] = *reinterpret_cast<const V*>(z + p1 * C + c);
            dfq[k] = *reinterpret_cast<const V*>(dF + p0 * ldF + c);
        }
        if (pooled) {
            gq0 = *reinterpret_cast<const V*>(dP + ((size_t)(b * Ho + yc) * Wo + xc) * C + c);
            gq1 = *reinterpret_cast<const V*>(dP + ((size_t)((B + b) * Ho + yc) * Wo + xc) * C + c);
        }
        // ---- pass 1: position of the FIRST maximum of each window (strict >, like ATen's max_pool2d)
        unsigned idx0 = 0, idx1 = 0;                               // 2 bits per channel
        if (pooled) {
            float m0[EPU], m1[EPU];
#pragma unroll
            for (int i = 0; i < EPU; i++) { m0[i] = -1.f; m1[i] = -1.f; }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float f0[EPU], f1[EPU];                                      // pooled windows are complete
                U::unpack(zq0[k], f0);
                U::unpack(zq1[k], f1);
#pragma unroll
                for (int i = 0; i < EPU; i++) {
                    const float a0 = act1<T>(f0[i], sc0[i], sh0[i]), a1 = act1<T>(f1[i], sc1[i], sh1[i]);
                    if (a0 > m0[i]) { m0[i] = a0; idx0 = (idx0 & ~(3u << (2 * i))) | ((unsigned)k << (2 * i)); }
                    if (a1 > m1[i]) { m1[i] = a1; idx1 = (idx1 & ~(3u << (2 * i))) | ((unsigned)k << (2 * i)); }
                }
            }
        }
        float g0[EPU], g1[EPU];
#pragma unroll
        for (int i = 0; i < EPU; i++) { g0[i] = 0.f; g1[i] = 0.f; }
        if (pooled) { U::unpack(gq0, g0); U::unpack(gq1, g1); }
        // ---- pass 2: gradients (and statistics)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int y = 2 * yc + (k >> 1), x = 2 * xc + (k & 1);
            if (y < H && x < W) {
                float f0[EPU], f1[EPU], df[EPU], o0[EPU], o1[EPU];
                const size_t p0 = ((size_t)(b * H + y) * W + x), p1 = ((size_t)((B + b) * H + y) * W + x);
                U::unpack(zq0[k], f0);
                U::unpack(zq1[k], f1);
                U::unpack(dfq[k], df);
#pragma unroll
                for (int i = 0; i < EPU; i++) {
                    const float a0 = act1<T>(f0[i], sc0[i], sh0[i]), a1 = act1<T>(f1[i], sc1[i], sh1[i]);
                    o0[i] = df[i] * a1;
                    o1[i] = df[i] * a0;
                    if (pooled && ((idx0 >> (2 * i)) & 3u) == (unsigned)k) o0[i] += g0[i];
                    if (pooled && ((idx1 >> (2 * i)) & 3u) == (unsigned)k) o1[i] += g1[i];
                }
                V u0 = U::pack(o0), u1 = U::pack(o1);
                if (bs) {
                    U::unpack(u0, o0); U::unpack(u1, o1);       // what BatchNorm backward will read back (rounded) ...
#pragma unroll
                    for (int i = 0; i < EPU; i++) {
                        const float m0 = fmaf(f0[i], sc0[i], sh0[i]) > 0.f ? o0[i] : 0.f;
                        const float m1 = fmaf(f1[i], sc1[i], sh1[i]) > 0.f ? o1[i] : 0.f;
                        t00[i] += m0; t01[i] = fmaf(m0, f0[i], t01[i]);
                        t10[i] += m1; t11[i] = fmaf(m1, f1[i], t11[i]);
                        o0[i] = m0; o1[i] = m1;
                    }
                    u0 = U::pack(o0); u1 = U::pack(o1);         // ... and MASKED: the stored gradient is g (bidate_hip.h, bdn_conv3x3_dgrad_bs)
                }
                *reinterpret_cast<V*>(dA + p0 * C + c) = u0;
                *reinterpret_cast<V*>(dA + p1 * C + c) = u1;
            }
        }
    }
    if (bs) {
        // bs_partial[date][block][2][C]: rows lanes of one channel unit meet in LDS, fixed order
#pragma unroll
        for (int i = 0; i < EPU; i++) {
            sred[(tid * EPU + i) * 4 + 0] = t00[i]; sred[(tid * EPU + i) * 4 + 1] = t01[i];
            sred[(tid * EPU + i) * 4 + 2] = t10[i]; sred[(tid * EPU + i) * 4 + 3] = t11[i];
        }
        __syncthreads();
        for (int o = tid; o < C * 4; o += 256) {
            const int k = o & 3, cc = o >> 2, ccu = cc / EPU, i = cc % EPU;
            float v = 0.f;
            for (int r = 0; r < rows; r++) v += sred[((r * CU + ccu) * EPU + i) * 4 + k];
            bs_partial[(((size_t)(k >> 1) * gridDim.x + blockIdx.x) * 2 + (k & 1)) * C + cc] = v;
        }
    }
}

// elements per lane: bf16 lanes own 8 BYTES (four channels) -- with 16-byte units the kernel needs 196 registers (two waves per SIMD) and
// streams at 4.0-4.6 TB/s where its float32 instantiation (128 registers) reaches 5.4
static inline int enc_skip_epu(int dtype) { return 4; (void)dtype; }
// cells per lane: four where that still leaves >= 512 blocks, else two, else one (the 16x16 and 8x8 levels ran on 256 and 64 blocks:
// 2.9 and 1.0 TB/s)
static inline int enc_skip_bwd_it(int dtype, int B, int H, int W, int C) {
    const int ncell = B * ((H + 1) / 2) * ((W + 1) / 2);
    const int rows = 256 / (C / enc_skip_epu(dtype));
    for (int it = 4; it > 1; it >>= 1)
        if (ncell / (rows * it) >= 512) return it;
    return 1;
}
static inline int enc_skip_bwd_blocks(int dtype, int B, int H, int W, int C) {
    const int ncell = B * ((H + 1) / 2) * ((W + 1) / 2);
    const int per = 256 / (C / enc_skip_epu(dtype)) * enc_skip_bwd_it(dtype, B, H, W, C);
    return (ncell + per - 1) / per;
}

extern "C" int bdn_enc_skip_bwd_rows(int dtype, int B, int H, int W, int C) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 16 || C > 1024 || 1024 % C) return 0;
    return enc_skip_bwd_blocks(dtype, B, H, W, C);
}

extern "C" int bdn_enc_skip_bwd(int dtype, const void* dF, int ldF, const void* z, const float* bn,
                                const void* dP, void* dA, float* bs_partial, int B, int H, int W, int C, void* stream) {
    if (!dF || !z || !bn || !dA) BDN_FAIL(BDN_E_ARG, "enc_skip_bwd: null pointer");
    if (C % 16 || C > 1024 || 1024 % C || ldF < C || ldF % 16) BDN_FAIL(BDN_E_SHAPE, "enc_skip_bwd: bad shape");
    if (B <= 0 || H <= 0 || W <= 0) BDN_FAIL(BDN_E_SHAPE, "enc_skip_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int ncell = B * ((H + 1) / 2) * ((W + 1) / 2);
    const int grid = enc_skip_bwd_blocks(dtype, B, H, W, C);
    if (dtype == BDN_BF16)
        hipLaunchKernelGGL((enc_skip_bwd_kernel<bf16s, 4>), dim3(grid), dim3(256), bs_partial ? 256 * 4 * 4 * sizeof(float) : 0, st,
                           (const bf16s*)dF, ldF, (const bf16s*)z, bn, (const bf16s*)dP, (bf16s*)dA, bs_partial, B, H, W, C, ncell, enc_skip_bwd_it(dtype, B, H, W, C), FastDiv((W + 1) / 2), FastDiv((H + 1) / 2));
    else if (dtype == BDN_F32)
        hipLaunchKernelGGL((enc_skip_bwd_kernel<float, 4>), dim3(grid), dim3(256), bs_partial ? 256 * 4 * 4 * sizeof(float) : 0, st,
                           (const float*)dF, ldF, (const float*)z, bn, (const float*)dP, (float*)dA, bs_partial, B, H, W, C, ncell, enc_skip_bwd_it(dtype, B, H, W, C), FastDiv((W + 1) / 2), FastDiv((H + 1) / 2));
    else BDN_FAIL(BDN_E_ARG, "enc_skip_bwd: bad dtype");
    BDN_CHECK_LAUNCH("enc_skip_bwd");
    return BDN_OK;
}
