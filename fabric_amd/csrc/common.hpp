// Shared device/host helpers for libbidate_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <mutex>
#include "../../include/bidate_hip.h"

typedef uint16_t bf16s;                                   // bf16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// ---------------------------------------------------------------- errors
void bdn_set_error(const char* fmt, ...);
#define BDN_FAIL(code, ...) do { bdn_set_error(__VA_ARGS__); return (code); } while (0)
#define BDN_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) BDN_FAIL(BDN_E_HIP, "%s: %s", name, hipGetErrorString(e_)); } while (0)

// Raise a kernel's dynamic-LDS limit exactly once per instantiation (the macro sits inside a function template, so the
// statics are per kernel), safely under concurrent first calls from several threads: std::call_once, the result is kept.
#define BDN_SET_SMEM_ONCE(kern_, bytes_, name_) do {                                                           \
    static std::once_flag once_; static hipError_t err_ = hipSuccess;                                         \
    std::call_once(once_, [&] { err_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern_),              \
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (bytes_)); }); \
    if (err_ != hipSuccess) BDN_FAIL(BDN_E_HIP, "%s: hipFuncSetAttribute(%d): %s", name_, (int)(bytes_), hipGetErrorString(err_)); \
} while (0)

// bf16x3 helpers defined in x3.hip, used by the dtype dispatch of bdn_pack_weights / bdn_conv3x3_wgrad_ex
int bdn_pack_weights_x3(const float* w_oihw, void* wf, void* wd, int Cout, int Cin, int Cin_pad, hipStream_t st);
// bdn_pack_weights_multi's device record: one per layer
struct PackDesc { const float* w; void* wf; void* wd; int Cout, Cin, Cinp, pad_; };
int bdn_pack_weights_x3_multi(const PackDesc* desc, int n_layers, hipStream_t st);
int bdn_wgrad_x3_combine(const float* T, float* dw, int Cout, int Cinp, int Cin_real, int taps, int terms, hipStream_t st);

// ---------------------------------------------------------------- division by a launch-invariant integer
// n / d for 0 <= n < 2^31 as one v_mul_hi_u32 and a shift (the round-up multiplier; exhaustively checked against n / d on the host
// for every d <= 70 000 at the critical n and on 2e6 random pairs).  A runtime integer division is 15-20 VALU instructions, and the
// streaming kernels decode (image, row, column) from a linear pixel index once or three times per pixel.
struct FastDiv {
    unsigned mul, shr; int d;
    FastDiv() = default;
    __host__ __device__ explicit FastDiv(int d_) : d(d_) {
        if (d_ <= 1) { mul = 0; shr = 0; }
        else {
            unsigned l = 0; while ((1u << l) < (unsigned)d_) l++;              // ceil(log2 d)
            const unsigned p = 31 + l;
            mul = (unsigned)((((uint64_t)1 << p) + (unsigned)d_ - 1) / (unsigned)d_); shr = p - 32;
        }
    }
    __device__ __forceinline__ int div(int n) const { return d == 1 ? n : (int)(__umulhi((unsigned)n, mul) >> shr); }
    __device__ __forceinline__ void divmod(int n, int& q, int& r) const { q = div(n); r = n - q * d; }
};

// ---------------------------------------------------------------- element traits
template <typename T> struct ET;
template <> struct ET<float> { static constexpr int EPU = 4; };   // elements per 16-byte unit
template <> struct ET<bf16s> { static constexpr int EPU = 8; };

__device__ __forceinline__ float bf2f(uint32_t h) { return __uint_as_float(h << 16); }
// round to nearest even through the hardware converter (v_cvt_pk_bf16_f32); a bit-twiddling version with a
// NaN branch cost a divergent exec-mask sequence per element in every staging loop
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t f2bf(float f) {
    __bf16 r = (__bf16)f;
    return (uint32_t)__builtin_bit_cast(uint16_t, r);
}
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {  // two values -> one packed dword
    f32x2_t v = {lo, hi};
    bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16s v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16s from_f<bf16s>(float v) { return (bf16s)f2bf(v); }

// 16-byte unit <-> float[EPU]
template <typename T> struct Unit;
template <> struct Unit<float> {
    static constexpr int N = 4;
    __device__ __forceinline__ static void unpack(const uint4& u, float* f) {
        f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y);
        f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
    }
    __device__ __forceinline__ static uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct Unit<bf16s> {
    static constexpr int N = 8;
    __device__ __forceinline__ static void unpack(const uint4& u, float* f) {
        f[0] = bf2f(u.x & 0xffffu); f[1] = bf2f(u.x >> 16);
        f[2] = bf2f(u.y & 0xffffu); f[3] = bf2f(u.y >> 16);
        f[4] = bf2f(u.z & 0xffffu); f[5] = bf2f(u.z >> 16);
        f[6] = bf2f(u.w & 0xffffu); f[7] = bf2f(u.w >> 16);
    }
    __device__ __forceinline__ static uint4 pack(const float* f) {
        return make_uint4(f2bf2(f[0], f[1]), f2bf2(f[2], f[3]), f2bf2(f[4], f[5]), f2bf2(f[6], f[7]));
    }
};

// bf16x3: a float32 kernel's output stored directly as the [hi | lo] bf16 operand of the convolution that consumes it (bdn_split_pack's
// layout, possibly a channel window of a wider two-source operand): [pixel][ld], hi at off + c, lo at half + off + c.  p == nullptr: off.
struct SplitOut { bf16s* p; int ld, off, half; };
__device__ __forceinline__ void store_split4(const SplitOut& so, size_t pix, int c, const float* o_) {
    // the values are pinned first: without it the compiler contracts the product that formed o with the subtraction below into one FMA,
    // and lo would be the residual of the UNROUNDED product (not what splitting the stored float32 tensor gives)
    float o[4] = {o_[0], o_[1], o_[2], o_[3]};
    asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
    const uint32_t h01 = f2bf2(o[0], o[1]), h23 = f2bf2(o[2], o[3]);
    const uint32_t l01 = f2bf2(o[0] - bf2f(h01 & 0xffffu), o[1] - bf2f(h01 >> 16));      // exact differences in float32
    const uint32_t l23 = f2bf2(o[2] - bf2f(h23 & 0xffffu), o[3] - bf2f(h23 >> 16));
    *reinterpret_cast<uint2*>(so.p + pix * so.ld + so.off + c) = make_uint2(h01, h23);
    *reinterpret_cast<uint2*>(so.p + pix * so.ld + so.half + so.off + c) = make_uint2(l01, l23);
}
// store 16 bytes of output: as T at dst, or (float32 only) as the split pair when so.p is set
template <typename T>
__device__ __forceinline__ void store_out(T* dst_base, size_t elem_index, const SplitOut& so, size_t pix, int c, const float* o) {
    if constexpr (sizeof(T) == 4) {
        if (so.p) { store_split4(so, pix, c, o); return; }
    }
    *reinterpret_cast<uint4*>(dst_base + elem_index) = Unit<T>::pack(o);
}

// BatchNorm table layout: bn[g][k][c], k = 0 mean, 1 invstd, 2 scale, 3 shift
__device__ __forceinline__ const float* bn_row(const float* bn, int g, int k, int C) {
    return bn + ((size_t)g * 4 + k) * C;
}

// relu(z*scale+shift) on one 16-byte unit; sc/sh point at the unit's first channel
template <typename T>
__device__ __forceinline__ uint4 bnrelu_unit(const uint4& u, const float* sc, const float* sh) {
    constexpr int N = Unit<T>::N;
    float f[N];
    Unit<T>::unpack(u, f);
#pragma unroll
    for (int i = 0; i < N; i++) f[i] = fmaxf(fmaf(f[i], sc[i], sh[i]), 0.f);
    return Unit<T>::pack(f);
}
// bf16: two channels per dword -- two scalar v_fma_f32 (NOT one v_pk_fma_f32: packed f32 math beside MFMAs costs the matrix
// pipe, MI355X_MICROARCH.md; measured here: the BatchNorm-on-load convolutions +2-4 %, the weight-gradient staging +4 %), one
// v_cvt_pk_bf16_f32, and the ReLU as a signed 16-bit max with 0 on the rounded pair (a negative bf16 is a negative int16;
// rounding keeps the sign, so round(max(x,0)) == max(round(x),0)).
typedef __attribute__((ext_vector_type(2))) short s16x2_t;
__device__ __forceinline__ uint32_t bnrelu_pair(uint32_t u, float s0, float s1, float t0, float t1) {
    f32x2_t y;
    y.x = __builtin_fmaf(__uint_as_float(u << 16), s0, t0);
    y.y = __builtin_fmaf(__uint_as_float(u & 0xffff0000u), s1, t1);
    asm("" : "+v"(y.x)); asm("" : "+v"(y.y));                   // keep the SLP vectoriser from re-packing the two FMAs into v_pk_fma_f32
    const bf16x2_t r = __builtin_convertvector(y, bf16x2_t);
    const s16x2_t zero = {0, 0};
    const s16x2_t q = __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, r), zero);
    return __builtin_bit_cast(uint32_t, q);
}
template <>
__device__ __forceinline__ uint4 bnrelu_unit<bf16s>(const uint4& u, const float* sc, const float* sh) {
    return make_uint4(bnrelu_pair(u.x, sc[0], sc[1], sh[0], sh[1]), bnrelu_pair(u.y, sc[2], sc[3], sh[2], sh[3]),
                      bnrelu_pair(u.z, sc[4], sc[5], sh[4], sh[5]), bnrelu_pair(u.w, sc[6], sc[7], sh[6], sh[7]));
}

template <int N> __device__ __forceinline__ float dpp_row_shl(float v) {     // lane i <- lane i + N of its row of 16 (0 beyond the row)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xf, 0xf, true));
}
template <int CU> __device__ __forceinline__ float first_lane_sum(float v) {
    static_assert(CU == 8 || CU == 16, "lanes per pixel");
    if constexpr (CU == 16) v += dpp_row_shl<8>(v);
    v += dpp_row_shl<4>(v); v += dpp_row_shl<2>(v); v += dpp_row_shl<1>(v);
    return v;
}

// element-wise maximum of two 16-byte units of NON-NEGATIVE values (post-ReLU activations): for bf16 the order of non-negative values is the
// order of their bit patterns as signed 16-bit integers (a -0.0 = 0x8000 is the smallest), so four v_pk_max_i16 do it without unpacking
template <typename T> __device__ __forceinline__ uint4 unit_max_nonneg(const uint4& a, const uint4& b);
template <> __device__ __forceinline__ uint4 unit_max_nonneg<float>(const uint4& a, const uint4& b) {
    return make_uint4(__float_as_uint(fmaxf(__uint_as_float(a.x), __uint_as_float(b.x))), __float_as_uint(fmaxf(__uint_as_float(a.y), __uint_as_float(b.y))),
                      __float_as_uint(fmaxf(__uint_as_float(a.z), __uint_as_float(b.z))), __float_as_uint(fmaxf(__uint_as_float(a.w), __uint_as_float(b.w))));
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, a), __builtin_bit_cast(s16x2_t, b)));
}
template <> __device__ __forceinline__ uint4 unit_max_nonneg<bf16s>(const uint4& a, const uint4& b) {
    return make_uint4(pk_max_i16(a.x, b.x), pk_max_i16(a.y, b.y), pk_max_i16(a.z, b.z), pk_max_i16(a.w, b.w));
}

// Filter image in MFMA fragment order (what conv3x3_kernel streams from L2): one contiguous 1 KB record per
// (32 output channels, tap, k-group of 32 bytes of input channels); inside a record lane l owns 16 bytes:
// output channel 32*cb + (l & 31), input channels kgroup*KCH + (l >> 5)*EPU + [0, EPU).
// Returns the element index of weight (co, tap, c) for a filter with Cin (padded) input channels.
template <typename T>
__host__ __device__ __forceinline__ size_t wfrag_index(int co, int tap, int c, int Cin) {
    constexpr int EPU = 16 / (int)sizeof(T), KCH = 32 / (int)sizeof(T);
    const size_t rec = ((size_t)(co >> 5) * 9 + tap) * (Cin / KCH) + c / KCH;
    const int lane = (co & 31) + 32 * ((c % KCH) / EPU);
    return rec * (64 * EPU) + lane * EPU + c % EPU;
}

// XCD-aware bijective block remap: consecutive logical ids stay on one XCD (private L2)
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
    int q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---------------------------------------------------------------- spatial tile geometry (shared by fwd/dgrad and wgrad)
// A block covers TI images x TH x TW output pixels ("slots"); its input patch has a 1-pixel halo.
template <int TH, int TW, int TI> struct Tile {
    static constexpr int BM = TI * TH * TW;
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int NPIX = TI * PH * PW;
    __device__ __forceinline__ static int slot_to_pix(int s) {           // patch index of the slot's (r=0,c=0) tap
        int ti = s / (TH * TW), rem = s % (TH * TW);
        return (ti * PH + rem / TW) * PW + rem % TW;
    }
    __device__ __forceinline__ static void slot_to_nyx(int s, int& ti, int& py, int& px) {
        ti = s / (TH * TW); int rem = s % (TH * TW); py = rem / TW; px = rem % TW;
    }
};

struct TileGeom { int tiles_y, tiles_x, n_mtiles, TI, TH, TW; };
static inline TileGeom pick_tile(int N, int H, int W, int imgs_per_group) {
    TileGeom g;
    if (W <= 8 && H <= 8 && imgs_per_group % 2 == 0) { g.TI = 2; g.TH = 8; g.TW = 8; }
    else { g.TI = 1; g.TH = 8; g.TW = 16; }
    g.tiles_y = (H + g.TH - 1) / g.TH;
    g.tiles_x = (W + g.TW - 1) / g.TW;
    g.n_mtiles = ((N + g.TI - 1) / g.TI) * g.tiles_y * g.tiles_x;
    return g;
}

// Stage the input patch of one channel chunk into LDS (zero outside the image = conv zero padding).
// CKB = bytes of channels per pixel in the chunk, PSTR = LDS pixel stride in bytes.
// cvalid = number of valid channels in this chunk (others zero-filled); sc/sh: BN scale/shift rows
// already offset to the chunk's first channel, or nullptr for plain input.
template <typename T, int CKB, int PSTR, int TH, int TW, int TI>
__device__ __forceinline__ void stage_patch(unsigned char* patch, const T* src, int Csrc, int cs, int cvalid,
                                            const float* sc, const float* sh,
                                            int n0, int y0, int x0, int N, int H, int W, int tid) {
    using TL = Tile<TH, TW, TI>;
    constexpr int EPU = ET<T>::EPU;
    constexpr int UPP = CKB / 16;
    for (int u = tid; u < TL::NPIX * UPP; u += 256) {
        int pix = u / UPP, sub = u % UPP;
        int xx = pix % TL::PW; int t = pix / TL::PW; int yy = t % TL::PH; int ti = t / TL::PH;
        int n = n0 + ti, y = y0 + yy - 1, x = x0 + xx - 1;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (n < N && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && sub * EPU < cvalid) {
            const T* p = src + ((size_t)(n * H + y) * W + x) * Csrc + cs + sub * EPU;
            v = *reinterpret_cast<const uint4*>(p);
            if (sc) v = bnrelu_unit<T>(v, sc + sub * EPU, sh + sub * EPU);
        }
        *reinterpret_cast<uint4*>(patch + pix * PSTR + sub * 16) = v;
    }
}
