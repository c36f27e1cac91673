// 3x3 / stride 1 / pad 1 convolution as an NHWC implicit GEMM on MFMA (gfx950).
// Replaces nn.Conv2d(ci, co, 3, padding=1) forward and its data gradient
// (reference models/unet_parts.py:13,16 and autograd thereof).
//
// GEMM view: M = N*H*W output pixels, N = Cout, K = 9*Cin.
// One 256-thread block (4 waves) owns BM = 128 output pixels (a TI x TH x TW spatial tile)
// x BN output channels.  Per channel chunk the (TH+2)x(TW+2) halo patch is staged ONCE into
// LDS (BatchNorm+ReLU of the producer applied on the way in); the nine taps then read it at
// shifted pixel offsets, so activations cross HBM/L2 -> LDS once instead of nine times.
// The filter slice of one tap x chunk is double-buffered through registers -> LDS.
// Epilogue: + bias, per-tile sum / sum^2 for the following BatchNorm, transpose through
// LDS, 16-byte coalesced NHWC stores.
#include "common.hpp"

struct ConvArgs {
    const void* in0; const void* in1; int C0, C1;
    const float* in_bn;          // [G][4][C0] or null
    int imgs_per_group;
    const void* w;               // [Cout][9][Cin]
    const float* bias;           // [Cout] or null
    void* out;                   // [N,H,W,Cout]
    float* stats_partial;        // [n_mtiles][2][Cout] or null
    int N, H, W, Cout;
    int tiles_y, tiles_x, n_ntiles;
};

template <typename T> struct Mma;
template <> struct Mma<bf16s> {
    __device__ __forceinline__ static void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // 16 bytes per lane-half = 4 k-values per half -> four 32x32x2 steps (k order is a permutation
    // shared by A and B, which leaves the sum unchanged)
    __device__ __forceinline__ static void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

template <typename T, int CKB, int TH, int TW, int TI, int BN, int WM, int WN>
struct ConvCfg {
    using TL = Tile<TH, TW, TI>;
    static constexpr int ES = sizeof(T);
    static constexpr int EPU = 16 / ES;
    static constexpr int CK = CKB / ES;
    static constexpr int UPP = CKB / 16;
    static constexpr int BM = TL::BM;
    static constexpr int PSTR = CKB + 16;
    static constexpr int WSTR = CKB + 16;
    static constexpr int MI = BM / (WM * 32), NJ = BN / (WN * 32);
    static constexpr int KG = CKB / 32;
    static constexpr int PATCH_BYTES = TL::NPIX * PSTR;
    static constexpr int WBUF_BYTES = BN * WSTR;
    static constexpr int OSTR = BN * ES + 16;
    static constexpr int NWU = (BN * UPP + 255) / 256;
    static constexpr int MAIN_BYTES = PATCH_BYTES + 2 * WBUF_BYTES;
    static constexpr int EPI_BYTES = BM * OSTR + WM * BN * 2 * 4;
    static constexpr int SMEM = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tiling");
};

template <typename T, int CKB, int TH, int TW, int TI, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv3x3_kernel(ConvArgs a) {
    using CF = ConvCfg<T, CKB, TH, TW, TI, BN, WM, WN>;
    using TL = typename CF::TL;
    constexpr int MI = CF::MI, NJ = CF::NJ, KG = CF::KG, PSTR = CF::PSTR, WSTR = CF::WSTR;
    constexpr int EPU = CF::EPU, UPP = CF::UPP, NWU = CF::NWU, CK = CF::CK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem;
    unsigned char* wbuf = smem + CF::PATCH_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = logical % a.n_ntiles, mtile = logical / a.n_ntiles;
    const int tx = mtile % a.tiles_x, ty = (mtile / a.tiles_x) % a.tiles_y, ib = mtile / (a.tiles_x * a.tiles_y);
    const int n0 = ib * TI, y0 = ty * TH, x0 = tx * TW, col0 = ntile * BN;
    const int Cin = a.C0 + a.C1;
    const int grp = n0 / a.imgs_per_group;

    // per-lane LDS offsets of the A rows (pixel slots) and B rows (output channels)
    int a_off[MI], b_off[NJ];
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
        a_off[mi] = TL::slot_to_pix((wm * MI + mi) * 32 + l31) * PSTR + half * 16;
#pragma unroll
    for (int nj = 0; nj < NJ; nj++)
        b_off[nj] = ((wn * NJ + nj) * 32 + l31) * WSTR + half * 16;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int nj = 0; nj < NJ; nj++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mi][nj][r] = 0.f;

    const T* wbase = reinterpret_cast<const T*>(a.w);
    // filter slice of one (tap, chunk): BN rows x CKB bytes, NWU 16-byte units per thread, held in registers
    // between the global load (issued before the MFMAs of the previous tap) and the LDS store (after them).
    // Plain unrolled code on purpose: a lambda capturing the register array by reference sent it to scratch.
    constexpr bool W_EXACT = (BN * UPP) % 256 == 0;
    uint4 wreg[NWU];
    const T* wsrc[NWU];
    int wdst[NWU];
#pragma unroll
    for (int i = 0; i < NWU; i++) {
        const int u = tid + i * 256;
        const int row = (W_EXACT || u < BN * UPP) ? u / UPP : 0, sub = u % UPP;
        wsrc[i] = wbase + (size_t)(col0 + row) * 9 * Cin + sub * EPU;
        wdst[i] = row * WSTR + sub * 16;
    }
#define LOAD_W(tap_, c0_)                                                                              \
    _Pragma("unroll") for (int i = 0; i < NWU; i++)                                                     \
        if (W_EXACT || tid + i * 256 < BN * UPP)                                                        \
            wreg[i] = *reinterpret_cast<const uint4*>(wsrc[i] + (size_t)(tap_) * Cin + (c0_));
#define STORE_W(buf_)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NWU; i++)                                                     \
        if (W_EXACT || tid + i * 256 < BN * UPP)                                                        \
            *reinterpret_cast<uint4*>(wbuf + (buf_) * CF::WBUF_BYTES + wdst[i]) = wreg[i];

    for (int c0 = 0; c0 < Cin; c0 += CK) {
        // ---- stage the activation patch of this channel chunk (LDS free: previous chunk ended with a barrier)
        {
            const T* src; int Csrc, cs; const float *sc = nullptr, *sh = nullptr;
            if (c0 < a.C0) {
                src = reinterpret_cast<const T*>(a.in0); Csrc = a.C0; cs = c0;
                if (a.in_bn) { sc = bn_row(a.in_bn, grp, 2, a.C0) + c0; sh = bn_row(a.in_bn, grp, 3, a.C0) + c0; }
            } else {
                src = reinterpret_cast<const T*>(a.in1); Csrc = a.C1; cs = c0 - a.C0;
            }
            stage_patch<T, CKB, PSTR, TH, TW, TI>(patch, src, Csrc, cs, CK, sc, sh, n0, y0, x0, a.N, a.H, a.W, tid);
        }
        LOAD_W(0, c0)
        STORE_W(0)
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            if (tap < 8) { LOAD_W(tap + 1, c0) }                 // global -> regs, in flight during the MFMAs
            const unsigned char* wb = wbuf + (tap & 1) * CF::WBUF_BYTES;
            const int tapoff = ((tap / 3) * TL::PW + (tap % 3)) * PSTR;
#pragma unroll
            for (int kg = 0; kg < KG; kg++) {
                uint4 af[MI], bf[NJ];
#pragma unroll
                for (int mi = 0; mi < MI; mi++) af[mi] = *reinterpret_cast<const uint4*>(patch + a_off[mi] + tapoff + kg * 32);
#pragma unroll
                for (int nj = 0; nj < NJ; nj++) bf[nj] = *reinterpret_cast<const uint4*>(wb + b_off[nj] + kg * 32);
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
#pragma unroll
                    for (int nj = 0; nj < NJ; nj++) Mma<T>::run(af[mi], bf[nj], acc[mi][nj]);
            }
            if (tap < 8) { STORE_W((tap + 1) & 1) }              // other buffer: last read before the previous barrier
            __syncthreads();
        }
    }
#undef LOAD_W
#undef STORE_W

    // ------------------------------------------------------------------ epilogue (LDS reused)
    unsigned char* otile = smem;
    float* red = reinterpret_cast<float*>(smem + CF::BM * CF::OSTR);
    const bool do_stats = a.stats_partial != nullptr;
#pragma unroll
    for (int nj = 0; nj < NJ; nj++) {
        const int col = (wn * NJ + nj) * 32 + l31;
        const float bias = a.bias ? a.bias[col0 + col] : 0.f;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int slot = (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                int ti, py, px; TL::slot_to_nyx(slot, ti, py, px);
                const bool valid = (n0 + ti < a.N) && (y0 + py < a.H) && (x0 + px < a.W);
                const float v = acc[mi][nj][r] + bias;
                if (valid) { s += v; q += v * v; }
                *reinterpret_cast<T*>(otile + slot * CF::OSTR + col * CF::ES) = from_f<T>(v);
            }
        }
        if (do_stats) {
            s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
            if (half == 0) { red[(wm * BN + col) * 2] = s; red[(wm * BN + col) * 2 + 1] = q; }
        }
    }
    __syncthreads();
    if (do_stats && tid < BN) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int m = 0; m < WM; m++) { s += red[(m * BN + tid) * 2]; q += red[(m * BN + tid) * 2 + 1]; }
        a.stats_partial[((size_t)mtile * 2 + 0) * a.Cout + col0 + tid] = s;
        a.stats_partial[((size_t)mtile * 2 + 1) * a.Cout + col0 + tid] = q;
    }
    constexpr int UPR = BN * CF::ES / 16;                    // 16-byte units per output pixel row
    T* outp = reinterpret_cast<T*>(a.out);
    for (int u = tid; u < CF::BM * UPR; u += 256) {
        const int slot = u / UPR, sub = u % UPR;
        int ti, py, px; TL::slot_to_nyx(slot, ti, py, px);
        const int n = n0 + ti, y = y0 + py, x = x0 + px;
        if (n < a.N && y < a.H && x < a.W) {
            uint4 v = *reinterpret_cast<const uint4*>(otile + slot * CF::OSTR + sub * 16);
            *reinterpret_cast<uint4*>(outp + ((size_t)(n * a.H + y) * a.W + x) * a.Cout + col0 + sub * EPU) = v;
        }
    }
}

template <typename T, int CKB, int TH, int TW, int TI, int BN, int WM, int WN>
static int launch_conv(const ConvArgs& a, int n_mtiles, hipStream_t st) {
    using CF = ConvCfg<T, CKB, TH, TW, TI, BN, WM, WN>;
    auto kern = conv3x3_kernel<T, CKB, TH, TW, TI, BN, WM, WN>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
        if (e != hipSuccess) BDN_FAIL(BDN_E_HIP, "conv3x3: hipFuncSetAttribute(%d): %s", CF::SMEM, hipGetErrorString(e));
        attr_set = true;
    }
    ConvArgs b = a;
    b.n_ntiles = a.Cout / BN;
    hipLaunchKernelGGL(kern, dim3(n_mtiles * b.n_ntiles), dim3(256), CF::SMEM, st, b);
    BDN_CHECK_LAUNCH("conv3x3");
    return BDN_OK;
}

template <typename T, int CKB>
static int dispatch_conv(const ConvArgs& a, const TileGeom& g, hipStream_t st) {
    const bool wide = (a.Cout % 128 == 0);
    if (g.TI == 1) {
        if (wide) return launch_conv<T, CKB, 8, 16, 1, 128, 2, 2>(a, g.n_mtiles, st);
        return launch_conv<T, CKB, 8, 16, 1, 64, 2, 2>(a, g.n_mtiles, st);
    }
    if (wide) return launch_conv<T, CKB, 8, 8, 2, 128, 2, 2>(a, g.n_mtiles, st);
    return launch_conv<T, CKB, 8, 8, 2, 64, 2, 2>(a, g.n_mtiles, st);
}

extern "C" int bdn_conv3x3_num_mtiles(int N, int H, int W, int imgs_per_group) {
    if (N <= 0 || H <= 0 || W <= 0 || imgs_per_group <= 0) return 0;
    return pick_tile(N, H, W, imgs_per_group).n_mtiles;
}

extern "C" int bdn_conv3x3(int dtype, const void* in0, int C0, const void* in1, int C1,
                           int in_mode, const float* in_bn, int imgs_per_group,
                           const void* w, const float* bias, void* out, float* stats_partial,
                           int N, int H, int W, int Cout, void* stream) {
    if (!in0 || !w || !out) BDN_FAIL(BDN_E_ARG, "conv3x3: null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || imgs_per_group <= 0 || N % imgs_per_group)
        BDN_FAIL(BDN_E_SHAPE, "conv3x3: bad N=%d H=%d W=%d imgs_per_group=%d", N, H, W, imgs_per_group);
    if (Cout <= 0 || Cout % 64) BDN_FAIL(BDN_E_SHAPE, "conv3x3: Cout=%d must be a multiple of 64", Cout);
    if (in1 == nullptr) C1 = 0;
    if (C1 < 0 || (in1 && C1 == 0)) BDN_FAIL(BDN_E_SHAPE, "conv3x3: bad C1=%d", C1);
    if (in_mode == BDN_IN_BNRELU && !in_bn) BDN_FAIL(BDN_E_ARG, "conv3x3: BNRELU input needs in_bn");
    if (in_mode != BDN_IN_BNRELU && in_mode != BDN_IN_PLAIN) BDN_FAIL(BDN_E_ARG, "conv3x3: bad in_mode %d", in_mode);
    if (in_mode == BDN_IN_BNRELU && in1) BDN_FAIL(BDN_E_ARG, "conv3x3: two-source input must be plain");
    ConvArgs a;
    a.in0 = in0; a.in1 = in1; a.C0 = C0; a.C1 = C1;
    a.in_bn = in_mode == BDN_IN_BNRELU ? in_bn : nullptr;
    a.imgs_per_group = imgs_per_group; a.w = w; a.bias = bias; a.out = out; a.stats_partial = stats_partial;
    a.N = N; a.H = H; a.W = W; a.Cout = Cout;
    TileGeom g = pick_tile(N, H, W, imgs_per_group);
    a.tiles_y = g.tiles_y; a.tiles_x = g.tiles_x; a.n_ntiles = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int Cin = C0 + C1;
    if (dtype == BDN_BF16) {
        // channel chunk: 64 channels (128 B) when both sources allow it, else 16 channels (32 B)
        if (C0 % 64 == 0 && C1 % 64 == 0) return dispatch_conv<bf16s, 128>(a, g, st);
        if (C0 % 16 == 0 && C1 % 16 == 0) return dispatch_conv<bf16s, 32>(a, g, st);
        BDN_FAIL(BDN_E_SHAPE, "conv3x3(bf16): C0=%d C1=%d must be multiples of 16", C0, C1);
    } else if (dtype == BDN_F32) {
        if (C0 % 32 == 0 && C1 % 32 == 0) return dispatch_conv<float, 128>(a, g, st);
        if (C0 % 16 == 0 && C1 % 16 == 0) return dispatch_conv<float, 64>(a, g, st);
        BDN_FAIL(BDN_E_SHAPE, "conv3x3(f32): C0=%d C1=%d must be multiples of 16", C0, C1);
    }
    (void)Cin;
    BDN_FAIL(BDN_E_ARG, "conv3x3: bad dtype %d", dtype);
}
