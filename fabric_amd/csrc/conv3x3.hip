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

// experiment switches (defaults = shipped configuration)
#ifndef F_SCHED
#define F_SCHED 1      // pin the W / patch global loads at the top of the tap
#endif
#ifndef F_FRAGDB
#define F_FRAGDB 0     // fragment double buffer inside a tap
#endif
#ifndef F_PREFP
#define F_PREFP 1      // prefetch the next chunk's patch into registers
#endif
#ifndef F_ROWPAD
#define F_ROWPAD 1     // bank-conflict-free patch row pitch
#endif
#ifndef F_LB2
#define F_LB2 1        // __launch_bounds__(256, 2)
#endif

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
    static constexpr int PSTR = CKB + 16;              // LDS pixel stride: odd number of 16-byte slots
    // LDS row pitch: a 16-lane ds_read_b128 group spans two (TW=16) or four (TW=8) patch rows; the row
    // pitch is padded so that those rows land on disjoint bank slots (pitch/16 = 0 resp. 8 mod 16).
    static constexpr int ROWSLOTS = TL::PW * (PSTR / 16);
    static constexpr int RPAD = F_ROWPAD ? (((TW == 16 ? 0 : 8) - ROWSLOTS % 16 + 16) % 16) * 16 : 0;
    static constexpr int ROWP = TL::PW * PSTR + RPAD;
    static constexpr int WSTR = CKB + 16;
    static constexpr int MI = BM / (WM * 32), NJ = BN / (WN * 32);
    static constexpr int KG = CKB / 32;
    static constexpr int PATCH_BYTES = TI * TL::PH * ROWP;
    static constexpr int WBUF_BYTES = BN * WSTR;
    static constexpr int OSTR = BN * ES + 16;
    static constexpr int NWU = (BN * UPP + 255) / 256;
    static constexpr int NPU = (TL::NPIX * UPP + 255) / 256;
    static constexpr int MAIN_BYTES = PATCH_BYTES + 2 * WBUF_BYTES;
    static constexpr int EPI_BYTES = BM * OSTR + WM * BN * 2 * 4;
    static constexpr int SMEM = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tiling");
    __device__ __forceinline__ static int slot_off(int s) {       // LDS byte offset of a slot's (r=0,c=0) tap
        int ti, py, px; TL::slot_to_nyx(s, ti, py, px);
        return (ti * TL::PH + py) * ROWP + px * PSTR;
    }
};

template <typename T, int CKB, int TH, int TW, int TI, int BN, int WM, int WN>
#if F_LB2
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(ConvArgs a) {
#else
__global__ __launch_bounds__(256) void conv3x3_kernel(ConvArgs a) {
#endif
    using CF = ConvCfg<T, CKB, TH, TW, TI, BN, WM, WN>;
    using TL = typename CF::TL;
    constexpr int MI = CF::MI, NJ = CF::NJ, KG = CF::KG, PSTR = CF::PSTR, WSTR = CF::WSTR, ROWP = CF::ROWP;
    constexpr int EPU = CF::EPU, UPP = CF::UPP, NWU = CF::NWU, NPU = CF::NPU, CK = CF::CK;
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
    for (int mi = 0; mi < MI; mi++) a_off[mi] = CF::slot_off((wm * MI + mi) * 32 + l31) + half * 16;
#pragma unroll
    for (int nj = 0; nj < NJ; nj++) b_off[nj] = ((wn * NJ + nj) * 32 + l31) * WSTR + half * 16;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int nj = 0; nj < NJ; nj++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mi][nj][r] = 0.f;

    // ---- activation patch: every thread owns NPU 16-byte units whose pixel / LDS offsets never change
    int p_pix[NPU], p_lds[NPU];                          // global pixel index (-1 = zero padding), LDS byte offset
#pragma unroll
    for (int i = 0; i < NPU; i++) {
        const int u = tid + i * 256;
        const int pix = u / UPP, sub = u % UPP;
        const int xx = pix % TL::PW, t = pix / TL::PW, yy = t % TL::PH, ti = t / TL::PH;
        const int n = n0 + ti, y = y0 + yy - 1, x = x0 + xx - 1;
        const bool ok = u < TL::NPIX * UPP && n < a.N && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
        p_pix[i] = ok ? (n * a.H + y) * a.W + x : -1;
        p_lds[i] = u < TL::NPIX * UPP ? (ti * TL::PH + yy) * ROWP + xx * PSTR + sub * 16 : -1;
    }
    const int p_sub = (tid % UPP) * EPU;                 // channel offset of this thread's units (256 % UPP == 0)
    uint4 preg[NPU];
#define LOAD_PATCH(c0_)                                                                                  \
    {                                                                                                   \
        const T* src_; int cs_, Cs_;                                                                    \
        if ((c0_) < a.C0) { src_ = reinterpret_cast<const T*>(a.in0); Cs_ = a.C0; cs_ = (c0_); }        \
        else { src_ = reinterpret_cast<const T*>(a.in1); Cs_ = a.C1; cs_ = (c0_) - a.C0; }              \
        _Pragma("unroll") for (int i = 0; i < NPU; i++)                                                  \
            if (p_pix[i] >= 0) preg[i] = *reinterpret_cast<const uint4*>(src_ + (size_t)p_pix[i] * Cs_ + cs_ + p_sub); \
    }
#define STORE_PATCH(c0_)                                                                                 \
    {                                                                                                   \
        const bool bn_ = a.in_bn != nullptr && (c0_) < a.C0;                                            \
        float sc_[EPU], sh_[EPU];                          /* all units of a thread share one channel group */ \
        if (bn_) {                                                                                      \
            const float* ps_ = bn_row(a.in_bn, grp, 2, a.C0) + (c0_) + p_sub;                           \
            const float* ph_ = bn_row(a.in_bn, grp, 3, a.C0) + (c0_) + p_sub;                           \
            _Pragma("unroll") for (int e = 0; e < EPU; e++) { sc_[e] = ps_[e]; sh_[e] = ph_[e]; }        \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < NPU; i++)                                                  \
            if (p_lds[i] >= 0) {                                                                        \
                uint4 v_ = make_uint4(0, 0, 0, 0);                                                      \
                if (p_pix[i] >= 0) v_ = bn_ ? bnrelu_unit<T>(preg[i], sc_, sh_) : preg[i];              \
                *reinterpret_cast<uint4*>(patch + p_lds[i]) = v_;                                       \
            }                                                                                           \
    }

    // ---- filter slice of one (tap, chunk): BN rows x CKB bytes, NWU units per thread, global -> regs -> LDS.
    // Plain unrolled code on purpose: a lambda capturing the register array by reference sent it to scratch.
    const T* wbase = reinterpret_cast<const T*>(a.w);
    constexpr bool W_EXACT = (BN * UPP) % 256 == 0;
    static_assert(NWU <= 4, "filter slice: at most four units per thread");
    // named scalars, not an array: an array that meets a compiler memory fence (or a by-reference lambda)
    // is kept in scratch memory
    uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0, w2 = w0, w3 = w0;
    const T* wsrc0; const T* wsrc1; const T* wsrc2; const T* wsrc3;
    int wdst0, wdst1, wdst2, wdst3;
    {
        auto setup = [&](int i, const T*& src, int& dst) {
            const int u = tid + i * 256;
            const int row = (W_EXACT || u < BN * UPP) ? u / UPP : 0, sub = u % UPP;
            src = wbase + (size_t)(col0 + row) * 9 * Cin + sub * EPU;
            dst = row * WSTR + sub * 16;
        };
        setup(0, wsrc0, wdst0); setup(1, wsrc1, wdst1); setup(2, wsrc2, wdst2); setup(3, wsrc3, wdst3);
    }
#define W_ON(i_) ((i_) < NWU && (W_EXACT || tid + (i_) * 256 < BN * UPP))
#define LOAD_W(tap_, c0_)                                                                              \
    {                                                                                                  \
        const size_t o_ = (size_t)(tap_) * Cin + (c0_);                                                 \
        if (W_ON(0)) w0 = *reinterpret_cast<const uint4*>(wsrc0 + o_);                                  \
        if (W_ON(1)) w1 = *reinterpret_cast<const uint4*>(wsrc1 + o_);                                  \
        if (W_ON(2)) w2 = *reinterpret_cast<const uint4*>(wsrc2 + o_);                                  \
        if (W_ON(3)) w3 = *reinterpret_cast<const uint4*>(wsrc3 + o_);                                  \
    }
#define STORE_W(buf_)                                                                                  \
    {                                                                                                  \
        unsigned char* b_ = wbuf + (buf_) * CF::WBUF_BYTES;                                             \
        if (W_ON(0)) *reinterpret_cast<uint4*>(b_ + wdst0) = w0;                                        \
        if (W_ON(1)) *reinterpret_cast<uint4*>(b_ + wdst1) = w1;                                        \
        if (W_ON(2)) *reinterpret_cast<uint4*>(b_ + wdst2) = w2;                                        \
        if (W_ON(3)) *reinterpret_cast<uint4*>(b_ + wdst3) = w3;                                        \
    }

    LOAD_PATCH(0)
    for (int c0 = 0; c0 < Cin; c0 += CK) {
        // LDS is free here: the previous chunk ended with a barrier
        if (!F_PREFP && c0 > 0) { LOAD_PATCH(c0) }
        STORE_PATCH(c0)
        LOAD_W(0, c0)
        STORE_W(0)
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            // issue the global loads FIRST and pin them there (left alone, the scheduler sinks them below the
            // MFMAs, right in front of their LDS store, which exposes the full L2 latency every tap)
            if (tap < 8) { LOAD_W(tap + 1, c0) }
            if (F_PREFP && tap == 5 && c0 + CK < Cin) { LOAD_PATCH(c0 + CK) }   // next chunk's activations: 3 taps of cover
#if F_SCHED == 1
            __builtin_amdgcn_sched_barrier(0);
#elif F_SCHED == 2
            asm volatile("" ::: "memory");      // memory ops may not cross: keeps the global loads above the LDS reads
#endif
            const unsigned char* wb = wbuf + (tap & 1) * CF::WBUF_BYTES;
            const int tapoff = (tap / 3) * ROWP + (tap % 3) * PSTR;
#if F_FRAGDB
            uint4 fa[2][MI], fb[2][NJ];                  // fragment double buffer: reads of k-group g+1 fly under MFMAs of g
#pragma unroll
            for (int mi = 0; mi < MI; mi++) fa[0][mi] = *reinterpret_cast<const uint4*>(patch + a_off[mi] + tapoff);
#pragma unroll
            for (int nj = 0; nj < NJ; nj++) fb[0][nj] = *reinterpret_cast<const uint4*>(wb + b_off[nj]);
#pragma unroll
            for (int kg = 0; kg < KG; kg++) {
                if (kg + 1 < KG) {
#pragma unroll
                    for (int mi = 0; mi < MI; mi++) fa[(kg + 1) & 1][mi] = *reinterpret_cast<const uint4*>(patch + a_off[mi] + tapoff + (kg + 1) * 32);
#pragma unroll
                    for (int nj = 0; nj < NJ; nj++) fb[(kg + 1) & 1][nj] = *reinterpret_cast<const uint4*>(wb + b_off[nj] + (kg + 1) * 32);
                }
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
#pragma unroll
                    for (int nj = 0; nj < NJ; nj++) Mma<T>::run(fa[kg & 1][mi], fb[kg & 1][nj], acc[mi][nj]);
            }
#else
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
#endif
            if (tap < 8) { STORE_W((tap + 1) & 1) }              // other buffer: last read before the previous barrier
            __syncthreads();
        }
    }
#undef LOAD_W
#undef STORE_W
#undef W_ON
#undef LOAD_PATCH
#undef STORE_PATCH

    // ------------------------------------------------------------------ epilogue (LDS reused)
    unsigned char* otile = smem;
    float* red = reinterpret_cast<float*>(smem + CF::BM * CF::OSTR);
    const bool do_stats = a.stats_partial != nullptr;
    const bool full_tile = (n0 + TI <= a.N) && (y0 + TH <= a.H) && (x0 + TW <= a.W);   // block-uniform
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
                bool valid = true;
                if (!full_tile) {
                    int ti, py, px; TL::slot_to_nyx(slot, ti, py, px);
                    valid = (n0 + ti < a.N) && (y0 + py < a.H) && (x0 + px < a.W);
                }
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

// Tile choice (shared with bdn_conv3x3_num_mtiles so the caller can size the statistics buffer):
//   8x8x2 images   for maps up to 8x8;
//   16x16 (BM=256) for 64-wide outputs on larger maps: one filter slice feeds twice the pixels;
//   8x16  (BM=128) otherwise.  BN = 128 when Cout allows it and the grid still has >= 512 blocks.
struct ConvPlan { TileGeom g; int BN; };
static ConvPlan conv_plan(int N, int H, int W, int Cout, int imgs_per_group) {
    ConvPlan p;
    TileGeom& g = p.g;
    const bool narrow = (Cout % 128 != 0);
    if (W <= 8 && H <= 8 && imgs_per_group % 2 == 0) { g.TI = 2; g.TH = 8; g.TW = 8; }
    else if (narrow && H >= 12 && W >= 12) { g.TI = 1; g.TH = 16; g.TW = 16; }
    else { g.TI = 1; g.TH = 8; g.TW = 16; }
    g.tiles_y = (H + g.TH - 1) / g.TH;
    g.tiles_x = (W + g.TW - 1) / g.TW;
    g.n_mtiles = ((N + g.TI - 1) / g.TI) * g.tiles_y * g.tiles_x;
    p.BN = (!narrow && (long)g.n_mtiles * (Cout / 128) >= 512) ? 128 : 64;
    if (g.TH == 16) p.BN = 64;
    return p;
}

template <typename T, int CKB>
static int dispatch_conv(const ConvArgs& a, const ConvPlan& p, hipStream_t st) {
    const TileGeom& g = p.g;
    if (g.TH == 16) return launch_conv<T, CKB, 16, 16, 1, 64, 4, 1>(a, g.n_mtiles, st);
    if (g.TI == 1) {
        if (p.BN == 128) return launch_conv<T, CKB, 8, 16, 1, 128, 2, 2>(a, g.n_mtiles, st);
        return launch_conv<T, CKB, 8, 16, 1, 64, 2, 2>(a, g.n_mtiles, st);
    }
    if (p.BN == 128) return launch_conv<T, CKB, 8, 8, 2, 128, 2, 2>(a, g.n_mtiles, st);
    return launch_conv<T, CKB, 8, 8, 2, 64, 2, 2>(a, g.n_mtiles, st);
}

extern "C" int bdn_conv3x3_num_mtiles(int N, int H, int W, int Cout, int imgs_per_group) {
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || imgs_per_group <= 0) return 0;
    return conv_plan(N, H, W, Cout, imgs_per_group).g.n_mtiles;
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
    const ConvPlan g = conv_plan(N, H, W, Cout, imgs_per_group);
    a.tiles_y = g.g.tiles_y; a.tiles_x = g.g.tiles_x; a.n_ntiles = 0;
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
