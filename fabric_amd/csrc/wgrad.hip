// Weight gradient of the 3x3 / stride 1 / pad 1 convolution (autograd of reference
// models/unet_parts.py:13,16) as an implicit GEMM whose reduction runs over output PIXELS:
//
//   dW[co][tap][ci] = sum_{n,y,x} dz[n,y,x,co] * a[n, y+r-1, x+c-1, ci]        tap = 3r+c
//
// One 256-thread block owns a 64(co) x 64(ci) tile for ALL nine taps (9 x 32x32 accumulators per
// wave) and a contiguous range of 128-pixel spatial chunks (split over pixels; partial tiles are
// written to a workspace and summed by wgrad_reduce_kernel, which also emits the reference's
// OIHW f32 layout -- deterministic, no atomics).
// Per chunk the dz tile [128 px][64 co] and the activation halo patch [(8+2)x(16+2) px][64 ci]
// are staged into LDS in their natural NHWC order; the MFMA operands need 8 consecutive *pixels*
// per lane, which on gfx950 is exactly what ds_read_b64_tr_b16 delivers from a channel-minor
// image (lane-group semantics pinned by tools/archive/probe_hw.hip): no transposed copy ever exists.
// The f32 variant feeds v_mfma_f32_32x32x2_f32, whose one-value-per-lane operands are plain
// conflict-free ds_read_b32.
#include "common.hpp"

struct WgradArgs {
    const void* dz; int Cout;
    const void* in0; const void* in1; int C0, C1;
    const float* in_bn; int imgs_per_group;
    float* partial;                // [S][9][Cout][Cin]
    int N, H, W;
    int tiles_y, tiles_x, n_mtiles;
    int S, per_split, n_cot, n_cit;
    // 3x3x3 mode (bdn_conv3d_wgrad): the N images are depth slices of N/Dz samples; this launch is depth tap kd = dshift + 1, i.e.
    // dz of slice n pairs with the activation slice n + dshift, and chunks whose partner slice lies outside the sample are skipped
    // (8x16 tiles of one slice only).  Dz = 0: 2-D.
    int Dz, dshift;
    // bf16x3 (the doubled-operand GEMM [dz_hi | dz_lo] x [a_hi | a_lo]): x3h > 0 = number of 64-wide co tiles of the hi half; the
    // lo x lo quadrant (co tile >= x3h and ci tile >= n_cit / 2) contributes 2^-16 of the product and is neither computed nor read
    int x3h, n_tiles;
};

// tile index -> (co tile, ci tile); with x3h the lo x lo quadrant is left out of the enumeration
__device__ __forceinline__ void wg_tile(const WgradArgs& a, int tile, int& cot, int& cit) {
    if (a.x3h == 0) { cot = tile / a.n_cit; cit = tile % a.n_cit; return; }
    const int top = a.x3h * a.n_cit, hc = a.n_cit >> 1;
    if (tile < top) { cot = tile / a.n_cit; cit = tile % a.n_cit; }
    else { const int u = tile - top; cot = a.x3h + u / hc; cit = u % hc; }
}

template <typename T, int TH, int TW, int TI>
struct WgCfg {
    using TL = Tile<TH, TW, TI>;
    static constexpr int ES = sizeof(T);
    static constexpr int CKB = 64 * ES;                 // 64 channels per operand row
    static constexpr int STR = CKB + 64;                // LDS pixel stride: 192 B (bf16) keeps the 4 pixel rows a
                                                        // transposing read touches on disjoint banks
    static constexpr int PATCH_BYTES = TL::NPIX * STR;
    static constexpr int DZ_BYTES = TL::BM * STR;
    static constexpr int SMEM = PATCH_BYTES + DZ_BYTES;
};

__device__ __forceinline__ uint4 tr_pair(const unsigned char* p0, const unsigned char* p1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// KSPLIT: for inputs with <= 32 channels the second half of the 64-wide ci tile is empty; the two waves
// that would own it take every other 16-pixel k-step instead and write their own partial slice.
// f32: nine 32x32 accumulators + 20 staging units of 16 bytes do not fit 256 registers (33-51 spilled in round 1), so the f32
// instantiations are compiled for one block per CU -- and so is KSPLIT (the 3-band configuration's first layer; 5 registers short of
// two blocks per CU, and a spill's scratch traffic queues behind the prefetch loads)
template <typename T, int TH, int TW, int TI, bool KSPLIT>
__global__ __launch_bounds__(256, (sizeof(T) == 4 || KSPLIT) ? 1 : 2) void wgrad_kernel(WgradArgs a) {
    using CF = WgCfg<T, TH, TW, TI>;
    using TL = typename CF::TL;
    constexpr int STR = CF::STR, EPU = ET<T>::EPU, UPP = CF::CKB / 16;
    constexpr int NPU = (TL::NPIX * UPP + 255) / 256, NDU = TL::BM * UPP / 256;
    static_assert((TL::BM * UPP) % 256 == 0 && 256 % UPP == 0, "unit ownership");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem;
    unsigned char* dzt = smem + CF::PATCH_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = KSPLIT ? 0 : (wave & 1);   // wave tile: co [wm*32,+32) x ci [wn*32,+32)
    const int kpar = KSPLIT ? (wave & 1) : 0;
    const int half = lane >> 5, l31 = lane & 31;

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = a.n_tiles;
    const int tile = logical % ntile, split = logical / ntile;
    int cot_, cit_;
    wg_tile(a, tile, cot_, cit_);
    const int co0 = cot_ * 64, ci0 = cit_ * 64;
    const int Cin = a.C0 + a.C1;

    const T* src; int Csrc, cs; bool use_bn = false;
    if (ci0 < a.C0) { src = reinterpret_cast<const T*>(a.in0); Csrc = a.C0; cs = ci0; use_bn = a.in_bn != nullptr; }
    else { src = reinterpret_cast<const T*>(a.in1); Csrc = a.C1; cs = ci0 - a.C0; }
    const int cvalid = min(64, Csrc - cs);
    const T* dzp = reinterpret_cast<const T*>(a.dz);
    const int sub_e = (tid % UPP) * EPU;                      // channel offset of this thread's units
    const bool sub_ok = sub_e < cvalid;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    // per-thread staging registers: next chunk's activation-patch units and dz units travel global -> regs while
    // the current chunk is in the MFMAs, regs -> LDS after the barrier
    uint4 preg[NPU], dreg[NDU];
    unsigned p_ok = 0;                                       // bit i: patch unit i lies inside the image
    int grp_next = 0;
#define LOAD_CHUNK(q_)                                                                                   \
    {                                                                                                   \
        const int tx_ = (q_) % a.tiles_x, ty_ = ((q_) / a.tiles_x) % a.tiles_y, ib_ = (q_) / (a.tiles_x * a.tiles_y); \
        const int n0_ = ib_ * TI, y0_ = ty_ * TH, x0_ = tx_ * TW;                                       \
        const bool dlive_ = a.Dz == 0 || (unsigned)(n0_ % a.Dz + a.dshift) < (unsigned)a.Dz;           \
        grp_next = n0_ / a.imgs_per_group; p_ok = 0;                                                    \
        _Pragma("unroll") for (int i = 0; i < NPU; i++) {                                                \
            const int u = tid + i * 256, pix = u / UPP;                                                  \
            const int xx = pix % TL::PW, t_ = pix / TL::PW, yy = t_ % TL::PH, ti = t_ / TL::PH;          \
            const int n = n0_ + ti, y = y0_ + yy - 1, x = x0_ + xx - 1;                                  \
            const bool ok_ = dlive_ && u < TL::NPIX * UPP && sub_ok && n < a.N && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W; \
            p_ok |= (ok_ ? 1u : 0u) << i;                                                                \
            if (ok_) preg[i] = *reinterpret_cast<const uint4*>(src + ((size_t)((n + a.dshift) * a.H + y) * a.W + x) * Csrc + cs + sub_e); \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < NDU; i++) {                                                \
            const int slot = (tid + i * 256) / UPP;                                                      \
            int ti, py, px; TL::slot_to_nyx(slot, ti, py, px);                                           \
            const int n = n0_ + ti, y = y0_ + py, x = x0_ + px;                                          \
            dreg[i] = make_uint4(0, 0, 0, 0);                                                            \
            if (n < a.N && y < a.H && x < a.W)                                                           \
                dreg[i] = *reinterpret_cast<const uint4*>(dzp + ((size_t)(n * a.H + y) * a.W + x) * a.Cout + co0 + sub_e); \
        }                                                                                               \
    }
#define STORE_CHUNK()                                                                                    \
    {                                                                                                   \
        const float* sc_ = use_bn ? bn_row(a.in_bn, grp_next, 2, a.C0) + cs + sub_e : nullptr;          \
        const float* sh_ = use_bn ? bn_row(a.in_bn, grp_next, 3, a.C0) + cs + sub_e : nullptr;          \
        _Pragma("unroll") for (int i = 0; i < NPU; i++) {                                                \
            const int u = tid + i * 256;                                                                 \
            if (u < TL::NPIX * UPP) {                                                                    \
                uint4 v_ = make_uint4(0, 0, 0, 0);                                                       \
                if ((p_ok >> i) & 1u) v_ = use_bn ? bnrelu_unit<T>(preg[i], sc_, sh_) : preg[i];                  \
                *reinterpret_cast<uint4*>(patch + (u / UPP) * STR + (u % UPP) * 16) = v_;                \
            }                                                                                           \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < NDU; i++) {                                                \
            const int u = tid + i * 256;                                                                 \
            *reinterpret_cast<uint4*>(dzt + (u / UPP) * STR + (u % UPP) * 16) = dreg[i];                 \
        }                                                                                               \
    }

    const int q_begin = split * a.per_split;
    const int q_end = min(a.n_mtiles, q_begin + a.per_split);
    if (q_begin < q_end) LOAD_CHUNK(q_begin)
    for (int q = q_begin; q < q_end; q++) {
        __syncthreads();                                   // previous chunk's LDS reads are done
        STORE_CHUNK()
        __syncthreads();
        if (q + 1 < q_end) LOAD_CHUNK(q + 1)
        __builtin_amdgcn_sched_barrier(0);          // keep the prefetch loads above the MFMAs (the scheduler sinks them otherwise)

        if constexpr (sizeof(T) == 2) {
            // lane's transposing-read role: pixel (lane&15)>>2 of a 4-pixel group, 4-channel piece (lane&3)
            // of the 16-channel block ((lane>>4)&1) of this wave's 32 channels
            const int chan_b = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
            const int kpix = (lane & 15) >> 2;
#define WG_KSTEP()                                                                                       \
            {                                                                                           \
                const int s0 = ks * 16 + half * 8 + kpix;            /* k = 8*half + [0,4) ; +4 for the second read */ \
                const uint4 af = tr_pair(dzt + s0 * STR + wm * 64 + chan_b, dzt + (s0 + 4) * STR + wm * 64 + chan_b); \
                const unsigned char* pb0 = patch + TL::slot_to_pix(s0) * STR + wn * 64 + chan_b;        \
                const unsigned char* pb1 = patch + TL::slot_to_pix(s0 + 4) * STR + wn * 64 + chan_b;    \
                _Pragma("unroll") for (int tap = 0; tap < 9; tap++) {                                    \
                    const int tapoff = ((tap / 3) * TL::PW + (tap % 3)) * STR;                           \
                    const uint4 bfr = tr_pair(pb0 + tapoff, pb1 + tapoff);                              \
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af),  \
                                                                       __builtin_bit_cast(bf16x8, bfr), acc[tap], 0, 0, 0); \
                }                                                                                       \
            }
            if constexpr (KSPLIT) {
                // <= 32 input channels (the 3-band configuration): two k-steps in flight cost this instantiation 5 spilled registers
#pragma unroll 1
                for (int ks = kpar; ks < TL::BM / 16; ks += 2) WG_KSTEP()
            } else {
#pragma unroll 2
                for (int ks = 0; ks < TL::BM / 16; ks++) WG_KSTEP()
            }
#undef WG_KSTEP
        } else {
#pragma unroll 2
            for (int ks = kpar; ks < TL::BM / 2; ks += KSPLIT ? 2 : 1) {
                const int s = ks * 2 + half;
                const float av = *reinterpret_cast<const float*>(dzt + s * STR + (wm * 32 + l31) * 4);
                const unsigned char* pb = patch + TL::slot_to_pix(s) * STR + (wn * 32 + l31) * 4;
#pragma unroll
                for (int tap = 0; tap < 9; tap++) {
                    const int tapoff = ((tap / 3) * TL::PW + (tap % 3)) * STR;
                    const float bv = *reinterpret_cast<const float*>(pb + tapoff);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
                }
            }
        }
    }
#undef LOAD_CHUNK
#undef STORE_CHUNK

    // partial[psplit][tap][co][ci]
    const int ci = ci0 + wn * 32 + l31;
    const int psplit = KSPLIT ? split * 2 + kpar : split;
    if (ci < Cin) {
#pragma unroll
        for (int tap = 0; tap < 9; tap++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                a.partial[(((size_t)psplit * 9 + tap) * a.Cout + co) * Cin + ci] = acc[tap][r];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// LDS image of the pipelined weight-gradient kernel, per chunk buffer: the halo patch (10 x 18 pixels, padded to 192) and the
// dz tile (128 pixels), 128 bytes (64 channels) per pixel, no padding: the two 64-byte halves of a pixel are swapped when bit 1
// of the pixel index is set, so the four consecutive pixels a transposing read touches land on the four bank quarters.
struct Wg6 {
    static constexpr int PW = 18, PH = 10, STR = 128;
    static constexpr int PATCH_BYTES = 6 * 32 * STR;             // 180 patch pixels padded to 192 (24 DMA pieces of 8 pixels)
    static constexpr int DZ_BYTES = 128 * STR;                   // 16 DMA pieces
    static constexpr int BUF = PATCH_BYTES + DZ_BYTES;
    static constexpr unsigned NUM_RECORDS = 0x40000000u, OOB = 0x80000000u;
};

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
// one LDS-DMA piece: every lane's 16 bytes at (descriptor base + voff) -> LDS byte (lds_dst + 16 lane); a lane whose offset lies
// beyond the descriptor's num_records reads 0, and the DMA writes that 0 (tools/archive/probe_dma.hip) -- zero padding, ragged tiles and
// chunks past a split's end cost no instruction.  Inline asm on purpose: hipcc's wait-count pass drains the VM queue (vmcnt(0))
// before the next LDS access behind an LDS-DMA it can see, which serialises the pipeline; this one is invisible to it and is
// waited for by hand.  M0 (the DMA's LDS base) is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void lds_dma16(u32x4_t rsrc, unsigned lds_dst, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ u32x4_t raw_rsrc(const void* base, unsigned num_records) {
    const unsigned long long p_ = reinterpret_cast<unsigned long long>(base);
    u32x4_t r = {(unsigned)p_, (unsigned)(p_ >> 32) & 0xffffu, num_records, 0x00020000u};   // stride 0, raw 32-bit dwords
    return r;
}

// ---------------------------------------------------------------------------------------------------------
// wgrad7 (BDN_WG_ROLE): the bf16 / 8x16-tile / full 64-channel case (16 of the 18 layers of BiDateNet).  Block tile 64 (co) x 64 (ci)
// x all nine taps (9 accumulators per MFMA wave), a contiguous range of 128-pixel chunks per block, three LDS chunk buffers (120 KB),
// ONE barrier per chunk.  The MFMAs walk the patch ROW by ROW: the three fragments of patch row pr feed tile rows pr, pr-1, pr-2 (taps
// r = 0, 1, 2), so every patch fragment is read from LDS once (30 fragment reads per chunk instead of 72) and the reads of row pr+1
// are issued ahead of the MFMAs of row pr (ds_read_b64_tr_b16: lane-group semantics pinned by tools/archive/probe_hw.hip).
// The block is split by ROLE -- made for the half-chip grid, where a weight-gradient block owns its CU anyway (the retired four-wave
// kernels wgrad2 / wgrad6, tools/experimental/wgrad_v2_v6.hip.inc, compute the same values bit for bit):
//   waves 0-3  CONSUMERS, one per SIMD: 76 transposing fragment reads + 72 MFMAs per chunk and nothing else;
//   waves 4-7  PRODUCERS, one per SIMD beside a consumer: everything that stalls a lone MFMA wave in wgrad2 -- the global
//              loads of the halo patch (two chunks ahead, through two register sets), BatchNorm+ReLU of the producing layer,
//              zero masks, ds_write_b128 -- and the LDS-DMA of the dz tile (plain, two chunks ahead: three LDS buffers).
// USE_BN = false: the patch goes by LDS-DMA too and a producer's chunk is ten DMA instructions.
// Every vector-memory instruction of the producer path is inline asm and is waited for by hand: hipcc's wait-count pass cannot
// see the LDS-DMAs, so any wait it derived for a visible load would be off by the DMAs in flight (an over-wait of a whole HBM
// latency per chunk).  Issue order per chunk q: DMA dz(q+2), loads patch(q+3), `s_waitcnt vmcnt(10)` (= patch(q+2) and
// everything older, i.e. dz(q+1) too, has landed), BatchNorm + stores of patch(q+2), `lgkmcnt(0)`, barrier.
__device__ __forceinline__ u32x4_t gload16_asm(const void* sbase, unsigned voff) {
    u32x4_t r;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(r) : "v"(voff), "s"(sbase) : "memory");
    return r;
}
template <bool USE_BN>
__global__ __launch_bounds__(512, 1) void wgrad7_kernel(WgradArgs a) {
    constexpr int PW = Wg6::PW, STR = Wg6::STR, BUF = Wg6::BUF, PATCH_BYTES = Wg6::PATCH_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = a.n_tiles;
    const int tile = logical % ntile, split = logical / ntile;
    int cot_, cit_;
    wg_tile(a, tile, cot_, cit_);
    const int co0 = cot_ * 64, ci0 = cit_ * 64;
    const int Cin = a.C0 + a.C1;
    const int q_begin = split * a.per_split;
    const int q_end = min(a.n_mtiles, q_begin + a.per_split);
    if (q_begin >= q_end) return;                              // (never: the plan leaves no empty split) -- uniform for the block

    if (wave >= 4) {
        // ================================================= producer
        const int pw = wave - 4;
        const unsigned char* src; int Csrc, cs;
        if (ci0 < a.C0) { src = reinterpret_cast<const unsigned char*>(a.in0); Csrc = a.C0; cs = ci0; }
        else { src = reinterpret_cast<const unsigned char*>(a.in1); Csrc = a.C1; cs = ci0 - a.C0; }
        const unsigned char* dzp = reinterpret_cast<const unsigned char*>(a.dz);
        // ownership: piece i = LDS pixels 8 pw + 32 i .. +7; lane = (pixel u_pix of the piece, 16-byte slot sub)
        const int u_pix = pw * 8 + (lane >> 3), sub = lane & 7;
        const int swz = ((u_pix >> 1) & 1) << 2;                // pieces keep bit 1 of the pixel index
        const int unit = sub ^ swz;                             // DMA: lane-linear LDS slot `sub` holds source channel unit `unit`
        const unsigned wbase = u_pix * STR + unit * 16;         // register path: the lane loads channel unit `sub` and stores it at slot sub ^ swz
        int pyx[6];
        unsigned poff[6], poff_dma[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int pix = u_pix + 32 * i, yy = pix / PW, xx = pix % PW;
            pyx[i] = pix < Wg6::PH * PW ? (((yy - 1) << 16) | ((xx - 1) & 0xffff)) : (-4096 << 16);   // never inside
            poff[i] = (unsigned)(((yy * a.W + xx) * Csrc + cs + sub * 8) * 2);
            poff_dma[i] = (unsigned)(((yy * a.W + xx) * Csrc + cs + unit * 8) * 2);
        }
        const unsigned poff_c = (unsigned)(((a.W + 1) * Csrc + cs + sub * 8) * 2);    // the tile's origin pixel: always inside
        const int dpx = u_pix & 15, dpy0 = u_pix >> 4;             // dz pieces: tile pixel (dpy0 + 2 i, dpx)
        const unsigned drow2 = (unsigned)(2 * a.W * a.Cout * 2);
        const unsigned doff0 = (unsigned)(((dpy0 * a.W + dpx) * a.Cout + co0 + unit * 8) * 2);
        const unsigned lds_piece0 = (unsigned)(pw * 8 * STR);
        const unsigned smem_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);

        u32x4_t pA[6], pB[6];
        unsigned mA = 0, mB = 0;
        int gA = 0, gB = 0, cur_grp = -1;
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; e++) { sc[e] = 1.f; sh[e] = 0.f; }
        // cursor: the next chunk in sequence and its tile coordinates; c_* = the chunk it last evaluated
        int lq = q_begin;
        int ltx = q_begin % a.tiles_x, lty = (q_begin / a.tiles_x) % a.tiles_y, ln = q_begin / (a.tiles_x * a.tiles_y);
        int lg = ln / a.imgs_per_group;
        int ldep = a.Dz ? ln % a.Dz : 0;                           // 3x3x3 mode: depth index of slice ln inside its sample
        const long dslice = (long)a.dshift * a.H * a.W;            // pixels between a dz slice and its partner activation slice
        bool c_live = false; int c_y0 = 0, c_x0 = 0, c_grp = 0; long c_pix = 0;
#define W7_CUR()                                                                                         \
        {                                                                                               \
            const bool inside_ = lq < q_end;                 /* a chunk whose partner slice lies outside the sample is skipped (all zero) */ \
            c_live = inside_ && (a.Dz == 0 || (unsigned)(ldep + a.dshift) < (unsigned)a.Dz);            \
            c_y0 = lty * 8; c_x0 = ltx * 16;                                                            \
            if (c_live) { c_pix = (long)(ln * a.H + c_y0) * a.W + c_x0; c_grp = lg; }                   \
            if (inside_) {                                                                              \
                lq++;                                                                                   \
                if (++ltx == a.tiles_x) { ltx = 0; if (++lty == a.tiles_y) { lty = 0; ln++; if (++ldep == a.Dz) ldep = 0; if (ln - lg * a.imgs_per_group == a.imgs_per_group) lg++; } } \
            }                                                                                           \
        }
        // dz tile of the evaluated chunk -> buffer at byte offset wb_ (4 LDS-DMA pieces per wave)
#define W7_DMA_D(wb_)                                                                                    \
        {                                                                                               \
            const u32x4_t rs_ = raw_rsrc(dzp + c_pix * a.Cout * 2, Wg6::NUM_RECORDS);                   \
            _Pragma("unroll") for (int i = 0; i < 4; i++) {                                              \
                const bool ok_ = c_live && (c_y0 + dpy0 + 2 * i) < a.H && (c_x0 + dpx) < a.W;           \
                lds_dma16(rs_, smem_base + (wb_) + PATCH_BYTES + lds_piece0 + i * 32 * STR, ok_ ? doff0 + (unsigned)i * drow2 : Wg6::OOB); \
            }                                                                                           \
        }
        // plain halo patch of the evaluated chunk -> buffer wb_ (6 LDS-DMA pieces per wave)
#define W7_DMA_P(wb_)                                                                                    \
        {                                                                                               \
            const u32x4_t rs_ = raw_rsrc(src + (c_pix + dslice - a.W - 1) * Csrc * 2, Wg6::NUM_RECORDS); \
            _Pragma("unroll") for (int i = 0; i < 6; i++) {                                              \
                const int y_ = c_y0 + (pyx[i] >> 16), x_ = c_x0 + (short)(pyx[i] & 0xffff);             \
                const bool ok_ = c_live && (unsigned)y_ < (unsigned)a.H && (unsigned)x_ < (unsigned)a.W; \
                lds_dma16(rs_, smem_base + (wb_) + lds_piece0 + i * 32 * STR, ok_ ? poff_dma[i] : Wg6::OOB); \
            }                                                                                           \
        }
        // halo patch of the evaluated chunk -> register set (P, M, G): six global loads per lane
#define W7_LOAD_P(P, M, G)                                                                               \
        {                                                                                               \
            const unsigned char* sp_ = src + (c_pix + dslice - a.W - 1) * Csrc * 2;                     \
            if (c_live) G = c_grp;                                                                      \
            unsigned m_ = 0;                                                                            \
            _Pragma("unroll") for (int i = 0; i < 6; i++) {                                              \
                const int y_ = c_y0 + (pyx[i] >> 16), x_ = c_x0 + (short)(pyx[i] & 0xffff);             \
                const bool ok_ = c_live && (unsigned)y_ < (unsigned)a.H && (unsigned)x_ < (unsigned)a.W; \
                P[i] = gload16_asm(sp_, ok_ ? poff[i] : poff_c);                                        \
                m_ |= (ok_ ? 1u : 0u) << i;                                                             \
            }                                                                                           \
            M = m_;                                                                                     \
        }
        // wait until at most n_ vector-memory operations of this wave are in flight; the set's registers are operands so that
        // no use of them is scheduled above the wait
#define W7_WAIT_P(n_, P) asm volatile("s_waitcnt vmcnt(" #n_ ")" : "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]), "+v"(P[4]), "+v"(P[5]) :: "memory");
#define W7_GROUP(g_)                                                                                     \
        if ((g_) != cur_grp) {                                                                          \
            cur_grp = (g_);                                                                             \
            const float* ps_ = bn_row(a.in_bn, cur_grp, 2, a.C0) + cs + sub * 8;                        \
            const float* ph_ = bn_row(a.in_bn, cur_grp, 3, a.C0) + cs + sub * 8;                        \
            _Pragma("unroll") for (int e = 0; e < 8; e++) { sc[e] = ps_[e]; sh[e] = ph_[e]; }            \
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(sc[3]), "+v"(sc[4]), "+v"(sc[5]), "+v"(sc[6]), "+v"(sc[7]), \
                         "+v"(sh[0]), "+v"(sh[1]), "+v"(sh[2]), "+v"(sh[3]), "+v"(sh[4]), "+v"(sh[5]), "+v"(sh[6]), "+v"(sh[7]) :: "memory"); \
        }
        // BatchNorm+ReLU, zero mask and LDS store of the six units of a landed register set
#define W7_STAGE(P, M, G, wb_)                                                                           \
        {                                                                                               \
            W7_GROUP(G)                                                                                 \
            _Pragma("unroll") for (int i = 0; i < 6; i++) {                                              \
                const u32x4_t r_ = P[i];                                                                \
                const bool ok_ = ((M) >> i) & 1u;                                                       \
                const unsigned b0_ = bnrelu_pair(r_.x, sc[0], sc[1], sh[0], sh[1]), b1_ = bnrelu_pair(r_.y, sc[2], sc[3], sh[2], sh[3]); \
                const unsigned b2_ = bnrelu_pair(r_.z, sc[4], sc[5], sh[4], sh[5]), b3_ = bnrelu_pair(r_.w, sc[6], sc[7], sh[6], sh[7]); \
                u32x4_t v_;                                                                             \
                v_.x = ok_ ? b0_ : 0u; v_.y = ok_ ? b1_ : 0u; v_.z = ok_ ? b2_ : 0u; v_.w = ok_ ? b3_ : 0u; \
                *reinterpret_cast<u32x4_t*>(smem + (wb_) + wbase + i * 32 * STR) = v_;                  \
            }                                                                                           \
        }
#define W7_BARRIER() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }

        if constexpr (USE_BN) {
            // prologue: chunks q_begin / q_begin + 1 into buffers 0 / 1, the patch of chunk q_begin + 2 into set A
            W7_CUR() W7_DMA_D(0) W7_LOAD_P(pA, mA, gA)
            W7_CUR() W7_DMA_D(BUF) W7_LOAD_P(pB, mB, gB)
            W7_WAIT_P(10, pA)
            W7_STAGE(pA, mA, gA, 0)
            W7_CUR() W7_LOAD_P(pA, mA, gA)
            W7_WAIT_P(6, pB)
            W7_STAGE(pB, mB, gB, BUF)
            unsigned nxt = 2 * BUF;                                // buffer filled for chunk q + 2
            for (int q = q_begin; q < q_end; q += 2) {
                W7_BARRIER()
                W7_DMA_D(nxt)                                      // dz(q+2): the chunk set A holds
                W7_CUR() W7_LOAD_P(pB, mB, gB)                     // patch(q+3)
                W7_WAIT_P(10, pA)
                W7_STAGE(pA, mA, gA, nxt)
                nxt = nxt == 2 * BUF ? 0 : nxt + BUF;
                if (q + 1 >= q_end) break;
                W7_BARRIER()
                W7_DMA_D(nxt)
                W7_CUR() W7_LOAD_P(pA, mA, gA)
                W7_WAIT_P(10, pB)
                W7_STAGE(pB, mB, gB, nxt)
                nxt = nxt == 2 * BUF ? 0 : nxt + BUF;
            }
        } else {
            W7_CUR() W7_DMA_P(0) W7_DMA_D(0)
            W7_CUR() W7_DMA_P(BUF) W7_DMA_D(BUF)
            unsigned nxt = 2 * BUF;
            for (int q = q_begin; q < q_end; q++) {
                // chunk q has landed when at most chunk q+1's ten pieces are still in flight
                asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                W7_CUR() W7_DMA_P(nxt) W7_DMA_D(nxt)
                nxt = nxt == 2 * BUF ? 0 : nxt + BUF;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // trailing (all-zero) fetches: nothing may land after exit
#undef W7_CUR
#undef W7_DMA_D
#undef W7_DMA_P
#undef W7_LOAD_P
#undef W7_WAIT_P
#undef W7_GROUP
#undef W7_STAGE
#undef W7_BARRIER
        return;
    }

    // ===================================================== consumer: the row walk, fragment reads and MFMAs only
    const int wm = wave >> 1, wn = wave & 1;                  // wave tile: co [wm*32,+32) x ci [wn*32,+32)
    const int half = lane >> 5, l31 = lane & 31;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    const int chan_b = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const int kpix = (lane & 15) >> 2;
    const unsigned a_base = PATCH_BYTES + (half * 8 + kpix) * STR + ((wm ^ ((kpix >> 1) & 1)) * 64) + chan_b;
    const unsigned b_lin = (half * 8 + kpix) * STR + chan_b;
    const unsigned b_base0 = b_lin + ((wn ^ (((kpix + 0) >> 1) & 1)) * 64), b_base1 = b_lin + ((wn ^ (((kpix + 1) >> 1) & 1)) * 64);
    const unsigned b_base2 = b_lin + ((wn ^ (((kpix + 2) >> 1) & 1)) * 64), b_base3 = b_lin + ((wn ^ (((kpix + 3) >> 1) & 1)) * 64);
#define B_BASE(pr_, c_) ((((c_) + 2 * (pr_)) & 3) == 0 ? b_base0 : (((c_) + 2 * (pr_)) & 3) == 1 ? b_base1 : (((c_) + 2 * (pr_)) & 3) == 2 ? b_base2 : b_base3)
    uint4 af[4], bq[2][3];
#define TRP(addr_) __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(addr_)))
#define LDA(dst_, ks_) { const uint2 l_ = TRP(rb + a_base + (ks_) * 16 * STR), h_ = TRP(rb + a_base + ((ks_) * 16 + 4) * STR); dst_ = make_uint4(l_.x, l_.y, h_.x, h_.y); }
#define LDB(dst_, pr_, c_) { const uint2 l_ = TRP(rb + B_BASE(pr_, c_) + ((pr_) * PW + (c_)) * STR), h_ = TRP(rb + B_BASE(pr_, c_) + ((pr_) * PW + (c_) + 4) * STR); dst_ = make_uint4(l_.x, l_.y, h_.x, h_.y); }
#define WG_MMA(t_, ks_, pr_, c_) acc[t_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[(ks_) & 3]), __builtin_bit_cast(bf16x8, bq[(pr_) & 1][c_]), acc[t_], 0, 0, 0);
#define WG_ROW(pr_)                                                                                      \
    {                                                                                                   \
        if ((pr_) + 1 < 10) { LDB(bq[((pr_) + 1) & 1][0], (pr_) + 1, 0) LDB(bq[((pr_) + 1) & 1][1], (pr_) + 1, 1) LDB(bq[((pr_) + 1) & 1][2], (pr_) + 1, 2) } \
        if ((pr_) + 1 < 8) { LDA(af[((pr_) + 1) & 3], (pr_) + 1) }                                      \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        if ((pr_) < 8) { WG_MMA(0, (pr_), (pr_), 0) WG_MMA(1, (pr_), (pr_), 1) WG_MMA(2, (pr_), (pr_), 2) } \
        if ((pr_) >= 1 && (pr_) < 9) { WG_MMA(3, (pr_) - 1, (pr_), 0) WG_MMA(4, (pr_) - 1, (pr_), 1) WG_MMA(5, (pr_) - 1, (pr_), 2) } \
        if ((pr_) >= 2) { WG_MMA(6, (pr_) - 2, (pr_), 0) WG_MMA(7, (pr_) - 2, (pr_), 1) WG_MMA(8, (pr_) - 2, (pr_), 2) } \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    }
    unsigned cur = 0;
    for (int q = q_begin; q < q_end; q++) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned char* rb = smem + cur;
        LDB(bq[0][0], 0, 0) LDB(bq[0][1], 0, 1) LDB(bq[0][2], 0, 2) LDA(af[0], 0)
        WG_ROW(0) WG_ROW(1) WG_ROW(2) WG_ROW(3) WG_ROW(4) WG_ROW(5) WG_ROW(6) WG_ROW(7) WG_ROW(8) WG_ROW(9)
        cur = cur == 2 * BUF ? 0 : cur + BUF;
    }
#undef TRP
#undef LDA
#undef LDB
#undef B_BASE
#undef WG_MMA
#undef WG_ROW
    const int ci = ci0 + wn * 32 + l31;
    if (ci < Cin) {
        const unsigned lane_off = (unsigned)((wm * 32 + 4 * half) * Cin + ci);
        const float* __restrict__ base0 = a.partial + ((size_t)split * 9 * a.Cout + co0) * Cin;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            float* pt = const_cast<float*>(base0) + (size_t)tap * a.Cout * Cin;                 // wave-uniform
#pragma unroll
            for (int r = 0; r < 16; r++)
                pt[(size_t)((r & 3) + 8 * (r >> 2)) * Cin + lane_off] = acc[tap][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// wgrad7x (round 6): the bf16x3 / bf16x2 weight gradient with the terms of the split product summed in ONE accumulator.
// Operands: dz [N,H,W,2 Cout] and in0 [N,H,W,2 Cin] bf16 = hi | lo (bdn_split_pack's layout); a.Cout / a.C0 are the LOGICAL widths.  A block owns
// a 64 (co) x 64 (ci) tile of the LOGICAL weight gradient and, per 128-pixel chunk, stages FOUR slabs -- patch_hi, dz_hi, patch_lo, dz_lo
// (TERMS = 2: no dz_lo) -- by LDS-DMA; every (tile row, tap) then issues
//     acc += dz_hi a_hi;  acc += dz_hi a_lo;  [acc += dz_lo a_hi]
// Against wgrad7 on the doubled operands (three 64 x 64 tiles of [2 Cout] x [2 Cin], one term each): 20 DMA pieces and 76 fragment reads feed 216
// MFMAs instead of 30 / 114, the partial tiles are [Cout][Cin] (a quarter), and the reduction writes dw directly -- no [2 Cout][2 Cin][9] tile, no
// quadrant-sum launch.  LDS: TWO chunk buffers of 80 KB (64 KB for two terms): the DMA of chunk q + 1 is issued behind the barrier that hands
// chunk q to the consumers and has the 216 MFMAs of chunk q (~3.3 us) to land.
template <int TERMS>
__global__ __launch_bounds__(512, 1) void wgrad7x_kernel(WgradArgs a) {
    constexpr int PW = Wg6::PW, STR = Wg6::STR, BUF = Wg6::BUF, PATCH_BYTES = Wg6::PATCH_BYTES;
    constexpr unsigned XBUF = TERMS == 3 ? 2 * BUF : BUF + PATCH_BYTES;          // [patch_hi | dz_hi | patch_lo | dz_lo]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = a.n_tiles;
    const int tile = logical % ntile, split = logical / ntile;
    const int cot_ = tile / a.n_cit, cit_ = tile % a.n_cit;
    const int co0 = cot_ * 64, ci0 = cit_ * 64;
    const int Cin = a.C0;                                      // logical widths; pixel strides are 2 Cin / 2 Cout elements
    const int q_begin = split * a.per_split;
    const int q_end = min(a.n_mtiles, q_begin + a.per_split);
    if (q_begin >= q_end) return;

    if (wave >= 4) {
        // ================================================= producer: 20 (16) LDS-DMA pieces per chunk and wave
        const int pw = wave - 4;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(a.in0);
        const unsigned char* dzp = reinterpret_cast<const unsigned char*>(a.dz);
        const int Ps = 2 * Cin, Ds = 2 * a.Cout;               // pixel strides in elements
        const int u_pix = pw * 8 + (lane >> 3), sub = lane & 7;
        const int swz = ((u_pix >> 1) & 1) << 2;
        const int unit = sub ^ swz;
        int pyx[6];
        unsigned poff_dma[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int pix = u_pix + 32 * i, yy = pix / PW, xx = pix % PW;
            pyx[i] = pix < Wg6::PH * PW ? (((yy - 1) << 16) | ((xx - 1) & 0xffff)) : (-4096 << 16);
            poff_dma[i] = (unsigned)(((yy * a.W + xx) * Ps + ci0 + unit * 8) * 2);
        }
        const unsigned p_lo = (unsigned)(Cin * 2), d_lo = (unsigned)(a.Cout * 2);       // byte offset of the lo half inside a pixel
        const int dpx = u_pix & 15, dpy0 = u_pix >> 4;
        const unsigned drow2 = (unsigned)(2 * a.W * Ds * 2);
        const unsigned doff0 = (unsigned)(((dpy0 * a.W + dpx) * Ds + co0 + unit * 8) * 2);
        const unsigned lds_piece0 = (unsigned)(pw * 8 * STR);
        const unsigned smem_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
        int lq = q_begin;
        int ltx = q_begin % a.tiles_x, lty = (q_begin / a.tiles_x) % a.tiles_y, ln = q_begin / (a.tiles_x * a.tiles_y);
        bool c_live = false; int c_y0 = 0, c_x0 = 0; long c_pix = 0;
#define X7_CUR()                                                                                         \
        {                                                                                               \
            c_live = lq < q_end;                                                                        \
            c_y0 = lty * 8; c_x0 = ltx * 16;                                                            \
            if (c_live) {                                                                               \
                c_pix = (long)(ln * a.H + c_y0) * a.W + c_x0;                                           \
                lq++;                                                                                   \
                if (++ltx == a.tiles_x) { ltx = 0; if (++lty == a.tiles_y) { lty = 0; ln++; } }         \
            }                                                                                           \
        }
#define X7_DMA(wb_)                                                                                      \
        {                                                                                               \
            const u32x4_t rp_ = raw_rsrc(src + (c_pix - a.W - 1) * Ps * 2, Wg6::NUM_RECORDS);           \
            const u32x4_t rd_ = raw_rsrc(dzp + c_pix * Ds * 2, Wg6::NUM_RECORDS);                       \
            _Pragma("unroll") for (int i = 0; i < 6; i++) {                                              \
                const int y_ = c_y0 + (pyx[i] >> 16), x_ = c_x0 + (short)(pyx[i] & 0xffff);             \
                const bool ok_ = c_live && (unsigned)y_ < (unsigned)a.H && (unsigned)x_ < (unsigned)a.W; \
                lds_dma16(rp_, smem_base + (wb_) + lds_piece0 + i * 32 * STR, ok_ ? poff_dma[i] : Wg6::OOB); \
                lds_dma16(rp_, smem_base + (wb_) + BUF + lds_piece0 + i * 32 * STR, ok_ ? poff_dma[i] + p_lo : Wg6::OOB); \
            }                                                                                           \
            _Pragma("unroll") for (int i = 0; i < 4; i++) {                                              \
                const bool ok_ = c_live && (c_y0 + dpy0 + 2 * i) < a.H && (c_x0 + dpx) < a.W;           \
                lds_dma16(rd_, smem_base + (wb_) + PATCH_BYTES + lds_piece0 + i * 32 * STR, ok_ ? doff0 + (unsigned)i * drow2 : Wg6::OOB); \
                if (TERMS == 3)                                                                         \
                    lds_dma16(rd_, smem_base + (wb_) + BUF + PATCH_BYTES + lds_piece0 + i * 32 * STR, ok_ ? doff0 + (unsigned)i * drow2 + d_lo : Wg6::OOB); \
            }                                                                                           \
        }
        X7_CUR() X7_DMA(0)
        unsigned nxt = XBUF;
        for (int q = q_begin; q < q_end; q++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // chunk q has landed (nothing else of this wave is in flight)
            __builtin_amdgcn_s_barrier();                          // ... the consumers take it; they are done with the other buffer
            asm volatile("" ::: "memory");
            X7_CUR() X7_DMA(nxt)                                   // chunk q + 1 (past the end: all-zero pieces)
            nxt = nxt == XBUF ? 0 : XBUF;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef X7_CUR
#undef X7_DMA
        return;
    }

    // ===================================================== consumer
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    const int chan_b = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const int kpix = (lane & 15) >> 2;
    const unsigned a_base = PATCH_BYTES + (half * 8 + kpix) * STR + ((wm ^ ((kpix >> 1) & 1)) * 64) + chan_b;
    const unsigned b_lin = (half * 8 + kpix) * STR + chan_b;
    const unsigned b_base0 = b_lin + ((wn ^ (((kpix + 0) >> 1) & 1)) * 64), b_base1 = b_lin + ((wn ^ (((kpix + 1) >> 1) & 1)) * 64);
    const unsigned b_base2 = b_lin + ((wn ^ (((kpix + 2) >> 1) & 1)) * 64), b_base3 = b_lin + ((wn ^ (((kpix + 3) >> 1) & 1)) * 64);
#define B_BASE(pr_, c_) ((((c_) + 2 * (pr_)) & 3) == 0 ? b_base0 : (((c_) + 2 * (pr_)) & 3) == 1 ? b_base1 : (((c_) + 2 * (pr_)) & 3) == 2 ? b_base2 : b_base3)
    uint4 afh[4], afl[TERMS == 3 ? 4 : 1], bqh[2][3], bql[2][3];
#define TRP(addr_) __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(addr_)))
#define LDA(dst_, base_, ks_) { const uint2 l_ = TRP((base_) + a_base + (ks_) * 16 * STR), h_ = TRP((base_) + a_base + ((ks_) * 16 + 4) * STR); dst_ = make_uint4(l_.x, l_.y, h_.x, h_.y); }
#define LDB(dst_, base_, pr_, c_) { const uint2 l_ = TRP((base_) + B_BASE(pr_, c_) + ((pr_) * PW + (c_)) * STR), h_ = TRP((base_) + B_BASE(pr_, c_) + ((pr_) * PW + (c_) + 4) * STR); dst_ = make_uint4(l_.x, l_.y, h_.x, h_.y); }
#define X_MMA(t_, ks_, pr_, c_)                                                                          \
    {                                                                                                   \
        acc[t_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afh[(ks_) & 3]), __builtin_bit_cast(bf16x8, bqh[(pr_) & 1][c_]), acc[t_], 0, 0, 0); \
        acc[t_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afh[(ks_) & 3]), __builtin_bit_cast(bf16x8, bql[(pr_) & 1][c_]), acc[t_], 0, 0, 0); \
        if (TERMS == 3)                                                                                 \
            acc[t_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afl[TERMS == 3 ? (ks_) & 3 : 0]), __builtin_bit_cast(bf16x8, bqh[(pr_) & 1][c_]), acc[t_], 0, 0, 0); \
    }
#define X_ROW(pr_)                                                                                       \
    {                                                                                                   \
        if ((pr_) + 1 < 10) {                                                                           \
            LDB(bqh[((pr_) + 1) & 1][0], rb, (pr_) + 1, 0) LDB(bqh[((pr_) + 1) & 1][1], rb, (pr_) + 1, 1) LDB(bqh[((pr_) + 1) & 1][2], rb, (pr_) + 1, 2) \
            LDB(bql[((pr_) + 1) & 1][0], rl, (pr_) + 1, 0) LDB(bql[((pr_) + 1) & 1][1], rl, (pr_) + 1, 1) LDB(bql[((pr_) + 1) & 1][2], rl, (pr_) + 1, 2) \
        }                                                                                               \
        if ((pr_) + 1 < 8) { LDA(afh[((pr_) + 1) & 3], rb, (pr_) + 1) if (TERMS == 3) LDA(afl[TERMS == 3 ? ((pr_) + 1) & 3 : 0], rl, (pr_) + 1) } \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        if ((pr_) < 8) { X_MMA(0, (pr_), (pr_), 0) X_MMA(1, (pr_), (pr_), 1) X_MMA(2, (pr_), (pr_), 2) } \
        if ((pr_) >= 1 && (pr_) < 9) { X_MMA(3, (pr_) - 1, (pr_), 0) X_MMA(4, (pr_) - 1, (pr_), 1) X_MMA(5, (pr_) - 1, (pr_), 2) } \
        if ((pr_) >= 2) { X_MMA(6, (pr_) - 2, (pr_), 0) X_MMA(7, (pr_) - 2, (pr_), 1) X_MMA(8, (pr_) - 2, (pr_), 2) } \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    }
    unsigned cur = 0;
    for (int q = q_begin; q < q_end; q++) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned char* rb = smem + cur;
        const unsigned char* rl = rb + BUF;
        LDB(bqh[0][0], rb, 0, 0) LDB(bqh[0][1], rb, 0, 1) LDB(bqh[0][2], rb, 0, 2)
        LDB(bql[0][0], rl, 0, 0) LDB(bql[0][1], rl, 0, 1) LDB(bql[0][2], rl, 0, 2)
        LDA(afh[0], rb, 0)
        if (TERMS == 3) LDA(afl[0], rl, 0)
        X_ROW(0) X_ROW(1) X_ROW(2) X_ROW(3) X_ROW(4) X_ROW(5) X_ROW(6) X_ROW(7) X_ROW(8) X_ROW(9)
        cur = cur == XBUF ? 0 : XBUF;
    }
#undef TRP
#undef LDA
#undef LDB
#undef B_BASE
#undef X_MMA
#undef X_ROW
    const int ci = ci0 + wn * 32 + l31;
    if (ci < Cin) {
        const unsigned lane_off = (unsigned)((wm * 32 + 4 * half) * Cin + ci);
        const float* __restrict__ base0 = a.partial + ((size_t)split * 9 * a.Cout + co0) * Cin;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            float* pt = const_cast<float*>(base0) + (size_t)tap * a.Cout * Cin;
#pragma unroll
            for (int r = 0; r < 16; r++)
                pt[(size_t)((r & 3) + 8 * (r >> 2)) * Cin + lane_off] = acc[tap][r];
        }
    }
}

template <int TERMS>
static int launch_wgrad7x(const WgradArgs& a, hipStream_t st) {
    constexpr int SMEM = 2 * (TERMS == 3 ? 2 * Wg6::BUF : Wg6::BUF + Wg6::PATCH_BYTES);
    auto kern = wgrad7x_kernel<TERMS>;
    BDN_SET_SMEM_ONCE(kern, SMEM, "wgrad7x");
    hipLaunchKernelGGL(kern, dim3(a.S * a.n_tiles), dim3(512), SMEM, st, a);
    BDN_CHECK_LAUNCH("wgrad7x");
    return BDN_OK;
}

template <bool USE_BN>
static int launch_wgrad7(const WgradArgs& a, hipStream_t st) {
    auto kern = wgrad7_kernel<USE_BN>;
    BDN_SET_SMEM_ONCE(kern, 3 * Wg6::BUF, "wgrad7");
    hipLaunchKernelGGL(kern, dim3(a.S * a.n_tiles), dim3(512), 3 * Wg6::BUF, st, a);
    BDN_CHECK_LAUNCH("wgrad7");
    return BDN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// wgrad_first: the FIRST layer's weight gradient with its BatchNorm+ReLU backward fused in.  The first conv has no data
// gradient, so its dz (the largest tensor of the step, 2B x 128 x 128 x 64) has exactly one reader: this GEMM.  Writing it
// with bn_bwd_apply and reading it back is a 0.8 GB round trip at the very end of the step, where nothing of the dz chain
// is left to overlap it.  Here the staging loads dA and z instead, applies bn_bwd_apply's formula (same expression, so the
// bf16 dz values -- and the result -- are the same as the two-kernel path's) and writes the rounded dz straight into the
// LDS operand tile.  Shape class: bf16, Cin_pad = 16, Cout = 64, 8x16 tiles.  With 16 input channels a 32-wide MFMA column
// block holds TWO taps (lanes' columns 0-15 read tap 2j, columns 16-31 tap 2j+1 of the same 16-channel patch row), so a
// k-step is 5 MFMAs instead of 9 and a wave carries 80 accumulator registers.  Two waves per co half split the k-steps of a
// chunk and are added through LDS at the end; two or three blocks share a CU, which is what hides the staging here.
struct WgFirstArgs {
    const bf16s* dA; int ldA; const bf16s* z; const float* bn; const float* sums; const bf16s* x;
    float* partial;                // [S][9][64][16]
    int N, H, W, imgs_per_group;
    int tiles_y, tiles_x, n_mtiles, S, per_split;
    float invM;
    // z == nullptr: dA IS dz (no BatchNorm backward on load).  Dz > 0 (bdn_conv3d_wgrad, 16-channel inputs): the N images are depth slices of
    // N / Dz samples and this launch is depth tap dshift + 1 -- dz of slice n pairs with the input slice n + dshift; a partner outside the
    // sample contributes nothing (its patch is staged as zeros)
    int Dz, dshift;
};
struct WgF {
    using TL = Tile<8, 16, 1>;
    static constexpr int PSTR = 32;                            // 16 bf16 channels per patch pixel
    static constexpr int PATCH_BYTES = TL::NPIX * PSTR;        // 5760
    static constexpr int DSTR = 192;                           // dz pixel stride (128 B + 64: conflict-free transposing reads)
    static constexpr int DZ_BYTES = TL::BM * DSTR;             // 24576
    static constexpr int RED_BYTES = 2 * 5 * 16 * 64 * 4;      // the two odd-k waves' accumulators
    static constexpr int SMEM = (PATCH_BYTES + DZ_BYTES) > RED_BYTES ? (PATCH_BYTES + DZ_BYTES) : RED_BYTES;
};

// P3: the plain-dz / depth-tap form (see WgFirstArgs); false = the 2-D training kernel, compiled without those branches
template <bool P3>
__global__ __launch_bounds__(256, 2) void wgrad_first_kernel(WgFirstArgs a) {
    using TL = WgF::TL;
    constexpr int C = 64, PSTR = WgF::PSTR, DSTR = WgF::DSTR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem;
    unsigned char* dzt = smem + WgF::PATCH_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, kpar = wave & 1;                // co [wm*32,+32); k-steps kpar, kpar+2, ...
    const int half = lane >> 5, l31 = lane & 31;
    const int split = xcd_remap(blockIdx.x, gridDim.x);

    // staging ownership: dz units u = tid + 256 i -> (tile slot u >> 3, channels 8 (tid & 7) ..); patch units
    // u = tid + 256 i < 360 -> (patch pixel u >> 1, channels 8 (u & 1) ..)
    const int c8 = (tid & 7) * 8;
    uint4 gq[4], zq[4], pq[2];
    unsigned ok = 0;                                           // bits 0-3: dz unit inside the image; bits 8-9: patch unit
    int grp_next = 0, grp_cur = -1;
    float mean[8], inv[8], sc[8], sh[8], k0[8], k1[8];

    f32x16 acc[5];
#pragma unroll
    for (int j = 0; j < 5; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;

#define LOAD_CHUNK(q_)                                                                                   \
    {                                                                                                   \
        const int tx_ = (q_) % a.tiles_x, ty_ = ((q_) / a.tiles_x) % a.tiles_y, n_ = (q_) / (a.tiles_x * a.tiles_y); \
        const int y0_ = ty_ * 8, x0_ = tx_ * 16;                                                        \
        grp_next = n_ / a.imgs_per_group; ok = 0;                                                       \
        const bool pv_ = !P3 || (unsigned)(n_ % a.Dz + a.dshift) < (unsigned)a.Dz;   /* 3x3x3: the partner slice lies inside the sample */ \
        const int np_ = P3 ? n_ + (pv_ ? a.dshift : 0) : n_;                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                  \
            const int slot = (tid + i * 256) >> 3, y = y0_ + (slot >> 4), x = x0_ + (slot & 15);         \
            const bool ok_ = y < a.H && x < a.W;                                                         \
            const size_t pix = ((size_t)n_ * a.H + (ok_ ? y : y0_)) * a.W + (ok_ ? x : x0_);             \
            gq[i] = *reinterpret_cast<const uint4*>(a.dA + pix * a.ldA + c8);                            \
            if (!P3) zq[i] = *reinterpret_cast<const uint4*>(a.z + pix * C + c8);             \
            ok |= (ok_ ? 1u : 0u) << i;                                                                  \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 2; i++) {                                                  \
            const int u = tid + i * 256, pp = u >> 1, yy = pp / TL::PW, xx = pp % TL::PW;                \
            const int y = y0_ + yy - 1, x = x0_ + xx - 1;                                                \
            const bool ok_ = pv_ && u < TL::NPIX * 2 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W; \
            const size_t pix = ((size_t)np_ * a.H + (ok_ ? y : y0_)) * a.W + (ok_ ? x : x0_);            \
            pq[i] = *reinterpret_cast<const uint4*>(a.x + pix * 16 + (u & 1) * 8);                       \
            ok |= (ok_ ? 1u : 0u) << (8 + i);                                                            \
        }                                                                                               \
    }
#define STORE_CHUNK()                                                                                    \
    {                                                                                                   \
        if (!P3 && grp_next != grp_cur) {                                                    \
            grp_cur = grp_next;                                                                         \
            _Pragma("unroll") for (int e = 0; e < 8; e++) {                                              \
                mean[e] = bn_row(a.bn, grp_cur, 0, C)[c8 + e]; inv[e] = bn_row(a.bn, grp_cur, 1, C)[c8 + e]; \
                sc[e] = bn_row(a.bn, grp_cur, 2, C)[c8 + e]; sh[e] = bn_row(a.bn, grp_cur, 3, C)[c8 + e]; \
                k0[e] = a.sums[((size_t)grp_cur * 2 + 0) * C + c8 + e] * a.invM;                         \
                k1[e] = a.sums[((size_t)grp_cur * 2 + 1) * C + c8 + e] * a.invM;                         \
            }                                                                                           \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                  \
            float fz[8], fg[8], o[8];                                                                   \
            Unit<bf16s>::unpack(zq[i], fz);                                                              \
            Unit<bf16s>::unpack(gq[i], fg);                                                              \
            _Pragma("unroll") for (int e = 0; e < 8; e++) {                                              \
                const float gm = fmaf(fz[e], sc[e], sh[e]) > 0.f ? fg[e] : 0.f;                          \
                const float xhat = (fz[e] - mean[e]) * inv[e];                                          \
                o[e] = sc[e] * (gm - k0[e] - xhat * k1[e]);                                             \
            }                                                                                           \
            uint4 v_ = !P3 ? Unit<bf16s>::pack(o) : gq[i];      /* plain mode: dA is dz already */ \
            const bool ok_ = (ok >> i) & 1u;                                                             \
            v_.x = ok_ ? v_.x : 0u; v_.y = ok_ ? v_.y : 0u; v_.z = ok_ ? v_.z : 0u; v_.w = ok_ ? v_.w : 0u; \
            *reinterpret_cast<uint4*>(dzt + ((tid + i * 256) >> 3) * DSTR + (tid & 7) * 16) = v_;        \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 2; i++) {                                                  \
            const int u = tid + i * 256;                                                                 \
            uint4 v_ = pq[i];                                                                           \
            const bool ok_ = (ok >> (8 + i)) & 1u;                                                       \
            v_.x = ok_ ? v_.x : 0u; v_.y = ok_ ? v_.y : 0u; v_.z = ok_ ? v_.z : 0u; v_.w = ok_ ? v_.w : 0u; \
            if (u < TL::NPIX * 2) *reinterpret_cast<uint4*>(patch + (u >> 1) * PSTR + (u & 1) * 16) = v_; \
        }                                                                                               \
    }

    // MFMA operand addressing.  A (dz): pixel kpix of a 4-pixel group, 4-channel piece (lane & 3) of 16-channel block
    // ((lane >> 4) & 1) of the wave's 32 co.  B (patch): the same pixel / piece roles over the 16 input channels; the lane's
    // column block ((lane >> 4) & 1) selects the tap of the pair instead of a second 16-channel block.
    const int kpix = (lane & 15) >> 2, upper = (lane >> 4) & 1;
    const unsigned a_off = (half * 8 + kpix) * DSTR + wm * 64 + (16 * upper + 4 * (lane & 3)) * 2;
    const unsigned b_off = (lane & 3) * 8;
    unsigned tapoff[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int tap = min(2 * j + upper, 8);
        tapoff[j] = ((tap / 3) * TL::PW + tap % 3) * PSTR + b_off;
    }

    const int q_begin = split * a.per_split;
    const int q_end = min(a.n_mtiles, q_begin + a.per_split);
    if (q_begin < q_end) LOAD_CHUNK(q_begin)
    for (int q = q_begin; q < q_end; q++) {
        __syncthreads();                                   // previous chunk's LDS reads are done
        STORE_CHUNK()
        __syncthreads();
        if (q + 1 < q_end) LOAD_CHUNK(q + 1)
        __builtin_amdgcn_sched_barrier(0);                 // keep the prefetch loads above the MFMAs
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const int kk = 2 * ks + kpar;                  // 16-pixel k-step = tile row kk
            const uint4 af = tr_pair(dzt + a_off + kk * 16 * DSTR, dzt + a_off + (kk * 16 + 4) * DSTR);
            const unsigned char* pb = patch + (kk * TL::PW + half * 8 + kpix) * PSTR;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const uint4 bfr = tr_pair(pb + tapoff[j], pb + tapoff[j] + 4 * PSTR);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bfr), acc[j], 0, 0, 0);
            }
        }
    }
#undef LOAD_CHUNK
#undef STORE_CHUNK

    // odd-k waves -> LDS -> even-k waves add and store partial[split][tap][co][ci]
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(smem) + (size_t)wm * 20 * 64 + lane;           // [wm][j][quad][lane]
    if (kpar == 1) {
#pragma unroll
        for (int j = 0; j < 5; j++)
#pragma unroll
            for (int qd = 0; qd < 4; qd++)
                red[(j * 4 + qd) * 64] = make_float4(acc[j][4 * qd], acc[j][4 * qd + 1], acc[j][4 * qd + 2], acc[j][4 * qd + 3]);
    }
    __syncthreads();
    if (kpar == 0) {
        const int ci = l31 & 15;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int tap = 2 * j + (l31 >> 4);
            float* pt = a.partial + (((size_t)split * 9 + tap) * C + wm * 32 + 4 * half) * 16 + ci;
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                const float4 o = red[(j * 4 + qd) * 64];
                if (tap < 9) {
                    pt[(8 * qd + 0) * 16] = acc[j][4 * qd + 0] + o.x;
                    pt[(8 * qd + 1) * 16] = acc[j][4 * qd + 1] + o.y;
                    pt[(8 * qd + 2) * 16] = acc[j][4 * qd + 2] + o.z;
                    pt[(8 * qd + 3) * 16] = acc[j][4 * qd + 3] + o.w;
                }
            }
        }
    }
}

// ---- bf16x3 / bf16x2 form of the first layer's weight gradient (round 6).  float32 dA and z; the staging forms dz in float32 with
// bn_bwd_apply_kernel<float>'s expression, splits it into bf16 hi + lo (bdn_split_pack's arithmetic) into TWO LDS tiles, the input patch comes
// split already ([pixel][hi 16 | lo 16] bf16: what bdn_pack_input(BDN_BF16X3) stores), and every (k-step, tap pair) issues
//     acc += dz_hi x_hi;  acc += dz_hi x_lo;  [acc += dz_lo x_hi]        (TERMS = 3; 2 = dz rounded to bf16: BDN_BF16X2)
// into the SAME accumulator -- no doubled operands, no quadrant tile, no combine pass; the split dz of the layer is never written, the
// bn_bwd_apply_split pass over dA and z (1.6 GB at B = 64) and the generic k-split GEMM behind it do not run.
struct WgFirstX3Args {
    const float* dA; int ldA; const float* z; const float* bn; const float* sums; const bf16s* xs;       // xs [N,H,W,32]: hi(16) | lo(16)
    float* partial;                // [S][9][64][16]
    int N, H, W, imgs_per_group;
    int tiles_y, tiles_x, n_mtiles, S, per_split;
    float invM;
};
struct WgFX {
    static constexpr int PATCH_BYTES = WgF::PATCH_BYTES, DZ_BYTES = WgF::DZ_BYTES;
    static constexpr int MAIN_BYTES = 2 * PATCH_BYTES + 2 * DZ_BYTES;
    static constexpr int SMEM = MAIN_BYTES > WgF::RED_BYTES ? MAIN_BYTES : WgF::RED_BYTES;
};

template <int TERMS>
__global__ __launch_bounds__(256, 2) void wgrad_first_x3_kernel(WgFirstX3Args a) {
    using TL = WgF::TL;
    constexpr int C = 64, PSTR = WgF::PSTR, DSTR = WgF::DSTR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch_h = smem;
    unsigned char* patch_l = smem + WgFX::PATCH_BYTES;
    unsigned char* dzt_h = smem + 2 * WgFX::PATCH_BYTES;
    unsigned char* dzt_l = dzt_h + WgFX::DZ_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, kpar = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    const int split = xcd_remap(blockIdx.x, gridDim.x);

    const int c8 = (tid & 7) * 8;
    uint4 gq[4][2], zq[4][2], ph[2], pl[2];
    unsigned ok = 0;
    int grp_next = 0, grp_cur = -1;
    float mean[8], inv[8], sc[8], sh[8], k0[8], k1[8];

    f32x16 acc[5];
#pragma unroll
    for (int j = 0; j < 5; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;

#define LOAD_CHUNK(q_)                                                                                   \
    {                                                                                                   \
        const int tx_ = (q_) % a.tiles_x, ty_ = ((q_) / a.tiles_x) % a.tiles_y, n_ = (q_) / (a.tiles_x * a.tiles_y); \
        const int y0_ = ty_ * 8, x0_ = tx_ * 16;                                                        \
        grp_next = n_ / a.imgs_per_group; ok = 0;                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                  \
            const int slot = (tid + i * 256) >> 3, y = y0_ + (slot >> 4), x = x0_ + (slot & 15);         \
            const bool ok_ = y < a.H && x < a.W;                                                         \
            const size_t pix = ((size_t)n_ * a.H + (ok_ ? y : y0_)) * a.W + (ok_ ? x : x0_);             \
            gq[i][0] = *reinterpret_cast<const uint4*>(a.dA + pix * a.ldA + c8);                         \
            gq[i][1] = *reinterpret_cast<const uint4*>(a.dA + pix * a.ldA + c8 + 4);                     \
            zq[i][0] = *reinterpret_cast<const uint4*>(a.z + pix * C + c8);                              \
            zq[i][1] = *reinterpret_cast<const uint4*>(a.z + pix * C + c8 + 4);                          \
            ok |= (ok_ ? 1u : 0u) << i;                                                                  \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 2; i++) {                                                  \
            const int u = tid + i * 256, pp = u >> 1, yy = pp / TL::PW, xx = pp % TL::PW;                \
            const int y = y0_ + yy - 1, x = x0_ + xx - 1;                                                \
            const bool ok_ = u < TL::NPIX * 2 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W; \
            const size_t pix = ((size_t)n_ * a.H + (ok_ ? y : y0_)) * a.W + (ok_ ? x : x0_);             \
            ph[i] = *reinterpret_cast<const uint4*>(a.xs + pix * 32 + (u & 1) * 8);                      \
            pl[i] = *reinterpret_cast<const uint4*>(a.xs + pix * 32 + 16 + (u & 1) * 8);                 \
            ok |= (ok_ ? 1u : 0u) << (8 + i);                                                            \
        }                                                                                               \
    }
#define STORE_CHUNK()                                                                                    \
    {                                                                                                   \
        if (grp_next != grp_cur) {                                                                      \
            grp_cur = grp_next;                                                                         \
            _Pragma("unroll") for (int e = 0; e < 8; e++) {                                              \
                mean[e] = bn_row(a.bn, grp_cur, 0, C)[c8 + e]; inv[e] = bn_row(a.bn, grp_cur, 1, C)[c8 + e]; \
                sc[e] = bn_row(a.bn, grp_cur, 2, C)[c8 + e]; sh[e] = bn_row(a.bn, grp_cur, 3, C)[c8 + e]; \
                k0[e] = a.sums[((size_t)grp_cur * 2 + 0) * C + c8 + e] * a.invM;                         \
                k1[e] = a.sums[((size_t)grp_cur * 2 + 1) * C + c8 + e] * a.invM;                         \
            }                                                                                           \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                  \
            float fz[8], fg[8], o[8], hf[8], r[8];                                                      \
            Unit<float>::unpack(zq[i][0], fz); Unit<float>::unpack(zq[i][1], fz + 4);                   \
            Unit<float>::unpack(gq[i][0], fg); Unit<float>::unpack(gq[i][1], fg + 4);                   \
            _Pragma("unroll") for (int e = 0; e < 8; e++) {                                              \
                const float gm = fmaf(fz[e], sc[e], sh[e]) > 0.f ? fg[e] : 0.f;                          \
                const float xhat = (fz[e] - mean[e]) * inv[e];                                          \
                o[e] = sc[e] * (gm - k0[e] - xhat * k1[e]);                                             \
                asm volatile("" : "+v"(o[e]));              /* the float32 dz, pinned: lo is the residual of the ROUNDED product (store_split4) */ \
            }                                                                                           \
            uint4 vh_ = Unit<bf16s>::pack(o);                                                            \
            Unit<bf16s>::unpack(vh_, hf);                                                                \
            _Pragma("unroll") for (int e = 0; e < 8; e++) r[e] = o[e] - hf[e];                           \
            uint4 vl_ = Unit<bf16s>::pack(r);                                                            \
            const bool ok_ = (ok >> i) & 1u;                                                             \
            vh_.x = ok_ ? vh_.x : 0u; vh_.y = ok_ ? vh_.y : 0u; vh_.z = ok_ ? vh_.z : 0u; vh_.w = ok_ ? vh_.w : 0u; \
            vl_.x = ok_ ? vl_.x : 0u; vl_.y = ok_ ? vl_.y : 0u; vl_.z = ok_ ? vl_.z : 0u; vl_.w = ok_ ? vl_.w : 0u; \
            const unsigned o_ = ((tid + i * 256) >> 3) * DSTR + (tid & 7) * 16;                          \
            *reinterpret_cast<uint4*>(dzt_h + o_) = vh_;                                                 \
            if (TERMS == 3) *reinterpret_cast<uint4*>(dzt_l + o_) = vl_;                                 \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 2; i++) {                                                  \
            const int u = tid + i * 256;                                                                 \
            uint4 vh_ = ph[i], vl_ = pl[i];                                                              \
            const bool ok_ = (ok >> (8 + i)) & 1u;                                                       \
            vh_.x = ok_ ? vh_.x : 0u; vh_.y = ok_ ? vh_.y : 0u; vh_.z = ok_ ? vh_.z : 0u; vh_.w = ok_ ? vh_.w : 0u; \
            vl_.x = ok_ ? vl_.x : 0u; vl_.y = ok_ ? vl_.y : 0u; vl_.z = ok_ ? vl_.z : 0u; vl_.w = ok_ ? vl_.w : 0u; \
            if (u < TL::NPIX * 2) {                                                                     \
                *reinterpret_cast<uint4*>(patch_h + (u >> 1) * PSTR + (u & 1) * 16) = vh_;               \
                *reinterpret_cast<uint4*>(patch_l + (u >> 1) * PSTR + (u & 1) * 16) = vl_;               \
            }                                                                                           \
        }                                                                                               \
    }

    const int kpix = (lane & 15) >> 2, upper = (lane >> 4) & 1;
    const unsigned a_off = (half * 8 + kpix) * DSTR + wm * 64 + (16 * upper + 4 * (lane & 3)) * 2;
    const unsigned b_off = (lane & 3) * 8;
    unsigned tapoff[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int tap = min(2 * j + upper, 8);
        tapoff[j] = ((tap / 3) * TL::PW + tap % 3) * PSTR + b_off;
    }

    const int q_begin = split * a.per_split;
    const int q_end = min(a.n_mtiles, q_begin + a.per_split);
    if (q_begin < q_end) LOAD_CHUNK(q_begin)
    for (int q = q_begin; q < q_end; q++) {
        __syncthreads();
        STORE_CHUNK()
        __syncthreads();
        if (q + 1 < q_end) LOAD_CHUNK(q + 1)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const int kk = 2 * ks + kpar;
            const unsigned ao = a_off + kk * 16 * DSTR;
            const uint4 afh = tr_pair(dzt_h + ao, dzt_h + ao + 4 * DSTR);
            uint4 afl = afh;
            if (TERMS == 3) afl = tr_pair(dzt_l + ao, dzt_l + ao + 4 * DSTR);
            const unsigned po = (kk * TL::PW + half * 8 + kpix) * PSTR;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const uint4 bh = tr_pair(patch_h + po + tapoff[j], patch_h + po + tapoff[j] + 4 * PSTR);
                const uint4 bl = tr_pair(patch_l + po + tapoff[j], patch_l + po + tapoff[j] + 4 * PSTR);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afh), __builtin_bit_cast(bf16x8, bh), acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afh), __builtin_bit_cast(bf16x8, bl), acc[j], 0, 0, 0);
                if (TERMS == 3)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afl), __builtin_bit_cast(bf16x8, bh), acc[j], 0, 0, 0);
            }
        }
    }
#undef LOAD_CHUNK
#undef STORE_CHUNK

    __syncthreads();
    float4* red = reinterpret_cast<float4*>(smem) + (size_t)wm * 20 * 64 + lane;           // [wm][j][quad][lane]
    if (kpar == 1) {
#pragma unroll
        for (int j = 0; j < 5; j++)
#pragma unroll
            for (int qd = 0; qd < 4; qd++)
                red[(j * 4 + qd) * 64] = make_float4(acc[j][4 * qd], acc[j][4 * qd + 1], acc[j][4 * qd + 2], acc[j][4 * qd + 3]);
    }
    __syncthreads();
    if (kpar == 0) {
        const int ci = l31 & 15;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int tap = 2 * j + (l31 >> 4);
            float* pt = a.partial + (((size_t)split * 9 + tap) * C + wm * 32 + 4 * half) * 16 + ci;
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                const float4 o = red[(j * 4 + qd) * 64];
                if (tap < 9) {
                    pt[(8 * qd + 0) * 16] = acc[j][4 * qd + 0] + o.x;
                    pt[(8 * qd + 1) * 16] = acc[j][4 * qd + 1] + o.y;
                    pt[(8 * qd + 2) * 16] = acc[j][4 * qd + 2] + o.z;
                    pt[(8 * qd + 3) * 16] = acc[j][4 * qd + 3] + o.w;
                }
            }
        }
    }
}

// dw[co][ci][tap] (OIHW f32, ci < Cin_real) = sum_s partial[s][tap][co][ci].
// Block = SL split lanes x (256/SL) (co,ci) pairs: reads are coalesced along ci, the SL lanes walk the
// splits in parallel (fixed order -> deterministic), an LDS tree combines them, and each pair's nine taps
// leave as 36 contiguous bytes of the reference's OIHW layout.
template <int SL>
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                    int S, int Cout, int Cin, int Cin_real, int ostride, int ooff) {
    constexpr int PAIRS = 256 / SL;
    __shared__ float sm[SL][PAIRS][9];
    const size_t plane = (size_t)Cout * Cin;
    const int pl = threadIdx.x % PAIRS, sl = threadIdx.x / PAIRS;
    const size_t i = (size_t)blockIdx.x * PAIRS + pl;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++) acc[t] = 0.f;
    if (i < plane)
        for (int k = sl; k < S; k += SL) {
            const float* p = partial + (size_t)k * 9 * plane + i;
#pragma unroll
            for (int t = 0; t < 9; t++) acc[t] += p[(size_t)t * plane];
        }
#pragma unroll
    for (int t = 0; t < 9; t++) sm[sl][pl][t] = acc[t];
    __syncthreads();
    if (sl == 0 && i < plane) {
        const int ci = i % Cin, co = i / Cin;
        if (ci < Cin_real) {
#pragma unroll
            for (int t = 0; t < 9; t++) {
                float v = acc[t];
                for (int k = 1; k < SL; k++) v += sm[k][pl][t];
                dw[((size_t)co * Cin_real + ci) * ostride + ooff + t] = v;      // OIHW (stride 9) or tap kd of OIDHW (stride 27, offset 9 kd)
            }
        }
    }
}

static void launch_wgrad_reduce(const float* partial, float* dw, int S, int Cout, int Cin, int Cin_real, hipStream_t st,
                                int ostride = 9, int ooff = 0) {
    const size_t plane = (size_t)Cout * Cin;
    // split lanes give the small filters some parallelism -- up to ~256 blocks, never more lanes than splits.  More, thinner
    // blocks (the first version went to 2048) read 64-byte pieces of every partial tile: the same speed alone, but inside
    // the step, where this kernel shares HBM with the dz chain, the fatter blocks make the step 0.9 % shorter.
    int SL = 1;
    while (SL < 16 && SL * 2 <= S && plane / (256 / (SL * 2)) < 256) SL *= 2;
    const unsigned grid = (unsigned)((plane + 256 / SL - 1) / (256 / SL));
    switch (SL) {
        case 1: hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3(grid), dim3(256), 0, st, partial, dw, S, Cout, Cin, Cin_real, ostride, ooff); break;
        case 2: hipLaunchKernelGGL(wgrad_reduce_kernel<2>, dim3(grid), dim3(256), 0, st, partial, dw, S, Cout, Cin, Cin_real, ostride, ooff); break;
        case 4: hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3(grid), dim3(256), 0, st, partial, dw, S, Cout, Cin, Cin_real, ostride, ooff); break;
        case 8: hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3(grid), dim3(256), 0, st, partial, dw, S, Cout, Cin, Cin_real, ostride, ooff); break;
        default: hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3(grid), dim3(256), 0, st, partial, dw, S, Cout, Cin, Cin_real, ostride, ooff); break;
    }
}

// ---- plan: tile geometry, split count, kernel variant.  A pure function of the shape and of the caller's `flags` word
// (no process-wide tuning state: two threads may size and launch different plans concurrently).
//   flags bits 0-1   phases (bdn_conv3x3_wgrad_ex)
//   flags bits 8-11  kernel override: 0 = the library's choice, BDN_WG_SIMPLE forces the one-chunk-at-a-time kernel
//   flags bits 16-28 target grid size of the GEMM (0 = default, one block per CU)
#ifndef BDN_WG_X3_FUSED
#define BDN_WG_X3_FUSED 1     /* A/B switch of the round (tools/build_lib_variant.sh old "-DBDN_WG_X3_FUSED=0") */
#endif
constexpr int WG_SIMPLE_MULT = 2;
constexpr int WG_X3_SKIP = 1 << 30;          // internal plan flag, see wgrad_plan
constexpr int WG_X3_TWO = 1 << 29;           // internal plan flag (BDN_BF16X2): only the hi half of dz -- the [lo, hi] quadrant is left out as well
struct WgPlan { TileGeom g; int S, per_split, n_cot, n_cit; bool ksplit; int variant; int x3h, n_tiles; };
static WgPlan wgrad_plan(int dtype, int N, int H, int W, int Cout, int C0, int C1, int imgs_per_group, int in_mode, int flags) {
    WgPlan p;
    const int Cin = C0 + C1;
    p.g = pick_tile(N, H, W, imgs_per_group);
    p.n_cot = Cout / 64;
    p.n_cit = (Cin + 63) / 64;
    p.ksplit = Cin <= 32;
    // internal flag (bit 30, set by the bf16x3 entry on its doubled-operand call): leave the lo x lo quadrant out
    p.x3h = ((flags >> 30) & 1) && p.n_cot % 2 == 0 && p.n_cit % 2 == 0 ? p.n_cot / 2 : 0;
    const int tiles = (p.x3h && ((flags >> 29) & 1)) ? p.x3h * p.n_cit : p.n_cot * p.n_cit - (p.x3h ? p.x3h * (p.n_cit / 2) : 0);
    p.n_tiles = tiles;
    // the pipelined kernels cover full 64-channel input tiles on 8x16 spatial tiles whose tensors stay below 2^31 elements
    const size_t cmax = (size_t)(Cout > C0 ? (Cout > C1 ? Cout : C1) : (C0 > C1 ? C0 : C1));
    const bool pipe_ok = dtype == BDN_BF16 && !p.ksplit && p.g.TI == 1 && C0 % 64 == 0 && C1 % 64 == 0 &&
                         (size_t)N * H * W * cmax < ((size_t)1 << 31);
    const int want = (flags >> 8) & 15;
    p.variant = pipe_ok ? BDN_WG_ROLE : BDN_WG_SIMPLE;
    if (want == BDN_WG_SIMPLE) p.variant = BDN_WG_SIMPLE;
    // the simple kernel (first layer / 8x8 maps / f32) has no software pipeline: it hides latency with a second block per CU
    int blocks = (flags >> 16) & 0x1fff;
    if (blocks == 0) blocks = 128;                          // HALF the CUs: the GEMM runs beside the dz chain on a second stream, and two MFMA kernels
                                                            // sharing a CU lose throughput -- with 128 blocks (16 per XCD) the chain's convolutions
                                                            // always find free CUs and the partial-sum traffic halves again (A/B inside the step:
                                                            // 512 -> 256 = -2.7 %, 256 -> 128 = -2.6 %; 112 / 144 / 96: +3.5 % / +3.7 % / +4 %)
    int S = ((p.variant == BDN_WG_SIMPLE ? WG_SIMPLE_MULT : 1) * blocks + tiles - 1) / tiles;
    if (S > p.g.n_mtiles) S = p.g.n_mtiles;
    if (S < 1) S = 1;
    p.per_split = (p.g.n_mtiles + S - 1) / S;
    p.S = (p.g.n_mtiles + p.per_split - 1) / p.per_split;   // no empty splits
    return p;
}

extern "C" size_t bdn_wgrad_workspace_bytes_ex(int dtype, int N, int H, int W, int Cout, int C0, int C1, int imgs_per_group,
                                               int in_mode, int flags) {
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || C0 <= 0 || C1 < 0 || imgs_per_group <= 0) return 0;
    if (dtype == BDN_BF16X3 || dtype == BDN_BF16X2)      // doubled operands ([hi | lo] x [hi | lo]) through the bf16 plan + the [2 Cout][2 Cin][9] tile the quadrants are summed from
    {
        const size_t doubled = bdn_wgrad_workspace_bytes_ex(BDN_BF16, N, H, W, 2 * Cout, 2 * (C0 + C1), 0, imgs_per_group, BDN_IN_PLAIN,
                                                            flags | WG_X3_SKIP | (dtype == BDN_BF16X2 ? WG_X3_TWO : 0))
                               + (size_t)4 * Cout * (C0 + C1) * 9 * sizeof(float);
        const size_t fused = bdn_wgrad_workspace_bytes_ex(BDN_BF16, N, H, W, Cout, C0 + C1, 0, imgs_per_group, BDN_IN_PLAIN, flags);   // wgrad7x: the logical plan
        return doubled > fused ? doubled : fused;
    }
    const WgPlan p = wgrad_plan(dtype, N, H, W, Cout, C0, C1, imgs_per_group, in_mode, flags);
    return (size_t)p.S * (p.ksplit ? 2 : 1) * 9 * Cout * (C0 + C1) * sizeof(float);
}

extern "C" size_t bdn_wgrad_workspace_bytes(int N, int H, int W, int Cout, int Cin, int imgs_per_group) {
    // default flags, any dtype / source split / input mode: the largest plan
    const size_t a = bdn_wgrad_workspace_bytes_ex(BDN_F32, N, H, W, Cout, Cin, 0, imgs_per_group, BDN_IN_PLAIN, 0);
    const size_t b = bdn_wgrad_workspace_bytes_ex(BDN_BF16, N, H, W, Cout, Cin, 0, imgs_per_group, BDN_IN_PLAIN, 0);
    const size_t c = bdn_wgrad_workspace_bytes_ex(BDN_BF16X3, N, H, W, Cout, Cin, 0, imgs_per_group, BDN_IN_PLAIN, 0);
    // bdn_conv3x3_wgrad_bnbwd (first layer, end of backward) plans its own, larger grid
    const size_t d = Cin <= 32 ? bdn_wgrad_workspace_bytes_ex(BDN_BF16, N, H, W, Cout, Cin, 0, imgs_per_group, BDN_IN_PLAIN, BDN_WG_FLAGS(0, 0, 256)) : 0;
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    return m > d ? m : d;
}

template <typename T, int TH, int TW, int TI, bool KSPLIT>
static int launch_wgrad(const WgradArgs& a, hipStream_t st) {
    using CF = WgCfg<T, TH, TW, TI>;
    auto kern = wgrad_kernel<T, TH, TW, TI, KSPLIT>;
    BDN_SET_SMEM_ONCE(kern, CF::SMEM, "wgrad");
    hipLaunchKernelGGL(kern, dim3(a.S * a.n_tiles), dim3(256), CF::SMEM, st, a);
    BDN_CHECK_LAUNCH("wgrad");
    return BDN_OK;
}

// flags bits 0-1 = phases: bit 0 = the split-K GEMM (partial tiles into `partial`), bit 1 = the fixed-order reduction of the
// partial tiles into dw_oihw.  bdn_conv3x3_wgrad runs both; a profiler that wants the GEMM's own duration calls them apart.
// Higher bits: per-call plan overrides (wgrad_plan above); `partial` must then hold bdn_wgrad_workspace_bytes_ex(flags).
extern "C" int bdn_conv3x3_wgrad_ex(int dtype, const void* dz, int Cout,
                                    const void* in0, int C0, const void* in1, int C1,
                                    int in_mode, const float* in_bn, int imgs_per_group,
                                    float* partial, float* dw_oihw, int Cin_real,
                                    int N, int H, int W, int phases, void* stream) {
    if (!dz || !in0 || !partial || !dw_oihw) BDN_FAIL(BDN_E_ARG, "wgrad: null pointer");
    if (!(phases & 3)) BDN_FAIL(BDN_E_ARG, "wgrad: phases must select the GEMM (1), the reduction (2) or both (3)");
    if (dtype == BDN_BF16X3 || dtype == BDN_BF16X2) {
        const int xfl = WG_X3_SKIP | (dtype == BDN_BF16X2 ? WG_X3_TWO : 0);          // BDN_BF16X2: dw = T[hi,hi] + T[hi,lo] (dz rounded to bf16)
        // dz = split operand [N,H,W,2 Cout] (hi | lo), in0 = split operand [N,H,W,2 C0] from bdn_split_pack (which did any cat /
        // BatchNorm+ReLU): the bf16 GEMM on the doubled operands yields T = [2 Cout][2 C0][9]; dw = T[hi,hi] + T[hi,lo] + T[lo,hi].
        if (in1 || in_mode != BDN_IN_PLAIN) BDN_FAIL(BDN_E_ARG, "wgrad(bf16x3): one split-packed, plain operand");
        if (Cout <= 0 || Cout % 32 || C0 <= 0 || C0 % 8 || Cin_real <= 0 || Cin_real > C0)
            BDN_FAIL(BDN_E_SHAPE, "wgrad(bf16x3): Cout=%d must be a multiple of 32, C0=%d of 8, Cin_real=%d <= C0", Cout, C0, Cin_real);
        if (N <= 0 || H <= 0 || W <= 0 || imgs_per_group <= 0 || N % imgs_per_group)
            BDN_FAIL(BDN_E_SHAPE, "wgrad(bf16x3): bad N=%d H=%d W=%d imgs_per_group=%d", N, H, W, imgs_per_group);
        if (BDN_WG_X3_FUSED && Cout % 64 == 0 && C0 % 64 == 0 && (size_t)N * H * W * 2 * (size_t)(Cout > C0 ? Cout : C0) < ((size_t)1 << 31)) {
            // full 64-channel tiles on 8 x 16 spatial tiles: the terms of the split product in one accumulator (wgrad7x_kernel), the plan of
            // the LOGICAL [Cout] x [C0] problem, the plain split-K reduction straight into dw
            const WgPlan pf = wgrad_plan(BDN_BF16, N, H, W, Cout, C0, 0, imgs_per_group, BDN_IN_PLAIN, phases);
            if (pf.variant == BDN_WG_ROLE) {
                WgradArgs a;
                a.dz = dz; a.Cout = Cout; a.in0 = in0; a.in1 = nullptr; a.C0 = C0; a.C1 = 0; a.in_bn = nullptr; a.imgs_per_group = imgs_per_group;
                a.partial = partial; a.N = N; a.H = H; a.W = W;
                a.tiles_y = pf.g.tiles_y; a.tiles_x = pf.g.tiles_x; a.n_mtiles = pf.g.n_mtiles;
                a.S = pf.S; a.per_split = pf.per_split; a.n_cot = pf.n_cot; a.n_cit = pf.n_cit; a.x3h = 0; a.n_tiles = pf.n_tiles;
                a.Dz = 0; a.dshift = 0;
                hipStream_t st = reinterpret_cast<hipStream_t>(stream);
                if (phases & 1) {
                    const int rc = dtype == BDN_BF16X3 ? launch_wgrad7x<3>(a, st) : launch_wgrad7x<2>(a, st);
                    if (rc) return rc;
                }
                if (phases & 2) {
                    launch_wgrad_reduce(partial, dw_oihw, pf.S, Cout, C0, Cin_real, st);
                    BDN_CHECK_LAUNCH("wgrad_reduce");
                }
                return BDN_OK;
            }
        }
        const size_t gemm_bytes = bdn_wgrad_workspace_bytes_ex(BDN_BF16, N, H, W, 2 * Cout, 2 * C0, 0, imgs_per_group, BDN_IN_PLAIN, phases | xfl);
        float* tile = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(partial) + gemm_bytes);
        const int rc = bdn_conv3x3_wgrad_ex(BDN_BF16, dz, 2 * Cout, in0, 2 * C0, nullptr, 0, BDN_IN_PLAIN, nullptr, imgs_per_group,
                                            partial, tile, 2 * C0, N, H, W, phases | xfl, stream);
        if (rc) return rc;
        if (phases & 2) return bdn_wgrad_x3_combine(tile, dw_oihw, Cout, C0, Cin_real, 9, dtype == BDN_BF16X2 ? 2 : 3, reinterpret_cast<hipStream_t>(stream));
        return BDN_OK;
    }
    if (N <= 0 || H <= 0 || W <= 0 || imgs_per_group <= 0 || N % imgs_per_group)
        BDN_FAIL(BDN_E_SHAPE, "wgrad: bad N=%d H=%d W=%d imgs_per_group=%d", N, H, W, imgs_per_group);
    if (Cout <= 0 || Cout % 64) BDN_FAIL(BDN_E_SHAPE, "wgrad: Cout=%d must be a multiple of 64", Cout);
    if (in1 == nullptr) C1 = 0;
    if (C0 <= 0 || C0 % 16 || C1 % 64 || (in1 && C0 % 64))
        BDN_FAIL(BDN_E_SHAPE, "wgrad: C0=%d must be a multiple of 16 (64 with a second source), C1=%d of 64", C0, C1);
    if (in_mode == BDN_IN_BNRELU && (!in_bn || in1)) BDN_FAIL(BDN_E_ARG, "wgrad: BNRELU input needs in_bn and a single source");
    const int Cin = C0 + C1;
    if (Cin_real <= 0 || Cin_real > Cin) BDN_FAIL(BDN_E_SHAPE, "wgrad: Cin_real=%d out of range", Cin_real);
    if (dtype != BDN_BF16 && dtype != BDN_F32) BDN_FAIL(BDN_E_ARG, "wgrad: bad dtype %d", dtype);
    const WgPlan p = wgrad_plan(dtype, N, H, W, Cout, C0, C1, imgs_per_group, in_mode, phases);
    WgradArgs a;
    a.dz = dz; a.Cout = Cout; a.in0 = in0; a.in1 = in1; a.C0 = C0; a.C1 = C1;
    a.in_bn = in_mode == BDN_IN_BNRELU ? in_bn : nullptr; a.imgs_per_group = imgs_per_group;
    a.partial = partial; a.N = N; a.H = H; a.W = W;
    a.tiles_y = p.g.tiles_y; a.tiles_x = p.g.tiles_x; a.n_mtiles = p.g.n_mtiles;
    a.S = p.S; a.per_split = p.per_split; a.n_cot = p.n_cot; a.n_cit = p.n_cit; a.x3h = p.x3h; a.n_tiles = p.n_tiles;
    a.Dz = 0; a.dshift = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc = BDN_OK;
    if (!(phases & 1)) {
    } else if (dtype == BDN_BF16) {
        if (p.variant == BDN_WG_ROLE) rc = a.in_bn ? launch_wgrad7<true>(a, st) : launch_wgrad7<false>(a, st);
        else if (p.ksplit) rc = p.g.TI == 1 ? launch_wgrad<bf16s, 8, 16, 1, true>(a, st) : launch_wgrad<bf16s, 8, 8, 2, true>(a, st);
        else rc = p.g.TI == 1 ? launch_wgrad<bf16s, 8, 16, 1, false>(a, st) : launch_wgrad<bf16s, 8, 8, 2, false>(a, st);
    } else {
        if (p.ksplit) rc = p.g.TI == 1 ? launch_wgrad<float, 8, 16, 1, true>(a, st) : launch_wgrad<float, 8, 8, 2, true>(a, st);
        else rc = p.g.TI == 1 ? launch_wgrad<float, 8, 16, 1, false>(a, st) : launch_wgrad<float, 8, 8, 2, false>(a, st);
    }
    if (rc) return rc;
    if (phases & 2) {
        launch_wgrad_reduce(partial, dw_oihw, p.S * (p.ksplit ? 2 : 1), Cout, Cin, Cin_real, st);
        BDN_CHECK_LAUNCH("wgrad_reduce");
    }
    return BDN_OK;
}

extern "C" int bdn_conv3x3_wgrad(int dtype, const void* dz, int Cout,
                                 const void* in0, int C0, const void* in1, int C1,
                                 int in_mode, const float* in_bn, int imgs_per_group,
                                 float* partial, float* dw_oihw, int Cin_real,
                                 int N, int H, int W, void* stream) {
    return bdn_conv3x3_wgrad_ex(dtype, dz, Cout, in0, C0, in1, C1, in_mode, in_bn, imgs_per_group, partial, dw_oihw, Cin_real,
                                N, H, W, 3, stream);
}

// The first layer's weight gradient straight from dA and z (BatchNorm+ReLU backward applied while staging; `sums` from
// bdn_bn_bwd_finalize).  Returns BDN_E_SHAPE outside its shape class -- ask bdn_conv3x3_wgrad_bnbwd_supported first.
extern "C" int bdn_conv3x3_wgrad_bnbwd_supported(int dtype, int N, int H, int W, int Cout, int C0, int imgs_per_group) {
    if ((dtype != BDN_BF16 && dtype != BDN_BF16X3 && dtype != BDN_BF16X2) || Cout != 64 || C0 != 16 || N <= 0 || H <= 0 || W <= 0 ||
        imgs_per_group <= 0 || N % imgs_per_group) return 0;
    if (pick_tile(N, H, W, imgs_per_group).TI != 1) return 0;
    return (size_t)N * H * W * 64 < ((size_t)1 << 31) ? 1 : 0;
}

extern "C" int bdn_conv3x3_wgrad_bnbwd(int dtype, const void* dA, int ldA, const void* z, const float* bn, const float* sums,
                                       int imgs_per_group, int Cout, const void* in0, int C0,
                                       float* partial, float* dw_oihw, int Cin_real, int N, int H, int W, void* stream) {
    if (!dA || !z || !bn || !sums || !in0 || !partial || !dw_oihw) BDN_FAIL(BDN_E_ARG, "wgrad_bnbwd: null pointer");
    if (!bdn_conv3x3_wgrad_bnbwd_supported(dtype, N, H, W, Cout, C0, imgs_per_group))
        BDN_FAIL(BDN_E_SHAPE, "wgrad_bnbwd: only bf16 / bf16x3 / bf16x2, Cout=64, C0=16, 8x16 tiles (got dtype %d Cout %d C0 %d)", dtype, Cout, C0);
    if (ldA < 64 || ldA % 8 || Cin_real <= 0 || Cin_real > 16) BDN_FAIL(BDN_E_SHAPE, "wgrad_bnbwd: bad ldA=%d / Cin_real=%d", ldA, Cin_real);
    // runs at the very end of backward on the MAIN stream (two blocks per CU, nothing of the chain left): its own grid of 512
    const WgPlan p = wgrad_plan(BDN_BF16, N, H, W, Cout, C0, 0, imgs_per_group, BDN_IN_PLAIN, BDN_WG_FLAGS(0, 0, 256));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype != BDN_BF16) {
        // float32 dA [.., ldA] and z [.., 64]; in0 = the first convolution's split operand [N,H,W,32] bf16 = hi(16) | lo(16)
        WgFirstX3Args x;
        x.dA = (const float*)dA; x.ldA = ldA; x.z = (const float*)z; x.bn = bn; x.sums = sums; x.xs = (const bf16s*)in0;
        x.partial = partial; x.N = N; x.H = H; x.W = W; x.imgs_per_group = imgs_per_group;
        x.tiles_y = p.g.tiles_y; x.tiles_x = p.g.tiles_x; x.n_mtiles = p.g.n_mtiles; x.S = p.S; x.per_split = p.per_split;
        x.invM = 1.f / (float)((size_t)imgs_per_group * H * W);
        if (dtype == BDN_BF16X3) {
            auto kern = wgrad_first_x3_kernel<3>;
            BDN_SET_SMEM_ONCE(kern, WgFX::SMEM, "wgrad_first_x3");
            hipLaunchKernelGGL(kern, dim3(p.S), dim3(256), WgFX::SMEM, st, x);
        } else {
            auto kern = wgrad_first_x3_kernel<2>;
            BDN_SET_SMEM_ONCE(kern, WgFX::SMEM, "wgrad_first_x3");
            hipLaunchKernelGGL(kern, dim3(p.S), dim3(256), WgFX::SMEM, st, x);
        }
        BDN_CHECK_LAUNCH("wgrad_first_x3");
        launch_wgrad_reduce(partial, dw_oihw, p.S, Cout, C0, Cin_real, st);
        BDN_CHECK_LAUNCH("wgrad_reduce");
        return BDN_OK;
    }
    WgFirstArgs a;
    a.dA = (const bf16s*)dA; a.ldA = ldA; a.z = (const bf16s*)z; a.bn = bn; a.sums = sums; a.x = (const bf16s*)in0;
    a.partial = partial; a.N = N; a.H = H; a.W = W; a.imgs_per_group = imgs_per_group;
    a.tiles_y = p.g.tiles_y; a.tiles_x = p.g.tiles_x; a.n_mtiles = p.g.n_mtiles; a.S = p.S; a.per_split = p.per_split;
    a.invM = 1.f / (float)((size_t)imgs_per_group * H * W);
    a.Dz = 0; a.dshift = 0;
    hipLaunchKernelGGL(wgrad_first_kernel<false>, dim3(p.S), dim3(256), WgF::SMEM, st, a);
    BDN_CHECK_LAUNCH("wgrad_first");
    launch_wgrad_reduce(partial, dw_oihw, p.S, Cout, C0, Cin_real, st);
    BDN_CHECK_LAUNCH("wgrad_reduce");
    return BDN_OK;
}

// which kernel the GEMM phase runs for `flags` (bench.py names its roofline line after it): BDN_WG_SIMPLE / BDN_WG_ROLE
extern "C" int bdn_conv3x3_wgrad_variant(int dtype, int N, int H, int W, int Cout, int C0, int C1, int imgs_per_group, int in_mode, int flags) {
    if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || C0 <= 0 || imgs_per_group <= 0) return 0;
    if (dtype == BDN_BF16X3) return wgrad_plan(BDN_BF16, N, H, W, 2 * Cout, 2 * (C0 + C1), 0, imgs_per_group, BDN_IN_PLAIN, flags).variant;
    return wgrad_plan(dtype, N, H, W, Cout, C0, C1, imgs_per_group, in_mode, flags).variant;
}

// ---- weight gradient of the 3x3x3 convolution (bdn_conv3d): dw[co][ci][kd][kh][kw], f32 OIDHW.
// Three runs of the 2-D split-K GEMM, one per depth tap: dz slice n against the activation slice n + kd - 1 (chunks whose partner
// lies outside the sample are skipped), each reduced into its 9 of the 27 taps.  Plain inputs only (materialise relu(bn(z)) with
// bdn_bnrelu first).  partial: bdn_wgrad_workspace_bytes_ex(dtype, N*D, H, W, Cout, C, 0, 1, BDN_IN_PLAIN, 0) bytes.
extern "C" int bdn_conv3d_wgrad(int dtype, const void* dz, int Cout, const void* in, int C,
                                float* partial, float* dw_oidhw, int Cin_real, int N, int D, int H, int W, void* stream) {
    if (!dz || !in || !partial || !dw_oidhw) BDN_FAIL(BDN_E_ARG, "conv3d_wgrad: null pointer");
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0) BDN_FAIL(BDN_E_SHAPE, "conv3d_wgrad: bad N=%d D=%d H=%d W=%d", N, D, H, W);
    if (Cout <= 0 || Cout % 64 || C <= 0 || C % 16 || Cin_real <= 0 || Cin_real > C)
        BDN_FAIL(BDN_E_SHAPE, "conv3d_wgrad: Cout=%d must be a multiple of 64, C=%d of 16, Cin_real=%d <= C", Cout, C, Cin_real);
    if (dtype == BDN_BF16X3) {
        // dz [N,D,H,W,2 Cout] and in [N,D,H,W,2 C] are bdn_split_pack operands (hi | lo): the bf16 GEMM on the doubled operands gives
        // T = [2 Cout][2 C][27] behind the split-K partials; dw = T[hi,hi] + T[hi,lo] + T[lo,hi]
        if (Cout % 32 || C % 8) BDN_FAIL(BDN_E_SHAPE, "conv3d_wgrad(bf16x3): Cout=%d must be a multiple of 32, C=%d of 8", Cout, C);
        const size_t gemm_bytes = bdn_wgrad_workspace_bytes_ex(BDN_BF16, N * D, H, W, 2 * Cout, 2 * C, 0, 1, BDN_IN_PLAIN, 0);
        float* tile = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(partial) + gemm_bytes);
        const int rc = bdn_conv3d_wgrad(BDN_BF16, dz, 2 * Cout, in, 2 * C, partial, tile, 2 * C, N, D, H, W, stream);
        if (rc) return rc;
        return bdn_wgrad_x3_combine(tile, dw_oidhw, Cout, C, Cin_real, 27, 3, reinterpret_cast<hipStream_t>(stream));
    }
    if (dtype != BDN_BF16 && dtype != BDN_F32) BDN_FAIL(BDN_E_ARG, "conv3d_wgrad: bad dtype %d (bf16 / f32 / bf16x3)", dtype);
    const int NS = N * D;
    const WgPlan p = wgrad_plan(dtype, NS, H, W, Cout, C, 0, 1 /* one slice per tile */, BDN_IN_PLAIN, 0);
    if (p.g.TI != 1) BDN_FAIL(BDN_E_SHAPE, "conv3d_wgrad: internal plan error");
    if (dtype == BDN_BF16 && Cout == 64 && C == 16 && (size_t)NS * H * W * 64 < ((size_t)1 << 31)) {
        // the first layer of a multi-date stack (13 -> 64): the 2-D first-layer kernel in its plain-dz form, once per depth tap -- 64 x 16 x 9
        // accumulators per block, HBM-bound (dz is read three times, 4.6 TB/s) where the generic k-split GEMM ran at 1.6 TB/s
        WgFirstArgs f;
        f.dA = (const bf16s*)dz; f.ldA = Cout; f.z = nullptr; f.bn = nullptr; f.sums = nullptr; f.x = (const bf16s*)in;
        f.partial = partial; f.N = NS; f.H = H; f.W = W; f.imgs_per_group = 1;
        f.tiles_y = p.g.tiles_y; f.tiles_x = p.g.tiles_x; f.n_mtiles = p.g.n_mtiles; f.S = p.S * (p.ksplit ? 2 : 1);
        f.per_split = (p.g.n_mtiles + f.S - 1) / f.S;
        f.S = (p.g.n_mtiles + f.per_split - 1) / f.per_split;      // no empty splits (their partial tiles would be read uninitialised)
        f.invM = 0.f; f.Dz = D;
        hipStream_t st1 = reinterpret_cast<hipStream_t>(stream);
        for (int kd = 0; kd < 3; kd++) {
            f.dshift = kd - 1;
            hipLaunchKernelGGL(wgrad_first_kernel<true>, dim3(f.S), dim3(256), WgF::SMEM, st1, f);
            BDN_CHECK_LAUNCH("conv3d_wgrad_first");
            launch_wgrad_reduce(partial, dw_oidhw, f.S, Cout, C, Cin_real, st1, 27, 9 * kd);
            BDN_CHECK_LAUNCH("conv3d_wgrad_reduce");
        }
        return BDN_OK;
    }
    WgradArgs a;
    a.dz = dz; a.Cout = Cout; a.in0 = in; a.in1 = nullptr; a.C0 = C; a.C1 = 0; a.in_bn = nullptr; a.imgs_per_group = 1;
    a.partial = partial; a.N = NS; a.H = H; a.W = W;
    a.tiles_y = p.g.tiles_y; a.tiles_x = p.g.tiles_x; a.n_mtiles = p.g.n_mtiles;
    a.S = p.S; a.per_split = p.per_split; a.n_cot = p.n_cot; a.n_cit = p.n_cit; a.x3h = 0; a.n_tiles = p.n_tiles;
    a.Dz = D;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int kd = 0; kd < 3; kd++) {
        a.dshift = kd - 1;
        int rc;
        if (dtype == BDN_BF16) {
            if (p.variant == BDN_WG_ROLE) rc = launch_wgrad7<false>(a, st);
            else rc = p.ksplit ? launch_wgrad<bf16s, 8, 16, 1, true>(a, st) : launch_wgrad<bf16s, 8, 16, 1, false>(a, st);
        } else {
            rc = p.ksplit ? launch_wgrad<float, 8, 16, 1, true>(a, st) : launch_wgrad<float, 8, 16, 1, false>(a, st);
        }
        if (rc) return rc;
        launch_wgrad_reduce(partial, dw_oidhw, p.S * (p.ksplit ? 2 : 1), Cout, C, Cin_real, st, 27, 9 * kd);
        BDN_CHECK_LAUNCH("conv3d_wgrad_reduce");
    }
    return BDN_OK;
}
