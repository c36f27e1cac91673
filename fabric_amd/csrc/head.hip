// Boundary kernels: NCHW<->NHWC / weight packing, the 1x1 classifier, the Tversky loss and the SGD
// update.  All activations NHWC, 16-byte vector access
// along channels; per-channel reductions go through per-block partials (deterministic, no atomics).
#include "common.hpp"

static inline unsigned grid_for(size_t n, int block = 256) { return (unsigned)((n + block - 1) / block); }

// ============================================================ pack_input
// reference boundary: BiDateNet.forward(x_d1, x_d2), models/bidate_model.py:22 (NCHW f32)
// One thread per (pixel, 16-byte output unit): the band planes are read coalesced along x, every store is 16 bytes
// (the first version wrote Cpad scalars per thread: 100 us for 176 MB).
template <typename T>
__global__ void pack_input_kernel(const float* __restrict__ x1, const float* __restrict__ x2, T* __restrict__ out,
                                  int B, int C, int H, int W, int Cpad, FastDiv dhw, FastDiv dupp) {
    constexpr int EPU = ET<T>::EPU;
    const int upp = Cpad / EPU;
    const size_t hw = (size_t)H * W, total = (size_t)2 * B * hw * upp;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // the unit index is the slow coordinate inside an image so that a wave reads 64 consecutive pixels of one plane set
    int ti, pi, n, u; dhw.divmod((int)i, ti, pi); dupp.divmod(ti, n, u);        // (total < 2^31: the entry point keeps the tensor below 4 GB)
    const size_t p = pi;
    const float* src = (n < B ? x1 + (size_t)n * C * hw : x2 + (size_t)(n - B) * C * hw) + p;
    float f[EPU];
#pragma unroll
    for (int e = 0; e < EPU; e++) { const int c = u * EPU + e; f[e] = c < C ? src[(size_t)c * hw] : 0.f; }
    *reinterpret_cast<uint4*>(out + ((size_t)n * hw + p) * Cpad + u * EPU) = Unit<T>::pack(f);
}

// bf16x3: the packed input leaves directly as the [hi | lo] bf16 operand of the first convolution ([2B,H,W,2 Cpad]: bdn_split_pack's layout)
__global__ void pack_input_split_kernel(const float* __restrict__ x1, const float* __restrict__ x2, bf16s* __restrict__ out,
                                        int B, int C, int H, int W, int Cpad, FastDiv dhw, FastDiv dupp) {
    // eight channels per thread: the hi and the lo unit leave as one 16-byte store each (four channels per thread wrote 8 bytes at a 64-byte
    // stride: 88.7 us for 243 MB at B = 64)
    const int upp = Cpad / 8;
    const size_t hw = (size_t)H * W, total = (size_t)2 * B * hw * upp;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int ti, pi, n, u; dhw.divmod((int)i, ti, pi); dupp.divmod(ti, n, u);
    const size_t p = pi;
    const float* src = (n < B ? x1 + (size_t)n * C * hw : x2 + (size_t)(n - B) * C * hw) + p;
    float f[8], h[8], r[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { const int c = u * 8 + e; f[e] = c < C ? src[(size_t)c * hw] : 0.f; }
    const uint4 hi = Unit<bf16s>::pack(f);                     // bdn_split_pack's arithmetic: hi = bf16(x), lo = bf16(x - hi)
    Unit<bf16s>::unpack(hi, h);
#pragma unroll
    for (int e = 0; e < 8; e++) r[e] = f[e] - h[e];
    bf16s* dst = out + ((size_t)n * hw + p) * 2 * Cpad + u * 8;
    *reinterpret_cast<uint4*>(dst) = hi;
    *reinterpret_cast<uint4*>(dst + Cpad) = Unit<bf16s>::pack(r);
}

extern "C" int bdn_pack_input(int dtype, const float* x_d1, const float* x_d2, void* out,
                              int B, int C, int H, int W, int Cpad, void* stream) {
    if (!x_d1 || !x_d2 || !out) BDN_FAIL(BDN_E_ARG, "pack_input: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C || Cpad % 16) BDN_FAIL(BDN_E_SHAPE, "pack_input: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)2 * B * H * W;
    if (npix * (Cpad / 4) >= ((size_t)1 << 31)) BDN_FAIL(BDN_E_SHAPE, "pack_input: 2*B*H*W*Cpad/4 = %zu reaches 2^31; split the batch", npix * (Cpad / 4));
    if (dtype == BDN_BF16) hipLaunchKernelGGL(pack_input_kernel<bf16s>, dim3(grid_for(npix * (Cpad / 8))), dim3(256), 0, st, x_d1, x_d2, (bf16s*)out, B, C, H, W, Cpad, FastDiv(H * W), FastDiv(Cpad / 8));
    else if (dtype == BDN_F32) hipLaunchKernelGGL(pack_input_kernel<float>, dim3(grid_for(npix * (Cpad / 4))), dim3(256), 0, st, x_d1, x_d2, (float*)out, B, C, H, W, Cpad, FastDiv(H * W), FastDiv(Cpad / 4));
    else if (dtype == BDN_BF16X3) hipLaunchKernelGGL(pack_input_split_kernel, dim3(grid_for(npix * (Cpad / 8))), dim3(256), 0, st, x_d1, x_d2, (bf16s*)out, B, C, H, W, Cpad, FastDiv(H * W), FastDiv(Cpad / 8));
    else BDN_FAIL(BDN_E_ARG, "pack_input: bad dtype");
    BDN_CHECK_LAUNCH("pack_input");
    return BDN_OK;
}

// ============================================================ pack_weights
// forward image:        W_f(co, tap, ci) = w[co][ci][r][c]               (tap = 3r+c)
// data-gradient image:  W_d(ci, tap, co) = w[co][ci][2-r][2-c]  i.e. the transposed filter with taps rotated 180 deg
// both stored in MFMA fragment order (common.hpp: wfrag_index); Cout and Cin_pad are multiples of 32 / 16.
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd,
                                    int Cout, int Cin, int Cinp) {
    const size_t total = (size_t)Cout * 9 * Cinp;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ci = i % Cinp; const size_t t = i / Cinp; const int tap = t % 9; const int co = t / 9;
    const float v = ci < Cin ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.f;
    if (wf) wf[wfrag_index<T>(co, tap, ci, Cinp)] = from_f<T>(v);
    if (wd) wd[wfrag_index<T>(ci, 8 - tap, co, Cout)] = from_f<T>(v);
}

extern "C" int bdn_pack_weights(int dtype, const float* w_oihw, void* wf, void* wd,
                                int Cout, int Cin, int Cin_pad, void* stream) {
    if (!w_oihw || (!wf && !wd)) BDN_FAIL(BDN_E_ARG, "pack_weights: null pointer");
    if (Cout <= 0 || Cin <= 0 || Cin_pad < Cin || Cout % 32 || Cin_pad % 16 || (wd && Cin_pad % 32))
        BDN_FAIL(BDN_E_SHAPE, "pack_weights: Cout must be a multiple of 32, Cin_pad of 16 (32 with a data-gradient image)");
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)Cout * 9 * Cin_pad;
    if (dtype == BDN_BF16X3) return bdn_pack_weights_x3(w_oihw, wf, wd, Cout, Cin, Cin_pad, st);   // images of 3x the reduction length (x3.hip)
    if (dtype == BDN_BF16) hipLaunchKernelGGL(pack_weights_kernel<bf16s>, dim3(grid_for(total)), dim3(256), 0, st, w_oihw, (bf16s*)wf, (bf16s*)wd, Cout, Cin, Cin_pad);
    else if (dtype == BDN_F32) hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, w_oihw, (float*)wf, (float*)wd, Cout, Cin, Cin_pad);
    else BDN_FAIL(BDN_E_ARG, "pack_weights: bad dtype");
    BDN_CHECK_LAUNCH("pack_weights");
    return BDN_OK;
}

// all layers in one launch: desc[l] = {w, wf, wd, Cout, Cin, Cin_pad}; grid.y = layer
// one thread per OUTPUT element (coalesced 2/4-byte writes in fragment order; the strided reads of the
// 54 MB fp32 master hit L2)
template <typename T>
__global__ void pack_weights_multi_kernel(const PackDesc* __restrict__ desc) {
    constexpr int EPU = 16 / (int)sizeof(T), KCH = 32 / (int)sizeof(T), REC = 64 * EPU;
    const PackDesc d = desc[blockIdx.y];
    const size_t total = (size_t)d.Cout * 9 * d.Cinp;
    T* wf = reinterpret_cast<T*>(d.wf); T* wd = reinterpret_cast<T*>(d.wd);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = i % EPU, lane = (i / EPU) % 64; const size_t rec = i / REC;
        if (wf) {       // rec = (cb*9 + tap)*(Cinp/KCH) + kgi
            const int kgi = rec % (d.Cinp / KCH); const size_t t = rec / (d.Cinp / KCH); const int tap = t % 9, cb = t / 9;
            const int co = cb * 32 + (lane & 31), ci = kgi * KCH + (lane >> 5) * EPU + e;
            wf[i] = from_f<T>(ci < d.Cin ? d.w[((size_t)co * d.Cin + ci) * 9 + tap] : 0.f);
        }
        if (wd) {       // roles swapped: "cout" = ci (padded), "cin" = co; taps rotated by 180 degrees
            const int kgi = rec % (d.Cout / KCH); const size_t t = rec / (d.Cout / KCH); const int tap = t % 9, cb = t / 9;
            const int ci = cb * 32 + (lane & 31), co = kgi * KCH + (lane >> 5) * EPU + e;
            wd[i] = from_f<T>(ci < d.Cin ? d.w[((size_t)co * d.Cin + ci) * 9 + (8 - tap)] : 0.f);
        }
    }
}

// Regular layers (bf16, Cin a multiple of 64 and unpadded, Cout of 32 -- 17 of BiDateNet's 18) go through LDS instead: a block
// takes 32 output channels x 64 input channels x 9 taps, reads them as 32 contiguous 2304-byte rows of the OIHW master
// (16-byte loads), and writes complete 1 KB fragment records of both images with 16-byte stores.  The element-wise kernel
// above gathers 4-byte values 36 bytes apart and stores 2 bytes per lane: 61 us for the 107 MB of the step, and that
// sits alone at the start of every step.
__device__ __forceinline__ bool pack_regular(const PackDesc& d) { return d.Cin == d.Cinp && d.Cin % 64 == 0 && d.Cout % 32 == 0; }
__global__ __launch_bounds__(256) void pack_weights_tiles_kernel(const PackDesc* __restrict__ desc, int n_layers) {
    constexpr int ROW = 9 * 64 + 8;                            // LDS elements per output channel: [tap][ci] + 16 bytes of padding
    __shared__ __attribute__((aligned(16))) bf16s t[32 * ROW];
    const int tid = threadIdx.x;
    for (int item = blockIdx.x;; item += gridDim.x) {
        // which (layer, 32-co block, 64-ci block) is this item?
        int l = 0, local = item;
        for (; l < n_layers; l++) {
            const int cnt = pack_regular(desc[l]) ? (desc[l].Cout / 32) * (desc[l].Cin / 64) : 0;
            if (local < cnt) break;
            local -= cnt;
        }
        if (l == n_layers) return;                             // uniform for the block
        const PackDesc d = desc[l];
        const int ncc = d.Cin / 64, cb = local / ncc, cc = local % ncc, co0 = cb * 32, ci0 = cc * 64;
        __syncthreads();                                       // the previous item's LDS reads are done
        for (int q = tid; q < 32 * 144; q += 256) {            // 144 float4 per output-channel row
            const int r = q / 144, o4 = (q % 144) * 4;
            const float4 v = *reinterpret_cast<const float4*>(d.w + ((size_t)(co0 + r) * d.Cin + ci0) * 9 + o4);
            const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int o = o4 + e, ci = o / 9, tap = o % 9;
                t[r * ROW + tap * 64 + ci] = from_f<bf16s>(f[e]);
            }
        }
        __syncthreads();
        bf16s* wf = reinterpret_cast<bf16s*>(d.wf); bf16s* wd = reinterpret_cast<bf16s*>(d.wd);
        for (int u = tid; u < 36 * 64; u += 256) {             // 36 records x 64 lanes, 16 bytes each
            const int rec_l = u >> 6, lane = u & 63;
            if (wf) {      // record (tap, kq): lane = co & 31 + 32 * (ci % 16) / 8, 8 consecutive ci
                const int tap = rec_l >> 2, kq = rec_l & 3;
                const size_t rec = ((size_t)cb * 9 + tap) * (d.Cin / 16) + (ci0 / 16 + kq);
                *reinterpret_cast<uint4*>(wf + rec * 512 + lane * 8) =
                    *reinterpret_cast<const uint4*>(t + (lane & 31) * ROW + tap * 64 + kq * 16 + (lane >> 5) * 8);
            }
            if (wd) {      // roles swapped, taps rotated: record (ci block, 8 - tap, co group of 16): 8 consecutive co
                const int cbi = rec_l / 18, rem = rec_l % 18, tap = rem >> 1, kg = rem & 1;
                const int ci = cbi * 32 + (lane & 31), co8 = kg * 16 + (lane >> 5) * 8;
                unsigned short h[8];
#pragma unroll
                for (int e = 0; e < 8; e++) h[e] = t[(co8 + e) * ROW + tap * 64 + ci];
                const size_t rec = ((size_t)(ci0 / 32 + cbi) * 9 + (8 - tap)) * (d.Cout / 16) + (co0 / 16 + kg);
                *reinterpret_cast<uint4*>(wd + rec * 512 + lane * 8) =
                    make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
            }
        }
    }
}
// the element-wise kernel restricted to the layers the tile kernel does not take
__global__ void pack_weights_irregular_kernel(const PackDesc* __restrict__ desc) {
    if (pack_regular(desc[blockIdx.y])) return;
    constexpr int EPU = 8, KCH = 16, REC = 64 * EPU;
    const PackDesc d = desc[blockIdx.y];
    const size_t total = (size_t)d.Cout * 9 * d.Cinp;
    bf16s* wf = reinterpret_cast<bf16s*>(d.wf); bf16s* wd = reinterpret_cast<bf16s*>(d.wd);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = i % EPU, lane = (i / EPU) % 64; const size_t rec = i / REC;
        if (wf) {
            const int kgi = rec % (d.Cinp / KCH); const size_t t = rec / (d.Cinp / KCH); const int tap = t % 9, cb = t / 9;
            const int co = cb * 32 + (lane & 31), ci = kgi * KCH + (lane >> 5) * EPU + e;
            wf[i] = from_f<bf16s>(ci < d.Cin ? d.w[((size_t)co * d.Cin + ci) * 9 + tap] : 0.f);
        }
        if (wd) {
            const int kgi = rec % (d.Cout / KCH); const size_t t = rec / (d.Cout / KCH); const int tap = t % 9, cb = t / 9;
            const int ci = cb * 32 + (lane & 31), co = kgi * KCH + (lane >> 5) * EPU + e;
            wd[i] = from_f<bf16s>(ci < d.Cin ? d.w[((size_t)co * d.Cin + ci) * 9 + (8 - tap)] : 0.f);
        }
    }
}

extern "C" int bdn_pack_weights_multi(int dtype, const void* desc, int n_layers, void* stream) {
    if (!desc || n_layers <= 0) BDN_FAIL(BDN_E_ARG, "pack_weights_multi: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BDN_BF16) {
        hipLaunchKernelGGL(pack_weights_tiles_kernel, dim3(1024), dim3(256), 0, st, (const PackDesc*)desc, n_layers);
        hipLaunchKernelGGL(pack_weights_irregular_kernel, dim3(32, n_layers), dim3(256), 0, st, (const PackDesc*)desc);
    }
    else if (dtype == BDN_F32) hipLaunchKernelGGL(pack_weights_multi_kernel<float>, dim3(256, n_layers), dim3(256), 0, st, (const PackDesc*)desc);
    else if (dtype == BDN_BF16X3) return bdn_pack_weights_x3_multi((const PackDesc*)desc, n_layers, st);      // images of 3x the reduction length (x3.hip)
    else BDN_FAIL(BDN_E_ARG, "pack_weights_multi: bad dtype");
    BDN_CHECK_LAUNCH("pack_weights_multi");
    return BDN_OK;
}

// ============================================================ outconv 1x1 (unet_parts.py:86)
constexpr int OUTC_MAXCLS = 8;
constexpr int OUTC_ITERS = 16;      // pixels per thread in the classifier forward
constexpr int OUTC_BWD_ITERS = 32;  // ... and backward: every block ends with 130 same-address global atomics, which
                                    // serialise per address, so fewer, fatter blocks (2048 -> 1024 at full resolution: 117 -> 88 us)
// NC = compile-time class-count bound (2 for the change / no-change head, 8 generic): loops over classes unroll
// without runtime predicates.  CU = C/EPU consecutive lanes share one pixel (each reads 16 contiguous bytes ->
// fully coalesced), partial dot products are combined with xor-shuffles inside the CU-lane group.
// CUC: the lanes per pixel as a compile-time constant (8 or 16: the 64-channel head in bf16 / float32), 0 = run-time.  With it the
// partial dot products of a pixel meet in its first lane by DPP row shifts (same association as the xor butterfly: 4, 2, 1 -- the
// same bits) instead of ds_bpermute round trips in a run-time loop, and the next four pixels of a lane are requested before the
// current four are reduced (round 5: 42 -> 3x us at B = 64).
// (dpp_row_shl / first_lane_sum: common.hpp -- the eval-mode convolution epilogue sums a pixel's classifier terms the same way)
template <typename T, int NC, int CUC = 0>
__global__ __launch_bounds__(256) void outc_fwd_kernel(const T* __restrict__ z, const float* __restrict__ bn, const float* __restrict__ w,
                                const float* __restrict__ bias, float* __restrict__ logits, int npix, int hw, int C, int ncls, FastDiv dhw) {
    constexpr int EPU = ET<T>::EPU;
    const int CU = CUC ? CUC : C / EPU, rows = 256 / CU, tid = threadIdx.x, cu = tid % CU, row = tid / CU, c = cu * EPU;
    float sc[EPU], sh[EPU], wk[NC][EPU], bk[NC];
#pragma unroll
    for (int i = 0; i < EPU; i++) { sc[i] = bn_row(bn, 0, 2, C)[c + i]; sh[i] = bn_row(bn, 0, 3, C)[c + i]; }
#pragma unroll
    for (int k = 0; k < NC; k++) {
        const int kk = k < ncls ? k : 0;                                           // unconditional loads, selected afterwards
        const float bv = bias[kk];
        bk[k] = k < ncls ? bv : 0.f;
#pragma unroll
        for (int i = 0; i < EPU; i++) { const float wv = w[kk * C + c + i]; wk[k][i] = k < ncls ? wv : 0.f; }
    }
    const int p_begin = blockIdx.x * rows * OUTC_ITERS, p_end = min(npix, p_begin + rows * OUTC_ITERS);
    // four pixels of a lane are requested before the first is used (with one 16-byte load in flight per lane the pass ran at 3.6 TB/s),
    // and the following four before these are reduced.  RAGGED_ = false: the block's whole range lies inside the tensor (every block
    // but possibly the last) -- no per-pixel bounds tests
    uint4 u[4], un[4];
#define OUTC_LOAD4(dst_, p0_, RAGGED_)                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                      \
        const int p_ = (p0_) + j * rows + row;                                                          \
        dst_[j] = *reinterpret_cast<const uint4*>(z + (size_t)(!(RAGGED_) || p_ < p_end ? p_ : (p0_)) * C + c); \
    }
#define OUTC_RUN(RAGGED_)                                                                                \
    {                                                                                                   \
        OUTC_LOAD4(u, p_begin, RAGGED_)                                                                 \
        for (int p0 = p_begin; p0 < p_end; p0 += 4 * rows) {                      /* block-uniform trip count */ \
            const bool more = p0 + 4 * rows < p_end;                                                    \
            /* unconditional (the last trip re-requests its own pixels): a branch here makes the compiler wait for ALL loads, */ \
            /* the new ones included, before the first use of the current four */                        \
            { const int pn_ = more ? p0 + 4 * rows : p0; OUTC_LOAD4(un, pn_, RAGGED_) }                  \
            _Pragma("unroll") for (int j = 0; j < 4; j++) {                                              \
                const int p = p0 + j * rows + row;                                                      \
                if ((RAGGED_) && p0 + j * rows >= p_end) break;                   /* block-uniform */   \
                const bool in_ = !(RAGGED_) || p < p_end;                                               \
                float f[EPU], acc[NC];                                                                  \
                _Pragma("unroll") for (int k = 0; k < NC; k++) acc[k] = 0.f;                             \
                Unit<T>::unpack(u[j], f);                                                               \
                _Pragma("unroll") for (int i = 0; i < EPU; i++) {                                        \
                    const float a = to_f(from_f<T>(fmaxf(fmaf(f[i], sc[i], sh[i]), 0.f)));              \
                    _Pragma("unroll") for (int k = 0; k < NC; k++) acc[k] = fmaf(a, wk[k][i], acc[k]);   \
                }                                                                                       \
                if constexpr (CUC != 0) {                                                               \
                    _Pragma("unroll") for (int k = 0; k < NC; k++) acc[k] = first_lane_sum<CUC ? CUC : 8>(acc[k]); \
                } else {                                                                                \
                    _Pragma("unroll") for (int k = 0; k < NC; k++)                                       \
                        for (int off = CU >> 1; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off);  \
                }                                                                                       \
                if (cu == 0 && in_) {                                                                   \
                    int b, q; dhw.divmod(p, b, q);                                                      \
                    _Pragma("unroll") for (int k = 0; k < NC; k++) if (k < ncls) logits[((size_t)b * ncls + k) * hw + q] = acc[k] + bk[k]; \
                }                                                                                       \
            }                                                                                           \
            _Pragma("unroll") for (int j = 0; j < 4; j++) u[j] = un[j];                                  \
        }                                                                                               \
    }
    if (p_begin >= p_end) return;
    if (p_begin + rows * OUTC_ITERS <= npix) OUTC_RUN(false) else OUTC_RUN(true)
#undef OUTC_RUN
#undef OUTC_LOAD4
}

extern "C" int bdn_outc_fwd(int dtype, const void* z, const float* bn, const float* w, const float* b,
                            float* logits, int B, int H, int W, int C, int ncls, void* stream) {
    if (!z || !bn || !w || !b || !logits) BDN_FAIL(BDN_E_ARG, "outc_fwd: null pointer");
    if (ncls < 1 || ncls > OUTC_MAXCLS || C % 16 || C > 512 || 512 % C) BDN_FAIL(BDN_E_SHAPE, "outc_fwd: ncls=%d (max %d), C=%d", ncls, OUTC_MAXCLS, C);
    hipStream_t st = (hipStream_t)stream; const int npix = B * H * W, hw = H * W;
    if (dtype == BDN_BF16) {
        const int per = 256 / (C / 8) * OUTC_ITERS; const unsigned grid = (npix + per - 1) / per;
        if (ncls <= 2 && C == 64) hipLaunchKernelGGL((outc_fwd_kernel<bf16s, 2, 8>), dim3(grid), dim3(256), 0, st, (const bf16s*)z, bn, w, b, logits, npix, hw, C, ncls, FastDiv(hw));
        else if (ncls <= 2) hipLaunchKernelGGL((outc_fwd_kernel<bf16s, 2>), dim3(grid), dim3(256), 0, st, (const bf16s*)z, bn, w, b, logits, npix, hw, C, ncls, FastDiv(hw));
        else hipLaunchKernelGGL((outc_fwd_kernel<bf16s, OUTC_MAXCLS>), dim3(grid), dim3(256), 0, st, (const bf16s*)z, bn, w, b, logits, npix, hw, C, ncls, FastDiv(hw));
    } else if (dtype == BDN_F32) {
        const int per = 256 / (C / 4) * OUTC_ITERS; const unsigned grid = (npix + per - 1) / per;
        if (ncls <= 2 && C == 64) hipLaunchKernelGGL((outc_fwd_kernel<float, 2, 16>), dim3(grid), dim3(256), 0, st, (const float*)z, bn, w, b, logits, npix, hw, C, ncls, FastDiv(hw));
        else if (ncls <= 2) hipLaunchKernelGGL((outc_fwd_kernel<float, 2>), dim3(grid), dim3(256), 0, st, (const float*)z, bn, w, b, logits, npix, hw, C, ncls, FastDiv(hw));
        else hipLaunchKernelGGL((outc_fwd_kernel<float, OUTC_MAXCLS>), dim3(grid), dim3(256), 0, st, (const float*)z, bn, w, b, logits, npix, hw, C, ncls, FastDiv(hw));
    }
    else BDN_FAIL(BDN_E_ARG, "outc_fwd: bad dtype");
    BDN_CHECK_LAUNCH("outc_fwd");
    return BDN_OK;
}

// backward: dA[p][c] = sum_k dl[k][p] w[k][c];  dw[k][c] = sum_p dl[k][p] a[p][c];  db[k] = sum_p dl[k][p]
// Thread t owns channel unit t % CU (its filter taps, BN constants and dw accumulators live in registers)
// and walks pixels t / CU, +rows, ...; the block partials of dw/db go to a workspace and are summed in a fixed order by
// outc_dw_reduce_kernel (the first version added them with float atomics: the only non-deterministic bits of a step).
template <typename T, int NC, int CUC = 0>          // CUC: lanes per pixel at compile time (see outc_fwd_kernel), 0 = run-time
__global__ __launch_bounds__(256) void outc_bwd_kernel(const float* __restrict__ dl, const T* __restrict__ z, const float* __restrict__ bn,
                                const float* __restrict__ w, T* __restrict__ dA, float* __restrict__ wpart,
                                float* __restrict__ bs_partial, int npix, int hw, int C, int ncls, FastDiv dhw) {
    constexpr int EPU = ET<T>::EPU;
    extern __shared__ float sm[];                             // [ncls][C+1] block sums + [256][EPU][2] reduction scratch
    const int CU = CUC ? CUC : C / EPU, rows = 256 / CU, tid = threadIdx.x, cu = tid % CU, row = tid / CU, c = cu * EPU;
    for (int i = tid; i < ncls * (C + 1); i += 256) sm[i] = 0.f;
    float sc[EPU], sh[EPU], wk[NC][EPU], acc[NC][EPU], accb[NC], t0[EPU], t1[EPU];
#pragma unroll
    for (int i = 0; i < EPU; i++) { sc[i] = bn_row(bn, 0, 2, C)[c + i]; sh[i] = bn_row(bn, 0, 3, C)[c + i]; t0[i] = 0.f; t1[i] = 0.f; }
    const bool bs = bs_partial != nullptr;
#pragma unroll
    for (int k = 0; k < NC; k++) {
        accb[k] = 0.f;
        const int kk = k < ncls ? k : 0;
#pragma unroll
        for (int i = 0; i < EPU; i++) { const float wv = w[kk * C + c + i]; wk[k][i] = k < ncls ? wv : 0.f; acc[k][i] = 0.f; }
    }
    __syncthreads();
    const int p_begin = blockIdx.x * rows * OUTC_BWD_ITERS, p_end = min(npix, p_begin + rows * OUTC_BWD_ITERS);
    // four pixels of a lane are requested before the first is used, and the following four before these are consumed (unconditionally:
    // the last trip re-requests its own -- a branch around the requests makes the compiler drain them all before the first use); the
    // pixels are still accumulated one after the other, in order
    uint4 zu[4], zn[4]; float gv[4][NC], gn[4][NC];
#define OUTC_BLOAD4(zd_, gd_, p0_)                                                                       \
    _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                      \
        const int pj_ = (p0_) + j * rows + row < p_end ? (p0_) + j * rows + row : p_begin + row;          \
        int b_, q_; dhw.divmod(pj_, b_, q_);                                                            \
        _Pragma("unroll") for (int k = 0; k < NC; k++) gd_[j][k] = dl[((size_t)b_ * ncls + (k < ncls ? k : 0)) * hw + q_]; \
        zd_[j] = *reinterpret_cast<const uint4*>(z + (size_t)pj_ * C + c);                              \
    }
    if (p_begin + row < p_end) {                              // (a block's first row of pixels is inside the tensor whenever the block has work)
        OUTC_BLOAD4(zu, gv, p_begin)
        for (int p0 = p_begin; p0 < p_end; p0 += 4 * rows) {  // block-uniform trip count
            { const int pn_ = p0 + 4 * rows < p_end ? p0 + 4 * rows : p0; OUTC_BLOAD4(zn, gn, pn_) }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int p = p0 + j * rows + row;
                if (p < p_end) {
                    float g[NC], f[EPU], o[EPU];
#pragma unroll
                    for (int k = 0; k < NC; k++) g[k] = k < ncls ? gv[j][k] : 0.f;
                    Unit<T>::unpack(zu[j], f);
#pragma unroll
                    for (int i = 0; i < EPU; i++) {
                        const float a = to_f(from_f<T>(fmaxf(fmaf(f[i], sc[i], sh[i]), 0.f)));
                        float s = 0.f;
#pragma unroll
                        for (int k = 0; k < NC; k++) if (k < ncls) { s = fmaf(g[k], wk[k][i], s); acc[k][i] = fmaf(g[k], a, acc[k][i]); }
                        o[i] = s;
                    }
                    if (cu == 0) {
#pragma unroll
                        for (int k = 0; k < NC; k++) accb[k] += g[k];
                    }
                    const uint4 uo = Unit<T>::pack(o);
                    if (dA) *reinterpret_cast<uint4*>(dA + (size_t)p * C + c) = uo;
                    if (bs) {                                             // BatchNorm-backward partial sums of this layer on the stored gradient
                        Unit<T>::unpack(uo, o);
#pragma unroll
                        for (int i = 0; i < EPU; i++) {
                            const float gm = fmaf(f[i], sc[i], sh[i]) > 0.f ? o[i] : 0.f;
                            t0[i] += gm; t1[i] = fmaf(gm, f[i], t1[i]);
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                zu[j] = zn[j];
#pragma unroll
                for (int k = 0; k < NC; k++) gv[j][k] = gn[j][k];
            }
        }
    }
#undef OUTC_BLOAD4
    // block sums of dw / db in a fixed order: per class every thread parks its partials in LDS, then one thread per
    // channel adds the block's `rows` pixel rows in order (LDS atomics would make the last bits depend on wave timing)
    {
        float* red = sm + ncls * (C + 1);                     // [256][EPU] dw partials + [rows] db partials
#pragma unroll
        for (int k = 0; k < NC; k++) {
            if (k < ncls) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < EPU; i++) red[tid * EPU + i] = acc[k][i];
                if (cu == 0) red[256 * EPU + row] = accb[k];
                __syncthreads();
                for (int o = tid; o <= C; o += 256) {
                    float v = 0.f;
                    if (o < C) { const int ccu = o / EPU, i = o % EPU; for (int r = 0; r < rows; r++) v += red[(r * CU + ccu) * EPU + i]; }
                    else for (int r = 0; r < rows; r++) v += red[256 * EPU + r];
                    sm[k * (C + 1) + o] = v;
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < ncls * (C + 1); i += 256)        // block partial [ncls][C+1]; outc_dw_reduce_kernel sums the blocks in order
        wpart[(size_t)blockIdx.x * ncls * (C + 1) + i] = sm[i];
    if (bs) {                                                 // bs_partial[block][2][C], fixed order
        float* sred = sm + ncls * (C + 1);
#pragma unroll
        for (int i = 0; i < EPU; i++) { sred[(tid * EPU + i) * 2] = t0[i]; sred[(tid * EPU + i) * 2 + 1] = t1[i]; }
        __syncthreads();
        for (int o = tid; o < C * 2; o += 256) {
            const int k = o & 1, cc = o >> 1, ccu = cc / EPU, i = cc % EPU;
            float v = 0.f;
            for (int r = 0; r < rows; r++) v += sred[((r * CU + ccu) * EPU + i) * 2 + k];
            bs_partial[((size_t)blockIdx.x * 2 + k) * C + cc] = v;
        }
    }
}

// dw[k][c] / db[k] = sum over the blocks' partials, fixed order: one block per output value, 256 lanes stride the rows,
// then an LDS tree.
__global__ __launch_bounds__(256) void outc_dw_reduce_kernel(const float* __restrict__ wpart, int rows, int C, int ncls,
                                                             float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float sm[256];
    const int o = blockIdx.x, n = ncls * (C + 1);
    float a = 0.f;
    for (int r = threadIdx.x; r < rows; r += 256) a += wpart[(size_t)r * n + o];
    sm[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int k = o / (C + 1), cc = o % (C + 1);
        if (cc < C) dw[k * C + cc] = sm[0]; else db[k] = sm[0];
    }
}

extern "C" int bdn_outc_bwd_rows(int dtype, int B, int H, int W, int C) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 16 || C > 1024 || 1024 % C) return 0;
    const int per = 256 / (C / (dtype == BDN_BF16 ? 8 : 4)) * OUTC_BWD_ITERS;
    return (B * H * W + per - 1) / per;
}

extern "C" size_t bdn_outc_bwd_workspace_bytes(int dtype, int B, int H, int W, int C, int ncls) {
    const int rows = bdn_outc_bwd_rows(dtype, B, H, W, C);
    if (rows <= 0 || ncls < 1 || ncls > OUTC_MAXCLS) return 0;
    return (size_t)rows * ncls * (C + 1) * sizeof(float);
}

extern "C" int bdn_outc_bwd(int dtype, const float* dlogits, const void* z, const float* bn, const float* w,
                            void* dA, float* dw, float* db, float* bs_partial, float* ws, int B, int H, int W, int C, int ncls, void* stream) {
    if (!dlogits || !z || !bn || !w || !dw || !db || !ws) BDN_FAIL(BDN_E_ARG, "outc_bwd: null pointer");
    if (!dA && !bs_partial) BDN_FAIL(BDN_E_ARG, "outc_bwd: dA may be omitted only together with bs_partial (bdn_outc_bn_bwd_apply recomputes it)");
    if (ncls < 1 || ncls > OUTC_MAXCLS || C % 16 || C > 1024 || 1024 % C) BDN_FAIL(BDN_E_SHAPE, "outc_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream; const int npix = B * H * W;
    const unsigned grid = bdn_outc_bwd_rows(dtype, B, H, W, C);
    const size_t smem = sizeof(float) * (ncls * (C + 1) + 256 * (dtype == BDN_BF16 ? 8 : 4) * 2);
    if (dtype == BDN_BF16) {
        if (ncls <= 2 && C == 64) hipLaunchKernelGGL((outc_bwd_kernel<bf16s, 2, 8>), dim3(grid), dim3(256), smem, st, dlogits, (const bf16s*)z, bn, w, (bf16s*)dA, ws, bs_partial, npix, H * W, C, ncls, FastDiv(H * W));
        else if (ncls <= 2) hipLaunchKernelGGL((outc_bwd_kernel<bf16s, 2>), dim3(grid), dim3(256), smem, st, dlogits, (const bf16s*)z, bn, w, (bf16s*)dA, ws, bs_partial, npix, H * W, C, ncls, FastDiv(H * W));
        else hipLaunchKernelGGL((outc_bwd_kernel<bf16s, OUTC_MAXCLS>), dim3(grid), dim3(256), smem, st, dlogits, (const bf16s*)z, bn, w, (bf16s*)dA, ws, bs_partial, npix, H * W, C, ncls, FastDiv(H * W));
    } else if (dtype == BDN_F32) {
        if (ncls <= 2 && C == 64) hipLaunchKernelGGL((outc_bwd_kernel<float, 2, 16>), dim3(grid), dim3(256), smem, st, dlogits, (const float*)z, bn, w, (float*)dA, ws, bs_partial, npix, H * W, C, ncls, FastDiv(H * W));
        else if (ncls <= 2) hipLaunchKernelGGL((outc_bwd_kernel<float, 2>), dim3(grid), dim3(256), smem, st, dlogits, (const float*)z, bn, w, (float*)dA, ws, bs_partial, npix, H * W, C, ncls, FastDiv(H * W));
        else hipLaunchKernelGGL((outc_bwd_kernel<float, OUTC_MAXCLS>), dim3(grid), dim3(256), smem, st, dlogits, (const float*)z, bn, w, (float*)dA, ws, bs_partial, npix, H * W, C, ncls, FastDiv(H * W));
    } else BDN_FAIL(BDN_E_ARG, "outc_bwd: bad dtype");
    BDN_CHECK_LAUNCH("outc_bwd");
    hipLaunchKernelGGL(outc_dw_reduce_kernel, dim3(ncls * (C + 1)), dim3(256), 0, st, ws, (int)grid, C, ncls, dw, db);
    BDN_CHECK_LAUNCH("outc_dw_reduce");
    return BDN_OK;
}

// BatchNorm+ReLU backward of the layer in front of the classifier, with the classifier's data gradient RECOMPUTED from
// dlogits (ncls values per pixel) instead of read back: dA = round_T(sum_k dl[k] w[k][c]) exactly as outc_bwd forms and
// rounds it, then bn_bwd_apply's expression.  outc_bwd then need not store dA at all: 2 x B*H*W*C elements of HBM traffic
// less, at a point of the step where nothing else runs.
template <typename T, int NC, bool SPLIT = false>
__global__ __launch_bounds__(256) void outc_bn_bwd_apply_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                                const T* __restrict__ z, const float* __restrict__ bn, const float* __restrict__ sums,
                                T* __restrict__ dz, int npix, int hw, int pix_per_group, int pix_per_block, int C, int ncls, FastDiv dhw, FastDiv dppg) {
    constexpr int EPU = ET<T>::EPU;
    const int CU = C / EPU, rows = 256 / CU, tid = threadIdx.x, cu = tid % CU, row = tid / CU, c = cu * EPU;
    const int p_begin = blockIdx.x * pix_per_block, p_end = min(npix, p_begin + pix_per_block);
    if (p_begin >= p_end) return;
    const float invM = 1.f / (float)pix_per_group;
    int gcur = -1;
    float mean[EPU], inv[EPU], sc[EPU], sh[EPU], k0[EPU], k1[EPU], wk[NC][EPU];
#pragma unroll
    for (int k = 0; k < NC; k++)
#pragma unroll
        for (int i = 0; i < EPU; i++) wk[k][i] = k < ncls ? w[k * C + c + i] : 0.f;
    for (int pb = p_begin + row; pb < p_end; pb += 4 * rows) {
        // four pixels of a lane are requested before the first is used
        uint4 zu[4]; float gv[4][NC];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int pj = pb + j * rows < p_end ? pb + j * rows : pb;
            int b, q; dhw.divmod(pj, b, q);
#pragma unroll
            for (int k = 0; k < NC; k++) gv[j][k] = k < ncls ? dl[((size_t)b * ncls + k) * hw + q] : 0.f;
            zu[j] = *reinterpret_cast<const uint4*>(z + (size_t)pj * C + c);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int p = pb + j * rows;
            if (p >= p_end) break;
            const int g = dppg.div(p);
            if (g != gcur) {                                   // (blocks never straddle groups in practice; correct if they do)
                gcur = g;
#pragma unroll
                for (int i = 0; i < EPU; i++) {
                    mean[i] = bn_row(bn, g, 0, C)[c + i]; inv[i] = bn_row(bn, g, 1, C)[c + i];
                    sc[i] = bn_row(bn, g, 2, C)[c + i]; sh[i] = bn_row(bn, g, 3, C)[c + i];
                    k0[i] = sums[((size_t)g * 2 + 0) * C + c + i] * invM;
                    k1[i] = sums[((size_t)g * 2 + 1) * C + c + i] * invM;
                }
            }
            float fz[EPU], o[EPU];
            Unit<T>::unpack(zu[j], fz);
#pragma unroll
            for (int i = 0; i < EPU; i++) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < NC; k++) if (k < ncls) s = fmaf(gv[j][k], wk[k][i], s);
                o[i] = s;
            }
            Unit<T>::unpack(Unit<T>::pack(o), o);              // the gradient as outc_bwd would have stored it
#pragma unroll
            for (int i = 0; i < EPU; i++) {
                const float gm = fmaf(fz[i], sc[i], sh[i]) > 0.f ? o[i] : 0.f;
                const float xhat = (fz[i] - mean[i]) * inv[i];
                o[i] = sc[i] * (gm - k0[i] - xhat * k1[i]);
            }
            if constexpr (SPLIT) {                             // bf16x3: dz leaves as the [hi | lo] operand of its two consumers ([pixel][2C])
                const SplitOut so = {reinterpret_cast<bf16s*>(dz), 2 * C, 0, C};
                store_split4(so, (size_t)p, c, o);
            } else
            *reinterpret_cast<uint4*>(dz + (size_t)p * C + c) = Unit<T>::pack(o);
        }
    }
}

extern "C" int bdn_outc_bn_bwd_apply(int dtype, const float* dlogits, const float* w, const void* z, const float* bn,
                                     int imgs_per_group, const float* sums, void* dz, int B, int H, int W, int C, int ncls, void* stream) {
    if (!dlogits || !w || !z || !bn || !sums || !dz) BDN_FAIL(BDN_E_ARG, "outc_bn_bwd_apply: null pointer");
    if (ncls < 1 || ncls > OUTC_MAXCLS || C % 16 || C > 1024 || 1024 % C || B <= 0 || H <= 0 || W <= 0 || imgs_per_group <= 0 || B % imgs_per_group)
        BDN_FAIL(BDN_E_SHAPE, "outc_bn_bwd_apply: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int npix = B * H * W, epu = dtype == BDN_BF16 ? 8 : 4, rows = 256 / (C / epu);
    int ppb = (npix + 2047) / 2048; if (ppb < 2 * rows) ppb = 2 * rows; ppb = (ppb + rows - 1) / rows * rows;
    const unsigned grid = (npix + ppb - 1) / ppb;
    if (dtype == BDN_BF16X3) {                 // float32 z, dz stored as the split operand
#define OUTC_APPLY_S(NC_) hipLaunchKernelGGL((outc_bn_bwd_apply_kernel<float, NC_, true>), dim3(grid), dim3(256), 0, st, dlogits, w, (const float*)z, bn, sums, \
                                               (float*)dz, npix, H * W, imgs_per_group * H * W, ppb, C, ncls, FastDiv(H * W), FastDiv(imgs_per_group * H * W))
        if (ncls <= 2) OUTC_APPLY_S(2); else OUTC_APPLY_S(OUTC_MAXCLS);
#undef OUTC_APPLY_S
        BDN_CHECK_LAUNCH("outc_bn_bwd_apply");
        return BDN_OK;
    }
#define OUTC_APPLY(T_, NC_) hipLaunchKernelGGL((outc_bn_bwd_apply_kernel<T_, NC_>), dim3(grid), dim3(256), 0, st, dlogits, w, (const T_*)z, bn, sums, \
                                               (T_*)dz, npix, H * W, imgs_per_group * H * W, ppb, C, ncls, FastDiv(H * W), FastDiv(imgs_per_group * H * W))
    if (dtype == BDN_BF16) { if (ncls <= 2) OUTC_APPLY(bf16s, 2); else OUTC_APPLY(bf16s, OUTC_MAXCLS); }
    else if (dtype == BDN_F32) { if (ncls <= 2) OUTC_APPLY(float, 2); else OUTC_APPLY(float, OUTC_MAXCLS); }
    else BDN_FAIL(BDN_E_ARG, "outc_bn_bwd_apply: bad dtype");
#undef OUTC_APPLY
    BDN_CHECK_LAUNCH("outc_bn_bwd_apply");
    return BDN_OK;
}

// ============================================================ Tversky loss (utils/metrics.py:130-171, dims == (0,2))
// sums[k][c][w], k = 0 TP, 1 FP, 2 FN, reduced over batch and H for every (class, column w).
// pass 1: grid (column blocks x row blocks) -> per-block partial sums;  pass 2: single block adds the blocks in a fixed
// order (no float atomics: the loss and dlogits are the same bits every run), then loss + coefficient tables;  pass 3: dlogits.
template <int NC>
__global__ void tversky_sums_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ labels,
                                    float* __restrict__ part, int32_t* __restrict__ pcounts, int B, int ncls, int H, int W,
                                    int rows_per_block, int We, FastDiv dH) {
    // block = 256 threads = RL row lanes x CW columns (CW = min(W rounded up to a power of two, 256));
    // grid.x = column blocks, grid.y = row blocks
    extern __shared__ float sm[];                         // [RL][3*NC][CW]
    const int CW = blockDim.y, RL = blockDim.x;           // launch: dim3(RL, CW) with x = row lane (slow), see host
    const int cl = threadIdx.y, rl = threadIdx.x;
    const int x = blockIdx.x * CW + cl;
    const size_t hw = (size_t)H * W;
    float tp[NC], fp[NC], fn[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) { tp[k] = 0.f; fp[k] = 0.f; fn[k] = 0.f; }
    int c_tp = 0, c_fp = 0, c_fn = 0, c_ok = 0;
    const int rows = B * H, r_end = min(rows, (int)(blockIdx.y + 1) * rows_per_block);
    if (x < W)
        // four rows of a lane are requested before the first is used (a lane walks 16 rows at B = 64: one dependent HBM round trip
        // per row made this pass 17.6 us for 9 MB); the rows are still ACCUMULATED one after the other, in the same order
        for (int r0 = blockIdx.y * rows_per_block + rl; r0 < r_end; r0 += 4 * RL) {
            float lv[4][NC]; int tv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = r0 + u * RL;
                const int rr = r < r_end ? r : r0;
                int b, y; dH.divmod(rr, b, y);
                const size_t q = (size_t)y * W + x;
#pragma unroll
                for (int k = 0; k < NC; k++) lv[u][k] = k < ncls ? logits[((size_t)b * ncls + k) * hw + q] : -INFINITY;
                tv[u] = labels[(size_t)b * hw + q];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (r0 + u * RL >= r_end) break;
                float l[NC]; float m = -INFINITY; int am = 0;
#pragma unroll
                for (int k = 0; k < NC; k++) { l[k] = lv[u][k]; if (l[k] > m) { m = l[k]; am = k; } }
                float den = 0.f;
#pragma unroll
                for (int k = 0; k < NC; k++) { l[k] = k < ncls ? expf(l[k] - m) : 0.f; den += l[k]; }
                const int t = tv[u];
                const float inv = 1.f / den;
#pragma unroll
                for (int k = 0; k < NC; k++) {
                    const float p = l[k] * inv;
                    if (t == k) { tp[k] += p; fn[k] += 1.f - p; } else fp[k] += p;
                }
                c_tp += (am == 1 && t == 1); c_fp += (am == 1 && t != 1); c_fn += (am != 1 && t == 1); c_ok += (am == t);
            }
        }
#pragma unroll
    for (int k = 0; k < NC; k++) {
        sm[(rl * 3 * NC + 0 * NC + k) * CW + cl] = tp[k];
        sm[(rl * 3 * NC + 1 * NC + k) * CW + cl] = fp[k];
        sm[(rl * 3 * NC + 2 * NC + k) * CW + cl] = fn[k];
    }
    __syncthreads();
    // block partials, no atomics: part[row block][cell] (cells [3][ncls][W]) or part[block][3*NC] when the columns are
    // reduced too; tversky_finish_kernel adds the blocks in a fixed order.  pcounts[block][4] likewise.
    const int nblk_lin = blockIdx.y * gridDim.x + blockIdx.x;
    if (We == W) {
        if (rl == 0 && x < W)
            for (int k = 0; k < ncls; k++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    float v = 0.f;
                    for (int r = 0; r < RL; r++) v += sm[(r * 3 * NC + j * NC + k) * CW + cl];
                    part[(size_t)blockIdx.y * 3 * ncls * W + (j * ncls + k) * W + x] = v;
                }
    } else {
        // [B,1,H,W] labels: the reference reduces over the columns too (dims == (0,2,3))
        const int tid = rl + RL * cl;
        if (tid < 3 * NC) {
            float v = 0.f;
            for (int i = 0; i < RL * CW; i++) {
                const int r = i / CW, c = i % CW;
                if (blockIdx.x * CW + c < W) v += sm[(r * 3 * NC + tid) * CW + c];
            }
            const int j = tid / NC, k = tid % NC;
            if (k < ncls) part[(size_t)nblk_lin * 3 * ncls + j * ncls + k] = v;
        }
    }
    {
        int* ism = reinterpret_cast<int*>(sm);
        __syncthreads();
        const int tid = rl * CW + cl;
        ism[tid * 4 + 0] = c_tp; ism[tid * 4 + 1] = c_fp; ism[tid * 4 + 2] = c_fn; ism[tid * 4 + 3] = c_ok;
        __syncthreads();
        if (tid < 4) { int v = 0; for (int i = 0; i < 256; i++) v += ism[i * 4 + tid]; pcounts[nblk_lin * 4 + tid] = v; }
    }
}

// sums[cell] = sum over the nblk block partials (cell-major rows of `part`), fixed order: thread = (float4 of cells or one
// cell, block lane); then loss = 1 - mean_{c,w} TP/(TP + a FP + b FN + eps).  Overwrites sums[0] with 1/D and sums[1] with TP/D^2.
__global__ __launch_bounds__(1024) void tversky_finish_kernel(float* __restrict__ sums, const float* __restrict__ part, int nblk,
                                      const int32_t* __restrict__ pcounts, int ncblk, int32_t* __restrict__ counts,
                                      float alpha, float beta, float eps, int ncls, int W, float* __restrict__ loss) {   // W = effective width (1 when the columns are reduced too)
    __shared__ double red[256];
    __shared__ float4 lane_sums[1024];
    const int n = 3 * ncls * W, tid = threadIdx.x;
    if (n % 4 == 0 && n / 4 <= 1024) {
        const int n4 = n / 4, LN = 1024 / n4 > 0 ? (1024 / n4 > 16 ? 16 : 1024 / n4) : 1;       // block lanes per float4 of cells
        const int q = tid % n4, l = tid / n4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (l < LN) {
#pragma unroll 8
            for (int b = l; b < nblk; b += LN) {
                const float4 v = *reinterpret_cast<const float4*>(part + (size_t)b * n + 4 * q);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            lane_sums[tid] = a;
        }
        __syncthreads();
        if (l == 0) {
            for (int k = 1; k < LN; k++) { const float4 v = lane_sums[k * n4 + q]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
            *reinterpret_cast<float4*>(sums + 4 * q) = a;
        }
    } else {
        for (int i = tid; i < n; i += 1024) {
            float a = 0.f;
            for (int b = 0; b < nblk; b++) a += part[(size_t)b * n + i];
            sums[i] = a;
        }
    }
    if (counts) {                                          // TP / FP / FN / correct counts: 256 block lanes x 4 counters, LDS tree (integers: any order)
        int* ired = reinterpret_cast<int*>(lane_sums);
        __syncthreads();                                   // lane_sums is free again
        const int j = tid & 3, l = tid >> 2;
        int v = 0;
        for (int b = l; b < ncblk; b += 256) v += pcounts[b * 4 + j];
        ired[tid] = v;
        __syncthreads();
        for (int s2 = 512; s2 >= 4; s2 >>= 1) { if (tid < s2) ired[tid] += ired[tid + s2]; __syncthreads(); }
        if (tid < 4) counts[tid] = ired[tid];
    }
    __syncthreads();
    double acc = 0.0;
    const int nc = ncls * W;
    if (tid < 256)
        for (int i = tid; i < nc; i += 256) {
            const float tp = sums[i], fp = sums[nc + i], fn = sums[2 * nc + i];
            const float D = tp + alpha * fp + beta * fn + eps;
            acc += (double)(tp / D);
            sums[i] = 1.f / D; sums[nc + i] = tp / (D * D);
        }
    if (tid < 256) red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    if (tid == 0) *loss = (float)(1.0 - red[0] / nc);
}

__global__ void tversky_bwd_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ labels,
                                   const float* __restrict__ coef, float alpha, float beta, float* __restrict__ dlogits,
                                   int B, int ncls, int H, int Wimg, int W, FastDiv dhw, FastDiv dWimg) {
    const size_t hw = (size_t)H * Wimg, npix = (size_t)B * hw;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    int bi, qi, yi, xi; dhw.divmod((int)p, bi, qi); dWimg.divmod(qi, yi, xi);      // (the entry point keeps B*H*W below 2^31)
    const size_t b = bi, q = qi; const int x = W == 1 ? 0 : xi;
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

struct OverlapPlan { int We, CW, RL, rpb, gx, gy, nblk, n; };
static OverlapPlan overlap_plan(int B, int ncls, int H, int W, int reduce_w) {
    OverlapPlan p;
    p.We = reduce_w ? 1 : W;
    p.CW = 1; while (p.CW < W && p.CW < 256) p.CW *= 2;
    p.RL = 256 / p.CW;
    const int rows = B * H;
    p.rpb = (rows + 255) / 256; if (p.rpb < p.RL) p.rpb = p.RL;                 // ~256 row blocks
    p.gx = (W + p.CW - 1) / p.CW; p.gy = (rows + p.rpb - 1) / p.rpb;
    p.nblk = reduce_w ? p.gx * p.gy : p.gy;                                     // partial rows the finish kernel adds up
    p.n = 3 * ncls * p.We;
    return p;
}
extern "C" size_t bdn_overlap_workspace_bytes(int B, int ncls, int H, int W, int reduce_w) {
    if (B <= 0 || H <= 0 || W <= 0 || ncls < 2 || ncls > OUTC_MAXCLS) return 0;
    const OverlapPlan p = overlap_plan(B, ncls, H, W, reduce_w);
    return sizeof(float) * ((size_t)p.n * (p.nblk + 1) + 8) + sizeof(int32_t) * 4 * p.gx * p.gy;
}

extern "C" int bdn_overlap_loss(const float* logits, const uint8_t* labels, float alpha, float beta, float eps,
                                int reduce_w, float* ws, float* loss, int32_t* counts, float* dlogits,
                                int B, int ncls, int H, int W, void* stream) {
    if (!logits || !labels || !ws || !loss) BDN_FAIL(BDN_E_ARG, "overlap_loss: null pointer");
    if (ncls < 2 || ncls > OUTC_MAXCLS) BDN_FAIL(BDN_E_SHAPE, "overlap_loss: ncls=%d unsupported (2..%d)", ncls, OUTC_MAXCLS);
    if (B <= 0 || H <= 0 || W <= 0 || (size_t)B * H * W >= ((size_t)1 << 31)) BDN_FAIL(BDN_E_SHAPE, "overlap_loss: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const OverlapPlan p = overlap_plan(B, ncls, H, W, reduce_w);
    const int We = p.We;
    float* part = ws + p.n;                                                    // [nblk][n] block partials behind the n final sums
    int32_t* pcounts = reinterpret_cast<int32_t*>(part + (size_t)p.nblk * p.n);  // [gx*gy][4]
    dim3 grid(p.gx, p.gy), block(p.RL, p.CW);
    if (ncls <= 2) hipLaunchKernelGGL(tversky_sums_kernel<2>, grid, block, sizeof(float) * 256 * 3 * 2, st, logits, labels, part, pcounts, B, ncls, H, W, p.rpb, We, FastDiv(H));
    else hipLaunchKernelGGL(tversky_sums_kernel<OUTC_MAXCLS>, grid, block, sizeof(float) * 256 * 3 * OUTC_MAXCLS, st, logits, labels, part, pcounts, B, ncls, H, W, p.rpb, We, FastDiv(H));
    BDN_CHECK_LAUNCH("tversky_sums");
    hipLaunchKernelGGL(tversky_finish_kernel, dim3(1), dim3(1024), 0, st, ws, part, p.nblk, pcounts, p.gx * p.gy, counts, alpha, beta, eps, ncls, We, loss);
    BDN_CHECK_LAUNCH("tversky_finish");
    if (dlogits) {
        hipLaunchKernelGGL(tversky_bwd_kernel, dim3(grid_for((size_t)B * H * W)), dim3(256), 0, st, logits, labels, ws, alpha, beta, dlogits, B, ncls, H, W, We, FastDiv(H * W), FastDiv(W));
        BDN_CHECK_LAUNCH("tversky_bwd");
    }
    return BDN_OK;
}

extern "C" int bdn_tversky(const float* logits, const uint8_t* labels, float alpha, float beta, float eps,
                           float* ws, float* loss, int32_t* counts, float* dlogits,
                           int B, int ncls, int H, int W, void* stream) {
    return bdn_overlap_loss(logits, labels, alpha, beta, eps, 0, ws, loss, counts, dlogits, B, ncls, H, W, stream);
}

// ============================================================ Focal loss (utils/metrics.py:8-48)
// loss_i = -(1 - pt)^gamma * a[t] * log pt with pt = softmax(l)[t]; the modulating factor is built from
// `logpt.data.exp()` (:35) and is therefore a constant for the gradient:
//   d loss_i / d l_k = -(1 - pt)^gamma * a[t] * ([k == t] - p_k)   (times 1/N when size_average).
// pass 1: per-pixel loss + dlogits, per-block partial sums (double) -> ws;  pass 2: fixed-order finish.
__global__ void focal_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ labels,
                             const float* __restrict__ alpha, float gamma, float gscale,
                             double* __restrict__ part, int32_t* __restrict__ counts, float* __restrict__ dlogits,
                             int B, int ncls, size_t hw) {
    __shared__ double red[256];
    __shared__ int ired[256 * 4];
    const size_t npix = (size_t)B * hw;
    double acc = 0.0;
    int c_tp = 0, c_fp = 0, c_fn = 0, c_ok = 0;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
        const size_t b = p / hw, q = p % hw;
        float l[OUTC_MAXCLS]; float m = -INFINITY; int am = 0;
#pragma unroll
        for (int k = 0; k < OUTC_MAXCLS; k++) if (k < ncls) { l[k] = logits[(b * ncls + k) * hw + q]; if (l[k] > m) { m = l[k]; am = k; } }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < OUTC_MAXCLS; k++) if (k < ncls) den += expf(l[k] - m);
        const int t = labels[p];
        const float lse = m + logf(den);
        float lt = 0.f;
#pragma unroll
        for (int k = 0; k < OUTC_MAXCLS; k++) if (k == t) lt = l[k];
        const float logpt = lt - lse, pt = expf(logpt);
        const float a = alpha ? alpha[t] : 1.f;
        const float mod = gamma == 0.f ? 1.f : powf(fmaxf(1.f - pt, 0.f), gamma);
        acc += (double)(-mod * a * logpt);
        if (dlogits) {
            const float c = -mod * a * gscale;
#pragma unroll
            for (int k = 0; k < OUTC_MAXCLS; k++) if (k < ncls)
                dlogits[(b * ncls + k) * hw + q] = c * ((k == t ? 1.f : 0.f) - expf(l[k] - lse));
        }
        c_tp += (am == 1 && t == 1); c_fp += (am == 1 && t != 1); c_fn += (am != 1 && t == 1); c_ok += (am == t);
    }
    red[threadIdx.x] = acc;
    ired[threadIdx.x * 4 + 0] = c_tp; ired[threadIdx.x * 4 + 1] = c_fp; ired[threadIdx.x * 4 + 2] = c_fn; ired[threadIdx.x * 4 + 3] = c_ok;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
    if (counts && threadIdx.x < 4) { int v = 0; for (int i = 0; i < 256; i++) v += ired[i * 4 + threadIdx.x]; atomicAdd(&counts[threadIdx.x], v); }
}

__global__ void focal_finish_kernel(const double* __restrict__ part, int nblk, double scale, float* __restrict__ loss) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) acc += part[i];
    red[threadIdx.x] = acc; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) *loss = (float)(red[0] * scale);
}

extern "C" size_t bdn_focal_workspace_bytes(void) { return sizeof(double) * 1024; }

extern "C" int bdn_focal(const float* logits, const uint8_t* labels, float gamma, const float* alpha, int size_average,
                         void* ws, float* loss, int32_t* counts, float* dlogits,
                         int B, int ncls, int H, int W, void* stream) {
    if (!logits || !labels || !ws || !loss) BDN_FAIL(BDN_E_ARG, "focal: null pointer");
    if (ncls < 2 || ncls > OUTC_MAXCLS) BDN_FAIL(BDN_E_SHAPE, "focal: ncls=%d unsupported (2..%d)", ncls, OUTC_MAXCLS);
    if (B <= 0 || H <= 0 || W <= 0 || gamma < 0.f) BDN_FAIL(BDN_E_SHAPE, "focal: bad shape or negative gamma");
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)B * H * W;
    int nblk = (int)((npix + 255) / 256); if (nblk > 1024) nblk = 1024;
    if (counts) hipMemsetAsync(counts, 0, sizeof(int32_t) * 4, st);
    const double inv = size_average ? 1.0 / (double)npix : 1.0;
    hipLaunchKernelGGL(focal_kernel, dim3(nblk), dim3(256), 0, st, logits, labels, alpha, gamma, (float)inv,
                       (double*)ws, counts, dlogits, B, ncls, (size_t)H * W);
    BDN_CHECK_LAUNCH("focal");
    hipLaunchKernelGGL(focal_finish_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nblk, inv, loss);
    BDN_CHECK_LAUNCH("focal_finish");
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

// ============================================================ streams of the training step
// The step runs on THREE HIP streams per device -- the dependency chain (high priority), the weight-gradient GEMMs, the host -> device
// copies -- which must sit on three distinct hardware queues: two streams that share a queue serialise, and the step is 8-14 %
// slower (round 2: successive TrainStep instances took whatever the next pool streams of the host framework were, and some of those
// share a queue).  The library creates them itself, once per role, so their placement does not depend on how many streams the host
// program created before.  priority: 0 = normal, 1 = high (mapped onto hipDeviceGetStreamPriorityRange).
extern "C" int bdn_stream_create(int priority, void** stream_out) {
    if (!stream_out) BDN_FAIL(BDN_E_ARG, "stream_create: null pointer");
    if (priority != 0 && priority != 1) BDN_FAIL(BDN_E_ARG, "stream_create: priority must be 0 (normal) or 1 (high)");
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) BDN_FAIL(BDN_E_HIP, "stream_create: hipDeviceGetStreamPriorityRange: %s", hipGetErrorString(e));
    hipStream_t s = nullptr;
    e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority ? greatest : 0);
    if (e != hipSuccess) BDN_FAIL(BDN_E_HIP, "stream_create: hipStreamCreateWithPriority: %s", hipGetErrorString(e));
    *stream_out = reinterpret_cast<void*>(s);
    return BDN_OK;
}

extern "C" int bdn_stream_destroy(void* stream) {
    if (!stream) BDN_FAIL(BDN_E_ARG, "stream_destroy: null pointer");
    const hipError_t e = hipStreamDestroy(reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) BDN_FAIL(BDN_E_HIP, "stream_destroy: %s", hipGetErrorString(e));
    return BDN_OK;
}

// Events for stream-to-stream hand-offs on ONE device (the chain releases a layer's weight-gradient GEMM to the second stream 17 times
// per backward).  Created with hipEventDisableTiming | hipEventDisableSystemFence: the default event performs a SYSTEM-scope release
// when it is recorded -- a cache write-back / invalidate that makes device memory visible to the host and to other devices -- which
// showed as a 6.5 us bubble on the recording stream at every hand-off; a consumer stream on the same device needs none of it.
// NOT for host-side synchronisation (hipEventSynchronize on such an event does not make device writes visible to the host).
extern "C" int bdn_event_create(void** event_out) {
    if (!event_out) BDN_FAIL(BDN_E_ARG, "event_create: null pointer");
    hipEvent_t e = nullptr;
    const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence);
    if (rc != hipSuccess) BDN_FAIL(BDN_E_HIP, "event_create: %s", hipGetErrorString(rc));
    *event_out = reinterpret_cast<void*>(e);
    return BDN_OK;
}
extern "C" int bdn_event_destroy(void* event) {
    if (!event) BDN_FAIL(BDN_E_ARG, "event_destroy: null pointer");
    const hipError_t rc = hipEventDestroy(reinterpret_cast<hipEvent_t>(event));
    if (rc != hipSuccess) BDN_FAIL(BDN_E_HIP, "event_destroy: %s", hipGetErrorString(rc));
    return BDN_OK;
}
// record `event` on `stream`, resp. make `stream` wait for the event's most recent record (both asynchronous)
extern "C" int bdn_event_record(void* event, void* stream) {
    if (!event) BDN_FAIL(BDN_E_ARG, "event_record: null pointer");
    const hipError_t rc = hipEventRecord(reinterpret_cast<hipEvent_t>(event), reinterpret_cast<hipStream_t>(stream));
    if (rc != hipSuccess) BDN_FAIL(BDN_E_HIP, "event_record: %s", hipGetErrorString(rc));
    return BDN_OK;
}
extern "C" int bdn_stream_wait_event(void* stream, void* event) {
    if (!event) BDN_FAIL(BDN_E_ARG, "stream_wait_event: null pointer");
    const hipError_t rc = hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), reinterpret_cast<hipEvent_t>(event), 0);
    if (rc != hipSuccess) BDN_FAIL(BDN_E_HIP, "stream_wait_event: %s", hipGetErrorString(rc));
    return BDN_OK;
}
