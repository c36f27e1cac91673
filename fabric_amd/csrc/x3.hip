// bf16x3: the 1e-3-parity setting at matrix-core speed (north star: "logits within 1e-3 of the reference").
// Tensors stay float32 in HBM (every HBM-bound kernel runs its f32 instantiation); only the GEMM operands change:
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi), and a product keeps three of the four cross terms,
//     a*w ~= a_hi*w_hi + a_lo*w_hi + a_hi*w_lo            (dropped: a_lo*w_lo ~ 2^-16 of the product)
// accumulated in the same f32 MFMA tile.  Nothing new runs on the matrix pipe: the tuned bf16 kernels see a reduction that is
// three times as long (conv3x3: K = [hi | lo | hi] against [w_hi | w_hi | w_lo]) or operands that are twice as wide
// (weight gradient: [dz_hi | dz_lo] x [a_hi | a_lo], three of the four quadrants summed afterwards).  This file holds what
// surrounds them: the operand split (with the cat / BatchNorm+ReLU the f32 kernels would apply on load), the split filter
// images, and the quadrant sum.  Reference arithmetic being matched: float32 nn.Conv2d, models/unet_parts.py:13,16.
#include "common.hpp"

__device__ __forceinline__ void split8(const float* f, uint4& hi, uint4& lo) {
    float r[8];
    hi = Unit<bf16s>::pack(f);
    float h[8];
    Unit<bf16s>::unpack(hi, h);
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = f[i] - h[i];          // exact in float32
    lo = Unit<bf16s>::pack(r);
}

// out[p][c] = hi(a[p][c]), out[p][Ct + c] = lo(a[p][c]),  a = [src0 (relu(bn) optional) | src1],  Ct = C0 + C1
template <bool BN>
__global__ void split_pack_kernel(const float* __restrict__ src0, int C0, const float* __restrict__ src1, int C1,
                                  const float* __restrict__ bn, int ppg, bf16s* __restrict__ out, size_t npix) {
    const int Ct = C0 + C1, U = Ct / 8;
    const size_t total = npix * U;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / U; const int c = (int)(i % U) * 8;
        float f[8];
        if (c < C0) {
            const float4 a = *reinterpret_cast<const float4*>(src0 + p * C0 + c), b = *reinterpret_cast<const float4*>(src0 + p * C0 + c + 4);
            f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
            if (BN) {
                const int g = (int)(p / ppg);
                const float* sc = bn_row(bn, g, 2, C0) + c; const float* sh = bn_row(bn, g, 3, C0) + c;
#pragma unroll
                for (int e = 0; e < 8; e++) f[e] = fmaxf(fmaf(f[e], sc[e], sh[e]), 0.f);      // same expression as bnrelu_unit<float>
            }
        } else {
            const float* q = src1 + p * C1 + (c - C0);
            const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
            f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
        }
        uint4 hi, lo;
        split8(f, hi, lo);
        *reinterpret_cast<uint4*>(out + p * 2 * Ct + c) = hi;
        *reinterpret_cast<uint4*>(out + p * 2 * Ct + Ct + c) = lo;
    }
}

extern "C" int bdn_split_pack(const float* src0, int C0, const float* src1, int C1, int in_mode, const float* in_bn,
                              int imgs_per_group, void* out, int N, int H, int W, void* stream) {
    if (!src0 || !out) BDN_FAIL(BDN_E_ARG, "split_pack: null pointer");
    if (src1 == nullptr) C1 = 0;
    if (N <= 0 || H <= 0 || W <= 0 || C0 <= 0 || C0 % 8 || C1 < 0 || C1 % 8 || (src1 && C1 == 0))
        BDN_FAIL(BDN_E_SHAPE, "split_pack: bad shape N=%d H=%d W=%d C0=%d C1=%d", N, H, W, C0, C1);
    if (in_mode != BDN_IN_PLAIN && in_mode != BDN_IN_BNRELU) BDN_FAIL(BDN_E_ARG, "split_pack: bad in_mode %d", in_mode);
    if (in_mode == BDN_IN_BNRELU && (!in_bn || imgs_per_group <= 0 || N % imgs_per_group))
        BDN_FAIL(BDN_E_ARG, "split_pack: BNRELU input needs in_bn and a valid imgs_per_group");
    const size_t npix = (size_t)N * H * W;
    const size_t total = npix * ((C0 + C1) / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipStream_t st = (hipStream_t)stream;
    if (in_mode == BDN_IN_BNRELU)
        hipLaunchKernelGGL(split_pack_kernel<true>, dim3(grid), dim3(256), 0, st, src0, C0, src1, C1, in_bn, imgs_per_group * H * W, (bf16s*)out, npix);
    else
        hipLaunchKernelGGL(split_pack_kernel<false>, dim3(grid), dim3(256), 0, st, src0, C0, src1, C1, in_bn, 1, (bf16s*)out, npix);
    BDN_CHECK_LAUNCH("split_pack");
    return BDN_OK;
}

// Filter images for the split reduction, in the fragment order of the bf16 kernels (common.hpp: wfrag_index).
//   forward        wf[co][tap][k'], k' in [0, 3 Cinp):  [hi(w) | hi(w) | lo(w)]  against operand channels [a_hi | a_lo | a_hi]
//   data gradient  wd[ci][8 - tap][k'], k' in [0, 3 Cout): the same over the output channels, taps rotated by 180 degrees
__global__ void pack_weights_x3_kernel(const float* __restrict__ w, bf16s* __restrict__ wf, bf16s* __restrict__ wd,
                                       int Cout, int Cin, int Cinp) {
    const size_t total = (size_t)Cout * 9 * Cinp;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ci = i % Cinp; const size_t t = i / Cinp; const int tap = t % 9; const int co = t / 9;
    const float v = ci < Cin ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.f;
    const bf16s hi = (bf16s)f2bf(v);
    const bf16s lo = (bf16s)f2bf(v - bf2f(hi));
    if (wf) {
        wf[wfrag_index<bf16s>(co, tap, ci, 3 * Cinp)] = hi;
        wf[wfrag_index<bf16s>(co, tap, Cinp + ci, 3 * Cinp)] = hi;
        wf[wfrag_index<bf16s>(co, tap, 2 * Cinp + ci, 3 * Cinp)] = lo;
    }
    if (wd) {
        wd[wfrag_index<bf16s>(ci, 8 - tap, co, 3 * Cout)] = hi;
        wd[wfrag_index<bf16s>(ci, 8 - tap, Cout + co, 3 * Cout)] = hi;
        wd[wfrag_index<bf16s>(ci, 8 - tap, 2 * Cout + co, 3 * Cout)] = lo;
    }
}

// all layers in one launch (grid.y = layer): the 18 per-layer launches opened every bf16x3 step with 0.28 ms of dependent tiny kernels
__global__ void pack_weights_x3_multi_kernel(const PackDesc* __restrict__ desc) {
    const PackDesc d = desc[blockIdx.y];
    const size_t total = (size_t)d.Cout * 9 * d.Cinp;
    bf16s* wf = reinterpret_cast<bf16s*>(d.wf); bf16s* wd = reinterpret_cast<bf16s*>(d.wd);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = i % d.Cinp; const size_t t = i / d.Cinp; const int tap = t % 9; const int co = t / 9;
        const float v = ci < d.Cin ? d.w[((size_t)co * d.Cin + ci) * 9 + tap] : 0.f;
        const bf16s hi = (bf16s)f2bf(v);
        const bf16s lo = (bf16s)f2bf(v - bf2f(hi));
        if (wf) {
            wf[wfrag_index<bf16s>(co, tap, ci, 3 * d.Cinp)] = hi;
            wf[wfrag_index<bf16s>(co, tap, d.Cinp + ci, 3 * d.Cinp)] = hi;
            wf[wfrag_index<bf16s>(co, tap, 2 * d.Cinp + ci, 3 * d.Cinp)] = lo;
        }
        if (wd) {
            wd[wfrag_index<bf16s>(ci, 8 - tap, co, 3 * d.Cout)] = hi;
            wd[wfrag_index<bf16s>(ci, 8 - tap, d.Cout + co, 3 * d.Cout)] = hi;
            wd[wfrag_index<bf16s>(ci, 8 - tap, 2 * d.Cout + co, 3 * d.Cout)] = lo;
        }
    }
}

// Regular layers (Cin a multiple of 64 and unpadded, Cout of 32: 17 of BiDateNet's 18) go through LDS, as pack_weights_tiles_kernel (head.hip)
// does for the bf16 setting: a block takes 32 output channels x 64 input channels x 9 taps, reads them as 32 contiguous 2304-byte rows of the
// OIHW master (16-byte loads) and writes complete 1 KB fragment records of both images with 16-byte stores -- the hi part twice (thirds 0 and 1
// of a row), then, in a second pass over the same rows (L2 hits), the lo part (third 2).  The element-wise kernel above gathers 4-byte values
// 36 bytes apart and stores 2 bytes per lane six times: 229 us for the 190 MB of a step, alone at the start of every bf16x3 step (round 6).
__device__ __forceinline__ bool pack_regular_x3(const PackDesc& d) { return d.Cin == d.Cinp && d.Cin % 64 == 0 && d.Cout % 32 == 0; }
__global__ __launch_bounds__(256) void pack_weights_x3_tiles_kernel(const PackDesc* __restrict__ desc, int n_layers) {
    constexpr int ROW = 9 * 64 + 8;                            // LDS elements per output channel: [tap][ci] + 16 bytes of padding
    __shared__ __attribute__((aligned(16))) bf16s t[32 * ROW];
    const int tid = threadIdx.x;
    for (int item = blockIdx.x;; item += gridDim.x) {
        int l = 0, local = item;
        for (; l < n_layers; l++) {
            const int cnt = pack_regular_x3(desc[l]) ? (desc[l].Cout / 32) * (desc[l].Cin / 64) : 0;
            if (local < cnt) break;
            local -= cnt;
        }
        if (l == n_layers) return;                             // uniform for the block
        const PackDesc d = desc[l];
        const int ncc = d.Cin / 64, cb = local / ncc, cc = local % ncc, co0 = cb * 32, ci0 = cc * 64;
        bf16s* wf = reinterpret_cast<bf16s*>(d.wf); bf16s* wd = reinterpret_cast<bf16s*>(d.wd);
#pragma unroll 1
        for (int part = 0; part < 2; part++) {                 // 0: hi (stored to thirds 0 and 1), 1: lo (third 2)
            __syncthreads();                                   // the previous pass's LDS reads are done
            for (int q = tid; q < 32 * 144; q += 256) {        // 144 float4 per output-channel row
                const int r = q / 144, o4 = (q % 144) * 4;
                const float4 v = *reinterpret_cast<const float4*>(d.w + ((size_t)(co0 + r) * d.Cin + ci0) * 9 + o4);
                const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int o = o4 + e, ci = o / 9, tap = o % 9;
                    const bf16s hi = (bf16s)f2bf(f[e]);
                    t[r * ROW + tap * 64 + ci] = part == 0 ? hi : (bf16s)f2bf(f[e] - bf2f(hi));
                }
            }
            __syncthreads();
            for (int u = tid; u < 36 * 64; u += 256) {         // 36 records x 64 lanes, 16 bytes each
                const int rec_l = u >> 6, lane = u & 63;
                if (wf) {      // record (tap, kq): lane = co & 31 + 32 * (ci % 16) / 8, 8 consecutive ci
                    const int tap = rec_l >> 2, kq = rec_l & 3;
                    const size_t row = ((size_t)cb * 9 + tap) * (3 * d.Cin / 16), k = ci0 / 16 + kq;
                    const uint4 v = *reinterpret_cast<const uint4*>(t + (lane & 31) * ROW + tap * 64 + kq * 16 + (lane >> 5) * 8);
                    if (part == 0) {
                        *reinterpret_cast<uint4*>(wf + (row + k) * 512 + lane * 8) = v;
                        *reinterpret_cast<uint4*>(wf + (row + d.Cin / 16 + k) * 512 + lane * 8) = v;
                    } else *reinterpret_cast<uint4*>(wf + (row + 2 * (d.Cin / 16) + k) * 512 + lane * 8) = v;
                }
                if (wd) {      // roles swapped, taps rotated: record (ci block, 8 - tap, co group of 16): 8 consecutive co
                    const int cbi = rec_l / 18, rem = rec_l % 18, tap = rem >> 1, kg = rem & 1;
                    const int ci = cbi * 32 + (lane & 31), co8 = kg * 16 + (lane >> 5) * 8;
                    unsigned short h[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) h[e] = t[(co8 + e) * ROW + tap * 64 + ci];
                    const uint4 v = make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
                    const size_t row = ((size_t)(ci0 / 32 + cbi) * 9 + (8 - tap)) * (3 * d.Cout / 16), k = co0 / 16 + kg;
                    if (part == 0) {
                        *reinterpret_cast<uint4*>(wd + (row + k) * 512 + lane * 8) = v;
                        *reinterpret_cast<uint4*>(wd + (row + d.Cout / 16 + k) * 512 + lane * 8) = v;
                    } else *reinterpret_cast<uint4*>(wd + (row + 2 * (d.Cout / 16) + k) * 512 + lane * 8) = v;
                }
            }
        }
    }
}
// the element-wise kernel restricted to the layers the tile kernel does not take (the 13-band first layer)
__global__ void pack_weights_x3_irregular_kernel(const PackDesc* __restrict__ desc) {
    if (pack_regular_x3(desc[blockIdx.y])) return;
    const PackDesc d = desc[blockIdx.y];
    const size_t total = (size_t)d.Cout * 9 * d.Cinp;
    bf16s* wf = reinterpret_cast<bf16s*>(d.wf); bf16s* wd = reinterpret_cast<bf16s*>(d.wd);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = i % d.Cinp; const size_t t = i / d.Cinp; const int tap = t % 9; const int co = t / 9;
        const float v = ci < d.Cin ? d.w[((size_t)co * d.Cin + ci) * 9 + tap] : 0.f;
        const bf16s hi = (bf16s)f2bf(v);
        const bf16s lo = (bf16s)f2bf(v - bf2f(hi));
        if (wf) {
            wf[wfrag_index<bf16s>(co, tap, ci, 3 * d.Cinp)] = hi;
            wf[wfrag_index<bf16s>(co, tap, d.Cinp + ci, 3 * d.Cinp)] = hi;
            wf[wfrag_index<bf16s>(co, tap, 2 * d.Cinp + ci, 3 * d.Cinp)] = lo;
        }
        if (wd) {
            wd[wfrag_index<bf16s>(ci, 8 - tap, co, 3 * d.Cout)] = hi;
            wd[wfrag_index<bf16s>(ci, 8 - tap, d.Cout + co, 3 * d.Cout)] = hi;
            wd[wfrag_index<bf16s>(ci, 8 - tap, 2 * d.Cout + co, 3 * d.Cout)] = lo;
        }
    }
}

int bdn_pack_weights_x3_multi(const PackDesc* desc, int n_layers, hipStream_t st) {
    hipLaunchKernelGGL(pack_weights_x3_tiles_kernel, dim3(1024), dim3(256), 0, st, desc, n_layers);
    hipLaunchKernelGGL(pack_weights_x3_irregular_kernel, dim3(32, n_layers), dim3(256), 0, st, desc);
    BDN_CHECK_LAUNCH("pack_weights_x3_multi");
    return BDN_OK;
}

int bdn_pack_weights_x3(const float* w_oihw, void* wf, void* wd, int Cout, int Cin, int Cin_pad, hipStream_t st) {
    const size_t total = (size_t)Cout * 9 * Cin_pad;
    hipLaunchKernelGGL(pack_weights_x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_oihw, (bf16s*)wf, (bf16s*)wd, Cout, Cin, Cin_pad);
    BDN_CHECK_LAUNCH("pack_weights_x3");
    return BDN_OK;
}

// dw[co][ci][t] = T[co][ci][t] + T[co][Cinp + ci][t] + T[Cout + co][ci][t],  T = f32 [2 Cout][2 Cinp][9] (the doubled-operand
// weight gradient: hi*hi + hi*lo + lo*hi; the lo*lo quadrant is dropped)
// (taps = 9, or 27 for the 3x3x3 convolution)
__global__ void wgrad_x3_combine_kernel(const float* __restrict__ T, float* __restrict__ dw, int Cout, int Cinp, int Cin_real, int taps, int terms) {
    const size_t total = (size_t)Cout * Cin_real * taps;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = i % taps; const size_t r = i / taps; const int ci = r % Cin_real; const int co = r / Cin_real;
    const size_t ld = (size_t)2 * Cinp * taps;
    float v = T[(size_t)co * ld + (size_t)ci * taps + t] + T[(size_t)co * ld + (size_t)(Cinp + ci) * taps + t];
    if (terms == 3) v += T[(size_t)(Cout + co) * ld + (size_t)ci * taps + t];      // (BDN_BF16X2 never computed the lo rows)
    dw[i] = v;
}

int bdn_wgrad_x3_combine(const float* T, float* dw, int Cout, int Cinp, int Cin_real, int taps, int terms, hipStream_t st) {
    const size_t total = (size_t)Cout * Cin_real * taps;
    hipLaunchKernelGGL(wgrad_x3_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, T, dw, Cout, Cinp, Cin_real, taps, terms);
    BDN_CHECK_LAUNCH("wgrad_x3_combine");
    return BDN_OK;
}
