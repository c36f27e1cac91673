// libbidate_hip: full-scene sliding-window inference helpers.
// The two dates of a scene stay resident in HBM as band planes; tiles are gathered straight into the packed
// NHWC batch and predictions are written straight into the scene mask -- no host-side patch stack.
#include "common.hpp"

static inline unsigned grid_for(size_t n, int block = 256) { return (unsigned)((n + block - 1) / block); }

// ============================================================ band upload
// One row band of ALL band planes of a host scene in ONE copy: [C] planes of H x W float32, rows [r0, r1) of each -- a 2-D copy of C "rows" of
// (r1 - r0) * W * 4 bytes at pitch H * W * 4 (what train.py:190-193 does per batch with .to(device), here per band of the resident scene).
// The round-5 feeder issued one copy per plane (26 per band for both dates); boxes whose DMA engines pay more per copy sustained 37 of 57 GB/s.
extern "C" int bdn_upload_band(float* dst_planes, const float* src_planes_host, int C, int H, int W, int r0, int r1, void* stream) {
    if (!dst_planes || !src_planes_host) BDN_FAIL(BDN_E_ARG, "upload_band: null pointer");
    if (C <= 0 || H <= 0 || W <= 0 || r0 < 0 || r1 <= r0 || r1 > H) BDN_FAIL(BDN_E_SHAPE, "upload_band: bad C=%d H=%d W=%d rows [%d, %d)", C, H, W, r0, r1);
    const size_t pitch = (size_t)H * W * sizeof(float), off = (size_t)r0 * W, width = (size_t)(r1 - r0) * W * sizeof(float);
    const hipError_t e = hipMemcpy2DAsync(dst_planes + off, pitch, src_planes_host + off, pitch, width, (size_t)C, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e != hipSuccess) BDN_FAIL(BDN_E_HIP, "upload_band: hipMemcpy2DAsync: %s", hipGetErrorString(e));
    return BDN_OK;
}

// ============================================================ gather_tiles
// reference: utils/inference.py:134-184 (_get_patches) + :61-66 (NHWC->NCHW transpose) + train.py:190-193
// (batch slice, host->device).  scene_d*: [C][H][W] f32 band planes.  origins: int32 [n][2] = (y0, x0).
// out: [2n][p][p][Cpad] T, date-1 tiles first (the layout bdn_pack_input produces).
// One thread per (tile pixel, 16-byte unit): plane reads are coalesced along x, every store is 16 bytes.
template <typename T>
__global__ void gather_tiles_kernel(const float* __restrict__ s1, const float* __restrict__ s2,
                                    const int* __restrict__ origins, T* __restrict__ out,
                                    int n, int C, int H, int W, int p, int Cpad) {
    constexpr int EPU = ET<T>::EPU;
    const int upp = Cpad / EPU;
    const size_t total = (size_t)2 * n * p * p * upp;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // unit index is the SLOW coordinate inside a tile row so that a wave reads 64 consecutive x of one plane set
    const int x = i % p; size_t t = i / p;
    const int u = t % upp; t /= upp;
    const int y = t % p; const int tile = t / p;
    const int date = tile >= n, tt = tile - date * n;
    const int y0 = origins[2 * tt], x0 = origins[2 * tt + 1];
    const size_t hw = (size_t)H * W;
    const float* src = (date ? s2 : s1) + (size_t)(y0 + y) * W + (x0 + x);
    float f[EPU];
#pragma unroll
    for (int e = 0; e < EPU; e++) {
        const int c = u * EPU + e;
        f[e] = c < C ? src[(size_t)c * hw] : 0.f;
    }
    T* dst = out + (((size_t)tile * p + y) * p + x) * Cpad + u * EPU;
    *reinterpret_cast<uint4*>(dst) = Unit<T>::pack(f);
}

extern "C" int bdn_gather_tiles(int dtype, const float* scene_d1, const float* scene_d2, const int32_t* origins,
                                void* out, int n_tiles, int C, int H, int W, int p, int Cpad, void* stream) {
    if (!scene_d1 || !scene_d2 || !origins || !out) BDN_FAIL(BDN_E_ARG, "gather_tiles: null pointer");
    if (n_tiles <= 0 || C <= 0 || p <= 0 || H < p || W < p || Cpad < C || Cpad % 16)
        BDN_FAIL(BDN_E_SHAPE, "gather_tiles: need H,W >= p, Cpad a multiple of 16 and >= C");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == BDN_BF16) {
        const size_t total = (size_t)2 * n_tiles * p * p * (Cpad / 8);
        hipLaunchKernelGGL(gather_tiles_kernel<bf16s>, dim3(grid_for(total)), dim3(256), 0, st,
                           scene_d1, scene_d2, origins, (bf16s*)out, n_tiles, C, H, W, p, Cpad);
    } else if (dtype == BDN_F32) {
        const size_t total = (size_t)2 * n_tiles * p * p * (Cpad / 4);
        hipLaunchKernelGGL(gather_tiles_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st,
                           scene_d1, scene_d2, origins, (float*)out, n_tiles, C, H, W, p, Cpad);
    } else BDN_FAIL(BDN_E_ARG, "gather_tiles: bad dtype");
    BDN_CHECK_LAUNCH("gather_tiles");
    return BDN_OK;
}

// ============================================================ ingest: per-band normalisation + resize to the label grid
// reference utils/dataloaders.py:86-111 (city_loader): band = (band.astype(float32) - mean) / std; band = cv2.resize(band,
// (W, H)) (default INTER_LINEAR).  Sentinel-2 bands come at 10 / 20 / 60 m, the label raster at 10 m, so most bands
// are upsampled 2x or 6x into their plane of the [C][H][W] scene that the tile gather reads.
// cv2's float INTER_LINEAR (OpenCV imgproc/resize.cpp): fx = (dx + 0.5) * (src_w / dst_w) - 0.5, sx = floor(fx),
// fx -= sx, clamped to [0, src_w - 1] with weight (1, 0) at either border; rows alike; horizontal pass first.
template <typename S>
__global__ void ingest_band_kernel(const S* __restrict__ src, int hs, int ws, float mean, float stdv,
                                   float* __restrict__ dst, int H, int W, double scale_y, double scale_x) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    float fy = (float)((y + 0.5) * scale_y - 0.5), fx = (float)((x + 0.5) * scale_x - 0.5);
    int sy = (int)floorf(fy), sx = (int)floorf(fx);
    fy -= (float)sy; fx -= (float)sx;
    if (sy < 0) { sy = 0; fy = 0.f; }
    if (sy >= hs - 1) { sy = hs - 1; fy = 0.f; }
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= ws - 1) { sx = ws - 1; fx = 0.f; }
    const int sy1 = min(sy + 1, hs - 1), sx1 = min(sx + 1, ws - 1);
    const float a00 = ((float)src[(size_t)sy * ws + sx] - mean) / stdv, a01 = ((float)src[(size_t)sy * ws + sx1] - mean) / stdv;
    const float a10 = ((float)src[(size_t)sy1 * ws + sx] - mean) / stdv, a11 = ((float)src[(size_t)sy1 * ws + sx1] - mean) / stdv;
    const float r0 = a00 * (1.f - fx) + a01 * fx, r1 = a10 * (1.f - fx) + a11 * fx;      // horizontal pass, then vertical
    dst[(size_t)y * W + x] = r0 * (1.f - fy) + r1 * fy;
}

extern "C" int bdn_ingest_band(int src_is_f32, const void* src, int hs, int ws, float mean, float stdv,
                               float* dst, int H, int W, void* stream) {
    if (!src || !dst) BDN_FAIL(BDN_E_ARG, "ingest_band: null pointer");
    if (hs <= 0 || ws <= 0 || H <= 0 || W <= 0 || !(stdv != 0.f)) BDN_FAIL(BDN_E_SHAPE, "ingest_band: bad shape or zero std");
    const dim3 grid((W + 255) / 256, H);
    const double sy = (double)hs / H, sx = (double)ws / W;
    if (src_is_f32) hipLaunchKernelGGL(ingest_band_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, hs, ws, mean, stdv, dst, H, W, sy, sx);
    else hipLaunchKernelGGL(ingest_band_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)src, hs, ws, mean, stdv, dst, H, W, sy, sx);
    BDN_CHECK_LAUNCH("ingest_band");
    return BDN_OK;
}

// ============================================================ argmax (+ stitch)
// reference: `_, cd_preds = torch.max(preds, 1)` train.py:199 (first maximum wins ties), then
// utils/inference.py:187-236 (_get_bands): main tiles, then last-column tiles, then last-row tiles, then the
// corner are pasted in that order, later pastes overwriting earlier ones.  Equivalent order-free rule used here:
// a pixel belongs to the far-edge band(s) it lies in (y >= H-p, x >= W-p) and only a tile anchored on exactly
// those bands writes it, so tiles of one launch never race with different values.
__global__ void argmax_stitch_kernel(const float* __restrict__ logits, const int* __restrict__ origins,
                                     unsigned char* __restrict__ out, int n, int ncls, int p, int H, int W, size_t pp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // pp: pixels per image (p * p for tiles; H * W for the plain argmax of [n,ncls,H,W])
    if (i >= (size_t)n * pp) return;
    const int tile = i / pp; const int rem = i % pp;
    const float* l = logits + (size_t)tile * ncls * pp + rem;
    float best = l[0]; int arg = 0;
    for (int c = 1; c < ncls; c++) {
        const float v = l[(size_t)c * pp];
        if (v > best) { best = v; arg = c; }
    }
    if (!origins) { out[i] = (unsigned char)arg; return; }
    const int y0 = origins[2 * tile], x0 = origins[2 * tile + 1];
    const int y = y0 + rem / p, x = x0 + rem % p;
    const bool ty = y0 == H - p, tx = x0 == W - p;
    const bool py = y >= H - p, px = x >= W - p;
    if (ty == py && tx == px) out[(size_t)y * W + x] = (unsigned char)arg;
}

extern "C" int bdn_argmax(const float* logits, uint8_t* out, int n, int ncls, int H, int W, void* stream) {
    if (!logits || !out) BDN_FAIL(BDN_E_ARG, "argmax: null pointer");
    if (n <= 0 || ncls <= 0 || ncls > 256 || H <= 0 || W <= 0) BDN_FAIL(BDN_E_SHAPE, "argmax: bad shape (<= 256 classes)");
    hipLaunchKernelGGL(argmax_stitch_kernel, dim3(grid_for((size_t)n * H * W)), dim3(256), 0, (hipStream_t)stream,
                       logits, (const int*)nullptr, out, n, ncls, H, H, W, (size_t)H * W);
    BDN_CHECK_LAUNCH("argmax");
    return BDN_OK;
}

extern "C" int bdn_argmax_stitch(const float* logits, const int32_t* origins, uint8_t* mask,
                                 int n_tiles, int ncls, int p, int H, int W, void* stream) {
    if (!logits || !origins || !mask) BDN_FAIL(BDN_E_ARG, "argmax_stitch: null pointer");
    if (n_tiles <= 0 || ncls <= 0 || ncls > 256 || p <= 0 || H < p || W < p) BDN_FAIL(BDN_E_SHAPE, "argmax_stitch: bad shape");
    hipLaunchKernelGGL(argmax_stitch_kernel, dim3(grid_for((size_t)n_tiles * p * p)), dim3(256), 0, (hipStream_t)stream,
                       logits, origins, mask, n_tiles, ncls, p, H, W, (size_t)p * p);
    BDN_CHECK_LAUNCH("argmax_stitch");
    return BDN_OK;
}
