/*
 * bidate_hip.h -- C ABI of libbidate_hip.so: hand-written gfx950 (MI355X / CDNA4)
 * kernels for the bi-date Siamese U-Net training path of granularai/fabric.
 *
 * The reference has no FFI: its hot path is a set of ATen call sites reached from
 * Python.  Each entry point below names the reference call site(s)
 * (file:line relative to the reference root) whose arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch caching
 *     allocator); the library never allocates, frees or synchronises;
 *   - every call enqueues on `stream` (a hipStream_t passed as void*) and returns;
 *   - return value: 0 = ok, negative = error (BDN_E_*); bdn_last_error() gives a
 *     thread-local message.  No C++ exception crosses the boundary;
 *   - activations are NHWC; `dtype` selects the element type of activations and
 *     packed weights: BDN_F32 (f32 storage, v_mfma_f32_32x32x2_f32: the
 *     "fp32-equivalent" parity setting) or BDN_BF16 (bf16 storage,
 *     v_mfma_f32_32x32x16_bf16, fp32 accumulate: the throughput setting);
 *   - BatchNorm statistics are kept per *group* (= date): images
 *     [g*imgs_per_group, (g+1)*imgs_per_group) of a batch form group g
 *     (models/bidate_model.py:23-33 runs the shared BN modules once per date).
 */
#ifndef BIDATE_HIP_H
#define BIDATE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { BDN_F32 = 0, BDN_BF16 = 1, BDN_BF16X3 = 2,     /* BDN_BF16X3: float32 tensors, GEMM operands split into bf16 hi + lo (3 MFMAs per product) */
       BDN_BF16X2 = 3 };  /* the BACKWARD GEMMs of the bf16x3 setting with two of the three terms, on the same operands and filter images as
                           * BDN_BF16X3 -- bdn_conv3x3 / bdn_conv3x3_dgrad_bs (data gradient): K = [dz_hi | dz_lo] against [w_hi | w_hi], i.e. the
                           * filter rounded to bf16, dz in full; bdn_conv3x3_wgrad_ex / bdn_wgrad_workspace_bytes_ex: dz_hi x [a_hi | a_lo], i.e.
                           * dz rounded, the activations in full (autograd of models/unet_parts.py:13,16 to ~1e-3 relative instead of ~2e-5).
                           * Accepted by those entry points only. */
enum { BDN_OK = 0, BDN_E_ARG = -1, BDN_E_SHAPE = -2, BDN_E_HIP = -3 };
enum { BDN_IN_PLAIN = 0, BDN_IN_BNRELU = 1 };

const char* bdn_last_error(void);
int bdn_version(void);

/* ---- streams of the training step (reference: none -- train.py:83-101 runs on PyTorch's default stream) ----
 * A HIP stream on the calling thread's current device, created by the library so that its hardware-queue placement does not
 * depend on the host framework's stream pool: the step's dependency chain (priority 1 = high), its weight-gradient GEMMs and
 * its host -> device copies (priority 0) each get one, shared by every step object of the process (fabric_amd/streams.py).
 * The caller owns the stream and destroys it with bdn_stream_destroy. */
int bdn_stream_create(int priority, void** stream_out);
int bdn_stream_destroy(void* stream);
/* Events for hand-offs between two streams of ONE device (the chain releases each layer's weight-gradient GEMM to the second stream):
 * created with hipEventDisableTiming | hipEventDisableSystemFence -- a default event performs a system-scope release (cache write-back /
 * invalidate towards the host) every time it is recorded.  Not for host-side synchronisation.  bdn_event_record(event, stream) and
 * bdn_stream_wait_event(stream, event) enqueue and return. */
int bdn_event_create(void** event_out);
int bdn_event_destroy(void* event);
int bdn_event_record(void* event, void* stream);
int bdn_stream_wait_event(void* stream, void* event);

/* ---- layout converters (boundary of BiDateNet.forward, models/bidate_model.py:22) ---- */
/* x_d1, x_d2: [B,C,H,W] f32 NCHW  ->  out: [2B,H,W,Cpad] (date-1 images first), channels >= C zeroed.
 * dtype BDN_BF16X3: out is the first convolution's split operand [2B,H,W,2 Cpad] bf16 = hi | lo (bdn_split_pack of the float32 image). */
int bdn_pack_input(int dtype, const float* x_d1, const float* x_d2, void* out,
                   int B, int C, int H, int W, int Cpad, void* stream);
/* OIHW f32 [Cout,Cin,3,3] -> forward GEMM image wf [Cout][9][Cin_pad] and the data-gradient image
 * wd [Cin_pad][9][Cout] (taps rotated by 180 degrees); either output may be NULL. */
int bdn_pack_weights(int dtype, const float* w_oihw, void* wf, void* wd,
                     int Cout, int Cin, int Cin_pad, void* stream);

/* The same for n_layers filters in one launch.  desc: DEVICE array of n_layers records
 * { const float* w_oihw; void* wf; void* wd; int32 Cout, Cin, Cin_pad, reserved; } (40 bytes each).  dtype BDN_BF16X3: the split images of
 * bdn_pack_weights(BDN_BF16X3) for every record. */
int bdn_pack_weights_multi(int dtype, const void* desc, int n_layers, void* stream);

/* ---- 3x3 convolution, stride 1, zero padding 1: nn.Conv2d(ci,co,3,padding=1), models/unet_parts.py:13,16 ----
 * Implicit GEMM on MFMA.  The A operand is gathered from in0 (channels [0,C0)) and optionally in1
 * (channels [C0,C0+C1), the never-materialised torch.cat of models/unet_parts.py:78).
 * in_mode BDN_IN_BNRELU applies relu(z*scale+shift) (BatchNorm2d+ReLU of the producing layer,
 * models/unet_parts.py:14-15,17-18) while loading in0; in_bn is that layer's [G][4][C0] f32 table
 * written by bdn_bn_finalize / bdn_bn_eval.
 * w: packed [Cout][9][C0+C1]; bias: [Cout] f32 or NULL.
 * out: [N,H,W,Cout].  stats_partial: NULL or [bdn_conv3x3_num_mtiles][2][Cout] f32 receiving
 * per-tile sum / sum-of-squares of the (bias-included, f32) outputs for the BatchNorm that follows.
 * The same entry point computes the data gradient when given wd and dz.
 * Every tensor must stay below 4 GB (the kernels address with one base + a 32-bit byte offset): BDN_E_SHAPE otherwise. */
int bdn_conv3x3(int dtype, const void* in0, int C0, const void* in1, int C1,
                int in_mode, const float* in_bn, int imgs_per_group,
                const void* w, const float* bias, void* out, float* stats_partial,
                int N, int H, int W, int Cout, void* stream);
int bdn_conv3x3_num_mtiles(int N, int H, int W, int Cout, int imgs_per_group);
/* The same by operand type: BDN_BF16X3 / BDN_BF16X2 launches with C0 (logical operand channels) a multiple of 64 run the kernels that fuse the
 * split product into one reduction (two LDS patches per chunk), which have their own tile plan; every other case equals bdn_conv3x3_num_mtiles. */
int bdn_conv3x3_num_mtiles_ex(int dtype, int N, int H, int W, int C0, int Cout, int imgs_per_group);
/* Name of the kernel instantiation bdn_conv3x3 runs for a shape, e.g. "conv3x3_kernel<bf16,128,8,16,1,128,1,4,false,bf16,false,false>"
 * (the rocprofv3 name with `unsigned short` spelled bf16); "" for an unsupported shape.  Thread-local buffer. */
const char* bdn_conv3x3_variant(int dtype, int N, int H, int W, int C0, int C1, int Cout, int imgs_per_group);

/* bf16x3 / bf16x2 convolution straight from a FLOAT32 operand (reference: models/unet_parts.py:14-16, BatchNorm -> ReLU -> Conv2d of a
 * double_conv's second half; round 6).  Equals bdn_split_pack(in, in_mode, in_bn) followed by bdn_conv3x3(dtype, ...) bit for bit, in one
 * launch: relu(z * scale + shift) and the bf16 hi / lo split are applied while the tile is staged, no split pass runs in front of the
 * convolution.  in [N,H,W,C0] float32, C0 a multiple of 64 and <= 512; w the bdn_pack_weights(BDN_BF16X3) forward image; out float32;
 * stats_partial as bdn_conv3x3 with bdn_conv3x3_num_mtiles_ex(dtype, ...) rows.
 * split_out: NULL, or [N,H,W,2 C0] bf16 that receives exactly bdn_split_pack's output (the layer's weight-gradient GEMM reads it later). */
int bdn_conv3x3_x3src(int dtype, const float* in, int C0, int in_mode, const float* in_bn, int imgs_per_group,
                      const void* w, const float* bias, float* out, float* stats_partial, void* split_out,
                      int N, int H, int W, int Cout, void* stream);
/* Name of the kernel instantiation bdn_conv3x3_x3src runs for a shape (as bdn_conv3x3_variant); "" for an unsupported shape. */
const char* bdn_conv3x3_x3src_variant(int dtype, int N, int H, int W, int C0, int Cout, int imgs_per_group);

/* Data gradient of nn.Conv2d(ci,co,3,padding=1) (autograd of models/unet_parts.py:13,16) with the BatchNorm-backward
 * statistics of the PRODUCING layer fused into the epilogue: dz [N,H,W,C0] x rotated filter image w_dgrad ->
 * dA [N,H,W,Cout] (gradient wrt relu(bn(z_prev))), and bs_partial [bdn_conv3x3_num_mtiles(N,H,W,Cout,ipg)][2][Cout] =
 * per-tile sum g, sum g*z_prev with g = dA * [scale*z_prev + shift > 0] (z_prev [N,H,W,Cout], bn_prev [G][4][Cout]).
 * What is STORED to dA is g, the masked gradient (zero where relu(bn(z_prev)) is off): every BatchNorm-backward consumer applies the
 * same mask again (idempotent), and bdn_conv3x3_dgrad_bb expects it.  The same holds for bdn_enc_skip_bwd (with bs_partial) and
 * bdn_upsample2x_bwd_bs.  Tiles are image-major, so a statistic group owns num_mtiles/G consecutive rows: feed bdn_bn_bwd_apply(raw_moment=1). */
int bdn_conv3x3_dgrad_bs(int dtype, const void* dz, int C0, const void* w_dgrad, void* dA,
                         const void* z_prev, const float* bn_prev, int imgs_per_group, float* bs_partial,
                         int N, int H, int W, int Cout, void* stream);

/* Data gradient of layer L's convolution with L's BatchNorm+ReLU backward (autograd of models/unet_parts.py:14-15,17-18) applied while the
 * operand is staged: dA [N,H,W,C0] = the MASKED gradient g = dA*[scale z + shift > 0] as the fused producers store it (see
 * bdn_conv3x3_dgrad_bs), z [N,H,W,C0], bn [G][4][C0], sums [G][2][C0] from bdn_bn_bwd_finalize; the kernel forms
 * dz = a g + b z + c with a = scale, b = -scale invstd s1/M, c = -scale s0/M - b mean (= bdn_bn_bwd_apply's scale*(g - s0/M - xhat*s1/M)
 * up to rounding: two FMAs per element, three per-channel constants, no compare), convolves it with w_dgrad into dA_prev [N,H,W,Cout]
 * and stores dz [N,H,W,C0] to dz_out (NULL: not stored) for the weight-gradient GEMM -- bdn_bn_bwd_apply's pass over dA, z and dz does
 * not run.  z_prev / bn_prev / bs_partial: as bdn_conv3x3_dgrad_bs, or all NULL.
 * bf16, C0 = 64 (a single channel chunk: the staging then runs once, in the kernel's prologue), maps larger than 8x8. */
int bdn_conv3x3_dgrad_bb(int dtype, const void* dA, int C0, const void* z, const float* bn, const float* sums, int imgs_per_group,
                         const void* w_dgrad, void* dA_prev, const void* z_prev, const float* bn_prev, float* bs_partial,
                         void* dz_out, int N, int H, int W, int Cout, void* stream);

/* Name of the kernel instantiation bdn_conv3x3_dgrad_bb runs for a shape (its dispatcher picks its own tile configurations); "" for an
 * unsupported shape.  Thread-local buffer, as bdn_conv3x3_variant. */
const char* bdn_conv3x3_dgrad_bb_variant(int N, int H, int W, int Cout, int imgs_per_group);

/* ---- 3x3x3 convolution, stride 1, zero padding 1 (BASELINE configs[3]: the multi-date 3-D U-Net stack) ----
 * The reference tree holds NO source for that model (UNetLSTM/ is an empty sub-module, README.md:5): parity is UNPINNED, the
 * oracle is torch.nn.functional.conv3d.  Implicit GEMM with K = 27 Cin on the 2-D kernels: tensors are [N,D,H,W,C] (the D
 * slices of a sample are consecutive NHWC images), a 3x3x3 window is three 3x3 windows on slices d-1, d, d+1 = three sources
 * of one reduction, a slice outside the sample is a zero mask.  No depth padding, no im2col.
 *   w      bdn_pack_weights image (wf) of the OIHW view [Cout][3 C][3][3] whose input channel kd*C + c holds w3d[co][c][kd][.][.]
 *          (data gradient: the same entry point on dz with the view [C][3 Cout][3][3], channel s*Cout + co = w3d[co][c][2-s][2-kh][2-kw])
 *   in_mode / in_bn / stats_partial / bias as in bdn_conv3x3; imgs_per_group counts samples;
 *   stats_partial rows: bdn_conv3d_num_mtiles(N, D, H, W).  dtype: BDN_BF16, BDN_F32, or BDN_BF16X3 (round 4) -- then `in` is the bf16
 *   operand [N,D,H,W,C] whose C = 3 x the logical width holds [hi | lo | hi] of the float32 tensor (bdn_split_pack's [hi | lo] plus its hi
 *   half again), `w` the plain BDN_BF16 image of the view with channels [w_hi | w_hi | w_lo] per depth tap, in_mode PLAIN, out float32.
 * bdn_conv3d_wgrad: dw [Cout][Cin_real][3][3][3] f32 from dz [N,D,H,W,Cout] and the PLAIN input [N,D,H,W,C];
 *   partial: bdn_wgrad_workspace_bytes_ex(dtype, N*D, H, W, Cout, C, 0, 1, BDN_IN_PLAIN, 0) bytes.
 *   BDN_BF16X3: dz [N,D,H,W,2 Cout] and in [N,D,H,W,2 C] are bdn_split_pack operands; partial: the BDN_BF16 size for (2 Cout, 2 C) plus
 *   4*Cout*C*27 floats (the doubled-operand tile the three quadrants are summed from). */
int bdn_conv3d_num_mtiles(int N, int D, int H, int W);
int bdn_conv3d(int dtype, const void* in, int C, int in_mode, const float* in_bn, int imgs_per_group,
               const void* w, const float* bias, void* out, float* stats_partial,
               int N, int D, int H, int W, int Cout, void* stream);
int bdn_conv3d_wgrad(int dtype, const void* dz, int Cout, const void* in, int C,
                     float* partial, float* dw_oidhw, int Cin_real, int N, int D, int H, int W, void* stream);

/* ---- bf16x3 operand split (dtype BDN_BF16X3 of bdn_conv3x3 / bdn_conv3x3_dgrad_bs / bdn_conv3x3_wgrad* / bdn_pack_weights) ----
 * The 1e-3-parity setting at matrix-core speed: tensors stay float32; a GEMM operand x is fed as hi = bf16(x) and
 * lo = bf16(x - hi) and a product keeps a_hi*w_hi + a_lo*w_hi + a_hi*w_lo (float32 accumulate).  bdn_split_pack builds the
 * operand every BDN_BF16X3 entry point expects: out [N,H,W,2(C0+C1)] bf16 = hi(a) | lo(a), a = cat(src0, src1) (f32 NHWC;
 * src1 may be NULL) with relu(bn(.)) applied to src0 when in_mode == BDN_IN_BNRELU (models/unet_parts.py:14-15,78).
 * With BDN_BF16X3: bdn_conv3x3 takes in0 = that tensor, C0 = the logical channel count, in1 = NULL, in_mode PLAIN, w from
 * bdn_pack_weights(BDN_BF16X3) (wf: [Cout][9][3 Cin_pad] bf16, wd: [Cin_pad][9][3 Cout]), and writes float32 out / statistics;
 * bdn_conv3x3_wgrad* take dz and in0 both split-packed and write the float32 OIHW gradient. */
int bdn_split_pack(const float* src0, int C0, const float* src1, int C1, int in_mode, const float* in_bn,
                   int imgs_per_group, void* out, int N, int H, int W, void* stream);

/* ---- weight gradient of the same convolution (autograd of models/unet_parts.py:13,16) ----
 * dz: [N,H,W,Cout]; inputs as in bdn_conv3x3.  partial: workspace of bdn_wgrad_workspace_bytes().
 * dw_oihw: f32 [Cout,Cin_real,3,3] (overwritten; channels >= Cin_real of a padded input are dropped).
 * bdn_wgrad_workspace_bytes is pure: it covers every dtype / source split / input mode of the shape under default flags. */
size_t bdn_wgrad_workspace_bytes(int N, int H, int W, int Cout, int Cin, int imgs_per_group);
int bdn_conv3x3_wgrad(int dtype, const void* dz, int Cout,
                      const void* in0, int C0, const void* in1, int C1,
                      int in_mode, const float* in_bn, int imgs_per_group,
                      float* partial, float* dw_oihw, int Cin_real,
                      int N, int H, int W, void* stream);
/* The same with a per-call `flags` word (there is no process-wide tuning state):
 *   bits 0-1   phases: bit 0 = split-K GEMM into `partial`, bit 1 = fixed-order reduction into dw_oihw (a profiler can
 *              bracket the GEMM alone by issuing the phases apart);
 *   bits 8-11  kernel override (0 = the library's choice): BDN_WG_SIMPLE forces the one-chunk-at-a-time kernel
 *              (bdn_conv3x3_wgrad_variant says what a call will run: BDN_WG_SIMPLE or BDN_WG_ROLE);
 *   bits 16-28 target number of blocks of the GEMM (0 = default 128: half the CUs, the GEMM shares the chip with the dz chain).
 * With non-default flags `partial` must hold bdn_wgrad_workspace_bytes_ex(same arguments).  Results are deterministic
 * for fixed flags; different plans differ only in the summation order of the partial tiles. */
#define BDN_WG_SIMPLE 1      /* one-chunk-at-a-time kernel (any dtype, first layer, 8x8 maps) */
#define BDN_WG_ROLE   5      /* role-split bf16 kernel (64-channel tiles, 8x16 spatial tiles): four MFMA waves fed by four staging waves --
                                dz tile by LDS-DMA, halo patch by LDS-DMA (plain) or through registers with BatchNorm+ReLU on load; 3 LDS buffers.
                                (2-4 were the round-1/2 kernels it replaced: tools/experimental/wgrad_v2_v6.hip.inc) */
#define BDN_WG_FLAGS(phases, kernel, blocks) ((phases) | ((kernel) << 8) | ((blocks) << 16))
size_t bdn_wgrad_workspace_bytes_ex(int dtype, int N, int H, int W, int Cout, int C0, int C1, int imgs_per_group,
                                    int in_mode, int flags);
int bdn_conv3x3_wgrad_ex(int dtype, const void* dz, int Cout,
                         const void* in0, int C0, const void* in1, int C1,
                         int in_mode, const float* in_bn, int imgs_per_group,
                         float* partial, float* dw_oihw, int Cin_real,
                         int N, int H, int W, int flags, void* stream);
int bdn_conv3x3_wgrad_variant(int dtype, int N, int H, int W, int Cout, int C0, int C1, int imgs_per_group,
                              int in_mode, int flags);
/* Weight gradient of the FIRST convolution (inc.conv.conv.0, models/unet_parts.py:13 via unet_model.py inconv) with the
 * BatchNorm+ReLU backward of its own output (unet_parts.py:14-15) fused into the staging: the layer has no data gradient,
 * so dz = bn_bwd(dA, z) is never written -- the kernel reads dA [N,H,W,ldA>=64] and z [N,H,W,64], applies
 * bdn_bn_bwd_apply's expression with `sums` from bdn_bn_bwd_finalize, rounds to bf16 and multiplies with the input patches
 * in0 [N,H,W,16].  The result equals bdn_bn_bwd_apply + bdn_conv3x3_wgrad up to the summation order of the partial tiles.
 * Shape class: Cout = 64, C0 = 16 (bdn_conv3x3_wgrad_bnbwd_supported); partial: bdn_wgrad_workspace_bytes().
 * dtype BDN_BF16X3 / BDN_BF16X2 (round 6): dA [N,H,W,ldA] and z [N,H,W,64] float32, in0 = the input's split operand [N,H,W,32] bf16 =
 * hi(16) | lo(16) (bdn_pack_input(BDN_BF16X3) / bdn_split_pack); dz is formed in float32, split into bf16 hi + lo inside the staging and the
 * three (two: dz rounded) terms of the split product go into one accumulator: equals bdn_bn_bwd_apply_split + bdn_conv3x3_wgrad(dtype) up to
 * summation order, without the pass over dA and z and without the split dz. */
int bdn_conv3x3_wgrad_bnbwd_supported(int dtype, int N, int H, int W, int Cout, int C0, int imgs_per_group);
int bdn_conv3x3_wgrad_bnbwd(int dtype, const void* dA, int ldA, const void* z, const float* bn, const float* sums,
                            int imgs_per_group, int Cout, const void* in0, int C0,
                            float* partial, float* dw_oihw, int Cin_real, int N, int H, int W, void* stream);

/* ---- BatchNorm2d training statistics: nn.BatchNorm2d, models/unet_parts.py:14,17 ----
 * Reduces the conv's per-tile partials and produces, per group g and channel c,
 * bn[g][0..3][c] = {mean, invstd, scale = gamma*invstd, shift = beta - mean*scale}  (layout [G][4][C]),
 * then updates running_mean/var (momentum 0.1, unbiased variance) once per group in order g=0,1,..
 * and adds G to num_batches_tracked -- the reference calls the module once per date.
 * ws: scratch of bdn_bn_finalize_workspace_bytes() (double partial sums of the two-stage reduction). */
size_t bdn_bn_finalize_workspace_bytes(int n_mtiles, int G, int C);
int bdn_bn_finalize(const float* stats_partial, int n_mtiles, int G, int C, int count_per_group,
                    const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float* bn, void* ws, void* stream);
/* Eval mode: bn[0][..] from the running buffers (mean=rm, invstd=rsqrt(rv+eps)), replicated for G groups. */
int bdn_bn_eval(const float* gamma, const float* beta, const float* running_mean,
                const float* running_var, float eps, int G, int C, float* bn, void* stream);

/* ---- eval-mode stages (round 6): nothing depends on batch statistics, so conv -> BatchNorm -> ReLU is ONE launch ----
 * Reference: model.eval() forward of double_conv, models/unet_parts.py:13-18, as validation (train.py:125-172) and full-scene
 * inference (train.py:182-205, utils/inference.py:134-236) run it.
 *
 * bdn_bn_eval_fold_multi: per layer  scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale  (bdn_bn_eval's values),
 *   out[0][c] = scale, out[1][c] = conv_bias[c] * scale + shift, for n_layers layers in one launch.  desc: DEVICE array of records
 *   { const float* gamma, *beta, *running_mean, *running_var, *conv_bias (may be NULL); float* out (= [2][C] floats); int32 C, reserved; }
 *   (56 bytes each); max_C = the largest C among them.
 *
 * bdn_conv3x3_eval: out = relu(conv3x3(in0 | in1) * ep_scale + ep_shift) as [N,H,W,Cout]; operands are PLAIN activations (what such a
 *   launch stores), w the bdn_pack_weights forward image.  Optional fused consumers (either may be NULL):
 *     mul  [N,H,W,Cout]     the other date's activation of the same layer: `out` then receives relu(x_d2 * x_d1) = a * mul
 *                           (models/bidate_model.py:35-38; both factors are rounded activations >= 0) instead of a;
 *     pool [N,H/2,W/2,Cout] nn.MaxPool2d(2) of a (models/unet_parts.py:40, floor mode).
 * bdn_conv3x3_eval_cls: the last double_conv stage (Cout = 64) with the 1x1 classifier (models/unet_parts.py:88-89, ncls <= 2) in its
 *   epilogue: logits [N,ncls,H,W] f32 (NULL: not stored; bit-identical to bdn_outc_fwd on the stored activation), mask = argmax over
 *   classes (first maximum wins, train.py:199) as uint8 -- [N,H,W] when origins is NULL, else stitched into the scene mask [Hs,Ws] at
 *   origins[n] = (y0, x0) with bdn_argmax_stitch's ownership rule (utils/inference.py:187-236).  act_out (NULL: not stored) = the activation.
 * bdn_conv3x3_eval_pair: the second convolution of an encoder level on BOTH dates of B patch pairs at once -- in [2B,H,W,C0] (date 1 first;
 *   C0 a multiple of 64 bf16 / 32 f32 channels) -- on tiles that hold the two dates of the same pixels: f_out [B,H,W,Cout] = relu(x_d2 * x_d1)
 *   (models/bidate_model.py:35-38), pool [2B,H/2,W/2,Cout] = nn.MaxPool2d(2) of both dates (NULL: none).  Neither date's activation is stored.
 * dtype BDN_BF16 or BDN_F32.  Every tensor below 4 GB. */
int bdn_bn_eval_fold_multi(const void* desc, int n_layers, int max_C, float eps, void* stream);
int bdn_conv3x3_eval(int dtype, const void* in0, int C0, const void* in1, int C1, const void* w,
                     const float* ep_scale, const float* ep_shift, void* out, const void* mul, void* pool,
                     int N, int H, int W, int Cout, void* stream);
int bdn_conv3x3_eval_pair(int dtype, const void* in, int C0, const void* w, const float* ep_scale, const float* ep_shift,
                          void* f_out, void* pool, int B, int H, int W, int Cout, void* stream);
int bdn_conv3x3_eval_cls(int dtype, const void* in0, int C0, const void* w, const float* ep_scale, const float* ep_shift, void* act_out,
                         const float* cls_w, const float* cls_b, int ncls, float* logits, uint8_t* mask, const int32_t* origins,
                         int Hs, int Ws, int N, int H, int W, int Cout, void* stream);

/* ---- BatchNorm2d + ReLU backward (autograd of models/unet_parts.py:14-15,17-18) ----
 * g = dA * [z*scale+shift > 0]; sums[g][0][c] = sum g, sums[g][1][c] = sum g*xhat (layout [G][2][C]);
 * dgamma/dbeta (accumulated over groups, overwritten) ; dz = scale*(g - s0/M - xhat*s1/M).
 * dA: [N,H,W,ldA] channel slice starting at dA pointer (ldA = channel stride of the dA tensor).
 * ws: workspace of bdn_bn_bwd_workspace_bytes(). */
size_t bdn_bn_bwd_workspace_bytes(int dtype, int N, int H, int W, int C, int imgs_per_group);
int bdn_bn_bwd(int dtype, const void* dA, int ldA, const void* z, const float* bn,
               int imgs_per_group, int N, int H, int W, int C,
               float* ws, float* sums, float* dgamma, float* dbeta, void* dz, void* stream);

/* Second half of bdn_bn_bwd for callers whose dA producer already emitted the per-tile partial sums (fused
 * BatchNorm-backward statistics): partial is [G][rows_per_group][2][C] holding sum_p g and, with raw_moment != 0,
 * sum_p g*z (converted to sum_p g*xhat in double here), with g = dA * [relu(bn(z)) > 0].  Runs the fixed-order
 * reduction (dgamma, dbeta, sums) and the dz pass.  scratch: NULL or bdn_bn_bwd_scratch_bytes(G, C) bytes; with it,
 * more than 512 rows per group are pre-reduced by many blocks (same result, fixed order either way). */
size_t bdn_bn_bwd_scratch_bytes(int G, int C);
int bdn_bn_bwd_apply(int dtype, const void* dA, int ldA, const void* z, const float* bn,
                     int imgs_per_group, int N, int H, int W, int C,
                     const float* partial, int rows_per_group, int raw_moment,
                     float* sums, float* dgamma, float* dbeta, void* dz, void* scratch, void* stream);
/* bf16x3 setting (float32 tensors): the same, with dz stored directly as the split operand of its two consumers -- dz_split
 * [N,H,W,2 C] bf16 = hi(dz) | lo(dz), bdn_split_pack's layout -- instead of float32: no bdn_split_pack pass over dz. */
int bdn_bn_bwd_apply_split(const void* dA, int ldA, const void* z, const float* bn,
                           int imgs_per_group, int N, int H, int W, int C,
                           const float* partial, int rows_per_group, int raw_moment,
                           float* sums, float* dgamma, float* dbeta, void* dz_split, void* scratch, void* stream);
/* The reduction half of bdn_bn_bwd_apply alone: partial rows -> sums [G][2][C], dgamma, dbeta (same argument meaning),
 * for a consumer that applies the backward while it stages dz (bdn_conv3x3_wgrad_bnbwd). */
int bdn_bn_bwd_finalize(const float* bn, int G, int C, const float* partial, int rows_per_group, int raw_moment,
                        float* sums, float* dgamma, float* dbeta, void* scratch, void* stream);

/* a = relu(z*scale + shift) written out (nn.BatchNorm2d + nn.ReLU, models/unet_parts.py:14-15): z, out [N,H,W,C]; bn [G][4][C].
 * Rounded exactly like the on-load application inside bdn_conv3x3 / bdn_conv3x3_wgrad; bdn_conv3d_wgrad takes plain operands only. */
int bdn_bnrelu(int dtype, const void* z, const float* bn, int imgs_per_group, void* out,
               int N, int H, int W, int C, void* stream);
/* ---- nn.MaxPool2d(2) on relu(bn(z)): models/unet_parts.py:40 (floor mode) ---- */
int bdn_bnrelu_pool(int dtype, const void* z, const float* bn, int imgs_per_group,
                    void* out, int N, int H, int W, int C, void* stream);

/* ---- date fusion torch.relu(x_d2 * x_d1): models/bidate_model.py:35-38 ----
 * z: [2B,H,W,C] raw conv outputs of both dates, bn: [2][4][C]; f: [B,H,W,C]. */
int bdn_fuse_product(int dtype, const void* z, const float* bn, void* f,
                     int B, int H, int W, int C, void* stream);
/* Both of the above in one pass over z (every encoder level but the last needs the skip AND the pooled maps):
 * f [B,H,W,C] = relu(a_d2*a_d1), pool [2B,H/2,W/2,C] = MaxPool2d(2)(a), a = relu(bn(z)), z [2B,H,W,C] date 1 first. */
int bdn_product_pool(int dtype, const void* z, const float* bn, void* f, void* pool,
                     int B, int H, int W, int C, void* stream);
/* The same pass writing the pooled maps of SOME dates only (pool_dates: bit 0 = date 1, bit 1 = date 2; f is always written): the
 * two-chain forward runs the dates of models/bidate_model.py:23-33 on two streams, where the chain of date 2 forms the skip and its own
 * pooled map while date 1's chain pools its own map with bdn_bnrelu_pool (same values, bit for bit). */
int bdn_product_pool_dates(int dtype, const void* z, const float* bn, void* f, void* pool, int pool_dates,
                           int B, int H, int W, int C, void* stream);

/* bf16x3 setting (float32 z): both outputs stored directly as the [hi | lo] bf16 operands of the convolutions that consume them -- f into
 * channels [0, C) of the decoder stage's two-source operand f_split [B,H,W,f_ld] (lo half f_half channels further), pool as
 * pool_split [2B,H/2,W/2,2C] -- bdn_split_pack's layout, without the float32 tensors and the split pass over them. */
int bdn_product_pool_split(const void* z, const float* bn, void* f_split, int f_ld, int f_half, void* pool_split,
                           int B, int H, int W, int C, void* stream);

/* ---- nn.Upsample(scale_factor=2, bilinear, align_corners=True) + F.pad: models/unet_parts.py:56-58,68-72 ----
 * src: [B,h,w,C] (plain, or raw z with bn when in_mode = BDN_IN_BNRELU); out: [B,H,W,C] with the
 * 2h x 2w map placed at offset ((H-2h)/2, (W-2w)/2) and zeros elsewhere. */
int bdn_upsample2x(int dtype, const void* src, int in_mode, const float* bn,
                   void* out, int B, int h, int w, int H, int W, int C, void* stream);
/* bf16x3 setting (float32 src): the upsampled map stored as channels [off, off + C) of the decoder stage's split operand
 * out_split [B,H,W,ld] bf16 (lo half `half` channels further). */
int bdn_upsample2x_split(const void* src, int in_mode, const float* bn, void* out_split, int ld, int off, int half,
                         int B, int h, int w, int H, int W, int C, void* stream);
/* Transpose of the above: dU: [B,H,W,ldU] channel slice -> dsrc: [B,h,w,C]. */
int bdn_upsample2x_bwd(int dtype, const void* dU, int ldU, void* dsrc,
                       int B, int h, int w, int H, int W, int C, void* stream);
/* The same with the BatchNorm-backward partial sums of the layer whose relu(bn(z_prev)) had been upsampled (up's input x1 is the
 * previous double_conv's output, models/unet_parts.py:64-66 after :16-18) fused in: bs_partial f32
 * [bdn_upsample2x_bwd_rows(dtype,B,h,w,C)][2][C] = per block sum g, sum g*z_prev with g = dsrc * [scale*z_prev + shift > 0]
 * (z_prev [B,h,w,C], bn_prev [1][4][C]; one statistic group) -> bdn_bn_bwd_apply(raw_moment = 1); dsrc holds g (masked).  rows() is 0 for shapes the
 * tiled kernel does not take (maps below 8x8, C not a multiple of 32 (bf16) / 16 (f32)): use bdn_upsample2x_bwd + bdn_bn_bwd there. */
int bdn_upsample2x_bwd_rows(int dtype, int B, int h, int w, int C);
int bdn_upsample2x_bwd_bs(int dtype, const void* dU, int ldU, void* dsrc, const void* z_prev, const float* bn_prev,
                          float* bs_partial, int B, int h, int w, int H, int W, int C, void* stream);

/* ---- backward of the date fusion and of MaxPool2d into the encoder outputs ----
 * dF: [B,H,W,ldF] slice; z: [2B,H,W,C], bn [2][4][C]; dP: NULL or [2B,H/2,W/2,C] gradient of the
 * pooled map; dA: [2B,H,W,C] = gradient wrt relu(bn(z)) of each date:
 * dA_d1 = dF * a_d2 + unpool(dP_d1), dA_d2 = dF * a_d1 + unpool(dP_d2)  (first maximum wins ties). */
int bdn_enc_skip_bwd(int dtype, const void* dF, int ldF, const void* z, const float* bn,
                     const void* dP, void* dA, float* bs_partial, int B, int H, int W, int C, void* stream);
/* bs_partial: NULL, or f32 [2][bdn_enc_skip_bwd_rows(dtype,B,H,W,C)][2][C] receiving the BatchNorm-backward partial
 * sums of the layer (per block: sum g, sum g*z; date 1 rows then date 2 rows) -> bdn_bn_bwd_apply(raw_moment = 1); dA then holds
 * the MASKED gradient g = dA * [relu(bn(z)) > 0] (see bdn_conv3x3_dgrad_bs). */
int bdn_enc_skip_bwd_rows(int dtype, int B, int H, int W, int C);

/* ---- outconv: nn.Conv2d(64, n_classes, 1), models/unet_parts.py:86 ----
 * z: [B,H,W,C] raw output of up4's second conv, bn: [1][4][C]; w: [ncls][C] f32, b: [ncls];
 * logits: [B,ncls,H,W] f32 NCHW (the reference's output layout). */
int bdn_outc_fwd(int dtype, const void* z, const float* bn, const float* w, const float* b,
                 float* logits, int B, int H, int W, int C, int ncls, void* stream);
/* dlogits: [B,ncls,H,W] f32 -> dA [B,H,W,C] (wrt relu(bn(z))), dw [ncls][C], db [ncls] (overwritten).
 * ws: bdn_outc_bwd_workspace_bytes() of scratch -- the blocks' partial dw / db, summed in a fixed order (deterministic). */
size_t bdn_outc_bwd_workspace_bytes(int dtype, int B, int H, int W, int C, int ncls);
int bdn_outc_bwd(int dtype, const float* dlogits, const void* z, const float* bn, const float* w,
                 void* dA, float* dw, float* db, float* bs_partial, float* ws, int B, int H, int W, int C, int ncls, void* stream);
/* BatchNorm+ReLU backward of the layer in front of the classifier with the classifier's data gradient recomputed from
 * dlogits (outconv, models/unet_parts.py:83-90, after double_conv's BN+ReLU, :16-18): dz = bn_bwd(round(sum_k dlogits[k] w[k][c]), z)
 * with `sums` from bdn_bn_bwd_finalize.  Call bdn_outc_bwd with dA = NULL (it still leaves the partial sums) and this instead
 * of bdn_bn_bwd_apply: the gradient tensor in between is never written or read.
 * dtype BDN_BF16X3: z float32, dz the split operand [B,H,W,2 C] bf16 = hi | lo of the float32 result (bdn_bn_bwd_apply_split's form). */
int bdn_outc_bn_bwd_apply(int dtype, const float* dlogits, const float* w, const void* z, const float* bn,
                          int imgs_per_group, const float* sums, void* dz, int B, int H, int W, int C, int ncls, void* stream);
/* bs_partial: NULL, or f32 [bdn_outc_bwd_rows(dtype,B,H,W,C)][2][C]: BatchNorm-backward partial sums of the layer that
 * produced z (sum g, sum g*z on the stored dA) -> bdn_bn_bwd_apply(raw_moment = 1, one statistic group). */
int bdn_outc_bwd_rows(int dtype, int B, int H, int W, int C);

/* ---- TverskyLoss.forward, utils/metrics.py:130-171, for [B,H,W] labels (dims == (0,2)) ----
 * labels: uint8 [B,H,W].  ws: bdn_overlap_workspace_bytes(B, ncls, H, W, 0) bytes of f32 scratch (16-byte aligned): the
 * blocks' partial sums are added in a fixed order, no float atomics -- loss and dlogits are the same bits on every run.
 * loss: f32 scalar.  counts: NULL or int32[4] = {TP, FP, FN, correct} of argmax(logits) vs labels
 * (class 1 positive; train.py:96-106).  dlogits: NULL or [B,ncls,H,W] = d loss / d logits. */
size_t bdn_overlap_workspace_bytes(int B, int ncls, int H, int W, int reduce_w);
int bdn_tversky(const float* logits, const uint8_t* labels, float alpha, float beta, float eps,
                float* ws, float* loss, int32_t* counts, float* dlogits,
                int B, int ncls, int H, int W, void* stream);

/* ---- dice_loss / jaccard_loss / TverskyLoss in either label rank, utils/metrics.py:51-171 ----
 * One ratio TP / (TP + alpha FP + beta FN + eps) per reduction cell, averaged:
 *   TverskyLoss(alpha, beta, eps) as is;  jaccard_loss(eps) = (1, 1, eps);  dice_loss(eps) = (0.5, 0.5, eps / 2)
 *   (2I / (2I + FP + FN + eps), utils/metrics.py:80-83).
 * reduce_w = 0: [B,H,W] labels, reference dims == (0,2): one cell per (class, column);
 * reduce_w = 1: [B,1,H,W] labels, dims == (0,2,3): one cell per class.  ws: bdn_overlap_workspace_bytes(.., reduce_w).
 * Other arguments as bdn_tversky. */
int bdn_overlap_loss(const float* logits, const uint8_t* labels, float alpha, float beta, float eps,
                     int reduce_w, float* ws, float* loss, int32_t* counts, float* dlogits,
                     int B, int ncls, int H, int W, void* stream);

/* ---- FocalLoss(gamma, alpha, size_average).forward, utils/metrics.py:8-48 (criterion 'focal', utils/helpers.py:305) ----
 * labels uint8 [B,H,W] (or [B,1,H,W], same memory).  alpha: NULL or f32[ncls] class weights (device).
 * The modulating factor (1-pt)^gamma is a constant for the gradient exactly as in the reference (:35).
 * ws: bdn_focal_workspace_bytes() bytes.  loss, counts, dlogits as bdn_tversky. */
size_t bdn_focal_workspace_bytes(void);
int bdn_focal(const float* logits, const uint8_t* labels, float gamma, const float* alpha, int size_average,
              void* ws, float* loss, int32_t* counts, float* dlogits,
              int B, int ncls, int H, int W, void* stream);

/* ---- OSCD ingest: utils/dataloaders.py:86-111 city_loader, per band ----
 * dst [H][W] f32 (one plane of a [C][H][W] scene) = cv2.resize((src - mean) / std, (W, H)) with cv2's default float
 * INTER_LINEAR sampling (half-pixel centres, border weights (1,0)).  src: [hs][ws] uint16 (src_is_f32 = 0) or f32, on
 * the device.  cv2 is not installed in the build image, so this row's parity is pinned only against the written-out
 * algorithm (oracle/ingest_oracle.py), not against cv2 itself. */
int bdn_ingest_band(int src_is_f32, const void* src, int hs, int ws, float mean, float stdv,
                    float* dst, int H, int W, void* stream);

/* ---- full-scene sliding-window inference: train.py:182-205, utils/inference.py:134-236 ----
 * The scene stays in HBM as band planes scene_d*: [C][H][W] f32.  origins: device int32 [n_tiles][2] = (y0, x0)
 * in the reference's tile order (utils/inference.py:160-184: hs*ws main tiles, lc last-column tiles, lr last-row
 * tiles, corner).  out: [2*n_tiles][p][p][Cpad] packed batch (date-1 tiles first) = the encoder input. */
int bdn_gather_tiles(int dtype, const float* scene_d1, const float* scene_d2, const int32_t* origins,
                     void* out, int n_tiles, int C, int H, int W, int p, int Cpad, void* stream);
/* Rows [r0, r1) of all C band planes of a HOST scene [C][H][W] f32 (pinned for an asynchronous copy) into the resident device planes, as one
 * 2-D copy on `stream` (the host -> device staging of train.py:190-193 / utils/dataloaders.py:86-101, per row band of the scene instead of per batch). */
int bdn_upload_band(float* dst_planes, const float* src_planes_host, int C, int H, int W, int r0, int r1, void* stream);
/* `_, cd_preds = torch.max(preds, 1)` (train.py:199; first maximum wins): logits [n][ncls][H][W] f32 -> uint8 [n][H][W]. */
int bdn_argmax(const float* logits, uint8_t* out, int n, int ncls, int H, int W, void* stream);
/* argmax + _get_bands (utils/inference.py:187-236): class index of every tile pixel written to mask [H][W] uint8 at
 * the tile origin; far-edge tiles own the far-edge bands exactly as the reference's paste order leaves them. */
int bdn_argmax_stitch(const float* logits, const int32_t* origins, uint8_t* mask,
                      int n_tiles, int ncls, int p, int H, int W, void* stream);

/* ---- optim.SGD(lr) step, train.py:55,95: p -= lr * grad_scale * g over a flat f32 buffer ---- */
int bdn_sgd_step(float* params, const float* grads, float lr, float grad_scale, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BIDATE_HIP_H */
