// BatchNorm2d (reference models/unet_parts.py:14,17) pieces that are not fused into a convolution:
//   * finalize: per-tile sum / sum^2 partials written by the conv epilogue -> per-(group, channel)
//     mean / invstd / scale / shift table + running-statistics update;
//   * backward of BatchNorm+ReLU: masked reductions (sum g, sum g*xhat), then
//     dz = scale * (g - s0/M - xhat*s1/M).
// HBM-bound streaming kernels: every thread owns one 16-byte channel unit and walks pixels, so the
// per-channel constants live in registers; reductions go through per-block partials in a fixed order
// (deterministic, no atomics) and are summed in double.
#include "common.hpp"

static inline unsigned grid_for(size_t n, int block = 256) { return (unsigned)((n + block - 1) / block); }

// ---------------------------------------------------------------- row reduction helpers
// partial: [n_rows][2][C] f32.  Stage A (many rows): grid (ceil(C/64), G, RS), 256 threads = 16 row lanes x 16 channel
// quads (16-byte loads); block (cb, g, s) sums rows [s*rps, (s+1)*rps) of group g in double -> part2[g][s][2][C].
// Four rows per lane are in flight at a time (the first version walked one row per iteration: 39 us for 8 MB).
__global__ void reduce_rows_kernel(const float* __restrict__ partial, int rows_per_group, int rps, int RS, int C,
                                   double* __restrict__ part2) {
    __shared__ double sm[16][16][8];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4, g = blockIdx.y, s = blockIdx.z;
    const int r0 = s * rps, r1 = min(rows_per_group, r0 + rps);
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = 0.0;
    if (c < C) {
        const float* base = partial + (size_t)g * rows_per_group * 2 * C + c;
        for (int r = r0 + rl; r < r1; r += 64) {
            float4 v[4][2];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int rr = r + 16 * u;
                v[u][0] = v[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rr < r1) {
                    v[u][0] = *reinterpret_cast<const float4*>(base + (size_t)rr * 2 * C);
                    v[u][1] = *reinterpret_cast<const float4*>(base + ((size_t)rr * 2 + 1) * C);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {               // fixed order: row r, r+16, r+32, r+48
                a[0] += v[u][0].x; a[1] += v[u][0].y; a[2] += v[u][0].z; a[3] += v[u][0].w;
                a[4] += v[u][1].x; a[5] += v[u][1].y; a[6] += v[u][1].z; a[7] += v[u][1].w;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) sm[rl][cq][i] = a[i];
    __syncthreads();
    if (rl == 0 && c < C) {
        for (int k = 1; k < 16; k++)
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] += sm[k][cq][i];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            part2[(((size_t)g * RS + s) * 2 + 0) * C + c + i] = a[i];
            part2[(((size_t)g * RS + s) * 2 + 1) * C + c + i] = a[4 + i];
        }
    }
}

struct RowPlan { int RS, rps; };
static inline RowPlan row_plan(int rows_per_group) {
    RowPlan p;
    p.RS = (rows_per_group + 127) / 128;
    if (p.RS > 64) p.RS = 64;
    if (p.RS < 1) p.RS = 1;
    p.rps = (rows_per_group + p.RS - 1) / p.RS;
    p.RS = (rows_per_group + p.rps - 1) / p.rps;
    return p;
}

// Stage B helper: block = 64 channels x 16 s-lanes (1024 threads).  Returns in lane sl == 0 the double sums over the RS
// stage-A partials of (group g, channel c) in a fixed order (lane-strided partial sums, then an LDS tree).
__device__ __forceinline__ void part2_sum(const double* __restrict__ part2, int RS, int g, int C, int c, int sl, int cl,
                                          double (*sm)[64][2], double& a0, double& a1) {
    a0 = 0.0; a1 = 0.0;
    if (c < C)
        for (int s = sl; s < RS; s += 16) {
            a0 += part2[(((size_t)g * RS + s) * 2 + 0) * C + c];
            a1 += part2[(((size_t)g * RS + s) * 2 + 1) * C + c];
        }
    __syncthreads();                                     // sm is reused per group
    sm[sl][cl][0] = a0; sm[sl][cl][1] = a1;
    __syncthreads();
    for (int st = 8; st >= 1; st >>= 1) {
        if (sl < st) { sm[sl][cl][0] += sm[sl + st][cl][0]; sm[sl][cl][1] += sm[sl + st][cl][1]; }
        __syncthreads();
    }
    a0 = sm[0][cl][0]; a1 = sm[0][cl][1];
}

// ---------------------------------------------------------------- finalize (training statistics)
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const double* __restrict__ part2, int RS, int G, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                   float* running_mean, float* running_var, int64_t* nbt, float* __restrict__ bn) {
    __shared__ double sm[16][64][2];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    {
        float rm = (running_mean && c < C) ? running_mean[c] : 0.f, rv = (running_var && c < C) ? running_var[c] : 0.f;
        for (int g = 0; g < G; g++) {                     // sequential: running stats see date 1 then date 2
            double s0, s1;
            part2_sum(part2, RS, g, C, c, sl, cl, sm, s0, s1);
            if (sl != 0 || c >= C) continue;
            const double mean = s0 / count;
            double var = s1 / count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float inv = (float)(1.0 / sqrt(var + (double)eps));
            const float scale = gamma[c] * inv;
            bn[((size_t)g * 4 + 0) * C + c] = (float)mean;
            bn[((size_t)g * 4 + 1) * C + c] = inv;
            bn[((size_t)g * 4 + 2) * C + c] = scale;
            bn[((size_t)g * 4 + 3) * C + c] = beta[c] - (float)mean * scale;
            const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
            rm = (1.f - momentum) * rm + momentum * (float)mean;
            rv = (1.f - momentum) * rv + momentum * (float)unb;
        }
        if (running_mean && sl == 0 && c < C) { running_mean[c] = rm; running_var[c] = rv; }
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += G;
}

// One-launch variant for up to 512 partial rows per group (above that the two-stage path is faster): 1024 threads = 64 row lanes x 16 channels reduce the
// rows in double (fixed-shape tree), then the same finalize arithmetic as bn_finalize_kernel.
__global__ __launch_bounds__(1024) void bn_finalize_direct_kernel(const float* __restrict__ partial, int rows_per_group, int G, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                   float* running_mean, float* running_var, int64_t* nbt, float* __restrict__ bn) {
    __shared__ double sm[64][16][2];
    const int rl = threadIdx.x >> 4, cl = threadIdx.x & 15;
    const int c = blockIdx.x * 16 + cl;
    float rm = (running_mean && c < C) ? running_mean[c] : 0.f, rv = (running_var && c < C) ? running_var[c] : 0.f;
    for (int g = 0; g < G; g++) {                         // sequential: running stats see date 1 then date 2
        double a0 = 0.0, a1 = 0.0;
        if (c < C)
            for (int r = rl; r < rows_per_group; r += 64) {
                const size_t row = (size_t)g * rows_per_group + r;
                a0 += partial[(row * 2 + 0) * C + c];
                a1 += partial[(row * 2 + 1) * C + c];
            }
        sm[rl][cl][0] = a0; sm[rl][cl][1] = a1;
        __syncthreads();
        for (int st = 32; st >= 1; st >>= 1) {
            if (rl < st) { sm[rl][cl][0] += sm[rl + st][cl][0]; sm[rl][cl][1] += sm[rl + st][cl][1]; }
            __syncthreads();
        }
        if (rl == 0 && c < C) {
            const double mean = sm[0][cl][0] / count;
            double var = sm[0][cl][1] / count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float inv = (float)(1.0 / sqrt(var + (double)eps));
            const float scale = gamma[c] * inv;
            bn[((size_t)g * 4 + 0) * C + c] = (float)mean;
            bn[((size_t)g * 4 + 1) * C + c] = inv;
            bn[((size_t)g * 4 + 2) * C + c] = scale;
            bn[((size_t)g * 4 + 3) * C + c] = beta[c] - (float)mean * scale;
            const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
            rm = (1.f - momentum) * rm + momentum * (float)mean;
            rv = (1.f - momentum) * rv + momentum * (float)unb;
        }
        __syncthreads();
    }
    if (rl == 0 && c < C && running_mean) { running_mean[c] = rm; running_var[c] = rv; }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += G;
}

extern "C" size_t bdn_bn_finalize_workspace_bytes(int n_mtiles, int G, int C) {
    if (n_mtiles <= 0 || G <= 0 || C <= 0) return 0;
    return (size_t)G * 64 * 2 * C * sizeof(double);
}

extern "C" int bdn_bn_finalize(const float* stats_partial, int n_mtiles, int G, int C, int count_per_group,
                               const float* gamma, const float* beta, float eps, float momentum,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked,
                               float* bn, void* ws, void* stream) {
    if (!stats_partial || !gamma || !beta || !bn || !ws) BDN_FAIL(BDN_E_ARG, "bn_finalize: null pointer");
    if (G <= 0 || C <= 0 || n_mtiles <= 0 || n_mtiles % G || count_per_group <= 0) BDN_FAIL(BDN_E_SHAPE, "bn_finalize: bad shape");
    if ((running_mean == nullptr) != (running_var == nullptr)) BDN_FAIL(BDN_E_ARG, "bn_finalize: running_mean/var must come together");
    hipStream_t st = (hipStream_t)stream;
    const int rpg = n_mtiles / G;
    if (rpg <= 512) {
        hipLaunchKernelGGL(bn_finalize_direct_kernel, dim3((C + 15) / 16), dim3(1024), 0, st, stats_partial, rpg, G, C, (double)count_per_group,
                           gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, bn);
        BDN_CHECK_LAUNCH("bn_finalize_direct");
        return BDN_OK;
    }
    const RowPlan p = row_plan(rpg);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((C + 63) / 64, G, p.RS), dim3(256), 0, st, stats_partial, rpg, p.rps, p.RS, C, (double*)ws);
    BDN_CHECK_LAUNCH("bn_reduce_rows");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, (const double*)ws, p.RS, G, C, (double)count_per_group,
                       gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, bn);
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

// Eval-mode BatchNorm folded with the bias of the convolution in front of it, for every layer of a network in ONE launch (round 6):
//   scale = gamma / sqrt(running_var + eps),  shift = beta - running_mean * scale     (exactly bn_eval_kernel's values)
//   out[0][c] = scale,  out[1][c] = conv_bias[c] * scale + shift                      (one FMA)
// so that the convolution's epilogue forms relu(acc * out[0] + out[1]) = relu(bn(acc + bias)).  nn.BatchNorm2d.eval() + nn.ReLU,
// models/unet_parts.py:14-15,17-18 on the conv of :13,16.
struct EvalFoldDesc { const float* gamma; const float* beta; const float* rm; const float* rv; const float* bias; float* out; int C, pad_; };
__global__ void bn_eval_fold_kernel(const EvalFoldDesc* __restrict__ desc, float eps) {
    const EvalFoldDesc d = desc[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.C) return;
    const float inv = 1.0f / sqrtf(d.rv[c] + eps);
    const float scale = d.gamma[c] * inv;
    const float shift = d.beta[c] - d.rm[c] * scale;
    d.out[c] = scale;
    d.out[d.C + c] = d.bias ? fmaf(d.bias[c], scale, shift) : shift;
}

extern "C" int bdn_bn_eval_fold_multi(const void* desc, int n_layers, int max_C, float eps, void* stream) {
    if (!desc) BDN_FAIL(BDN_E_ARG, "bn_eval_fold_multi: null pointer");
    if (n_layers <= 0 || max_C <= 0) BDN_FAIL(BDN_E_SHAPE, "bn_eval_fold_multi: n_layers=%d max_C=%d", n_layers, max_C);
    hipLaunchKernelGGL(bn_eval_fold_kernel, dim3((max_C + 255) / 256, n_layers), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const EvalFoldDesc*>(desc), eps);
    BDN_CHECK_LAUNCH("bn_eval_fold_multi");
    return BDN_OK;
}

// ---------------------------------------------------------------- backward (BatchNorm2d + ReLU)
// Thread t owns channel unit t % CU and pixel lane t / CU (CU = C / EPU divides 256); a block covers
// pix_per_block pixels of ONE group.
constexpr int BNB_MAXBLOCKS = 1024; // per-group partial rows (enough blocks to saturate HBM; one-stage finalize below)

template <typename T>
__global__ void bn_bwd_reduce_kernel(const T* __restrict__ dA, int ldA, const T* __restrict__ z, const float* __restrict__ bn,
                                     int pix_per_group, int blocks_per_group, int pix_per_block, int C, float* __restrict__ partial) {
    constexpr int EPU = ET<T>::EPU;
    extern __shared__ float sred[];                           // [256][EPU][2]
    const int CU = C / EPU, rows = 256 / CU;
    const int g = blockIdx.x / blocks_per_group, bg = blockIdx.x % blocks_per_group;
    const int p_begin = bg * pix_per_block, p_end = min(pix_per_group, p_begin + pix_per_block);
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

// per-block partials [G*bpg][2][C] (bpg <= 1024 rows per group) -> sums[g][2][C]; dgamma / dbeta are summed over
// groups.  1024 threads = 64 row lanes x 16 channels, double accumulation, fixed order.
// raw_bn != nullptr: the second partial is the raw moment sum g*z (fused producers: the conv data-gradient epilogue,
// enc_skip_bwd, outc_bwd), converted here in double:  sum g*xhat = invstd * (sum g*z - mean * sum g).
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int blocks_per_group, int G, int C,
                                       float* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       const float* __restrict__ raw_bn) {
    __shared__ double sm[64][16][2];
    const int rl = threadIdx.x >> 4, cl = threadIdx.x & 15;
    const int c = blockIdx.x * 16 + cl;
    double t0 = 0.0, t1 = 0.0;
    for (int g = 0; g < G; g++) {
        double a0 = 0.0, a1 = 0.0;
        if (c < C)
            for (int r = rl; r < blocks_per_group; r += 64) {
                const size_t row = (size_t)g * blocks_per_group + r;
                a0 += partial[(row * 2 + 0) * C + c];
                a1 += partial[(row * 2 + 1) * C + c];
            }
        sm[rl][cl][0] = a0; sm[rl][cl][1] = a1;
        __syncthreads();
        for (int st = 32; st >= 1; st >>= 1) {              // fixed-shape tree over the 64 row lanes
            if (rl < st) { sm[rl][cl][0] += sm[rl + st][cl][0]; sm[rl][cl][1] += sm[rl + st][cl][1]; }
            __syncthreads();
        }
        if (rl == 0 && c < C) {
            a0 = sm[0][cl][0]; a1 = sm[0][cl][1];
            if (raw_bn) a1 = (double)bn_row(raw_bn, g, 1, C)[c] * (a1 - (double)bn_row(raw_bn, g, 0, C)[c] * a0);
            sums[((size_t)g * 2 + 0) * C + c] = (float)a0;
            sums[((size_t)g * 2 + 1) * C + c] = (float)a1;
            t0 += a0; t1 += a1;
        }
        __syncthreads();
    }
    if (rl == 0 && c < C) {
        if (dbeta) dbeta[c] = (float)t0;
        if (dgamma) dgamma[c] = (float)t1;
    }
}

// Second stage for many partial rows (> BNB_DIRECT_ROWS per group): rows were pre-reduced by reduce_rows_kernel into
// part2[g][RS][2][C] doubles; one thread per channel finishes in a fixed order.
__global__ __launch_bounds__(1024) void bn_bwd_finalize_part2_kernel(const double* __restrict__ part2, int RS, int G, int C,
                                             float* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                             const float* __restrict__ raw_bn) {
    __shared__ double sm[16][64][2];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double t0 = 0.0, t1 = 0.0;
    for (int g = 0; g < G; g++) {
        double a0, a1;
        part2_sum(part2, RS, g, C, c, sl, cl, sm, a0, a1);
        if (sl != 0 || c >= C) continue;
        if (raw_bn) a1 = (double)bn_row(raw_bn, g, 1, C)[c] * (a1 - (double)bn_row(raw_bn, g, 0, C)[c] * a0);
        sums[((size_t)g * 2 + 0) * C + c] = (float)a0;
        sums[((size_t)g * 2 + 1) * C + c] = (float)a1;
        t0 += a0; t1 += a1;
    }
    if (sl == 0 && c < C) {
        if (dbeta) dbeta[c] = (float)t0;
        if (dgamma) dgamma[c] = (float)t1;
    }
}

constexpr int BNB_DIRECT_ROWS = 512;   // above this many partial rows per group the reduction is split over blocks first

// partial rows -> sums / dgamma / dbeta.  ws2: bdn_bn_bwd_scratch_bytes(G, C) bytes (doubles), used only for many rows.
static void launch_bn_bwd_finalize(const float* partial, int rows_per_group, int G, int C, float* sums, float* dgamma, float* dbeta,
                                   const float* raw_bn, void* ws2, hipStream_t st) {
    if (rows_per_group <= BNB_DIRECT_ROWS || ws2 == nullptr) {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, st, partial, rows_per_group, G, C, sums, dgamma, dbeta, raw_bn);
        return;
    }
    const RowPlan p = row_plan(rows_per_group);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((C + 63) / 64, G, p.RS), dim3(256), 0, st, partial, rows_per_group, p.rps, p.RS, C, (double*)ws2);
    hipLaunchKernelGGL(bn_bwd_finalize_part2_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, (const double*)ws2, p.RS, G, C, sums, dgamma, dbeta, raw_bn);
}

extern "C" size_t bdn_bn_bwd_scratch_bytes(int G, int C) {
    if (G <= 0 || C <= 0) return 0;
    return (size_t)G * 64 * 2 * C * sizeof(double);
}

// dz = scale * (g - s0/M - xhat * s1/M)
// SPLIT (float32 tensors, bf16x3 setting): dz is not stored as float32 but directly as the [hi | lo] bf16 operand its two consumers (the
// data-gradient conv and the weight-gradient GEMM) take -- out [pixels][2 C] = hi(dz) | lo(dz), the layout of bdn_split_pack -- which
// saves the separate split pass over dz (a float32 read + the same bytes written again) per layer.
template <typename T, bool SPLIT = false>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dA, int ldA, const T* __restrict__ z, const float* __restrict__ bn,
                                    const float* __restrict__ sums, int pix_per_group, int blocks_per_group, int pix_per_block, int C,
                                    T* __restrict__ dz) {
    constexpr int EPU = ET<T>::EPU;
    static_assert(!SPLIT || sizeof(T) == 4, "the split output belongs to the float32 setting");
    const int CU = C / EPU, rows = 256 / CU;
    const int g = blockIdx.x / blocks_per_group, bg = blockIdx.x % blocks_per_group;
    const int p_begin = bg * pix_per_block, p_end = min(pix_per_group, p_begin + pix_per_block);
    const int tid = threadIdx.x, cu = tid % CU, row = tid / CU, c = cu * EPU;
    const float invM = 1.f / (float)pix_per_group;
    float mean[EPU], inv[EPU], sc[EPU], sh[EPU], k0[EPU], k1[EPU];
#pragma unroll
    for (int i = 0; i < EPU; i++) {
        mean[i] = bn_row(bn, g, 0, C)[c + i]; inv[i] = bn_row(bn, g, 1, C)[c + i];
        sc[i] = bn_row(bn, g, 2, C)[c + i]; sh[i] = bn_row(bn, g, 3, C)[c + i];
        k0[i] = sums[((size_t)g * 2 + 0) * C + c + i] * invM;
        k1[i] = sums[((size_t)g * 2 + 1) * C + c + i] * invM;
    }
    for (int p = p_begin + row; p < p_end; p += rows) {
        const size_t pix = (size_t)g * pix_per_group + p;
        float fz[EPU], fg[EPU], o[EPU];
        Unit<T>::unpack(*reinterpret_cast<const uint4*>(z + pix * C + c), fz);
        Unit<T>::unpack(*reinterpret_cast<const uint4*>(dA + pix * ldA + c), fg);
#pragma unroll
        for (int i = 0; i < EPU; i++) {
            const float gm = fmaf(fz[i], sc[i], sh[i]) > 0.f ? fg[i] : 0.f;
            const float xhat = (fz[i] - mean[i]) * inv[i];
            o[i] = sc[i] * (gm - k0[i] - xhat * k1[i]);
        }
        if constexpr (SPLIT) {
            const SplitOut so = {reinterpret_cast<bf16s*>(dz), 2 * C, 0, C};
            store_split4(so, pix, c, o);
        } else {
            *reinterpret_cast<uint4*>(dz + pix * C + c) = Unit<T>::pack(o);
        }
    }
}

// pixels per block: at most BNB_MAXBLOCKS blocks per group, never fewer than two pixel rows of the block
static inline int bnb_pix_per_block(int pix_per_group, int rows) {
    int p = (pix_per_group + BNB_MAXBLOCKS - 1) / BNB_MAXBLOCKS;
    if (p < 2 * rows) p = 2 * rows;
    return (p + rows - 1) / rows * rows;
}

static inline size_t bnb_partial_bytes(size_t blocks, int C) { return (blocks * 2 * C * sizeof(float) + 255) / 256 * 256; }

extern "C" size_t bdn_bn_bwd_workspace_bytes(int dtype, int N, int H, int W, int C, int imgs_per_group) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || imgs_per_group <= 0 || C % 16 || C > 1024) return 0;
    const int epu = dtype == BDN_BF16 ? 8 : 4;
    const int ppg = imgs_per_group * H * W;
    const int ppb = bnb_pix_per_block(ppg, 256 / (C / epu));
    const size_t blocks = (size_t)(N / imgs_per_group) * ((ppg + ppb - 1) / ppb);
    return bnb_partial_bytes(blocks, C) + bdn_bn_bwd_scratch_bytes(N / imgs_per_group, C);
}

template <typename T>
static int bn_bwd_impl(const void* dA, int ldA, const void* z, const float* bn, int imgs_per_group,
                       int N, int H, int W, int C, float* ws, float* sums, float* dgamma, float* dbeta, void* dz, hipStream_t st) {
    constexpr int EPU = ET<T>::EPU;
    const int G = N / imgs_per_group;
    const int ppg = imgs_per_group * H * W;
    const int ppb = bnb_pix_per_block(ppg, 256 / (C / EPU));
    const int bpg = (ppg + ppb - 1) / ppb;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<T>, dim3(G * bpg), dim3(256), 256 * EPU * 2 * sizeof(float), st,
                       (const T*)dA, ldA, (const T*)z, bn, ppg, bpg, ppb, C, ws);
    BDN_CHECK_LAUNCH("bn_bwd_reduce");
    launch_bn_bwd_finalize(ws, bpg, G, C, sums, dgamma, dbeta, nullptr,
                           reinterpret_cast<unsigned char*>(ws) + bnb_partial_bytes((size_t)G * bpg, C), st);
    BDN_CHECK_LAUNCH("bn_bwd_finalize");
    hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(G * bpg), dim3(256), 0, st,
                       (const T*)dA, ldA, (const T*)z, bn, sums, ppg, bpg, ppb, C, (T*)dz);
    BDN_CHECK_LAUNCH("bn_bwd_apply");
    return BDN_OK;
}

template <typename T, bool SPLIT = false>
static int bn_bwd_apply_impl(const void* dA, int ldA, const void* z, const float* bn, int imgs_per_group,
                             int N, int H, int W, int C, const float* partial, int rows_per_group, int raw_moment,
                             float* sums, float* dgamma, float* dbeta, void* dz, void* scratch, hipStream_t st) {
    constexpr int EPU = ET<T>::EPU;
    const int G = N / imgs_per_group;
    const int ppg = imgs_per_group * H * W;
    const int ppb = bnb_pix_per_block(ppg, 256 / (C / EPU));
    const int bpg = (ppg + ppb - 1) / ppb;
    launch_bn_bwd_finalize(partial, rows_per_group, G, C, sums, dgamma, dbeta, raw_moment ? bn : (const float*)nullptr, scratch, st);
    BDN_CHECK_LAUNCH("bn_bwd_finalize");
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, SPLIT>), dim3(G * bpg), dim3(256), 0, st,
                       (const T*)dA, ldA, (const T*)z, bn, sums, ppg, bpg, ppb, C, (T*)dz);
    BDN_CHECK_LAUNCH("bn_bwd_apply");
    return BDN_OK;
}

extern "C" int bdn_bn_bwd_apply(int dtype, const void* dA, int ldA, const void* z, const float* bn,
                                int imgs_per_group, int N, int H, int W, int C,
                                const float* partial, int rows_per_group, int raw_moment,
                                float* sums, float* dgamma, float* dbeta, void* dz, void* scratch, void* stream) {
    if (!dA || !z || !bn || !partial || !sums || !dz) BDN_FAIL(BDN_E_ARG, "bn_bwd_apply: null pointer");
    if (N <= 0 || imgs_per_group <= 0 || N % imgs_per_group || C % 16 || ldA < C || ldA % 16 || rows_per_group <= 0)
        BDN_FAIL(BDN_E_SHAPE, "bn_bwd_apply: bad shape");
    if (C > 1024 || 1024 % C) BDN_FAIL(BDN_E_SHAPE, "bn_bwd_apply: C=%d must divide 1024", C);
    if (dtype == BDN_BF16) return bn_bwd_apply_impl<bf16s>(dA, ldA, z, bn, imgs_per_group, N, H, W, C, partial, rows_per_group, raw_moment, sums, dgamma, dbeta, dz, scratch, (hipStream_t)stream);
    if (dtype == BDN_F32) return bn_bwd_apply_impl<float>(dA, ldA, z, bn, imgs_per_group, N, H, W, C, partial, rows_per_group, raw_moment, sums, dgamma, dbeta, dz, scratch, (hipStream_t)stream);
    BDN_FAIL(BDN_E_ARG, "bn_bwd_apply: bad dtype");
}

// bf16x3 setting: as bdn_bn_bwd_apply on float32 tensors, but dz leaves as the split operand [N,H,W,2 C] bf16 = hi | lo (bdn_split_pack's
// layout) that the data-gradient conv and the weight-gradient GEMM of the layer take: no float32 dz, no split pass over it.
extern "C" int bdn_bn_bwd_apply_split(const void* dA, int ldA, const void* z, const float* bn,
                                      int imgs_per_group, int N, int H, int W, int C,
                                      const float* partial, int rows_per_group, int raw_moment,
                                      float* sums, float* dgamma, float* dbeta, void* dz_split, void* scratch, void* stream) {
    if (!dA || !z || !bn || !partial || !sums || !dz_split) BDN_FAIL(BDN_E_ARG, "bn_bwd_apply_split: null pointer");
    if (N <= 0 || imgs_per_group <= 0 || N % imgs_per_group || C % 16 || ldA < C || ldA % 16 || rows_per_group <= 0)
        BDN_FAIL(BDN_E_SHAPE, "bn_bwd_apply_split: bad shape");
    if (C > 1024 || 1024 % C) BDN_FAIL(BDN_E_SHAPE, "bn_bwd_apply_split: C=%d must divide 1024", C);
    return bn_bwd_apply_impl<float, true>(dA, ldA, z, bn, imgs_per_group, N, H, W, C, partial, rows_per_group, raw_moment, sums, dgamma, dbeta,
                                          dz_split, scratch, (hipStream_t)stream);
}

// The finalize step alone (partial rows -> sums / dgamma / dbeta): for a consumer that applies the backward itself
// (bdn_conv3x3_wgrad_bnbwd).
extern "C" int bdn_bn_bwd_finalize(const float* bn, int G, int C, const float* partial, int rows_per_group, int raw_moment,
                                   float* sums, float* dgamma, float* dbeta, void* scratch, void* stream) {
    if (!bn || !partial || !sums) BDN_FAIL(BDN_E_ARG, "bn_bwd_finalize: null pointer");
    if (G <= 0 || C <= 0 || C % 16 || C > 1024 || 1024 % C || rows_per_group <= 0) BDN_FAIL(BDN_E_SHAPE, "bn_bwd_finalize: bad shape");
    launch_bn_bwd_finalize(partial, rows_per_group, G, C, sums, dgamma, dbeta, raw_moment ? bn : (const float*)nullptr, scratch, (hipStream_t)stream);
    BDN_CHECK_LAUNCH("bn_bwd_finalize");
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
