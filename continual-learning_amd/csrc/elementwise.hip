// HBM-bound kernels of the UNet train step: BatchNorm finalise / apply
// (+ fused 2x2 max-pool, + write into a concat slice), the BatchNorm+ReLU backward reductions and apply
// (+ fused max-pool backward routing), layout conversion, per-channel sums.
// All activations are NHWC with an explicit channel pitch (ldc) so a tensor can be a channel slice of a
// concat buffer; every thread moves 8 channels (16 B bf16 / 32 B f32) per pixel, lanes along channels first.
#include "common.hip.h"
#include "clamd_internal.h"

namespace clamd {

// The HBM-bound passes of the critical chain (BatchNorm statistics / apply / backward, pooling) run beside the weight-gradient kernels of the
// second stream, whose waves keep the fp32 MFMA -- and with it the SIMD's vector issue -- busy for 64 cycles per instruction: at equal priority
// the (older) MFMA wave wins the arbitration whenever it is ready and a pass wave issues in what is left.  In the round-5 two-stream trace the
// passes of the main stream took 7.9 ms for 3.6 ms of work (bn_bwd_finalize: 42 us for 7).  They are few instructions per byte: with a raised
// wave priority they take the slots they need when their data arrives, and the MFMA kernel beside them loses a few per cent of its issue slots.

// ------------------------------------------------------------------------------------------------
// The finalize kernels sit on the critical chain between two HBM-bound passes and, in the backward pass, run BESIDE a weight-gradient
// kernel of the second stream that keeps 352-384 of a SIMD's 512 registers (wgrad_dma.hip: 2 waves x 192): a workgroup must fit into what
// is left of a CU or it waits for a weight-gradient workgroup to retire -- the 1024-thread form of round 3 (4 waves x 40 registers per
// SIMD) did, for 35-95 us per launch on 8 launches of a bf16 step (kernel trace, round 4).  256 threads = one 40-register wave per SIMD;
// FIN_CH channels per workgroup (more, smaller workgroups instead of more threads).
constexpr int FIN_THREADS = 256, FIN_CH = 2;
// Fixed-order sum of partial rows [row][NK][Cp] for the FIN_CH channels c0.. of this block: thread (row lane, column) adds its rows in
// ascending order into four interleaved fp64 chains, the row lanes are then added in ascending order.  The result depends only on
// (nrows, data): two runs are bit-identical (no float atomics anywhere).
template <int NK>
__device__ inline void sum_partial_rows(const float* __restrict__ rows, int nrows, int Cp, int c0, double* red, double* out) {
    constexpr int COLS = NK * FIN_CH, RL = FIN_THREADS / COLS;
    const int t = threadIdx.x, col = t % COLS, rl = t / COLS;
    if (rl < RL) {
        const float* p = rows + (size_t)(col / FIN_CH) * Cp + c0 + (col % FIN_CH);
        const size_t rs = (size_t)NK * Cp;
        double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
        int r = rl;
        for (; r + 3 * RL < nrows; r += 4 * RL) {
            a0 += (double)p[(size_t)r * rs]; a1 += (double)p[(size_t)(r + RL) * rs];
            a2 += (double)p[(size_t)(r + 2 * RL) * rs]; a3 += (double)p[(size_t)(r + 3 * RL) * rs];
        }
        for (; r < nrows; r += RL) a0 += (double)p[(size_t)r * rs];
        red[rl * COLS + col] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (t < COLS) {      // out[k * FIN_CH + channel]
        double v = 0.;
        for (int q = 0; q < RL; ++q) v += red[q * COLS + t];
        out[t] = v;
    }
    __syncthreads();
}

// BN finalise: partial rows of (sum, sumsq) -> mean / invstd / scale / shift, running-stat update.  One block per FIN_CH channels.
// Reference: nn.BatchNorm2d train mode, models/unet.py:15 (momentum 0.1, eps 1e-5, unbiased running var).
// Mean and variance are formed in fp64 from the fp64 row sums (E[x^2] - mean^2 cancels in fp32 on low-variance channels).
__global__ void __launch_bounds__(FIN_THREADS) bn_finalize_kernel(const float* __restrict__ stats, int nrows, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean, float* running_var,
                                   float* scale, float* shift, float* save_mean, float* save_istd,
                                   int Cp, int C, double count, double momentum, double eps, long long* num_batches_tracked) {
    PASS_PRIO();
    __shared__ double red[FIN_THREADS], tot[2 * FIN_CH];
    const int c0 = blockIdx.x * FIN_CH;
    if (stats) sum_partial_rows<2>(stats, nrows, Cp, c0, red, tot);
    if (threadIdx.x >= FIN_CH) return;
    if (num_batches_tracked && stats && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;      // nn.BatchNorm2d's counter (train mode)
    const int c = c0 + threadIdx.x;
    double mean, var;
    if (stats) {
        mean = tot[threadIdx.x] / count;
        var = tot[FIN_CH + threadIdx.x] / count - mean * mean;
        var = var > 0. ? var : 0.;
    } else {   // eval mode (trainer.py:271): normalise with the running statistics, update nothing
        mean = c < C ? (double)running_mean[c] : 0.;
        var = c < C ? (double)running_var[c] : 1.;
    }
    const double istd = 1.0 / sqrt(var + eps);
    const double g = c < C ? (double)gamma[c] : 0., b = c < C ? (double)beta[c] : 0.;
    const float sc = (float)(g * istd);
    scale[c] = sc;
    shift[c] = (float)(b - mean * (g * istd));
    save_mean[c] = (float)mean;
    save_istd[c] = (float)istd;
    if (c < C && running_mean && stats) {
        const double unb = count > 1. ? var * (count / (count - 1.)) : var;
        running_mean[c] = (float)((1. - momentum) * (double)running_mean[c] + momentum * mean);
        running_var[c] = (float)((1. - momentum) * (double)running_var[c] + momentum * unb);
    }
}

// ------------------------------------------------------------------------------------------------
// BN apply: out = y*scale + shift (written into `out` with its own pitch: possibly a concat slice), and
// optionally pooled = max over the 2x2 window of `out` (nn.MaxPool2d(2,2), models/unet.py:12,80).
template <typename T, bool POOL>
__global__ void bn_apply_kernel(const T* __restrict__ y, int y_ldc, const float* __restrict__ scale,
                                const float* __restrict__ shift, T* out, int out_ldc, T* pooled, int p_ldc,
                                int B, int H, int W, int Cp) {
    PASS_PRIO();
    const int G = Cp >> 3;
    const long long nitem = POOL ? (long long)B * (H / 2) * (W / 2) * G : (long long)B * H * W * G;
    // G is a power of two <= 256 and the grid stride a multiple of 256: a thread keeps its channel group for the whole
    // loop, so the per-channel constants are loaded ONCE (they were 16-24 extra load instructions per 16-byte item)
    const long long it0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = (int)(it0 % G);
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[cg * 8 + j]; sh[j] = shift[cg * 8 + j]; }
    for (long long it = it0; it < nitem; it += (long long)gridDim.x * blockDim.x) {
        const long long pix = it / G;
        if constexpr (!POOL) {
            float v[8];
            Vec8<T>::load(y + pix * y_ldc + cg * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __fmaf_rn(v[j], sc[j], sh[j]);
            Vec8<T>::store(out + pix * out_ldc + cg * 8, v);
        } else {
            const int w2 = W / 2, h2 = H / 2;
            const int px = (int)(pix % w2), py = (int)((pix / w2) % h2), b = (int)(pix / ((long long)w2 * h2));
            float m[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long long p = ((long long)b * H + 2 * py + (q >> 1)) * W + 2 * px + (q & 1);
                float v[8];
                Vec8<T>::load(y + p * y_ldc + cg * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = __fmaf_rn(v[j], sc[j], sh[j]);
                    m[j] = (q == 0 || v[j] > m[j]) ? v[j] : m[j];
                }
                Vec8<T>::store(out + p * out_ldc + cg * 8, v);
            }
            Vec8<T>::store(pooled + pix * p_ldc + cg * 8, m);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// nn.MaxPool2d(2,2) on its own (models/unet.py:12: the first layer of a DownBlock run as a stand-alone block, blocks.py; inside the
// UNet step the pool is part of bn_apply / bn_bwd_*).  Forward: window maximum; backward: the gradient goes to the FIRST maximum of
// the window (the tie rule of bn_apply_kernel / load_gu and of torch's CPU kernel), zeros elsewhere.
// sign (optional, [Cp]): channels with sign[c] < 0 take the window MINIMUM -- the max-pool of a tensor s * x + t that is never written
// (a BatchNorm folded into its consumers, bnfold.hip) taken on x itself: max(s x + t) = s min(x) + t for s < 0.
template <typename T, bool BWD>
__global__ void __launch_bounds__(256) maxpool2x2_kernel(const T* __restrict__ x, int x_ldc, T* __restrict__ out, int o_ldc,
                                                        const T* __restrict__ gp, int gp_ldc, const float* __restrict__ sign,
                                                        int B, int H, int W, int Cp) {
    PASS_PRIO();
    const int G = Cp >> 3, w2 = W / 2, h2 = H / 2;
    const long long nitem = (long long)B * h2 * w2 * G;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < nitem; it += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(it % G);
        const long long pix = it / G;
        const int px = (int)(pix % w2), py = (int)((pix / w2) % h2), b = (int)(pix / ((long long)w2 * h2));
        float m[8], v[4][8];
        int arg[8];
        bool neg[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) neg[j] = sign != nullptr && sign[cg * 8 + j] < 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long p = ((long long)b * H + 2 * py + (q >> 1)) * W + 2 * px + (q & 1);
            Vec8<T>::load(x + p * x_ldc + cg * 8, v[q]);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (q == 0 || (neg[j] ? v[q][j] < m[j] : v[q][j] > m[j])) { m[j] = v[q][j]; arg[j] = q; }
        }
        if constexpr (!BWD) Vec8<T>::store(out + pix * o_ldc + cg * 8, m);
        else {
            float g[8];
            Vec8<T>::load(gp + pix * gp_ldc + cg * 8, g);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long long p = ((long long)b * H + 2 * py + (q >> 1)) * W + 2 * px + (q & 1);
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = arg[j] == q ? g[j] : 0.f;
                Vec8<T>::store(out + p * o_ldc + cg * 8, o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of [ReLU -> BatchNorm(train)] (+ optional max-pool routing of a second gradient source).
//   g_u  = ga[p]  (+ gp[pool cell] if p is the arg-max of its 2x2 window of u = y*scale+shift)
//   sums: s0 = sum g_u, s1 = sum g_u*y, s2 = sum g_u*[y>0], s3 = sum [y>0], s4 = sum y
//   g_z  = [y>0] * (k0*g_u + k1*y + k2)
// The window arg-max is recomputed bit-identically to bn_apply_kernel (same fmaf, first max wins).
template <typename T>
__device__ inline void load_gu(const T* ga, int ga_ldc, const T* gp, int gp_ldc, const T* y, int y_ldc,
                               const float (&sc)[8], const float (&sh)[8], int b, int py, int px, int H, int W,
                               int cg, float (&g)[4][8], float (&yy)[4][8]) {
    float best[8];
    int arg[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long p = ((long long)b * H + 2 * py + (q >> 1)) * W + 2 * px + (q & 1);
        Vec8<T>::load(y + p * y_ldc + cg * 8, yy[q]);
        if (ga) Vec8<T>::load(ga + p * ga_ldc + cg * 8, g[q]);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) g[q][j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float u = __fmaf_rn(yy[q][j], sc[j], sh[j]);
            if (q == 0 || u > best[j]) { best[j] = u; arg[j] = q; }
        }
    }
    float gpv[8];
    const long long pp = ((long long)b * (H / 2) + py) * (W / 2) + px;
    Vec8<T>::load(gp + pp * gp_ldc + cg * 8, gpv);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) g[q][j] += (arg[j] == q) ? gpv[j] : 0.f;
}

constexpr int NSUM = 5;

template <typename T, bool POOL>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const T* __restrict__ ga, int ga_ldc,
                                                            const T* __restrict__ gp, int gp_ldc,
                                                            const T* __restrict__ y, int y_ldc,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* sums,
                                                            int B, int H, int W, int Cp) {
    PASS_PRIO();
    __shared__ float red[256 * 8];
    const int G = Cp >> 3;
    const int tid = threadIdx.x;
    float acc[NSUM][8];
#pragma unroll
    for (int s = 0; s < NSUM; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[s][j] = 0.f;
    const int cg = tid % G;
    const int rows = 256 / G, prow = tid / G;
    {
        float sc[8], sh[8];
        if constexpr (POOL) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { sc[j] = scale[cg * 8 + j]; sh[j] = shift[cg * 8 + j]; }
        }
        const long long npix = POOL ? (long long)B * (H / 2) * (W / 2) : (long long)B * H * W;
        // !POOL: two pixels per trip -- four 16-byte loads in flight per thread instead of two (bf16 storage: 3.0 -> TB/s of the 4-byte
        // dtypes, whose Vec8 loads are two instructions each); the terms are still added in pixel order
        const long long stride = (long long)gridDim.x * rows;
        for (long long pix = (long long)blockIdx.x * rows + prow; pix < npix; pix += POOL ? stride : 2 * stride) {
            if constexpr (!POOL) {
                float g[2][8], v[2][8];
                const bool two = pix + stride < npix;
                const long long pix1 = two ? pix + stride : pix;
                Vec8<T>::load(ga + pix * ga_ldc + cg * 8, g[0]);
                Vec8<T>::load(y + pix * y_ldc + cg * 8, v[0]);
                Vec8<T>::load(ga + pix1 * ga_ldc + cg * 8, g[1]);
                Vec8<T>::load(y + pix1 * y_ldc + cg * 8, v[1]);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (u == 1 && !two) break;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float pos = v[u][j] > 0.f ? 1.f : 0.f;
                        acc[0][j] += g[u][j]; acc[1][j] += g[u][j] * v[u][j]; acc[2][j] += g[u][j] * pos;
                        acc[3][j] += pos; acc[4][j] += v[u][j];
                    }
                }
            } else {
                const int w2 = W / 2, h2 = H / 2;
                const int px = (int)(pix % w2), py = (int)((pix / w2) % h2), b = (int)(pix / ((long long)w2 * h2));
                float g[4][8], v[4][8];
                load_gu<T>(ga, ga_ldc, gp, gp_ldc, y, y_ldc, sc, sh, b, py, px, H, W, cg, g, v);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float pos = v[q][j] > 0.f ? 1.f : 0.f;
                        acc[0][j] += g[q][j]; acc[1][j] += g[q][j] * v[q][j]; acc[2][j] += g[q][j] * pos;
                        acc[3][j] += pos; acc[4][j] += v[q][j];
                    }
            }
        }
    }
    // block reduction over the threads that share a channel group, one sum kind at a time; this block's partial row
    float* dst = sums + (size_t)blockIdx.x * NSUM * Cp;
#pragma unroll
    for (int s = 0; s < NSUM; ++s) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid * 8 + j] = acc[s][j];
        __syncthreads();
        // channel c's contributions live at red[(r*G + c/8)*8 + c%8], r = 0..rows-1
        for (int c = tid; c < Cp; c += 256) {
            const int g8 = c >> 3, j = c & 7;
            float t = 0.f;
            for (int r = 0; r < rows; ++r) t += red[(r * G + g8) * 8 + j];
            dst[s * Cp + c] = t;
        }
    }
}

// partial rows of the five sums -> k0,k1,k2 per channel, and the parameter gradients d_gamma, d_beta, d_convbias.
// One block per 8 channels; fixed-order fp64 row sums (sum_partial_rows), coefficients formed in fp64.
__global__ void __launch_bounds__(FIN_THREADS) bn_bwd_finalize_kernel(const float* __restrict__ sums, int nrows, const float* __restrict__ gamma,
                                       const float* __restrict__ save_mean, const float* __restrict__ save_istd,
                                       float* k012, float* dgamma, float* dbeta, float* dbias, int Cp, int C,
                                       double count) {
    PASS_PRIO();
    __shared__ double red[FIN_THREADS], tot[NSUM * FIN_CH];
    const int c0 = blockIdx.x * FIN_CH;
    sum_partial_rows<NSUM>(sums, nrows, Cp, c0, red, tot);
    if (threadIdx.x >= FIN_CH) return;
    const int c = c0 + threadIdx.x;
    double s[NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) s[k] = tot[k * FIN_CH + threadIdx.x];
    const double mu = save_mean[c], istd = save_istd[c];
    const double g = c < C ? (double)gamma[c] : 0.;
    const double inv_n = 1. / count;
    const double k0 = g * istd;
    const double c2 = istd * istd * (s[1] * inv_n - mu * s[0] * inv_n);
    const double k1 = -k0 * c2;
    const double k2 = k0 * (mu * c2 - s[0] * inv_n);
    k012[c] = (float)k0; k012[Cp + c] = (float)k1; k012[2 * Cp + c] = (float)k2;
    if (c < C) {
        dgamma[c] = (float)(istd * (s[1] - mu * s[0]));
        dbeta[c] = (float)s[0];
        if (dbias) dbias[c] = (float)(k0 * s[2] + k1 * s[4] + k2 * s[3]);
    }
}

// g_z = k0 g + k1 y + k2 where the ReLU was active: ONE expression (two fused multiply-adds) for every apply kernel, so the variants
// write the same bits
__device__ inline float bn_bwd_gz(float k0, float k1, float k2, float g, float y) { return fmaf(k0, g, fmaf(k1, y, k2)); }

template <typename T, bool POOL>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ ga, int ga_ldc, const T* __restrict__ gp, int gp_ldc,
                                    const T* __restrict__ y, int y_ldc, const float* __restrict__ scale,
                                    const float* __restrict__ shift, const float* __restrict__ k012, T* gz,
                                    int gz_ldc, int B, int H, int W, int Cp) {
    PASS_PRIO();
    const int G = Cp >> 3;
    const long long nitem = POOL ? (long long)B * (H / 2) * (W / 2) * G : (long long)B * H * W * G;
    if constexpr (!POOL) {
        // channel group fixed per thread (see bn_apply_kernel): k0, k1, k2 are loaded once, not per 16-byte item
        // (34 -> 27 us per launch in bf16).  The pooled variant below keeps them per item: hoisted, its 40 extra live
        // registers next to g[4][8], v[4][8] cost occupancy (62 -> 74 us).
        const long long it0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        const int cg = (int)(it0 % G);
        float k0[8], k1[8], k2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            k0[j] = k012[cg * 8 + j]; k1[j] = k012[Cp + cg * 8 + j]; k2[j] = k012[2 * Cp + cg * 8 + j];
        }
        for (long long it = it0; it < nitem; it += (long long)gridDim.x * blockDim.x) {
            const long long pix = it / G;
            float g[8], v[8];
            Vec8<T>::load(ga + pix * ga_ldc + cg * 8, g);
            Vec8<T>::load(y + pix * y_ldc + cg * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = v[j] > 0.f ? bn_bwd_gz(k0[j], k1[j], k2[j], g[j], v[j]) : 0.f;
            Vec8<T>::store(gz + pix * gz_ldc + cg * 8, g);
        }
        return;
    }
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < nitem;
         it += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(it % G);
        const long long pix = it / G;
        float k0[8], k1[8], k2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            k0[j] = k012[cg * 8 + j]; k1[j] = k012[Cp + cg * 8 + j]; k2[j] = k012[2 * Cp + cg * 8 + j];
        }
        if constexpr (!POOL) {
            float g[8], v[8];
            Vec8<T>::load(ga + pix * ga_ldc + cg * 8, g);
            Vec8<T>::load(y + pix * y_ldc + cg * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = v[j] > 0.f ? bn_bwd_gz(k0[j], k1[j], k2[j], g[j], v[j]) : 0.f;
            Vec8<T>::store(gz + pix * gz_ldc + cg * 8, g);
        } else {
            float sc[8], sh[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { sc[j] = scale[cg * 8 + j]; sh[j] = shift[cg * 8 + j]; }
            const int w2 = W / 2, h2 = H / 2;
            const int px = (int)(pix % w2), py = (int)((pix / w2) % h2), b = (int)(pix / ((long long)w2 * h2));
            float g[4][8], v[4][8];
            load_gu<T>(ga, ga_ldc, gp, gp_ldc, y, y_ldc, sc, sh, b, py, px, H, W, cg, g, v);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long long p = ((long long)b * H + 2 * py + (q >> 1)) * W + 2 * px + (q & 1);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    g[q][j] = v[q][j] > 0.f ? bn_bwd_gz(k0[j], k1[j], k2[j], g[q][j], v[q][j]) : 0.f;
                Vec8<T>::store(gz + p * gz_ldc + cg * 8, g[q]);
            }
        }
    }
}

// bn_bwd_apply without the pooling branch, plus the per-channel sum of the g_z it writes (= the gradient of the convolution's bias) as one
// partial row per workgroup: the launches whose producing data-gradient kernel takes only sum g and sum g y (igemm_pws.hip, CLM = 3) get
// the bias gradient here, where g_z exists anyway, instead of from three more running sums in the MFMA kernel's epilogue.
// A thread's channel group is fixed (grid stride % G == 0: G is a power of two <= 256); the 256 / G threads of a group are added in
// thread order through LDS: deterministic.
template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_apply_sums_kernel(const T* __restrict__ ga, int ga_ldc, const T* __restrict__ y, int y_ldc,
                                                                const float* __restrict__ k012, T* gz, int gz_ldc, float* rows,
                                                                long long npix, int Cp) {
    PASS_PRIO();
    __shared__ float red[256 * 8];
    const int G = Cp >> 3, tid = threadIdx.x;
    const long long nitem = npix * G;
    const long long it0 = (long long)blockIdx.x * 256 + tid;
    const int cg = (int)(it0 % G);
    float k0[8], k1[8], k2[8], acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        k0[j] = k012[cg * 8 + j]; k1[j] = k012[Cp + cg * 8 + j]; k2[j] = k012[2 * Cp + cg * 8 + j];
        acc[j] = 0.f;
    }
    for (long long it = it0; it < nitem; it += (long long)gridDim.x * 256) {
        const long long pix = it / G;
        float g[8], v[8];
        Vec8<T>::load(ga + pix * ga_ldc + cg * 8, g);
        Vec8<T>::load(y + pix * y_ldc + cg * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            g[j] = v[j] > 0.f ? bn_bwd_gz(k0[j], k1[j], k2[j], g[j], v[j]) : 0.f;
            acc[j] += g[j];
        }
        Vec8<T>::store(gz + pix * gz_ldc + cg * 8, g);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = acc[j];
    __syncthreads();
    const int per = 256 / G;                       // threads per channel group: tid = q * G + cg
    for (int c = tid; c < Cp; c += 256) {
        const int g8 = c >> 3, j = c & 7;
        float t = 0.f;
        for (int q = 0; q < per; ++q) t += red[(q * G + g8) * 8 + j];
        rows[(size_t)blockIdx.x * Cp + c] = t;
    }
}


// ------------------------------------------------------------------------------------------------
// Per-channel sum of an NHWC tensor (bias gradients of convT / head): every block writes its partial row [Cp] into the
// workspace, channel_sum_final_kernel adds the rows in a fixed order (fp64) and OVERWRITES out[c].
template <typename T>
__global__ void __launch_bounds__(256) channel_sum_kernel(const T* __restrict__ g, int ldc, float* partial,
                                                          long long npix, int Cp) {
    SIDE_PRIO();
    __shared__ float red[256 * 8];
    const int G = Cp >> 3, tid = threadIdx.x;
    const int cg = tid % G, rows = 256 / G, prow = tid / G;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long pix = (long long)blockIdx.x * rows + prow; pix < npix; pix += (long long)gridDim.x * rows) {
        float v[8];
        Vec8<T>::load(g + pix * ldc + cg * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = acc[j];
    __syncthreads();
    for (int c = tid; c < G * 8; c += 256) {
        const int g8 = c >> 3, j = c & 7;
        float t = 0.f;
        for (int r = 0; r < rows; ++r) t += red[(r * G + g8) * 8 + j];
        partial[(size_t)blockIdx.x * Cp + c] = t;
    }
}

__global__ void __launch_bounds__(FIN_THREADS) channel_sum_final_kernel(const float* __restrict__ partial, int nrows, float* out, int Cp, int C) {
    SIDE_PRIO();
    __shared__ double red[FIN_THREADS], tot[FIN_CH];
    const int c0 = blockIdx.x * FIN_CH;
    sum_partial_rows<1>(partial, nrows, Cp, c0, red, tot);
    if (threadIdx.x < FIN_CH && c0 + threadIdx.x < C) out[c0 + threadIdx.x] = (float)tot[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// Layout conversion at the drop-in boundary (visible tensors are NCHW fp32).
// NCHW f32 [B,C,H,W] -> NHWC T [B,H,W,ldc] with channels >= C zero-filled up to Cp.  One thread per pixel.
// One thread per PIXEL, the channel groups in turn: for every channel the 64 lanes read 64 consecutive pixels of one NCHW
// plane (256 contiguous bytes) and a lane writes its pixel's channel vector back to back, so a wave completes each
// 128-byte line of the NHWC tensor at once.  Measured per launch of an fp32 step (nchw_to_nhwc / nchw_im2col3): one
// thread per (pixel, 8 channels) 92 / 117 us; this 67 / 79 us; an LDS-tiled variant with fully contiguous stores 70 / 95 us.
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* dst, int ldc, int B, int C, int H, int W,
                                         int Cp, float mul) {
    const int G = Cp >> 3;
    const long long hw = (long long)H * W, npix = (long long)B * hw;
    for (long long pixb = (long long)blockIdx.x * blockDim.x + threadIdx.x; pixb < npix; pixb += (long long)gridDim.x * blockDim.x) {
        const long long b = pixb / hw, p = pixb % hw;
        for (int cg = 0; cg < G; ++cg) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = cg * 8 + j;
                v[j] = c < C ? src[(b * C + c) * hw + p] * mul : 0.f;
            }
            Vec8<T>::store(dst + pixb * ldc + cg * 8, v);
        }
    }
}

// The same for few channels (the head's gradient, Cp = 32) with H * W a multiple of 4: a thread takes FOUR consecutive pixels -- one 16-byte
// load per channel plane (64 lanes: 1 KB contiguous) instead of four 4-byte ones, CP / 8 x 4 vector stores back to back.
template <typename T, int CP>
__global__ void __launch_bounds__(256) nchw_to_nhwc4_kernel(const float* __restrict__ src, T* dst, int ldc, int B, int C, int H, int W, float mul) {
    const long long hw = (long long)H * W, nq = (long long)B * hw / 4;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long long)gridDim.x * blockDim.x) {
        const long long pix = 4 * q, b = pix / hw, p = pix - b * hw;
        float4 v[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < C) {
                v[c] = *reinterpret_cast<const float4*>(src + (b * C + c) * hw + p);
                v[c].x *= mul; v[c].y *= mul; v[c].z *= mul; v[c].w *= mul;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int cg = 0; cg < CP / 8; ++cg) {
                float o[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float4& t = v[cg * 8 + k];
                    o[k] = j == 0 ? t.x : j == 1 ? t.y : j == 2 ? t.z : t.w;
                }
                Vec8<T>::store(dst + (pix + j) * ldc + cg * 8, o);
            }
    }
}

// First-layer special case (enc1.0, models/unet.py:50: Cin = 3): the 3x3 neighbourhood is gathered ONCE into the
// channel dimension, k = c*9 + (ky*3+kx) < 9*C <= Cp (zero outside the image and for k >= 9*C), so the conv becomes a
// K=32 pointwise GEMM instead of a 9-tap conv over 3 channels padded to 32 (10x wasted MFMA work), and its weight
// gradient a plain [Cout] x [27] pixel contraction whose output IS the [Cout][Cin][3][3] layout.
template <typename T>
__global__ void nchw_im2col3_kernel(const float* __restrict__ src, T* dst, int ldc, int B, int C, int H, int W, int Cp) {
    const int G = Cp >> 3;
    const long long hw = (long long)H * W, npix = (long long)B * hw;
    for (long long pixb = (long long)blockIdx.x * blockDim.x + threadIdx.x; pixb < npix; pixb += (long long)gridDim.x * blockDim.x) {
        const long long b = pixb / hw, p = pixb % hw;
        const int y = (int)(p / W), x = (int)(p % W);
        for (int cg = 0; cg < G; ++cg) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = cg * 8 + j, c = k / 9, t = k - c * 9;
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                v[j] = (c < C && yy >= 0 && yy < H && xx >= 0 && xx < W) ? src[(b * C + c) * hw + (long long)yy * W + xx] : 0.f;
            }
            Vec8<T>::store(dst + pixb * ldc + cg * 8, v);
        }
    }
}

// The same for the network's own first layer (C <= 3, Cp = 32, W a multiple of 4): a thread takes FOUR pixels of a row -- per channel and
// tap row one 16-byte load plus the two neighbours instead of twelve scalar gathers, 4 x 128 bytes (fp32) of output back to back.
template <typename T>
__global__ void __launch_bounds__(256) nchw_im2col3x4_kernel(const float* __restrict__ src, T* dst, int ldc, int B, int C, int H, int W) {
    const long long hw = (long long)H * W, nq = (long long)B * hw / 4;
    const int wq = W / 4;
    // bf16, pitch 32: a lane owns 4 pixels x 64 bytes; its 16 pieces go through LDS (piece P = 16 lane + 4 j + cg at slot P ^ (lane & 7), undone
    // by the reader with (P >> 4) & 7 -- the ce4_kernel exchange, misc.hip) so that one store instruction writes 1 KB of contiguous output
    constexpr bool XCH = __is_same(T, bf16_t);
    __shared__ uint4 xbuf[XCH ? 4 : 1][XCH ? 1024 : 1];
    for (long long base = (long long)blockIdx.x * blockDim.x; base < nq; base += (long long)gridDim.x * blockDim.x) {
        const long long qi = base + threadIdx.x;
        const bool live = qi < nq;                          // block-uniform trip count: the exchange has barriers
        if (!(XCH && ldc == 32) && !live) continue;
        const long long q = live ? qi : nq - 1;
        const int x = (int)(q % wq) * 4;
        const long long row = q / wq;
        const int y = (int)(row % H);
        const long long b = row / H;
        float a[3][3][6];                                   // [channel][tap row][x - 1 .. x + 4]
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = y + ky - 1;
                const bool ok = c < C && yy >= 0 && yy < H;
                const float* r = src + (b * C + (ok ? c : 0)) * hw + (long long)(ok ? yy : 0) * W + x;
                const float4 m = ok ? *reinterpret_cast<const float4*>(r) : make_float4(0.f, 0.f, 0.f, 0.f);
                a[c][ky][0] = (ok && x > 0) ? r[-1] : 0.f;
                a[c][ky][1] = m.x; a[c][ky][2] = m.y; a[c][ky][3] = m.z; a[c][ky][4] = m.w;
                a[c][ky][5] = (ok && x + 4 < W) ? r[4] : 0.f;
            }
        const long long pix = (b * H + y) * W + x;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) {
                float o[8];
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) {
                    const int k = cg * 8 + k8, c = k / 9, t = k - c * 9;        // compile-time after unrolling
                    o[k8] = k < 27 ? a[c < 3 ? c : 0][t / 3][j + t % 3] : 0.f;
                }
                if constexpr (XCH) {
                    if (ldc == 32) {
                        const int P = 16 * lane + 4 * j + cg;
                        xbuf[wv][P ^ (lane & 7)] = make_uint4((unsigned)f2bf(o[0]) | ((unsigned)f2bf(o[1]) << 16), (unsigned)f2bf(o[2]) | ((unsigned)f2bf(o[3]) << 16),
                                                              (unsigned)f2bf(o[4]) | ((unsigned)f2bf(o[5]) << 16), (unsigned)f2bf(o[6]) | ((unsigned)f2bf(o[7]) << 16));
                        continue;
                    }
                }
                Vec8<T>::store(dst + (pix + j) * ldc + cg * 8, o);
            }
        if constexpr (XCH) {
            if (ldc == 32) {
                __syncthreads();
                const long long wave_pix = 4 * (base + 64 * wv);      // first pixel of this wave's 256 (pixels are linear over (b, y, x): W % 4 == 0)
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int P = 64 * it + lane;
                    const long long px = wave_pix + (P >> 2);
                    if (px < 4 * nq) *reinterpret_cast<uint4*>((uint16_t*)dst + px * 32 + (P & 3) * 8) = xbuf[wv][P ^ ((P >> 4) & 7)];
                }
                __syncthreads();
            }
        }
    }
}

// The same tensor with one thread per 8-channel PIECE of a pixel (4 lanes per pixel): every store instruction of a wave writes 16 pixels x 64
// bytes (bf16) = 1 KB of contiguous output, where the four-pixels-per-thread kernel above writes 16 bytes every 256 (64 partial lines per
// instruction: 1.7 TB/s).  The eight values of a piece are gathered with scalar loads -- the input is 12 MB, read nine times, L1/L2-resident.
template <typename T>
__global__ void __launch_bounds__(256) nchw_im2col3p_kernel(const float* __restrict__ src, T* dst, int ldc, int B, int C, int H, int W) {
    const long long hw = (long long)H * W, npiece = (long long)B * hw * 4;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < npiece; it += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(it & 3);
        const long long pix = it >> 2;
        const int x = (int)(pix % W);
        const long long row = pix / W;
        const int y = (int)(row % H);
        const long long b = row / H;
        float o[8];
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
            const int k = cg * 8 + k8, c = k / 9, t = k - c * 9, ky = t / 3, kx = t - ky * 3;
            const int yy = y + ky - 1, xx = x + kx - 1;
            const bool ok = k < 27 && c < C && yy >= 0 && yy < H && xx >= 0 && xx < W;
            o[k8] = ok ? src[(b * C + c) * hw + (long long)yy * W + xx] : 0.f;
        }
        Vec8<T>::store(dst + pix * ldc + cg * 8, o);
    }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, int ldc, float* dst, int B, int C, int H, int W) {
    const long long hw = (long long)H * W, n = (long long)B * C * hw;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < n;
         it += (long long)gridDim.x * blockDim.x) {
        const long long p = it % hw;
        const int c = (int)((it / hw) % C);
        const long long b = it / (hw * C);
        dst[it] = ld1<T>(src + (b * hw + p) * ldc + c);
    }
}

// Grid caps of the two per-channel reductions (clamd_tuning::bn_reduce_blocks / chsum_blocks, 0 = per-launch choice):
// every block ends with one partial row per sum kind that the finalize kernel has to add, so with many channels the tail,
// not the streaming part, sets the time (1024 channels at 16x16: 39.5 us with 2048 blocks, 8.8 us with 256).
static inline long long reduce_grid_cap(int forced, int Cp, int lo, int hi, int budget) {
    if (forced > 0) return forced;
    long long c = budget / Cp;
    return c < lo ? lo : (c > hi ? hi : c);
}

}  // namespace clamd

using namespace clamd;

static inline int ew_grid(long long nitem, int cap = 4096) {
    long long g = (nitem + 255) / 256;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
static bool pow2_channels(int Cp) { return Cp >= 32 && Cp <= 2048 && (Cp & (Cp - 1)) == 0; }

long long clamd_bn_bwd_reduce_rows(int B, int H, int W, int Cp, bool pooled, const clamd_tuning& tn) {
    const int rows = 256 / (Cp / 8) > 0 ? 256 / (Cp / 8) : 1;
    const long long npix = pooled ? (long long)B * (H / 2) * (W / 2) : (long long)B * H * W;
    const long long gb = (npix + rows - 1) / rows;
    const long long cap = reduce_grid_cap(tn.bn_reduce_blocks, Cp, 256, 1024, 131072);
    return gb > cap ? cap : (gb < 1 ? 1 : gb);
}
constexpr int CHSUM_MAX_BLOCKS = 1024;

extern "C" {

int clamd_bn_finalize(const float* stats, int stat_rows, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float* scale, float* shift, float* save_mean, float* save_istd,
                      int Cp, int C, double count, double momentum, double eps, long long* num_batches_tracked, void* stream) {
    if (Cp <= 0 || Cp % 8 || C > Cp) return clamd_fail("bn_finalize: bad channel counts");
    if (stats && stat_rows <= 0) return clamd_fail("bn_finalize: stat_rows must be the row count the producing launch wrote");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(Cp / FIN_CH), dim3(FIN_THREADS), 0, (hipStream_t)stream, stats, stat_rows, gamma,
                       beta, running_mean, running_var, scale, shift, save_mean, save_istd, Cp, C, count, momentum, eps, num_batches_tracked);
    return clamd_check_launch("bn_finalize");
}

int clamd_bn_apply(const void* y, int y_ldc, const float* scale, const float* shift, void* out, int out_ldc,
                   void* pooled, int p_ldc, int B, int H, int W, int Cp, int dtype, void* stream) {
    if (!pow2_channels(Cp)) return clamd_fail("bn_apply: physical channels must be a power of two in [32,2048]");
    if (pooled && ((H | W) & 1)) return clamd_fail("bn_apply: pooling needs even H, W");
    if (int e = clamd_check_split(dtype, y, y_ldc)) return e;
    if (int e = clamd_check_split(dtype, out, out_ldc)) return e;
    if (int e = clamd_check_split(dtype, pooled, p_ldc)) return e;
    const long long nitem = (pooled ? (long long)B * (H / 2) * (W / 2) : (long long)B * H * W) * (Cp / 8);
    dim3 g(ew_grid(nitem, 8192)), b(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, P) hipLaunchKernelGGL((bn_apply_kernel<T, P>), g, b, 0, s, (const T*)y, y_ldc, scale, shift, \
                                        (T*)out, out_ldc, (T*)pooled, p_ldc, B, H, W, Cp)
    if (dtype == CLAMD_BF16) { if (pooled) LAUNCH(bf16_t, true); else LAUNCH(bf16_t, false); }
    else if (dtype == CLAMD_F32) { if (pooled) LAUNCH(float, true); else LAUNCH(float, false); }
    else if (dtype == CLAMD_SPLIT) { if (pooled) LAUNCH(split_t, true); else LAUNCH(split_t, false); }
    else return clamd_fail("bn_apply: bad dtype");
#undef LAUNCH
    return clamd_check_launch("bn_apply");
}

static int launch_maxpool(bool bwd, const void* x, int x_ldc, void* out, int o_ldc, const void* gp, int gp_ldc, const float* sign, int B, int H,
                          int W, int Cp, int dtype, void* stream) {
    if (!x || !out || (bwd && !gp)) return clamd_fail("maxpool2x2: null argument");
    if (B <= 0 || H <= 0 || W <= 0 || ((H | W) & 1)) return clamd_fail("maxpool2x2: needs even H, W");
    if (Cp % 8 || x_ldc < Cp || o_ldc < Cp || (bwd && gp_ldc < Cp)) return clamd_fail("maxpool2x2: bad channel counts / pitches");
    if (int e = clamd_check_split(dtype, x, x_ldc)) return e;
    if (int e = clamd_check_split(dtype, out, o_ldc)) return e;
    if (int e = clamd_check_split(dtype, gp, gp_ldc)) return e;
    const long long nitem = (long long)B * (H / 2) * (W / 2) * (Cp / 8);
    dim3 g(ew_grid(nitem, 8192)), b(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T)                                                                                                                   \
    do {                                                                                                                            \
        if (bwd) hipLaunchKernelGGL((maxpool2x2_kernel<T, true>), g, b, 0, s, (const T*)x, x_ldc, (T*)out, o_ldc, (const T*)gp, gp_ldc, sign, B, H, W, Cp); \
        else hipLaunchKernelGGL((maxpool2x2_kernel<T, false>), g, b, 0, s, (const T*)x, x_ldc, (T*)out, o_ldc, (const T*)nullptr, 0, sign, B, H, W, Cp);   \
    } while (0)
    if (dtype == CLAMD_BF16) LAUNCH(bf16_t);
    else if (dtype == CLAMD_F32) LAUNCH(float);
    else if (dtype == CLAMD_SPLIT) LAUNCH(split_t);
    else return clamd_fail("maxpool2x2: bad dtype");
#undef LAUNCH
    return clamd_check_launch("maxpool2x2");
}

int clamd_maxpool2x2(const void* x, int x_ldc, const float* sign, void* pooled, int p_ldc, int B, int H, int W, int Cp, int dtype, void* stream) {
    return launch_maxpool(false, x, x_ldc, pooled, p_ldc, nullptr, 0, sign, B, H, W, Cp, dtype, stream);
}

int clamd_maxpool2x2_bwd(const void* x, int x_ldc, const float* sign, const void* gp, int gp_ldc, void* gx, int gx_ldc, int B, int H, int W, int Cp,
                         int dtype, void* stream) {
    return launch_maxpool(true, x, x_ldc, gx, gx_ldc, gp, gp_ldc, sign, B, H, W, Cp, dtype, stream);
}

int clamd_bn_bwd_reduce(const void* ga, int ga_ldc, const void* gp, int gp_ldc, const void* y, int y_ldc,
                        const float* scale, const float* shift, float* sums, int sum_rows, int B, int H, int W, int Cp,
                        int dtype, const clamd_tuning* tune, void* stream) {
    if (!pow2_channels(Cp)) return clamd_fail("bn_bwd_reduce: physical channels must be a power of two in [32,2048]");
    if (!gp && !ga) return clamd_fail("bn_bwd_reduce: no gradient source");
    if (int e = clamd_check_split(dtype, ga, ga_ldc)) return e;
    if (int e = clamd_check_split(dtype, gp, gp_ldc)) return e;
    if (int e = clamd_check_split(dtype, y, y_ldc)) return e;
    if (int e = clamd_check_tuning(tune)) return e;
    const long long nrows = clamd_bn_bwd_reduce_rows(B, H, W, Cp, gp != nullptr, clamd_tune(tune));
    if (sum_rows != nrows) return clamd_fail("bn_bwd_reduce: sum_rows does not match clamd_stat_rows(CLAMD_OP_BN_BWD_REDUCE, ...)");
    dim3 g((unsigned)nrows), b(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, P) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, P>), g, b, 0, s, (const T*)ga, ga_ldc, \
                                        (const T*)gp, gp_ldc, (const T*)y, y_ldc, scale, shift, sums, B, H, W, Cp)
    if (dtype == CLAMD_BF16) { if (gp) LAUNCH(bf16_t, true); else LAUNCH(bf16_t, false); }
    else if (dtype == CLAMD_F32) { if (gp) LAUNCH(float, true); else LAUNCH(float, false); }
    else if (dtype == CLAMD_SPLIT) { if (gp) LAUNCH(split_t, true); else LAUNCH(split_t, false); }
    else return clamd_fail("bn_bwd_reduce: bad dtype");
#undef LAUNCH
    return clamd_check_launch("bn_bwd_reduce");
}

int clamd_bn_bwd_finalize(const float* sums, int sum_rows, const float* gamma, const float* save_mean, const float* save_istd,
                          float* k012, float* dgamma, float* dbeta, float* dbias, int Cp, int C, double count,
                          void* stream) {
    if (Cp <= 0 || Cp % 8 || C > Cp) return clamd_fail("bn_bwd_finalize: bad channel counts");
    if (sum_rows <= 0) return clamd_fail("bn_bwd_finalize: sum_rows must be the row count the producing launch wrote");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(Cp / FIN_CH), dim3(FIN_THREADS), 0, (hipStream_t)stream, sums, sum_rows,
                       gamma, save_mean, save_istd, k012, dgamma, dbeta, dbias, Cp, C, count);
    return clamd_check_launch("bn_bwd_finalize");
}

int clamd_bn_bwd_apply(const void* ga, int ga_ldc, const void* gp, int gp_ldc, const void* y, int y_ldc,
                       const float* scale, const float* shift, const float* k012, void* gz, int gz_ldc, int B,
                       int H, int W, int Cp, int dtype, void* stream) {
    if (!pow2_channels(Cp)) return clamd_fail("bn_bwd_apply: physical channels must be a power of two in [32,2048]");
    if (int e = clamd_check_split(dtype, ga, ga_ldc)) return e;
    if (int e = clamd_check_split(dtype, gp, gp_ldc)) return e;
    if (int e = clamd_check_split(dtype, y, y_ldc)) return e;
    if (int e = clamd_check_split(dtype, gz, gz_ldc)) return e;
    const long long nitem = (gp ? (long long)B * (H / 2) * (W / 2) : (long long)B * H * W) * (Cp / 8);
    dim3 g(ew_grid(nitem, 8192)), b(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, P) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, P>), g, b, 0, s, (const T*)ga, ga_ldc, \
                                        (const T*)gp, gp_ldc, (const T*)y, y_ldc, scale, shift, k012, (T*)gz, gz_ldc, B, H, W, Cp)
    if (dtype == CLAMD_BF16) { if (gp) LAUNCH(bf16_t, true); else LAUNCH(bf16_t, false); }
    else if (dtype == CLAMD_F32) { if (gp) LAUNCH(float, true); else LAUNCH(float, false); }
    else if (dtype == CLAMD_SPLIT) { if (gp) LAUNCH(split_t, true); else LAUNCH(split_t, false); }
    else return clamd_fail("bn_bwd_apply: bad dtype");
#undef LAUNCH
    return clamd_check_launch("bn_bwd_apply");
}

static long long apply_sums_rows(long long npix, int Cp) {
    const long long g = (npix * (Cp / 8) + 255) / 256;
    return g < 1 ? 1 : (g > 2048 ? 2048 : g);
}

int clamd_bn_bwd_apply_sums_rows(int B, int H, int W, int Cp) {
    if (B <= 0 || H <= 0 || W <= 0 || !pow2_channels(Cp)) return clamd_fail("bn_bwd_apply_sums_rows: bad sizes");
    return (int)apply_sums_rows((long long)B * H * W, Cp);
}

int clamd_bn_bwd_apply_sums(const void* ga, int ga_ldc, const void* y, int y_ldc, const float* k012, void* gz, int gz_ldc,
                            float* gz_rows, int nrows, int B, int H, int W, int Cp, int dtype, void* stream) {
    if (!pow2_channels(Cp)) return clamd_fail("bn_bwd_apply_sums: physical channels must be a power of two in [32,2048]");
    if (!ga || !y || !k012 || !gz || !gz_rows) return clamd_fail("bn_bwd_apply_sums: null argument");
    if (int e = clamd_check_split(dtype, ga, ga_ldc)) return e;
    if (int e = clamd_check_split(dtype, y, y_ldc)) return e;
    if (int e = clamd_check_split(dtype, gz, gz_ldc)) return e;
    const long long npix = (long long)B * H * W;
    if (nrows != apply_sums_rows(npix, Cp)) return clamd_fail("bn_bwd_apply_sums: nrows must be clamd_bn_bwd_apply_sums_rows(B, H, W, Cp)");
    dim3 g((unsigned)nrows), b(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T) hipLaunchKernelGGL((bn_bwd_apply_sums_kernel<T>), g, b, 0, s, (const T*)ga, ga_ldc, (const T*)y, y_ldc, k012, (T*)gz, gz_ldc, gz_rows, npix, Cp)
    if (dtype == CLAMD_BF16) LAUNCH(bf16_t);
    else if (dtype == CLAMD_F32) LAUNCH(float);
    else if (dtype == CLAMD_SPLIT) LAUNCH(split_t);
    else return clamd_fail("bn_bwd_apply_sums: bad dtype");
#undef LAUNCH
    return clamd_check_launch("bn_bwd_apply_sums");
}

int clamd_rows_sum(const float* rows, int nrows, float* out, int Cp, int C, void* stream) {
    if (!rows || !out || nrows <= 0 || Cp <= 0 || Cp % 8 || C > Cp) return clamd_fail("rows_sum: bad arguments");
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3(Cp / FIN_CH), dim3(FIN_THREADS), 0, (hipStream_t)stream, rows, nrows, out, Cp, C);
    return clamd_check_launch("rows_sum");
}

size_t clamd_channel_sum_workspace_bytes(int Cp) { return (size_t)CHSUM_MAX_BLOCKS * (Cp > 0 ? Cp : 0) * sizeof(float); }

int clamd_channel_sum(const void* g, int ldc, float* out, long long npix, int Cp, int C, int dtype, float* workspace,
                      size_t ws_bytes, const clamd_tuning* tune, void* stream) {
    if (!pow2_channels(Cp)) return clamd_fail("channel_sum: physical channels must be a power of two in [32,2048]");
    if (C > Cp || npix <= 0) return clamd_fail("channel_sum: bad sizes");
    if (int e = clamd_check_split(dtype, g, ldc)) return e;
    if (int e = clamd_check_tuning(tune)) return e;
    const int rows = 256 / (Cp / 8) > 0 ? 256 / (Cp / 8) : 1;
    long long gb = (npix + rows - 1) / rows;
    // partial rows, no atomics: more blocks stream faster (tools/bn_reduce_ab.py: 128 channels @128^2 23.8 us at 256 blocks, 16.1 at 1024)
    const long long cap = reduce_grid_cap(clamd_tune(tune).chsum_blocks, Cp, 256, CHSUM_MAX_BLOCKS, 131072);
    if (gb > cap) gb = cap;
    if (!workspace || (size_t)gb * Cp * sizeof(float) > ws_bytes) return clamd_fail("channel_sum: workspace too small (clamd_channel_sum_workspace_bytes)");
    dim3 gr((unsigned)gb), b(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CLAMD_BF16)
        hipLaunchKernelGGL(channel_sum_kernel<bf16_t>, gr, b, 0, s, (const bf16_t*)g, ldc, workspace, npix, Cp);
    else if (dtype == CLAMD_F32)
        hipLaunchKernelGGL(channel_sum_kernel<float>, gr, b, 0, s, (const float*)g, ldc, workspace, npix, Cp);
    else if (dtype == CLAMD_SPLIT)
        hipLaunchKernelGGL(channel_sum_kernel<split_t>, gr, b, 0, s, (const split_t*)g, ldc, workspace, npix, Cp);
    else return clamd_fail("channel_sum: bad dtype");
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3(Cp / FIN_CH), dim3(FIN_THREADS), 0, s, workspace, (int)gb, out, Cp, C);
    return clamd_check_launch("channel_sum");
}

int clamd_nchw_to_nhwc(const float* src, void* dst, int ldc, int B, int C, int H, int W, int Cp, double mul,
                       int dtype, void* stream) {
    if (Cp % 8 || C > Cp) return clamd_fail("nchw_to_nhwc: bad channel counts");
    if (dtype == CLAMD_SPLIT && Cp % 16) return clamd_fail("nchw_to_nhwc: bf16x3 needs Cp % 16 == 0");
    if (int e = clamd_check_split(dtype, dst, ldc)) return e;
    const long long nitem = (long long)B * H * W;
    dim3 g(ew_grid(nitem, 8192)), b(256);
    hipStream_t s = (hipStream_t)stream;
    if (Cp == 32 && ((long long)H * W) % 4 == 0 && ((size_t)src % 16) == 0) {
        const dim3 g4(ew_grid(nitem / 4, 8192));
#define LAUNCH4(T) hipLaunchKernelGGL((nchw_to_nhwc4_kernel<T, 32>), g4, b, 0, s, src, (T*)dst, ldc, B, C, H, W, (float)mul)
        if (dtype == CLAMD_BF16) LAUNCH4(bf16_t);
        else if (dtype == CLAMD_F32) LAUNCH4(float);
        else if (dtype == CLAMD_SPLIT) LAUNCH4(split_t);
        else return clamd_fail("nchw_to_nhwc: bad dtype");
#undef LAUNCH4
        return clamd_check_launch("nchw_to_nhwc");
    }
#define LAUNCH(T) hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, g, b, 0, s, src, (T*)dst, ldc, B, C, H, W, Cp, (float)mul)
    if (dtype == CLAMD_BF16) LAUNCH(bf16_t);
    else if (dtype == CLAMD_F32) LAUNCH(float);
    else if (dtype == CLAMD_SPLIT) LAUNCH(split_t);
    else return clamd_fail("nchw_to_nhwc: bad dtype");
#undef LAUNCH
    return clamd_check_launch("nchw_to_nhwc");
}

int clamd_nchw_im2col3(const float* src, void* dst, int ldc, int B, int C, int H, int W, int Cp, int dtype, void* stream) {
    if (Cp % 8 || 9 * C > Cp) return clamd_fail("nchw_im2col3: needs 9*C <= Cp, Cp % 8 == 0");
    if (dtype == CLAMD_SPLIT && Cp % 16) return clamd_fail("nchw_im2col3: bf16x3 needs Cp % 16 == 0");
    if (int e = clamd_check_split(dtype, dst, ldc)) return e;
    const long long nitem = (long long)B * H * W;
    dim3 g(ew_grid(nitem, 8192)), b(256);
    hipStream_t s = (hipStream_t)stream;
#ifndef IM2COL_FOUR_PIXELS      // -DIM2COL_FOUR_PIXELS: A/B builds of the four-pixels-per-thread kernel
    // 4-byte storage: the piece kernel (fp32 60 -> 35 us, bf16x3 47 -> 38 us at 16 x 256 x 256); bf16: the four-pixel kernel with its LDS
    // exchange (the piece kernel's 8 scalar gathers per 16 bytes of output cost more than they save there: 42 us against 36)
    if (C <= 3 && Cp == 32 && (dtype != CLAMD_BF16 || W % 4 || ((size_t)src % 16))) {
        const dim3 gp(ew_grid(nitem * 4, 16384));
#define LAUNCHP(T) hipLaunchKernelGGL(nchw_im2col3p_kernel<T>, gp, b, 0, s, src, (T*)dst, ldc, B, C, H, W)
        if (dtype == CLAMD_BF16) LAUNCHP(bf16_t);
        else if (dtype == CLAMD_F32) LAUNCHP(float);
        else if (dtype == CLAMD_SPLIT) LAUNCHP(split_t);
        else return clamd_fail("nchw_im2col3: bad dtype");
#undef LAUNCHP
        return clamd_check_launch("nchw_im2col3");
    }
#endif
    if (C <= 3 && Cp == 32 && W % 4 == 0 && ((size_t)src % 16) == 0) {
        const dim3 g4(ew_grid(nitem / 4, 8192));
#define LAUNCH4(T) hipLaunchKernelGGL(nchw_im2col3x4_kernel<T>, g4, b, 0, s, src, (T*)dst, ldc, B, C, H, W)
        if (dtype == CLAMD_BF16) LAUNCH4(bf16_t);
        else if (dtype == CLAMD_F32) LAUNCH4(float);
        else if (dtype == CLAMD_SPLIT) LAUNCH4(split_t);
        else return clamd_fail("nchw_im2col3: bad dtype");
#undef LAUNCH4
        return clamd_check_launch("nchw_im2col3");
    }
#define LAUNCH(T) hipLaunchKernelGGL(nchw_im2col3_kernel<T>, g, b, 0, s, src, (T*)dst, ldc, B, C, H, W, Cp)
    if (dtype == CLAMD_BF16) LAUNCH(bf16_t);
    else if (dtype == CLAMD_F32) LAUNCH(float);
    else if (dtype == CLAMD_SPLIT) LAUNCH(split_t);
    else return clamd_fail("nchw_im2col3: bad dtype");
#undef LAUNCH
    return clamd_check_launch("nchw_im2col3");
}

int clamd_nhwc_to_nchw(const void* src, int ldc, float* dst, int B, int C, int H, int W, int dtype, void* stream) {
    const long long n = (long long)B * C * H * W;
    dim3 g(ew_grid(n, 8192)), b(256);
    if (int e = clamd_check_split(dtype, src, ldc)) return e;
    if (dtype == CLAMD_BF16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, g, b, 0, (hipStream_t)stream, (const bf16_t*)src, ldc, dst, B, C, H, W);
    else if (dtype == CLAMD_F32)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, g, b, 0, (hipStream_t)stream, (const float*)src, ldc, dst, B, C, H, W);
    else if (dtype == CLAMD_SPLIT)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<split_t>, g, b, 0, (hipStream_t)stream, (const split_t*)src, ldc, dst, B, C, H, W);
    else return clamd_fail("nhwc_to_nchw: bad dtype");
    return clamd_check_launch("nhwc_to_nchw");
}

}  // extern "C"
