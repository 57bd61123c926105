// BatchNorm finalize arithmetic shared by the stand-alone finalize kernels (elementwise.hip) and the in-launch tail of the bf16 / fp32-storage
// implicit-GEMM convolutions (igemm_pws.hip, igemm_ws.hip): the LAST workgroup of a launch to finish adds the partial rows the launch wrote
// and produces what clamd_bn_finalize / clamd_bn_bwd_finalize would -- the same fixed-order fp64 sums, the same bits -- so that the 5-7 us
// finalize launch (plus its boundary) leaves the critical chain conv -> finalize -> apply / fold-pack -> conv.  Measured upper bound (all 36
// finalize launches of a step skipped, wrong statistics): bf16 6.38 -> 6.07 ms per step, fp32 20.63 -> 20.51.
//
// Hand-off (cdna_hip_programming.md, in-launch split-K reduction recipe, write-through form): every workgroup writes its rows with sc1 stores
// (st_row), drains them (s_waitcnt vmcnt(0) in every wave, barrier) and lane 0 draws a ticket with a relaxed agent-scope fetch_add; the
// workgroup that draws gridDim.x - 1 issues one agent-scope acquire and reads every row with plain loads.  No spinning: a
// workgroup never waits for another.  The ticket counter is zero on entry and reset by the last workgroup.
#pragma once
#include "common.hip.h"
#include "../../include/clamd.h"

namespace clamd {

typedef ::clamd_bn_tail BnTail;      // include/clamd.h

constexpr int FIN_THREADS = 256, FIN_CH = 2;      // the stand-alone finalize kernels: 256 threads, two channels per workgroup
constexpr int NSUM = 5;                           // sum kinds of the BatchNorm-backward rows

// One row lane of the fixed-order sum: rows q, q + RL, ... of one (sum kind, channel) column in four interleaved fp64 chains.  `p` points at the
// column in row 0, `rs` is the row stride in floats.  RL = FIN_THREADS / (NK * FIN_CH) row lanes exist per column; they are added in ascending order.
template <int NK>
__device__ inline double fin_lane_sum(const float* __restrict__ p, size_t rs, int nrows, int q) {
    constexpr int RL = FIN_THREADS / (NK * FIN_CH);
    double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
    int r = q;
    for (; r + 3 * RL < nrows; r += 4 * RL) {
        a0 += (double)p[(size_t)r * rs]; a1 += (double)p[(size_t)(r + RL) * rs];
        a2 += (double)p[(size_t)(r + 2 * RL) * rs]; a3 += (double)p[(size_t)(r + 3 * RL) * rs];
    }
    for (; r < nrows; r += RL) a0 += (double)p[(size_t)r * rs];
    return (a0 + a1) + (a2 + a3);
}

// nn.BatchNorm2d train mode from the two sums of a channel (models/unet.py:15: momentum 0.1, eps 1e-5, unbiased running variance); eval mode
// (have_sums false) normalises with the running statistics and updates nothing.  Mean and variance in fp64 (E[x^2] - mean^2 cancels in fp32).
__device__ inline void bn_finalize_channel(int c, int C, bool have_sums, double s1, double s2, const float* gamma, const float* beta,
                                           float* running_mean, float* running_var, float* scale, float* shift, float* save_mean,
                                           float* save_istd, double count, double momentum, double eps) {
    double mean, var;
    if (have_sums) {
        mean = s1 / count;
        var = s2 / count - mean * mean;
        var = var > 0. ? var : 0.;
    } else {
        mean = c < C ? (double)running_mean[c] : 0.;
        var = c < C ? (double)running_var[c] : 1.;
    }
    const double istd = 1.0 / sqrt(var + eps);
    const double g = c < C ? (double)gamma[c] : 0., b = c < C ? (double)beta[c] : 0.;
    scale[c] = (float)(g * istd);
    shift[c] = (float)(b - mean * (g * istd));
    save_mean[c] = (float)mean;
    save_istd[c] = (float)istd;
    if (c < C && running_mean && have_sums) {
        const double unb = count > 1. ? var * (count / (count - 1.)) : var;
        running_mean[c] = (float)((1. - momentum) * (double)running_mean[c] + momentum * mean);
        running_var[c] = (float)((1. - momentum) * (double)running_var[c] + momentum * unb);
    }
}

// k0, k1, k2 of g_z = k0 g + k1 y + k2 and the parameter gradients from the five sums of a channel (sum g, sum g y, sum g [y > 0], sum [y > 0], sum y)
__device__ inline void bn_bwd_finalize_channel(int c, int C, int Cp, const double* s, const float* gamma, const float* save_mean,
                                               const float* save_istd, float* k012, float* dgamma, float* dbeta, float* dbias, double count) {
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

// A partial-row element as a WRITE-THROUGH (sc1) store: visible to every XCD once the storing wave's vmcnt reaches zero, so the hand-off needs no
// release fence -- an agent-scope release in each of 256 workgroups that have just written megabytes of activations writes back the whole L2 256
// times (measured: +50 us per launch, the bf16 step 6.39 -> 7.65 ms).
typedef __attribute__((address_space(1))) unsigned int gu32_t;
__device__ inline void st_row(float* p, float v) {
    __hip_atomic_store((gu32_t*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// True in every thread of the workgroup that drew the last ticket; its loads then see the rows of every workgroup of the launch.  `flag` is one int
// of an LDS array the kernel already owns (free at this point).  Called by ALL threads of every workgroup, after the rows were stored.
__device__ inline bool bn_tail_last(unsigned int* ticket, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {      // the rows were stored write-through (st_row) and drained above: no release fence
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = t == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    const bool last = *flag != 0;
    if (last) {
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    return last;
}

// The finalize itself, run by the NT threads of the last workgroup.  ONE workgroup has to add rows x kinds x channels values (32 K floats in the
// persistent kernel) behind an L2 invalidate, so what matters is memory-level parallelism, not instruction count: a thread owns one (kind,
// channel) column and one of S row slices (rows slice, slice + S, ...), fetches up to 64 of its rows with independent loads BEFORE it adds the
// first one, adds them in ascending row order in fp64, and the S slices of a column are then added in ascending order.  (The first version
// walked the stand-alone kernels' lane order with one dependent load per add: 50 us per launch.)  A fixed order, so two runs agree bit for bit;
// it is NOT the order of clamd_bn_finalize / clamd_bn_bwd_finalize (four interleaved chains per row lane): against those the sums agree to
// fp64 rounding and the fp32 results to an ulp.
// NKE: leading sum kinds that are added (the persistent kernel's two-sum rows carry zeros in kinds 2-4), NK: kinds per row.
template <int NKE, int NK, int NT>
__device__ inline void bn_tail_finalize(const clamd_bn_tail& t, const float* __restrict__ rows, int nrows, int Cp, double* scratch) {
    constexpr int RB = 64;                         // rows in flight per thread
    const int tid = threadIdx.x;
    const int cols = NKE * Cp;                     // (kind, channel) columns; kind-major like the rows
    const size_t rs = (size_t)NK * Cp;
    double* tot = scratch;                         // [NKE * Cp]
    double* part = scratch + cols;                 // [S][cols] when a column is split over S > 1 threads (cols < NT)
    int S = cols < NT ? NT / cols : 1;
    if (S > 8) S = 8;
    for (int col0 = 0; col0 < cols; col0 += NT) {
        const int col = col0 + (S > 1 ? tid % cols : tid), slice = S > 1 ? tid / cols : 0;
        const bool mine = col < cols && slice < S;
        double acc = 0.;
        for (int r0 = 0; r0 < nrows; r0 += RB * S) {
            float buf[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int r = r0 + slice + i * S;
                buf[i] = (mine && r < nrows) ? rows[(size_t)r * rs + col] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) acc += (double)buf[i];
        }
        if (S > 1) { if (mine) part[slice * cols + col] = acc; }
        else if (mine) tot[col] = acc;
    }
    __syncthreads();
    if (S > 1) {
        for (int col = tid; col < cols; col += NT) {
            double v = 0.;
            for (int q = 0; q < S; ++q) v += part[q * cols + col];
            tot[col] = v;
        }
        __syncthreads();
    }
    for (int c = tid; c < Cp; c += NT) {
        if (t.kind == 1)
            bn_finalize_channel(c, t.C, true, tot[c], tot[Cp + c], t.gamma, t.beta, t.running_mean, t.running_var, t.scale, t.shift,
                                t.save_mean, t.save_istd, t.count, t.momentum, t.eps);
        else {
            double s[NSUM];
#pragma unroll
            for (int k = 0; k < NSUM; ++k) s[k] = k < NKE ? tot[k * Cp + c] : 0.;
            bn_bwd_finalize_channel(c, t.C, Cp, s, t.gamma, t.save_mean, t.save_istd, t.k012, t.dgamma, t.dbeta, t.dbias, t.count);
        }
    }
    if (tid == 0) {
        if (t.kind == 1 && t.num_batches_tracked) *t.num_batches_tracked += 1;
        *t.ticket = 0u;                                   // the next launch on this stream finds it zero
    }
}
// scratch doubles: NKE * Cp totals + up to 8 slices of at most NT columns
__host__ __device__ constexpr int bn_tail_scratch_doubles(int cp_max, int nt) { return NSUM * cp_max + 8 * nt; }
constexpr int BN_TAIL_MAX_CHANNELS = 1024;      // wider launches get the finalize kernel behind them (clamd_conv3x3_tail)

}  // namespace clamd
