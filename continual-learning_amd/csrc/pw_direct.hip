// nn.ConvTranspose2d(k2, s2) forward and data gradient on the exact-fp32 path (models/unet.py:34; loss.backward(),
// trainer.py:175) as a register-blocked GEMM whose operands go straight from global memory into the MFMA operand registers.
//
// igemm_kernel<float, PW / UP2> stages 64-byte K-steps of both operands through LDS for a 256-pixel x 64-channel tile per
// 4-wave workgroup (two workgroups per CU): 0.61-0.67 of the fp32 MFMA pipe.  The two GEMMs have no halo and no reuse inside a
// workgroup that the caches cannot provide, so -- as in wino24g.hip -- every WAVE owns a tile of its own and loads fragments
// itself: a lane reads 16 bytes = 4 consecutive k of its row (activations, [pixel][K], K innermost) and of its column (packed
// weights, [N][K], K innermost), four fp32 MFMAs per pair of loads into one 32 x 32 accumulator (mma16<float>).  Tile = (32 MT)
// pixels x 128 output columns: MT + 4 loads per 16 MT MFMAs (MT = 4: 8 loads per 64 MFMAs = 4096 matrix-pipe cycles), 64 MT
// accumulator registers, no LDS, no barrier, no VALU in the loop.  A lane's loads of four consecutive 8-k chunks cover 64
// contiguous bytes (half-wave h takes bytes [64h, 64h + 64) of each 128-byte line of a 32-k block), so every line is
// consumed by consecutive instructions of one wave (L1) although a single instruction touches 32 lines; the k ORDER inside a
// block differs from igemm_kernel's (same products, other summation order: not bit-identical to it, same 2e-5 bound).
//   MODE 0  forward:        y[(b, 2y+dy, 2x+dx), co] = bias[co] + sum_ci x[(b,y,x), ci] * w[q = 2dy+dx][co][ci]     N = 4 Cout_p, K = Cin_p
//   MODE 1  data gradient:  gx[(b,y,x), ci] = sum_(q,co) gy[(b, 2y+dy, 2x+dx), co] * w[ci][q][co]                    N = Cin_p, K = 4 Cout_p
#include <algorithm>
#include "common.hip.h"
#include "clamd_internal.h"
#include "igemm_common.hip.h"

namespace clamd {

struct PwDirectParams {
    const float* a; int a_ldc;       // activations: x [B,H,W,ldc] (forward) or gy [B,2H,2W,ldc] (data gradient)
    const float* w;                  // packed weights [Np][Kp], K innermost
    const float* bias;               // forward: [Cout_p]
    float* y; int y_ldc;             // forward: [B,2H,2W,ldc] (channel slice), data gradient: [B,H,W,ldc]
    int B, H, W;                     // COARSE pixel grid (the ConvTranspose input)
    int Kp, Np, Cout_p;
    int mtiles, ntiles;
};

constexpr int PWD_RING = 4;          // 8-k chunks in flight (one 32-k block: a lane's 64 contiguous bytes)

template <int MT, int MODE>
__global__ void __launch_bounds__(256, 1) pw_direct_kernel(const PwDirectParams p) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    // work item = one wave's tile; the four waves of a workgroup take consecutive column tiles of one row tile (shared A lines)
    const int item = xcd_remap(blockIdx.x, gridDim.x) * 4 + wv;
    if (item >= p.mtiles * p.ntiles) return;                                  // whole wave; the kernel has no barrier
    const int tn = item % p.ntiles, tm = item / p.ntiles;
    const long long M = (long long)p.B * p.H * p.W;
    const int m0 = tm * 32 * MT, n0 = tn * 128;

    const int fw = 2 * p.W;                                                    // fine grid width
    unsigned a_vo[MT];
    unsigned fine_px[MT];                        // fine-grid pixel of (row r of row block mi, tap 0): the epilogue fetches it by shuffle
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const unsigned m = (unsigned)m0 + 32u * mi + r;                        // 32-bit: M < 2^31 (checked on the host)
        const unsigned xy = m / (unsigned)p.W, x = m - xy * (unsigned)p.W, b = xy / (unsigned)p.H, y = xy - b * (unsigned)p.H;
        fine_px[mi] = (b * 2u * p.H + 2u * y) * (unsigned)fw + 2u * x;
        if ((long long)m >= M) { a_vo[mi] = BUF_OOB; continue; }
        if constexpr (MODE == 0) a_vo[mi] = m * (unsigned)p.a_ldc * 4u + h * 64u;
        else a_vo[mi] = fine_px[mi] * (unsigned)p.a_ldc * 4u + h * 64u;
    }
    const unsigned b_vo = (unsigned)(((long long)(n0 + r) * p.Kp) * 4 + h * 64);
    const unsigned b_nstep = (unsigned)p.Kp * 32u * 4u;                        // 32 columns further
    const size_t a_bytes = (size_t)(MODE == 0 ? M : 4 * M) * p.a_ldc * 4;
    const __amdgpu_buffer_rsrc_t ars = make_rsrc(p.a, (unsigned)a_bytes), brs = make_rsrc(p.w, (unsigned)((size_t)p.Np * p.Kp * 4));
    const __amdgpu_buffer_rsrc_t ars_dead = make_rsrc(p.a, 0u), brs_dead = make_rsrc(p.w, 0u);
    const int nchunks = p.Kp >> 3;                                             // multiple of 4 (Kp % 32 == 0)
    const int cpb = p.Cout_p >> 5;                                             // 32-k blocks per tap (data gradient)

    f32x16 acc[MT][4];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    uint4 A[PWD_RING][MT], Bq[PWD_RING][4];
    auto load = [&](int s, int c) {                  // 8-k chunk c (wave-uniform) into ring set s; past the end: zeros, no traffic
        const bool live = c < nchunks;
        const int blk = c >> 2, q4 = c & 3;
        unsigned a_so;
        if constexpr (MODE == 0) a_so = (unsigned)(blk * 128 + q4 * 16);
        else {
            const int q = blk / cpb, cb = blk - q * cpb;                        // tap (dy, dx) and 32-channel block inside it
            a_so = (unsigned)((((q >> 1) * fw + (q & 1)) * p.a_ldc + cb * 32) * 4 + q4 * 16);
        }
        const unsigned b_so = (unsigned)(blk * 128 + q4 * 16);
        const __amdgpu_buffer_rsrc_t ar = live ? ars : ars_dead, br = live ? brs : brs_dead;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) A[s][mi] = buf_ld16(ar, a_vo[mi], a_so);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) Bq[s][ni] = buf_ld16(br, b_vo, b_so + (unsigned)ni * b_nstep);
    };
#pragma unroll
    for (int s = 0; s < PWD_RING; ++s) {
        load(s, s);
        asm volatile("" ::: "memory");               // ring order (see wino24g_wgrad_kernel: keeps the loop's waits counted)
    }
    for (int c0 = 0; c0 < nchunks; c0 += PWD_RING) {
#pragma unroll
        for (int s = 0; s < PWD_RING; ++s) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) mma16<float>(A[s][mi], Bq[s][ni], acc[mi][ni]);
            load(s, c0 + s + PWD_RING);
            __builtin_amdgcn_sched_group_barrier(0x008, 16 * MT, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, MT + 4, 0);
        }
    }

    // ---- epilogue: lane = output column (32 consecutive channels = 128 contiguous bytes per row and half-wave) ----------------
    if constexpr (MODE == 0) {
        int tap_off[4], co[4];                   // per column block: the tap's pixel offset in the fine grid and the channel
        float bv[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + 32 * ni + r;
            const int q = n / p.Cout_p;                                        // uniform over the 32 lanes (32 | Cout_p)
            co[ni] = n - q * p.Cout_p;
            tap_off[ni] = (q >> 1) * fw + (q & 1);
            bv[ni] = p.bias ? p.bias[co[ni]] : 0.f;
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = acc_row(e, h);
                const unsigned fine = (unsigned)__shfl((int)fine_px[mi], row);    // computed by the lane whose A row this is
                if ((long long)m0 + 32 * mi + row >= M) continue;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    p.y[(size_t)(fine + (unsigned)tap_off[ni]) * p.y_ldc + co[ni]] = acc[mi][ni][e] + bv[ni];
            }
    } else {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long long m = (long long)m0 + 32 * mi + acc_row(e, h);
                if (m >= M) continue;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) p.y[m * p.y_ldc + n0 + 32 * ni + r] = acc[mi][ni][e];
            }
    }
}

// 0 = launched, -1 = shape not supported (the caller falls back to igemm_kernel), < -1 = launch error
template <int MODE>
static int launch_pw_direct_mode(PwDirectParams p, hipStream_t s) {
    const long long M = (long long)p.B * p.H * p.W;
    if (p.Np % 128 || p.Kp % 32 || p.Cout_p % 32 || p.a_ldc % 32) return -1;   // a row's 128-byte lines must be whole
    if ((MODE == 0 ? M : 4 * M) * p.a_ldc * 4 >= (1ll << 32) || (long long)p.Np * p.Kp * 4 >= (1ll << 32) || 4 * M >= (1ll << 31)) return -1;
    p.ntiles = p.Np / 128;
    // the largest row tile that still gives every SIMD of the chip a wave
    const long long want = 4LL * clamd_num_cus();
    int mt = 4;
    while (mt > 1 && ((M + 32 * mt - 1) / (32 * mt)) * p.ntiles < want) mt >>= 1;
    p.mtiles = (int)((M + 32 * mt - 1) / (32 * mt));
    const long long items = (long long)p.mtiles * p.ntiles;
    if (items > 0x7fffffff) return -1;
    const unsigned grid = (unsigned)((items + 3) / 4);
#define PWD_LAUNCH(MT_) hipLaunchKernelGGL((pw_direct_kernel<MT_, MODE>), dim3(grid), dim3(256), 0, s, p)
    if (mt == 4) PWD_LAUNCH(4); else if (mt == 2) PWD_LAUNCH(2); else PWD_LAUNCH(1);
#undef PWD_LAUNCH
    const int e = clamd_check_launch(MODE == 0 ? "convT2x2_fwd (direct)" : "convT2x2_dgrad (direct)");
    return e ? e - 1 : 0;
}

int launch_pw_direct_convT_fwd(const float* x, int x_ldc, const float* w, const float* bias, float* y, int y_ldc, int B, int h, int w_,
                               int Cin_p, int Cout_p, hipStream_t s) {
    PwDirectParams p{x, x_ldc, w, bias, y, y_ldc, B, h, w_, Cin_p, 4 * Cout_p, Cout_p, 0, 0};
    return launch_pw_direct_mode<0>(p, s);
}

int launch_pw_direct_convT_dgrad(const float* gy, int gy_ldc, const float* w, float* gx, int gx_ldc, int B, int h, int w_, int Cin_p,
                                 int Cout_p, hipStream_t s) {
    PwDirectParams p{gy, gy_ldc, w, nullptr, gx, gx_ldc, B, h, w_, 4 * Cout_p, Cin_p, Cout_p, 0, 0};
    return launch_pw_direct_mode<1>(p, s);
}

}  // namespace clamd

using namespace clamd;

extern "C" {

int clamd_convT2x2_fwd_direct(const float* x, int x_ldc, const float* w_packed, const float* bias, float* y, int y_ldc, int B,
                              int h, int w, int Cin_p, int Cout_p, void* stream) {
    if (B <= 0 || h <= 0 || w <= 0 || !x || !w_packed || !y) return clamd_fail("convT2x2_fwd_direct: bad arguments");
    const int e = launch_pw_direct_convT_fwd(x, x_ldc, w_packed, bias, y, y_ldc, B, h, w, Cin_p, Cout_p, (hipStream_t)stream);
    if (e == -1) return clamd_fail("convT2x2_fwd_direct: needs Cin_p % 32 == 0, Cout_p % 32 == 0, x_ldc % 32 == 0 and operands below 2^32 bytes");
    return e;
}

int clamd_convT2x2_dgrad_direct(const float* gy, int gy_ldc, const float* w_packed, float* gx, int gx_ldc, int B, int h, int w,
                                int Cin_p, int Cout_p, void* stream) {
    if (B <= 0 || h <= 0 || w <= 0 || !gy || !w_packed || !gx) return clamd_fail("convT2x2_dgrad_direct: bad arguments");
    const int e = launch_pw_direct_convT_dgrad(gy, gy_ldc, w_packed, gx, gx_ldc, B, h, w, Cin_p, Cout_p, (hipStream_t)stream);
    if (e == -1) return clamd_fail("convT2x2_dgrad_direct: needs Cin_p % 128 == 0, Cout_p % 32 == 0, gy_ldc % 32 == 0 and operands below 2^32 bytes");
    return e;
}

}  // extern "C"
