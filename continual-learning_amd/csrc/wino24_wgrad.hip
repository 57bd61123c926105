// Winograd F(2x4, 3x3) weight gradient for the exact-fp32 path on gfx950 (loss.backward(), trainer.py:175; the
// convolutions of models/unet.py:13,16,...): dU[xi][r][c] = sum over 2x4 tiles of (A4 dY A6^T)[xi][tile][r] * (B4^T d B6)[xi][tile][c],
// then dg = G4^T dU G6.  24 multiply-adds per 8 output pixels instead of 32 (wino.hip's F(2x2,3x3) weight gradient) or 72
// (direct sum); fp32 error against fp64 1.5-3e-6 (F(2x2): 0.8-1.8e-6), held to the same 2e-5 test bound.
//
// Structure of wino_wgrad_kernel (wino.hip): one 256-thread workgroup per CU, wave w owns the vertical Winograd row i = w
// and forms BOTH transformed operands in registers from one float per lane (lanes = channels: conflict-free ds_read_b32),
// the MFMA contraction index is the tile.  With six horizontal positions j per row the accumulators of a 64 x 64 channel
// block would be 6 x 4 x 16 = 384 registers, so the block is 64 (r: channels of gz) x 32 (c: channels of x): 192
// accumulators, 12 MFMAs per k-step (two tiles) beside 42 transform VALU operations and 28 LDS reads.
// Pixel tile = 4 x 4 Winograd tiles = 8 x 16 output pixels (8 k-steps); gz tile [128 px][64 ch] and x halo [10 x 18 px][32 ch]
// staged as fp32 images, two stages.  Slabs are [split][i][r][c][8] (j = 0..5 + 2 pad: two 16-byte stores per element);
// wino24_wgrad_reduce_kernel sums the splits in a fixed order and applies G4^T . G6 (24 -> 9 taps).
#include <string.h>
#include "common.hip.h"
#include "clamd_internal.h"
#include "wino_common.hip.h"

namespace clamd {

struct Wino24WgradParams {
    const float* a; int a_ldc;       // gz  [B,H,W,a_ldc]
    const float* b; int b_ldc;       // x   [B,H,W,b_ldc]
    float* partial;                  // [nsplit][4][Rp][Cp][8]
    int B, H, W, Rp, Cp, nsplit, tiles_per_split;
};

constexpr int V4_TY = 4, V4_TX = 4;                                   // Winograd tiles (2 x 4 px) per pixel tile
constexpr int V4_PH = 2 * V4_TY, V4_PW = 4 * V4_TX;                   // 8 x 16 output pixels
constexpr int V4_APIX = V4_PH * V4_PW;                                // 128 gz pixels
constexpr int V4_BW = V4_PW + 2, V4_BH = V4_PH + 2, V4_BPIX = V4_BW * V4_BH;   // 18 x 10 = 180 halo pixels
constexpr int V4_APS = 256, V4_BPS = 128;                             // bytes per pixel: 64 / 32 fp32 channels
constexpr int V4_ABYTES = V4_APIX * V4_APS, V4_STAGE = V4_ABYTES + V4_BPIX * V4_BPS;   // 32 KB + 22.5 KB
constexpr int V4_NJA = V4_APIX * 16 / 256, V4_NJB = (V4_BPIX * 8 + 255) / 256;        // 8 + 6 staging loads per thread
constexpr int V4_STEPS = V4_TY * V4_TX / 2;                           // 8 k-steps (two tiles each) per pixel tile

template <bool RAGGED>
__global__ void __launch_bounds__(256, 1) wino24_wgrad_kernel(const Wino24WgradParams p) {
    static_assert(2 * V4_STAGE <= 160 * 1024, "two stages must fit one CU");
    __shared__ __attribute__((aligned(16))) char smem[2 * V4_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, kh = lane >> 5;

    const int rt = (p.Rp + 63) >> 6, ct = (p.Cp + 31) >> 5;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tr = bid % rt; bid /= rt;
    const int tc = bid % ct; bid /= ct;
    const int split = bid;
    const int r0 = tr * 64, c0 = tc * 32;
    const int tiles_x = (p.W + V4_PW - 1) / V4_PW, tiles_y = (p.H + V4_PH - 1) / V4_PH;
    const int ntiles = tiles_x * tiles_y * p.B;
    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(ntiles, t_begin + p.tiles_per_split);
    const int n = max(t_end - t_begin, 0);

    const unsigned a_img = (unsigned)p.H * p.W * p.a_ldc * 4u, b_img = (unsigned)p.H * p.W * p.b_ldc * 4u;
    const unsigned b_shift = (unsigned)(p.W + 1) * p.b_ldc * 4u;      // descriptor base sits one row + one pixel early
    uint4 ra[V4_NJA], rb[V4_NJB];
    // tile-invariant staging offsets (!RAGGED) and the border masks of the halo pieces (see wino_wgrad_kernel)
    unsigned a_vo[RAGGED ? 1 : V4_NJA], b_vo[RAGGED ? 1 : V4_NJB];
    unsigned mT = 0, mB = 0, mL = 0, mR = 0;
    if constexpr (!RAGGED) {
        const int ga = tid & 15, gb = tid & 7;
#pragma unroll
        for (int j = 0; j < V4_NJA; ++j) {
            const int pix = (tid >> 4) + 16 * j, py = pix / V4_PW, px = pix % V4_PW;
            a_vo[j] = r0 + 4 * ga < p.Rp ? (unsigned)(((py * p.W + px) * p.a_ldc + r0 + 4 * ga) * 4) : BUF_OOB;
        }
#pragma unroll
        for (int j = 0; j < V4_NJB; ++j) {
            int pix = (tid >> 3) + 32 * j;
            if (pix >= V4_BPIX) pix -= V4_BPIX;
            const int hy = pix / V4_BW, hx = pix % V4_BW;
            b_vo[j] = c0 + 4 * gb < p.Cp ? (unsigned)(((hy * p.W + hx) * p.b_ldc + c0 + 4 * gb) * 4) : BUF_OOB;
            mT |= (hy == 0 ? 1u : 0u) << j; mB |= (hy == V4_BH - 1 ? 1u : 0u) << j;
            mL |= (hx == 0 ? 1u : 0u) << j; mR |= (hx == V4_BW - 1 ? 1u : 0u) << j;
        }
    }
    auto gload = [&](int tile, bool live) {            // A: 16 lanes = the 256 bytes of one pixel; B: 8 lanes = 128 bytes
        const int x0 = (tile % tiles_x) * V4_PW, y0 = ((tile / tiles_x) % tiles_y) * V4_PH, b = live ? tile / (tiles_x * tiles_y) : 0;
        const unsigned a_so = (unsigned)((y0 * p.W + x0) * p.a_ldc) * 4u, b_so = (unsigned)((y0 * p.W + x0) * p.b_ldc) * 4u;
        if constexpr (!RAGGED) {
            // a dead load (past the last tile of this split) gets an EMPTY descriptor: every lane out of range, no traffic
            const __amdgpu_buffer_rsrc_t ars2 = make_rsrc((const char*)p.a + (size_t)b * a_img, live ? a_img : 0u);
            const __amdgpu_buffer_rsrc_t brs2 = make_rsrc((const char*)p.b + (size_t)b * b_img - b_shift, live ? b_img + b_shift : 0u);
            const unsigned em = (y0 == 0 ? mT : 0u) | (y0 + V4_PH == p.H ? mB : 0u) | (x0 == 0 ? mL : 0u) | (x0 + V4_PW == p.W ? mR : 0u);
#pragma unroll
            for (int j = 0; j < V4_NJA; ++j) ra[j] = buf_ld16(ars2, a_vo[j], a_so);
#pragma unroll
            for (int j = 0; j < V4_NJB; ++j) rb[j] = buf_ld16(brs2, (em >> j) & 1u ? BUF_OOB : b_vo[j], b_so);
            return;
        }
        const __amdgpu_buffer_rsrc_t ars = make_rsrc((const char*)p.a + (size_t)b * a_img, a_img);
        const __amdgpu_buffer_rsrc_t brs = make_rsrc((const char*)p.b + (size_t)b * b_img - b_shift, b_img + b_shift);
        const int ga = tid & 15, gb = tid & 7;
#pragma unroll
        for (int j = 0; j < V4_NJA; ++j) {
            const int pix = (tid >> 4) + 16 * j, py = pix / V4_PW, px = pix % V4_PW;
            const bool ok = live && y0 + py < p.H && x0 + px < p.W && r0 + 4 * ga < p.Rp;
            ra[j] = buf_ld16(ars, ok ? (unsigned)(((py * p.W + px) * p.a_ldc + r0 + 4 * ga) * 4) : BUF_OOB, a_so);
        }
#pragma unroll
        for (int j = 0; j < V4_NJB; ++j) {
            int pix = (tid >> 3) + 32 * j;
            if (pix >= V4_BPIX) pix -= V4_BPIX;                        // the ragged last pass re-stages the first pixels
            const int hy = pix / V4_BW, hx = pix % V4_BW;
            const int yy = y0 + hy - 1, xx = x0 + hx - 1;
            const bool ok = live && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W && c0 + 4 * gb < p.Cp;
            rb[j] = buf_ld16(brs, ok ? (unsigned)(((hy * p.W + hx) * p.b_ldc + c0 + 4 * gb) * 4) : BUF_OOB, b_so);
        }
    };
    auto lds_store = [&](int st) {
        char* sa = smem + st * V4_STAGE;
        char* sb = sa + V4_ABYTES;
        const int ga = tid & 15, gb = tid & 7;
#pragma unroll
        for (int j = 0; j < V4_NJA; ++j) *reinterpret_cast<uint4*>(sa + ((tid >> 4) + 16 * j) * V4_APS + 16 * ga) = ra[j];
#pragma unroll
        for (int j = 0; j < V4_NJB; ++j) {
            int pix = (tid >> 3) + 32 * j;
            if (pix >= V4_BPIX) pix -= V4_BPIX;
            *reinterpret_cast<uint4*>(sb + pix * V4_BPS + 16 * gb) = rb[j];
        }
    };

    // wave row i = w.  A side (gz tile rows p = 0, 1): rc[q] = gA[q] + c1f * g1[q], A4 rows (1,0), (1,1), (1,-1), (0,-1): gA = g0
    // (rows 0-2) or g1 (row 3, c1f = 0: the row comes out as +g1 and the reduce flips plane 3).  B side: t[c] = d[a1][c] + s2 *
    // d[a2][c], row 2 comes out negated and the reduce flips plane 2 back (one fma per element, as in wino_wgrad_kernel).
    const float c1f = w == 1 ? 1.f : (w == 2 ? -1.f : 0.f);
    const int a_rowA = w == 3 ? V4_PW * V4_APS : 0;                     // byte offset of the row that supplies gA
    const int a1 = w == 0 ? 0 : 1, a2 = w == 3 ? 3 : 2;
    const float s2 = w == 1 ? 1.f : -1.f;
    // lane part of every fragment address: tile parity kh -> 4 pixels to the right, channel r (+32 for the second r half)
    const int a_lane = (4 * kh) * V4_APS + r * 4;
    const int b_lane1 = V4_ABYTES + (a1 * V4_BW + 4 * kh) * V4_BPS + r * 4, b_lane2 = V4_ABYTES + (a2 * V4_BW + 4 * kh) * V4_BPS + r * 4;

    f32x16 acc[6][2];                                                  // [j][r half]
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

    gload(t_begin, n > 0);
    if (n > 0) lds_store(0);
    __syncthreads();
    for (int it = 0; it < n; ++it) {
        gload(t_begin + it + 1, it + 1 < n);                           // in flight under this tile's 96 MFMAs
        const char* sm = smem + (it & 1) * V4_STAGE;
        // One MFMA k-step = tiles (ty, 2 s2_ + kh): the 28 raw reads of step s+1 are issued before the first 6 MFMAs of step s
        // and transformed between its two halves (software pipeline for one wave per SIMD, as in wino_wgrad_kernel).
        float rawa[2][2][8];                                           // [set][r half][row A: 4 columns | row 1: 4 columns]
        float rawb[2][12];                                             // [set][row a1: 6 columns | row a2: 6 columns]
        float Yt[2][6][2], V[2][6];
#define V4_LOAD(s_, z_)                                                                                            \
    do {                                                                                                           \
        constexpr int ty_ = (s_) / (V4_TX / 2), tx0_ = 2 * ((s_) % (V4_TX / 2));                                   \
        constexpr int apix_ = (2 * ty_) * V4_PW + 4 * tx0_, bpix_ = (2 * ty_) * V4_BW + 4 * tx0_;                  \
        _Pragma("unroll") for (int hf_ = 0; hf_ < 2; ++hf_) {                                                      \
            const char* ap_ = sm + a_lane + apix_ * V4_APS + 128 * hf_;                                            \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                     \
                rawa[z_][hf_][q_] = *reinterpret_cast<const float*>(ap_ + a_rowA + q_ * V4_APS);                   \
                rawa[z_][hf_][4 + q_] = *reinterpret_cast<const float*>(ap_ + (V4_PW + q_) * V4_APS);              \
            }                                                                                                      \
        }                                                                                                          \
        const char* bp1_ = sm + b_lane1 + bpix_ * V4_BPS;                                                          \
        const char* bp2_ = sm + b_lane2 + bpix_ * V4_BPS;                                                          \
        _Pragma("unroll") for (int bc_ = 0; bc_ < 6; ++bc_) {                                                      \
            rawb[z_][bc_] = *reinterpret_cast<const float*>(bp1_ + bc_ * V4_BPS);                                  \
            rawb[z_][6 + bc_] = *reinterpret_cast<const float*>(bp2_ + bc_ * V4_BPS);                              \
        }                                                                                                          \
    } while (0)
        // A6 columns (A6 = [[1,0,0,0],[1,1,1,1],[1,-1,1,-1],[1,2,4,8],[1,-2,4,-8],[0,0,0,1]]): y0 = g0 | y1,2 = (g0+g2) +- (g1+g3) |
        // y3,4 = (g0+4g2) +- 2(g1+4g3) | y5 = g3.   B6^T as in wino24.hip.
#define V4_TRANSFORM(z_)                                                                                           \
    do {                                                                                                           \
        _Pragma("unroll") for (int hf_ = 0; hf_ < 2; ++hf_) {                                                      \
            const float g0_ = fmaf(c1f, rawa[z_][hf_][4], rawa[z_][hf_][0]), g1_ = fmaf(c1f, rawa[z_][hf_][5], rawa[z_][hf_][1]); \
            const float g2_ = fmaf(c1f, rawa[z_][hf_][6], rawa[z_][hf_][2]), g3_ = fmaf(c1f, rawa[z_][hf_][7], rawa[z_][hf_][3]); \
            const float sa_ = g0_ + g2_, sb_ = g1_ + g3_, sc_ = fmaf(4.f, g2_, g0_), sd_ = fmaf(4.f, g3_, g1_);     \
            Yt[z_][0][hf_] = g0_; Yt[z_][1][hf_] = sa_ + sb_; Yt[z_][2][hf_] = sa_ - sb_;                          \
            Yt[z_][3][hf_] = fmaf(2.f, sd_, sc_); Yt[z_][4][hf_] = fmaf(-2.f, sd_, sc_); Yt[z_][5][hf_] = g3_;     \
        }                                                                                                          \
        float t_[6];                                                                                               \
        _Pragma("unroll") for (int bc_ = 0; bc_ < 6; ++bc_) t_[bc_] = fmaf(s2, rawb[z_][6 + bc_], rawb[z_][bc_]);  \
        const float pq_ = fmaf(-4.f, t_[2], t_[4]), qq_ = fmaf(-4.f, t_[1], t_[3]);                                \
        const float rr_ = t_[4] - t_[2], ss_ = t_[3] - t_[1];                                                      \
        V[z_][0] = fmaf(4.f, t_[0], fmaf(-5.f, t_[2], t_[4]));                                                     \
        V[z_][1] = pq_ + qq_; V[z_][2] = pq_ - qq_;                                                                \
        V[z_][3] = fmaf(2.f, ss_, rr_); V[z_][4] = fmaf(-2.f, ss_, rr_);                                           \
        V[z_][5] = fmaf(4.f, t_[1], fmaf(-5.f, t_[3], t_[5]));                                                     \
    } while (0)
#define V4_MMA(z_, j0_, j1_)                                                                                       \
    do {                                                                                                           \
        _Pragma("unroll") for (int j_ = (j0_); j_ < (j1_); ++j_)                                                   \
            _Pragma("unroll") for (int rh_ = 0; rh_ < 2; ++rh_)                                                    \
                acc[j_][rh_] = __builtin_amdgcn_mfma_f32_32x32x2f32(Yt[z_][j_][rh_], V[z_][j_], acc[j_][rh_], 0, 0, 0); \
    } while (0)
#define V4_STEP(s_)                                                                                                \
    do {                                                                                                           \
        if constexpr ((s_) + 1 < V4_STEPS) V4_LOAD((s_) + 1, ((s_) + 1) & 1);                                      \
        __builtin_amdgcn_sched_barrier(0);                 /* keep the reads in front of these 6 MFMAs ... */      \
        V4_MMA((s_) & 1, 0, 3);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                 /* ... and their consumers behind them */               \
        if constexpr ((s_) + 1 < V4_STEPS) V4_TRANSFORM(((s_) + 1) & 1);                                           \
        V4_MMA((s_) & 1, 3, 6);                                                                                    \
        sched_mfma_slots<6, 0, 0, 0, 0, 0, 8>();           /* the ~44 transform VALU ops in the gaps of these 6 MFMAs */ \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    } while (0)
        V4_LOAD(0, 0);
        V4_TRANSFORM(0);
        V4_STEP(0); V4_STEP(1); V4_STEP(2); V4_STEP(3); V4_STEP(4); V4_STEP(5); V4_STEP(6); V4_STEP(7);
        static_assert(V4_STEPS == 8, "8 k-steps per pixel tile");
#undef V4_STEP
#undef V4_MMA
#undef V4_TRANSFORM
#undef V4_LOAD
        if (it + 1 < n) lds_store((it + 1) & 1);                       // that stage was released by the last barrier
        __syncthreads();
    }

    // ---- slab: plane i = w, [r][c][8]: lane (col c = r_lane, rows acc_row(e, kh) + 32*rh) stores j = 0..3 and j = 4,5 (+2 pad)
    float* const plane = p.partial + (((size_t)split * 4 + w) * p.Rp + r0) * (size_t)p.Cp * 8;
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
        const int col = c0 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 32 * rh + acc_row(e, kh);
            if (r0 + row < p.Rp && col < p.Cp) {
                float* d = plane + ((size_t)row * p.Cp + col) * 8;
                *reinterpret_cast<float4*>(d) = make_float4(acc[0][rh][e], acc[1][rh][e], acc[2][rh][e], acc[3][rh][e]);
                *reinterpret_cast<float2*>(d + 4) = make_float2(acc[4][rh][e], acc[5][rh][e]);
            }
        }
    }
}

// out[rl][cl][3][3] = G4^T (sum_s dU_s) G6.  256 threads = 16 (r,c) pairs x 4 planes x 4 split-phases (fixed order: deterministic).
struct Wino24ReduceParams {
    const float* partial; float* out;
    int nsplit, Rp, Cp, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p;
};

__global__ void __launch_bounds__(256) wino24_wgrad_reduce_kernel(const Wino24ReduceParams p) {
    SIDE_PRIO();
    __shared__ float red[4][4][16][6];                                 // [phase][plane][pair][j]
    const int pr = threadIdx.x & 15, pl = (threadIdx.x >> 4) & 3, ph = threadIdx.x >> 6;
    const long long npair = (long long)p.Rp * p.Cp;
    const size_t plane_sz = (size_t)npair * 8, split_sz = plane_sz * 4;
    for (long long base = (long long)blockIdx.x * 16; base < npair; base += (long long)gridDim.x * 16) {
        const long long e = base + pr;
        float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (e < npair)
            for (int k = ph; k < p.nsplit; k += 4) {
                const float* q = p.partial + (size_t)k * split_sz + pl * plane_sz + (size_t)e * 8;
                const float4 v = *reinterpret_cast<const float4*>(q);
                const float2 u = *reinterpret_cast<const float2*>(q + 4);
                s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w; s[4] += u.x; s[5] += u.y;
            }
#pragma unroll
        for (int j = 0; j < 6; ++j) red[ph][pl][pr][j] = s[j];
        __syncthreads();
        if (threadIdx.x < 16 && e < npair) {
            float U[4][6];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) U[i][j] = (red[0][i][pr][j] + red[1][i][pr][j]) + (red[2][i][pr][j] + red[3][i][pr][j]);
            // signs left out of the kernel's row transforms: B side row 2 negated, A side row 3 negated
#pragma unroll
            for (int j = 0; j < 6; ++j) { U[2][j] = -U[2][j]; U[3][j] = -U[3][j]; }
            // rows: t = G4^T U, G4 = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
            float t[3][6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                t[0][j] = U[0][j] + 0.5f * (U[1][j] + U[2][j]);
                t[1][j] = 0.5f * (U[1][j] - U[2][j]);
                t[2][j] = U[3][j] + 0.5f * (U[1][j] + U[2][j]);
            }
            const int cp = (int)(e % p.Cp), rp = (int)(e / p.Cp);
            const int rl = wn_phys2log(rp, p.r_seg0, p.r_seg0p, p.R), cl = wn_phys2log(cp, p.c_seg0, p.c_seg0p, p.C);
            if (rl >= 0 && cl >= 0) {
                float* o = p.out + ((size_t)rl * p.C + cl) * 9;
                // columns: dg = t G6, G6 = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float s12 = t[a][1] + t[a][2], d12 = t[a][2] - t[a][1], s34 = t[a][3] + t[a][4], d34 = t[a][3] - t[a][4];
                    o[a * 3 + 0] = 0.25f * t[a][0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
                    o[a * 3 + 1] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
                    o[a * 3 + 2] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + t[a][5];
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace clamd

using namespace clamd;

static int w24wg_nsplit(int B, int H, int W, int Rp, int Cp, const clamd_tuning& tn, int* per_out) {
    const int rt = (Rp + 63) / 64, ct = (Cp + 31) / 32;
    const int ntiles = ((W + V4_PW - 1) / V4_PW) * ((H + V4_PH - 1) / V4_PH) * B;
    int nsplit = (int)((long long)(tn.wgrad_blocks / 2) * clamd_usable_cus(tn) / clamd_num_cus()) / (rt * ct);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > ntiles) nsplit = ntiles;
    const int per = (ntiles + nsplit - 1) / nsplit;
    if (per_out) *per_out = per;
    return (ntiles + per - 1) / per;
}

extern "C" {

size_t clamd_wgrad_winograd24_workspace_bytes(int Rp, int Cp) {
    int nsplit = 512 / (((Rp + 63) / 64) * ((Cp + 31) / 32));      // upper bound over every value of wgrad_blocks (<= 1024)
    if (nsplit < 1) nsplit = 1;
    return (size_t)nsplit * 4 * 8 * Rp * Cp * sizeof(float);
}

int clamd_wgrad_winograd24(const float* gz, int gz_ldc, const float* x, int x_ldc, float* workspace, size_t ws_bytes, float* out,
                           int B, int H, int W, int Rp, int Cp, int R, int C, int r_seg0, int r_seg0p, int c_seg0, int c_seg0p,
                           const clamd_tuning* tune, void* stream) {
    if (int e = clamd_check_tuning(tune)) return e;
    if (B <= 0 || H <= 0 || W <= 0) return clamd_fail("wgrad_winograd24: empty problem");
    if ((H & 1) || (W & 3)) return clamd_fail("wgrad_winograd24: H must be even and W a multiple of 4");
    if (Rp % 32 || Cp % 32 || gz_ldc % 8 || x_ldc % 8) return clamd_fail("wgrad_winograd24: channel counts/pitches must be padded");
    if ((long long)H * W * gz_ldc * 4 >= (1ll << 30) || (long long)H * W * x_ldc * 4 >= (1ll << 30)) return clamd_fail("wgrad_winograd24: one image exceeds 2^30 bytes");
    const clamd_tuning& tn = clamd_tune(tune);
    const int rt = (Rp + 63) / 64, ct = (Cp + 31) / 32;
    int per = 0;
    const int nsplit = w24wg_nsplit(B, H, W, Rp, Cp, tn, &per);
    if ((size_t)nsplit * 4 * 8 * Rp * Cp * sizeof(float) > ws_bytes) return clamd_fail("wgrad_winograd24: workspace too small");
    Wino24WgradParams p{gz, gz_ldc, x, x_ldc, workspace, B, H, W, Rp, Cp, nsplit, per};
    hipStream_t s = (hipStream_t)stream;
    if (H % V4_PH || W % V4_PW) hipLaunchKernelGGL(wino24_wgrad_kernel<true>, dim3(rt * ct * nsplit), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(wino24_wgrad_kernel<false>, dim3(rt * ct * nsplit), dim3(256), 0, s, p);
    if (int e = clamd_check_launch("wgrad_winograd24")) return e;
    Wino24ReduceParams rp{workspace, out, nsplit, Rp, Cp, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p};
    long long g = ((long long)Rp * Cp + 15) / 16;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(wino24_wgrad_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, rp);
    return clamd_check_launch("wgrad_winograd24_reduce");
}

}  // extern "C"
