// fp32 3x3 convolution of the NARROW layers (64 / 128 channels, levels 0-1 of models/unet.py:49-72) as a ONE-dimensional Winograd
// F(4,3) along the image row with the three kernel rows summed directly -- forward and data gradient.
//
// Why another form.  The exact-fp32 MFMA issues at the vector-FMA rate, so beside it every VALU / LDS instruction costs matrix time
// (DESIGN.md section 4).  The 2-D F(2x4,3x3) kernels (wino24.hip, wino24g.hip) execute 3 multiply-adds per output but form B^T d B inside
// the K loop of every wave: 72 transform VALU + 24 LDS reads per 48 MFMAs, plus a four-wave exchange in the epilogue -- 0.42-0.58 of the
// matrix pipe on these layers, i.e. 5.2-7.1 pipe-cycles' worth per output.  Pre-transforming (3x the activation bytes) is HBM-bound here.
// The 1-D form executes 4.5 multiply-adds per output (6 per 4 outputs and kernel row) but
//   * its input transform (6 values from 6 pixels of one row) is shared by the three kernel rows AND by all four waves, so it is done
//     ONCE per input element while STAGING: the LDS stage holds V = B6^T d (1.5x the halo), and the K loop is fragment reads + MFMAs only
//     (0.4 other vector instructions per MFMA instead of 2.1);
//   * the output transform A6^T is in-lane: the six Winograd positions of an output segment are six accumulator planes of one lane.
//   Y[y][4s .. 4s+3][co] = A6^T [ sum_{ky, ci} U[j][ky][co][ci] * V[j][y + ky][s][ci] ]_j,   U = G6 g[ky],  V = B6^T d (Lavin & Gray F(4,3))
// -> six GEMMs with K = 3 Cin, M = (row, 4-pixel segment) "super-pixels", N = Cout.  fp32 error against fp64 ~1e-6 like F(2x4).
//
// Workgroup = 256 threads (one wave per SIMD), persistent: tile = 16 x 32 output pixels = 128 super-pixels x 64 output channels; wave w owns
// rows 4w .. 4w+3 (32 super-pixels), 6 planes x 2 channel blocks x 16 = 192 accumulators.  Per 8-channel K-chunk the stage holds
// V [j][halo row 18][segment 8][8 ch] (27.6 KB) and the filter slab [j*3+ky][64 co][8 ch] (36.9 KB); two stages.  The MFMA takes the FILTER
// fragment as its row operand (igemm_pws.hip, channels in the lane): a lane owns one super-pixel and two runs of 8 consecutive channels per
// 32-channel block, so the outputs leave as 16-byte stores, the bias is the initial value of plane 1 (the one column of A6^T that is all
// ones) and the statistics are running sums per accumulator register.
#include <algorithm>
#include <type_traits>
#include "common.hip.h"
#include "clamd_internal.h"
#include "wino_common.hip.h"

namespace clamd {

#ifdef CLAMD_DIAG
// diagnostic build only (python build.py --variant diag --diag; tools/w41_diag.py): cycles per phase, summed over workgroups (wave 0, lane 0)
__device__ unsigned long long g_w41_diag[8];
#define W41_T() __builtin_amdgcn_s_memtime()
#define W41_ADD(i_, v_) do { if (threadIdx.x == 0) atomicAdd(&g_w41_diag[i_], (unsigned long long)(v_)); } while (0)
#else
#define W41_T() 0ull
#define W41_ADD(i_, v_) do { } while (0)
#endif

constexpr int W41_TH = 16, W41_TW = 32, W41_SEG = W41_TW / 4;
constexpr int W41_HH = W41_TH + 2;
// LDS images in 16-byte slots, the two 4-channel groups of a chunk in separate planes so that the 32 lanes of a fragment read walk
// consecutive slots (conflict-free ds_read_b128); the planes are 4 slots out of phase so that staging stores, whose lanes alternate
// between the planes, hit disjoint banks:   V [group][j][halo row][segment]      filters [group][j * 3 + ky][output channel]
constexpr int W41_XG = 6 * W41_HH * W41_SEG + 4, W41_XS = 2 * W41_XG;
constexpr int W41_WG = 18 * 64 + 4, W41_WS = 2 * W41_WG;
constexpr int W41_STAGE = W41_XS + W41_WS;

template <int N, int I = 0, typename F>
__device__ inline void w41_unroll(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        w41_unroll<N, I + 1>(f);
    }
}

// EPI 0: plain (data gradient)   1: bias + ReLU + statistics rows   2: the same with the bias from a border-class table (folded BatchNorm)
template <bool RAGGED, int EPI>
__global__ void __launch_bounds__(256, 1) wino41_kernel(const WinoParams p, const int gm) {
    static_assert(2 * W41_STAGE * 16 <= 150 * 1024, "two stages");
    static_assert(2 * W41_STAGE * 16 >= 2 * 64 * 4 * 33 * 4, "the hand-over of the running sums reuses the stages");
    __shared__ uint4 smem[2 * W41_STAGE];
    __shared__ float cls_tab[EPI == 2 ? 9 * 64 : 1];
    __shared__ float bias_lds[EPI >= 1 ? 64 : 1];
    __shared__ float w41_dummy[256];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int tiles_x = p.W / W41_TW, tiles_y = (p.H + W41_TH - 1) / W41_TH;
    const int ntm = tiles_x * tiles_y * p.B, ntn = (p.Np + 63) >> 6;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);            // gridDim.x == ntn * gm
    const int tn = bid % ntn, mg = bid / ntn;                    // this workgroup: pixel tiles mg, mg + gm, ... of slab tn
    const int T_ = (ntm - mg + gm - 1) / gm;                     // >= 1
    const int n0 = tn * 64;
    const int nk = p.Kp >> 3;
    const bool nt1 = n0 + 32 < p.Np;

    // ---- staging.  Input: unit (halo row hy < 16, segment s, 4-channel group g) = thread tid: six pixels x 16 bytes -> B6^T -> six transformed
    // values x 16 bytes; the two last halo rows (32 more units) are split by channel over waves 0-1: thread (unit, channel) moves six dwords.
    // Filters: no transform, slot = tid + 256 q through registers (LDS-DMA was tried: hipcc makes a wave wait for its own pending DMA before the
    // first ds_read that follows, i.e. the whole memory latency sat in front of every chunk's MFMAs).
    const unsigned img = (unsigned)p.H * (unsigned)p.W * (unsigned)p.x_ldc * 4u;
    const unsigned pstep = (unsigned)p.x_ldc * 4u;
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(p.w, (unsigned)((size_t)18 * p.Np * p.Kp * 4));
    const __amdgpu_buffer_rsrc_t wrs_dead = make_rsrc(p.w, 0u), xrs_dead = make_rsrc(p.x, 0u);      // the step behind the last one: loads without traffic
    const bool low = wave < 2;                                   // waves 0-1 also take the channel-split units (wave-uniform); the others go
    //                                                              through the same instructions with out-of-range loads and a dummy LDS word
    unsigned vb0, vb1;                                           // byte offset of a unit's pixel i = 0 (BUF_OOB: the whole halo row is padding)
    bool ok0_first, ok0_last, ok1_first, ok1_last;               // pixel 0 / pixel 5 of the unit inside the image (columns -1 / W are padding)
    __amdgpu_buffer_rsrc_t xrs;
    auto set_tile = [&](int tm) {
        const int x0 = (tm % tiles_x) * W41_TW, y0 = ((tm / tiles_x) % tiles_y) * W41_TH, b = tm / (tiles_x * tiles_y);
        xrs = make_rsrc((const char*)p.x + (size_t)b * img, img);
        {
            const int g = tid & 1, s = (tid >> 1) & 7, hy = tid >> 4, yy = y0 + hy - 1;
            vb0 = (yy >= 0 && yy < p.H) ? (unsigned)(((yy * p.W + x0 + 4 * s - 1) * p.x_ldc + 4 * g) * 4) : BUF_OOB;
            ok0_first = x0 + 4 * s > 0; ok0_last = x0 + 4 * s + 4 < p.W;
        }
        {
            const int u = (tid >> 2) & 31, ch = tid & 3;         // waves 0-1: unit 256 + u, channel ch of its group
            const int g = u & 1, s = (u >> 1) & 7, hy = 16 + (u >> 4), yy = y0 + hy - 1;
            vb1 = (low && yy >= 0 && yy < p.H) ? (unsigned)(((yy * p.W + x0 + 4 * s - 1) * p.x_ldc + 4 * g + ch) * 4) : BUF_OOB;
            ok1_first = x0 + 4 * s > 0; ok1_last = x0 + 4 * s + 4 < p.W;
        }
    };
    uint4 rin[6];
    float rin1[6];
    const unsigned w_vo = n0 + ((tid & 127) >> 1) < p.Np ? (unsigned)((((size_t)(tid >> 7) * p.Np + n0 + ((tid & 127) >> 1)) * 8 + 4 * (tid & 1)) * 4) : BUF_OOB;
    const unsigned w_q = (unsigned)((size_t)2 * p.Np * 8 * 4);             // slot + 256 = two (j, ky) planes further
    const unsigned w_chunk = (unsigned)((size_t)18 * p.Np * 8 * 4);
    uint4 rw[9];
    // 21 loads, no branch (the whole step is ONE basic block so that they can be issued between its MFMAs)
    auto gload = [&](const __amdgpu_buffer_rsrc_t xr, const __amdgpu_buffer_rsrc_t wr, int c) {
        const unsigned so = (unsigned)(c * 32);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const bool ok = i == 0 ? ok0_first : (i == 5 ? ok0_last : true);
            rin[i] = buf_ld16(xr, ok ? vb0 + i * pstep : BUF_OOB, so);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const bool ok = i == 0 ? ok1_first : (i == 5 ? ok1_last : true);
            rin1[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, ok ? vb1 + i * pstep : BUF_OOB, so, 0));      // the builtin returns the raw dword
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) rw[q] = buf_ld16(wr, w_vo, (unsigned)c * w_chunk + q * w_q);
    };
    // B6^T: v0 = 4 d0 - 5 d2 + d4; v1, v2 = (d4 - 4 d2) +- (d3 - 4 d1); v3, v4 = (d4 - d2) +- 2 (d3 - d1); v5 = 4 d1 - 5 d3 + d5
    auto bt6 = [](const float (&d)[6], float (&o)[6]) {
        const float t0 = fmaf(-4.f, d[2], d[4]), t1 = fmaf(-4.f, d[1], d[3]), t2 = d[4] - d[2], t3 = 2.f * (d[3] - d[1]);
        o[0] = fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4])); o[1] = t0 + t1; o[2] = t0 - t1; o[3] = t2 + t3; o[4] = t2 - t3;
        o[5] = fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]));
    };
    // slots of this thread in a stage: its unit's six V values, its nine filter pieces, its channel-split word (waves 2-3: a dummy word behind the stages)
    const int xslot = (tid & 1) * W41_XG + (tid >> 1);                                       // + j * (HH * SEG)
    const int wslot = W41_XS + (tid & 1) * W41_WG + (tid >> 1);                              // + 128 q
    const int u1 = 256 + ((tid >> 2) & 31);
    const int xword = low ? ((u1 & 1) * W41_XG + (u1 >> 1)) * 4 + (tid & 3) : -1;            // float index; + j * (HH * SEG) * 4
    auto stage_store = [&](int stg) {
        uint4* st = smem + stg * W41_STAGE;
        uint4 v[6];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float d[6], o[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) d[i] = __uint_as_float(c == 0 ? rin[i].x : c == 1 ? rin[i].y : c == 2 ? rin[i].z : rin[i].w);
            bt6(d, o);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const unsigned bq = __float_as_uint(o[j]);
                if (c == 0) v[j].x = bq; else if (c == 1) v[j].y = bq; else if (c == 2) v[j].z = bq; else v[j].w = bq;
            }
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) st[xslot + j * (W41_HH * W41_SEG)] = v[j];
#pragma unroll
        for (int q = 0; q < 9; ++q) st[wslot + 128 * q] = rw[q];
        float o[6];
        bt6(rin1, o);
        float* sf = xword >= 0 ? reinterpret_cast<float*>(st) + xword : w41_dummy + tid;
#pragma unroll
        for (int j = 0; j < 6; ++j) sf[xword >= 0 ? j * (W41_HH * W41_SEG) * 4 : 0] = o[j];
    };

    // ---- consumer side: lane (r, h) = super-pixel 32 wave + r (row 4 wave + r / 8, segment r % 8), channel pieces 8 h
    const int rperm = (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1);      // filter row behind MFMA row r (igemm_pws.hip)
    const int sp = 32 * wave + r, row_l = sp >> 3, seg = sp & 7;
    const int xs_base = h * W41_XG + row_l * W41_SEG + seg;              // + (j * HH + ky) * 8
    float cs1[EPI >= 1 ? 2 : 1][16], cs2[EPI >= 1 ? 2 : 1][16];
    if constexpr (EPI >= 1) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { cs1[nt][e] = 0.f; cs2[nt][e] = 0.f; }
        if (tid < 64) bias_lds[tid] = (p.bias && n0 + tid < p.Np) ? p.bias[(EPI == 2 ? 4 * p.Np : 0) + n0 + tid] : 0.f;      // class 4: interior pixels
    }
    if constexpr (EPI == 2) {      // what a border pixel's bias differs by from the interior's (class 4)
        for (int i = tid; i < 9 * 64; i += 256) {
            const int n = n0 + (i & 63);
            cls_tab[i] = n < p.Np ? p.bias[(i >> 6) * p.Np + n] - p.bias[4 * p.Np + n] : 0.f;
        }
    }
    const float relu_lo = p.relu ? 0.f : -__builtin_inff();
    const unsigned y_img = (unsigned)p.H * (unsigned)p.W * (unsigned)p.y_ldc * 4u;
    const unsigned st_vo = (unsigned)(((row_l * p.W + 4 * seg) * p.y_ldc + n0 + 8 * h) * 4);      // output pixel x = 0 of the segment
    const unsigned px_step = (unsigned)p.y_ldc * 4u;

    // ---- (tile, chunk) pipeline, one running step counter: the loads / filter DMA of step s + 1 are in flight under the MFMAs of step s
    // (across tile boundaries too), transformed and stored behind them, one barrier per step
    const int S = T_ * nk;
    int l_tile = 0, l_c = 0;                                     // load cursor: (tile, chunk) of the step whose data the registers hold
    auto advance = [&]() {                                        // -> the next step's (tile, chunk); re-derives the tile's offsets when it changes
        if (++l_c == nk) { l_c = 0; ++l_tile; if (l_tile < T_) set_tile(mg + l_tile * gm); }
    };
    set_tile(mg);
    gload(xrs, wrs, 0);
    stage_store(0);                                              // step 0 -> stage 0
    advance();
    gload(S > 1 ? xrs : xrs_dead, S > 1 ? wrs : wrs_dead, l_c);  // step 1 -> registers
    __syncthreads();
    int s = 0;                                                   // step
    unsigned long long dg[5] = {0, 0, 0, 0, 0};
    (void)dg;
    for (int ti = 0; ti < T_; ++ti) {
        f32x16 acc[6][2];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][nt][e] = 0.f;
        if constexpr (EPI >= 1) {      // the bias is where plane 1 starts: column 1 of A6^T is all ones
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float4 a = *reinterpret_cast<const float4*>(bias_lds + 32 * nt + 16 * q + 8 * h), bq = *reinterpret_cast<const float4*>(bias_lds + 32 * nt + 16 * q + 8 * h + 4);
                    acc[1][nt][8 * q + 0] = a.x; acc[1][nt][8 * q + 1] = a.y; acc[1][nt][8 * q + 2] = a.z; acc[1][nt][8 * q + 3] = a.w;
                    acc[1][nt][8 * q + 4] = bq.x; acc[1][nt][8 * q + 5] = bq.y; acc[1][nt][8 * q + 6] = bq.z; acc[1][nt][8 * q + 7] = bq.w;
                }
        }
        for (int c = 0; c < nk; ++c, ++s) {
            const unsigned long long t0 = W41_T();
            advance();                                       // the registers hold step s + 1; the loads below fetch step s + 2
            const bool more2 = s + 2 < S;
            // ---- ONE basic block from here to the barrier: the 144 MFMAs of step s with their 54 fragment reads; the transform of step s + 1
            // (in registers since the previous block) and its 21 LDS stores; the 21 loads of step s + 2 (dead descriptors past the end).  Alone,
            // each of the non-MFMA parts cost 1.5-2.2k cycles per step beside 8.3k of MFMAs (stamps, tools/w41_diag.py: a load instruction is 64
            // scattered lines and issues at ~100 cycles); placed between the MFMAs (w41_sched) they cost issue slots only.
            const unsigned long long t1 = W41_T();
            {
                const uint4* xs = smem + (s & 1) * W41_STAGE + xs_base;
                const uint4* ws = smem + (s & 1) * W41_STAGE + W41_XS + h * W41_WG;
                uint4 f[2][3];
#define W41_FRAG(jk_, d_)                                                                        \
    do {                                                                                         \
        d_[0] = xs[(((jk_) / 3) * W41_HH + (jk_) % 3) * W41_SEG];                                \
        d_[1] = ws[(jk_) * 64 + rperm];                                                          \
        d_[2] = ws[(jk_) * 64 + 32 + rperm];                                                     \
    } while (0)
                // Side work of the block in 36 pieces, one per group of four MFMAs, fenced (sched_barrier) so that it STAYS between them -- the
                // sched_group_barrier pipelines that order the other kernels put these loads either all in front of the MFMAs or all behind them:
                //   0-3 transform channel c of the unit in rin (step s + 1)   4 the channel-split unit   5-8 the nine filter stores   9-11 the six V
                //   stores   12-13 the channel-split stores   14-34 the 21 loads of step s + 2, one each (first use 16 groups = 3.6k cycles later)
                uint4* const st = smem + ((s + 1) & 1) * W41_STAGE;
                const __amdgpu_buffer_rsrc_t xr = more2 ? xrs : xrs_dead, wr = more2 ? wrs : wrs_dead;
                const unsigned so = (unsigned)(l_c * 32), wo = (unsigned)l_c * w_chunk;
                uint4 v[6];
                float o1[6];
                auto side = [&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    if constexpr (g < 4) {
                        float d[6], o[6];
#pragma unroll
                        for (int i = 0; i < 6; ++i) d[i] = __uint_as_float(g == 0 ? rin[i].x : g == 1 ? rin[i].y : g == 2 ? rin[i].z : rin[i].w);
                        bt6(d, o);
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            const unsigned bq = __float_as_uint(o[j]);
                            if (g == 0) v[j].x = bq; else if (g == 1) v[j].y = bq; else if (g == 2) v[j].z = bq; else v[j].w = bq;
                        }
                    } else if constexpr (g == 4) {
                        bt6(rin1, o1);
                    } else if constexpr (g < 9) {
                        constexpr int q0 = (g - 5) * 9 / 4, q1 = (g - 4) * 9 / 4;
#pragma unroll
                        for (int q = q0; q < q1; ++q) st[wslot + 128 * q] = rw[q];
                    } else if constexpr (g < 12) {
                        st[xslot + (2 * (g - 9)) * (W41_HH * W41_SEG)] = v[2 * (g - 9)];
                        st[xslot + (2 * (g - 9) + 1) * (W41_HH * W41_SEG)] = v[2 * (g - 9) + 1];
                    } else if constexpr (g < 14) {
                        float* sf = xword >= 0 ? reinterpret_cast<float*>(st) + xword : w41_dummy + tid;
#pragma unroll
                        for (int j = 3 * (g - 12); j < 3 * (g - 12) + 3; ++j) sf[xword >= 0 ? j * (W41_HH * W41_SEG) * 4 : 0] = o1[j];
                    } else if constexpr (g < 20) {
                        constexpr int i = g - 14;
                        const bool ok = i == 0 ? ok0_first : (i == 5 ? ok0_last : true);
                        rin[i] = buf_ld16(xr, ok ? vb0 + i * pstep : BUF_OOB, so);
                    } else if constexpr (g < 26) {
                        constexpr int i = g - 20;
                        const bool ok = i == 0 ? ok1_first : (i == 5 ? ok1_last : true);
                        rin1[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, ok ? vb1 + i * pstep : BUF_OOB, so, 0));
                    } else if constexpr (g < 35) {
                        constexpr int q = g - 26;
                        rw[q] = buf_ld16(wr, w_vo, wo + q * w_q);
                    }
                };
                W41_FRAG(0, f[0]);
                w41_unroll<18>([&](auto jkc) {
                    constexpr int jk = decltype(jkc)::value;
                    if constexpr (jk + 1 < 18) W41_FRAG(jk + 1, f[(jk + 1) & 1]);      // seven MFMAs and more cover the latency
                    const uint4* q = f[jk & 1];
                    side(std::integral_constant<int, 2 * jk>{});
                    mma16<float>(q[1], q[0], acc[jk / 3][0]);
                    __builtin_amdgcn_sched_barrier(0);
                    side(std::integral_constant<int, 2 * jk + 1>{});
                    mma16<float>(q[2], q[0], acc[jk / 3][1]);
                    __builtin_amdgcn_sched_barrier(0);
                });
#undef W41_FRAG
            }
            const unsigned long long t3 = W41_T();
            __syncthreads();
            const unsigned long long t4 = W41_T();
            dg[0] += t1 - t0; dg[1] += t3 - t1; dg[3] += t4 - t3;
        }
        const unsigned long long t5 = W41_T();

        // ---- epilogue of tile ti: A6^T in-lane, (border-class bias,) ReLU, statistics, 16-byte stores
        const int tm = mg + ti * gm;
        const int x0 = (tm % tiles_x) * W41_TW, y0 = ((tm / tiles_x) % tiles_y) * W41_TH, b = tm / (tiles_x * tiles_y);
        const __amdgpu_buffer_rsrc_t yrs = make_rsrc((const char*)p.y + (size_t)b * y_img, y_img);
        const unsigned y_so = (unsigned)((y0 * p.W + x0) * p.y_ldc) * 4u;
        const bool pok = !RAGGED || y0 + row_l < p.H;
        const unsigned vo = pok ? st_vo : BUF_OOB;
        bool border = false;
        if constexpr (EPI == 2) border = x0 == 0 || y0 == 0 || x0 + W41_TW >= p.W || y0 + W41_TH >= p.H;      // wave-uniform
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (nt == 1 && !nt1) break;
#pragma unroll
            for (int q = 0; q < 2; ++q) {              // one run of 8 consecutive channels at a time (32 temporaries, not 64)
                float y[4][8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = 8 * q + i;
                    const float m0 = acc[0][nt][e], m1 = acc[1][nt][e], m2 = acc[2][nt][e], m3 = acc[3][nt][e], m4 = acc[4][nt][e], m5 = acc[5][nt][e];
                    const float sa = m1 + m2, sb = m1 - m2, sc = m3 + m4, sd = m3 - m4;
                    y[0][i] = m0 + sa + sc;
                    y[1][i] = fmaf(2.f, sd, sb);
                    y[2][i] = fmaf(4.f, sc, sa);
                    y[3][i] = fmaf(8.f, sd, sb) + m5;
                }
                if constexpr (EPI == 2) {
                    if (border) {
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const int cls = pok ? border_class(y0 + row_l, x0 + 4 * seg + x, p.H, p.W) : 4;
                            const float* row = cls_tab + cls * 64 + 32 * nt + 16 * q + 8 * h;
                            const float4 a = *reinterpret_cast<const float4*>(row), bq = *reinterpret_cast<const float4*>(row + 4);
                            y[x][0] += a.x; y[x][1] += a.y; y[x][2] += a.z; y[x][3] += a.w;
                            y[x][4] += bq.x; y[x][5] += bq.y; y[x][6] += bq.z; y[x][7] += bq.w;
                        }
                    }
                }
                if constexpr (EPI >= 1) {
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int i = 0; i < 8; ++i) y[x][i] = fmaxf(y[x][i], relu_lo);
                    if (p.stats) {
#pragma unroll
                        for (int x = 0; x < 4; ++x)
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const float vs = (!RAGGED || pok) ? y[x][i] : 0.f;
                                cs1[nt][8 * q + i] += vs;
                                cs2[nt][8 * q + i] = fmaf(vs, vs, cs2[nt][8 * q + i]);
                            }
                    }
                }
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    buf_st16(yrs, vo + (unsigned)((32 * nt + 16 * q) * 4), y_so + x * px_step,
                             make_uint4(__float_as_uint(y[x][0]), __float_as_uint(y[x][1]), __float_as_uint(y[x][2]), __float_as_uint(y[x][3])));
                    buf_st16(yrs, vo + (unsigned)((32 * nt + 16 * q + 4) * 4), y_so + x * px_step,
                             make_uint4(__float_as_uint(y[x][4]), __float_as_uint(y[x][5]), __float_as_uint(y[x][6]), __float_as_uint(y[x][7])));
                }
            }
        }
        dg[4] += W41_T() - t5;
    }

    W41_ADD(0, dg[0]); W41_ADD(1, dg[1]); W41_ADD(2, dg[2]); W41_ADD(3, dg[3]); W41_ADD(4, dg[4]); W41_ADD(7, 1);
    // [0] tile cursor  [1] the step's block (loads, MFMAs, transform, stage stores)  [3] barrier  [4] epilogues  [7] workgroups
    // ---- statistics: one row per workgroup, fixed order (igemm_pws.hip): [kind][channel][wave][pixel lane], pitch 33, through the free stages
    if constexpr (EPI >= 1) {
        if (p.stats) {
            float* fb = reinterpret_cast<float*>(smem);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int cch = 32 * nt + 16 * (e >> 3) + 8 * h + (e & 7);
                    fb[((0 * 64 + cch) * 4 + wave) * 33 + r] = cs1[nt][e];
                    fb[((1 * 64 + cch) * 4 + wave) * 33 + r] = cs2[nt][e];
                }
            __syncthreads();
            for (int idx = tid; idx < 512; idx += 256) {       // (kind, channel, wave): 32 lanes each, then the four waves by two xor shuffles
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < 32; ++i) t += fb[idx * 33 + i];
                t += __shfl_xor(t, 1);
                t += __shfl_xor(t, 2);
                const int kc = idx >> 2, k = kc >> 6, cch = kc & 63;
                if ((idx & 3) == 0 && n0 + cch < p.Np) p.stats[((size_t)mg * 2 + k) * p.Np + n0 + cch] = t;
            }
        }
    }
}

// ---- filter transform: dst[(k/8) * 18 + 3 j + ky][n][k % 8] = (G6 g[ky])[j]  (jobs as in wino.hip; data gradient: g = flip(w[k][n])) ----
__global__ void __launch_bounds__(256) wino41_pack_kernel(const WinoPackJob* __restrict__ jobs, int njobs, int nblocks, const FoldBias fold) {
    if ((int)blockIdx.x >= nblocks) {       // appended blocks: the border-class bias table of a folded BatchNorm (common.hip.h)
        __shared__ float T[9];
        fold_bias_block(fold, (int)blockIdx.x - nblocks, T);
        return;
    }
    int ji = 0;
    while (ji + 1 < njobs && (int)blockIdx.x >= jobs[ji + 1].block0) ++ji;
    const WinoPackJob J = jobs[ji];
    const long long idx = (long long)(blockIdx.x - J.block0) * 256 + threadIdx.x;
    if (idx >= (long long)J.Np * J.Kp) return;
    const int k8 = (int)(idx & 7), n = (int)((idx >> 3) % J.Np), kc = (int)((idx >> 3) / J.Np);
    const int k = kc * 8 + k8;
    const int nl = wn_phys2log(n, J.n_seg0, J.n_seg0p, J.N), kl = wn_phys2log(k, J.k_seg0, J.k_seg0p, J.K);
    float g[3][3];
#pragma unroll
    for (int i = 0; i < 9; ++i) g[i / 3][i % 3] = 0.f;
    if (nl >= 0 && kl >= 0) {
        const float* s = J.dgrad ? J.w + ((size_t)kl * J.N + nl) * 9 : J.w + ((size_t)nl * J.K + kl) * 9;
        const float ks = J.kscale ? J.kscale[k] : 1.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i / 3][i % 3] = (J.dgrad ? s[8 - i] : s[i]) * ks;
    }
    float* d = J.dst + ((size_t)kc * 18 * J.Np + n) * 8 + k8;
    const size_t xs = (size_t)J.Np * 8;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        // G6 = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
        const float a = g[ky][0], b = g[ky][1], c = g[ky][2];
        d[(3 * 0 + ky) * xs] = 0.25f * a;
        d[(3 * 1 + ky) * xs] = (-1.f / 6.f) * (a + b + c);
        d[(3 * 2 + ky) * xs] = (-1.f / 6.f) * (a - b + c);
        d[(3 * 3 + ky) * xs] = (1.f / 24.f) * a + (1.f / 12.f) * b + (1.f / 6.f) * c;
        d[(3 * 4 + ky) * xs] = (1.f / 24.f) * a - (1.f / 12.f) * b + (1.f / 6.f) * c;
        d[(3 * 5 + ky) * xs] = c;
    }
}

static long long w41_tiles(int B, int H, int W) { return (long long)B * ((H + W41_TH - 1) / W41_TH) * (W / W41_TW); }

// workgroups per 64-channel output slab (= partial statistics rows of a launch)
long long clamd_winograd41_stat_rows(int B, int H, int W, int Cout_p, const clamd_tuning& tn) {
    const long long ntm = w41_tiles(B, H, W), ntn = (Cout_p + 63) / 64;
    long long gm = clamd_usable_cus(tn) / ntn;
    if (gm < 1) gm = 1;
    return gm > ntm ? ntm : gm;
}

}  // namespace clamd

using namespace clamd;

#ifdef CLAMD_DIAG
extern "C" int clamd_debug_w41_diag(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(clamd::g_w41_diag), 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(clamd::g_w41_diag), z, 64) != hipSuccess) return -1; }
    return 0;
}
#endif

int clamd_launch_wino41_pack(const void* jobs_dev, int njobs, int total_blocks, const clamd::FoldBias* fold, hipStream_t stream) {
    if (njobs <= 0 || total_blocks <= 0) return clamd_fail("wino41_pack: empty job table");
    const clamd::FoldBias f = fold ? *fold : clamd::FoldBias{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 9};
    hipLaunchKernelGGL(clamd::wino41_pack_kernel, dim3(total_blocks + (fold ? fold->Cout_p : 0)), dim3(256), 0, stream, (const clamd::WinoPackJob*)jobs_dev, njobs,
                       total_blocks, f);
    return clamd_check_launch("wino41_pack");
}

extern "C" {

int clamd_wino41_pack(const void* jobs_dev, int njobs, int total_blocks, void* stream) {
    return clamd_launch_wino41_pack(jobs_dev, njobs, total_blocks, nullptr, (hipStream_t)stream);
}

int clamd_conv3x3_winograd41_ok(int B, int H, int W, int Cin_p, int Cout_p) {
    return B > 0 && H >= 1 && W >= W41_TW && W % W41_TW == 0 && Cin_p >= 8 && Cin_p % 8 == 0 && Cout_p >= 32 && Cout_p % 32 == 0 &&
           (long long)H * W * std::max(Cin_p, Cout_p) * 4 < (1ll << 31) && (long long)18 * Cout_p * Cin_p * 4 < (1ll << 31);
}

int clamd_conv3x3_winograd41(const float* x, int x_ldc, const float* w41, const float* bias, float* y, int y_ldc, float* stats, int stat_rows,
                             int B, int H, int W, int Cin_p, int Cout_p, int relu, const clamd_tuning* tune, void* stream) {
    if (!clamd_conv3x3_winograd41_ok(B, H, W, Cin_p, Cout_p))
        return clamd_fail("conv3x3_winograd41: needs W a multiple of 32, Cin_p % 8 == 0, Cout_p % 32 == 0 and images / filters below 2^31 bytes");
    if (x_ldc % 4 || y_ldc % 4 || x_ldc < Cin_p || y_ldc < Cout_p) return clamd_fail("conv3x3_winograd41: bad pitches");
    if (int e = clamd_check_tuning(tune)) return e;
    const clamd_tuning& tn = clamd_tune(tune);
    if ((relu & ~3) || ((relu & CLAMD_BIAS_BORDER_CLASSES) && (!bias || H < 2)))
        return clamd_fail("conv3x3_winograd41: bad relu flags (bit 1 needs the [9][Cout_p] bias table and H >= 2)");
    const long long gm = clamd_winograd41_stat_rows(B, H, W, Cout_p, tn);
    if (stats && stat_rows != gm) return clamd_fail("conv3x3_winograd41: stat_rows does not match clamd_stat_rows(CLAMD_OP_CONV3X3_WINOGRAD41, ...)");
    WinoParams p{x, x_ldc, w41, bias, y, y_ldc, stats, B, H, W, Cin_p, Cout_p, relu & 1, 1, 0};
    p.bias_classes = (relu & CLAMD_BIAS_BORDER_CLASSES) ? 1 : 0;
    const int ntn = (Cout_p + 63) / 64;
    const dim3 grid((unsigned)(gm * ntn));
    const bool ragged = H % W41_TH != 0;
    const bool plain = !bias && !(relu & 1) && !stats;
    hipStream_t s = (hipStream_t)stream;
#define W41_LAUNCH(RG_)                                                                                              \
    do {                                                                                                             \
        if (plain) hipLaunchKernelGGL((wino41_kernel<RG_, 0>), grid, dim3(256), 0, s, p, (int)gm);                   \
        else if (p.bias_classes) hipLaunchKernelGGL((wino41_kernel<RG_, 2>), grid, dim3(256), 0, s, p, (int)gm);     \
        else hipLaunchKernelGGL((wino41_kernel<RG_, 1>), grid, dim3(256), 0, s, p, (int)gm);                         \
    } while (0)
    if (ragged) W41_LAUNCH(true); else W41_LAUNCH(false);
#undef W41_LAUNCH
    return clamd_check_launch("conv3x3_winograd41");
}

}  // extern "C"
