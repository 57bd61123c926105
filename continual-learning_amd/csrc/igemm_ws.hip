// Producer/consumer ("wave-specialised") implicit-GEMM 3x3 convolution for gfx950.
//
// Ablation of the baseline kernel (igemm.hip; tools/conv_ab.py variants 3-5) showed it is NOT MFMA-bound: with the
// MFMAs removed it is only 28 % faster, with the global loads removed 39 % faster -- the four waves of a workgroup all
// stop issuing MFMAs while they wait for loads, write LDS and meet at two barriers per K-step, and a second resident
// workgroup only partly covers that.  Here a 512-thread workgroup has two roles:
//   * waves 4-7 (producers) stream the halo tile + filter slab of K-step k+1 from global memory through registers
//     into LDS stage (k+1)&1 and issue the loads of K-step k+2;
//   * waves 0-3 (consumers, one per SIMD) do nothing but ds_read_b128 fragments of stage k&1 and MFMAs, with the
//     fragment reads of the next tap step issued before the MFMAs of the current one.
// ONE barrier per K-step hands the stages over; LDS holds two stages (2 x 60 KB), one workgroup per CU.
// Tile, LDS image, fragment maps, epilogue and results are identical to igemm.hip's CONV3/NHWC kernel.
#include <string.h>
#include "common.hip.h"
#include "igemm_common.hip.h"
#include "clamd_internal.h"

namespace clamd {

#ifdef CLAMD_DIAG
// diagnostic build only (python build.py --diag): per-role cycle shares of the K loop, summed over workgroups
__device__ unsigned long long g_ws_diag[8];
#define DIAG_T() __builtin_amdgcn_s_memtime()
#define DIAG_ADD(i_, v_) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_ws_diag[i_], (unsigned long long)(v_)); } while (0)
#else
#define DIAG_T() 0ull
#define DIAG_ADD(i_, v_) do { } while (0)
#endif

template <typename T, int TW, int MT>
__global__ void __launch_bounds__(512, 2) igemm_ws_kernel(const IgemmParams p) {
    using G = WsGeo<TW, MT>;
    constexpr int TH = G::TH, NT = G::NT, HW_ = G::HW_, NPIX = G::NPIX, NPIXP = G::NPIXP, NJ = G::NJ;
    constexpr int KC = DT<T>::KC, VEC = DT<T>::VEC;
    constexpr bool SPLIT = __is_same(T, split_t);
    constexpr int STAGE = G::IN_SLOTS + G::WT_SLOTS;
    static_assert(2 * STAGE * 16 <= 160 * 1024, "two LDS stages must fit one CU");
    __shared__ uint4 smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int ltid = tid & 255;                    // index inside the role (4 waves each)
    const int cw = wave & 3;                       // consumer wave index (M quarter of the tile)
    const int r = lane & 31, h = lane >> 5;

    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int ntm = tiles_x * tiles_y * p.B, ntn = (p.Np + 63) >> 6;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    if (p.m_fastest) { tm = bid % ntm; tn = bid / ntm; } else { tn = bid % ntn; tm = bid / ntn; }
    const int x0 = (tm % tiles_x) * TW, y0 = ((tm / tiles_x) % tiles_y) * TH, b = tm / (tiles_x * tiles_y);
    const int n0 = tn * 64;
    const int nk = p.Kp / KC;

    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (producer) {
        // ------------------------------------------------------------------ producers: global -> registers -> LDS
        constexpr int ESZ = sizeof(T);
        const int g4 = ltid & 3;
        const unsigned img_elems = (unsigned)p.H * (unsigned)p.W * (unsigned)p.x_ldc;
        const __amdgpu_buffer_rsrc_t xrs = make_rsrc((const char*)p.x + (size_t)b * img_elems * ESZ, img_elems * ESZ);
        const __amdgpu_buffer_rsrc_t wrs = make_rsrc(p.w, (unsigned)(9u * p.Np * p.Kp * ESZ));
        // halo pixel of staging slot j; the ragged last pass wraps around and re-stages the first pixels (same data, same
        // LDS slot), so every load has an unconditional store (a store guarded by "slot < NPIX" lets the compiler sink
        // the load next to it, behind a full s_waitcnt vmcnt(0))
        auto slot_pix = [&](int j) { const int pix = (ltid >> 2) + 64 * j; return pix >= NPIX ? pix - NPIX : pix; };
        unsigned in_vo[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int pix = slot_pix(j);
            const int hy = pix / HW_, hx = pix - hy * HW_;
            const int yy = y0 + hy - 1, xx = x0 + hx - 1;
            in_vo[j] = (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W)
                           ? (unsigned)(((yy * p.W + xx) * p.x_ldc + g4 * VEC) * ESZ) : BUF_OOB;
        }
        const int wco = ltid >> 2;
        const unsigned w_vo = n0 + wco < p.Np ? (unsigned)(((n0 + wco) * KC + g4 * VEC) * ESZ) : BUF_OOB;
        const unsigned w_slab = (unsigned)(p.Np * KC * ESZ);
        // Input chunks are fetched in PAIRS (two back-to-back 64-byte pieces = one full 128-byte line per pixel while the
        // line is still in L1): fetching one 64-byte piece per K-step, a whole period apart, made every line cross the
        // L2->CU path twice and capped the producers at ~11.5 B/clk/CU (tools/ws_diag.py, diagnostic build).
        // live_ = false issues the same instructions with every lane out of range (a fixed load schedule lets the compiler
        // count vmcnt instead of draining it).
        uint4 rinA[NJ], rinB[NJ], rw[NT];

#define WS_GLOAD_IN2(kp_, RA, RB, live_)                                                                          \
    do {                                                                                                          \
        const unsigned so_ = (unsigned)(2 * (kp_) * KC * ESZ);                                                    \
        const bool l_ = (live_);                                   /* wave-uniform */                             \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                          \
            RA[j] = buf_ld16(xrs, l_ ? in_vo[j] : BUF_OOB, so_);                                                  \
            RB[j] = buf_ld16(xrs, l_ ? in_vo[j] : BUF_OOB, so_ + KC * ESZ);                                       \
        }                                                                                                         \
    } while (0)
#define WS_GLOAD_W(ks_, live_)                                                                                    \
    do {                                                                                                          \
        const unsigned wv_ = (live_) ? w_vo : BUF_OOB;                                                            \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                            \
            rw[t] = buf_ld16(wrs, wv_, (unsigned)((ks_) * NT + t) * w_slab);                                      \
    } while (0)
#define WS_STORE(st_, RIN)                                                                                        \
    do {                                                                                                          \
        uint4* sm_ = smem + (st_) * STAGE;                                                                        \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                          \
            const int pix_ = slot_pix(j);                                                                         \
            /* split_t: the 64-byte K-chunk already is [hi 0-7][hi 8-15][lo 0-7][lo 8-15] = LDS slots 0..3 */     \
            sm_[g4 * NPIXP + pix_] = RIN[j];                                                                      \
        }                                                                                                         \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) sm_[G::IN_SLOTS + (t * 4 + g4) * G::WG + wco] = rw[t];     \
    } while (0)

        unsigned long long t0 = DIAG_T(), t1, d_store = 0, d_bar = 0;
        (void)t1; (void)d_store; (void)d_bar;
#define WS_PERIOD_END()                                                                                           \
    do { t1 = DIAG_T(); __syncthreads(); d_store += t1 - t0; d_bar += DIAG_T() - t1; t0 = DIAG_T(); } while (0)
        if ((nk & 3) == 0) {
            // Deep schedule (K-steps in fours, every layer with >= 128 input channels): TWO pair sets, a pair is fetched
            // 3-4 periods before it is staged.  With one set the data of every other K-step had a single period
            // (2.3-4.6k cycles in bf16) to arrive from HBM; whenever it was late the consumers sat at the barrier -- 20-30 %
            // of their cycles on the long-K bf16 / bf16x3 layers (tools/ws_diag.py).  Filters come from L2: one set.
            uint4 rinC[NJ], rinD[NJ];
            const int npair = nk >> 1;
            WS_GLOAD_IN2(0, rinA, rinB, true);
            WS_GLOAD_IN2(1, rinC, rinD, true);
            WS_GLOAD_W(0, true);
            WS_STORE(0, rinA);
            WS_GLOAD_W(1, true);
            __syncthreads();                               // stage 0 is ready
            DIAG_ADD(0, DIAG_T() - t0);                    // [0] producer prologue
            t0 = DIAG_T();
            for (int ks = 0; ks < nk; ks += 4) {           // consumers are reading stage ks&1 in period ks
                const int pr = ks >> 1;                    // pair of steps ks, ks+1 (set A/B); pr + 1: steps ks+2, ks+3 (set C/D)
                WS_STORE(1, rinB);                         // step ks+1
                WS_GLOAD_W(ks + 2, true);
                WS_GLOAD_IN2(pr + 2, rinA, rinB, pr + 2 < npair);
                WS_PERIOD_END();
                WS_STORE(0, rinC);                         // step ks+2
                WS_GLOAD_W(ks + 3, true);
                WS_PERIOD_END();
                WS_STORE(1, rinD);                         // step ks+3
                WS_GLOAD_W(ks + 4, ks + 4 < nk);
                WS_GLOAD_IN2(pr + 3, rinC, rinD, pr + 3 < npair);
                WS_PERIOD_END();
                if (ks + 4 < nk) WS_STORE(0, rinA);        // step ks+4
                WS_GLOAD_W(ks + 5, ks + 5 < nk);
                WS_PERIOD_END();
            }
        } else {
            WS_GLOAD_IN2(0, rinA, rinB, true);             // (an odd last step loads one chunk past its pair: in range or zero)
            WS_GLOAD_W(0, true);
            WS_STORE(0, rinA);
            if (nk > 1) WS_GLOAD_W(1, true);
            __syncthreads();                               // stage 0 is ready
            DIAG_ADD(0, DIAG_T() - t0);                    // [0] producer prologue
            t0 = DIAG_T();
            for (int ks = 0; ks < nk; ++ks) {
                if (ks + 1 < nk) {                         // consumers are reading stage ks&1
                    if ((ks + 1) & 1) WS_STORE((ks + 1) & 1, rinB); else WS_STORE((ks + 1) & 1, rinA);
                    if (ks + 2 < nk) {
                        WS_GLOAD_W(ks + 2, true);          // lands during the next period
                        if ((ks + 1) & 1) WS_GLOAD_IN2((ks + 2) >> 1, rinA, rinB, true);   // both input buffers are free again
                    }
                }
                WS_PERIOD_END();
            }
        }
        DIAG_ADD(1, d_store); DIAG_ADD(2, d_bar);          // [1] producer wait-loads+store+issue, [2] producer at barrier
#undef WS_PERIOD_END
#undef WS_GLOAD_IN2
#undef WS_GLOAD_W
#undef WS_STORE
    } else {
        // ------------------------------------------------------------------ consumers: LDS fragments -> MFMA
        int apix[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = 32 * MT * cw + 32 * mt + r;
            apix[mt] = (m / TW) * HW_ + (m % TW);
        }
        unsigned long long c0, c1, d_cmp = 0, d_cbar = 0;
        c0 = DIAG_T();
        __syncthreads();                                   // stage 0 is ready
        DIAG_ADD(3, DIAG_T() - c0);                        // [3] consumer waits for the first stage
        for (int ks = 0; ks < nk; ++ks) {
            c0 = DIAG_T();
            const uint4* sm = smem + (ks & 1) * STAGE;
            if constexpr (SPLIT) {
                uint4 f[2][2 * MT + 4];      // [buffer][a_hi[mt], a_lo[mt] ..., bh0, bl0, bh1, bl1]
#define WS_FRAG_S(t_, d_)                                                                                         \
    do {                                                                                                          \
        const int ib_ = ((t_) / 3) * HW_ + ((t_) % 3);                                                            \
        _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) {                                                    \
            d_[2 * mt_] = sm[ib_ + h * NPIXP + apix[mt_]];                                                        \
            d_[2 * mt_ + 1] = sm[ib_ + (2 + h) * NPIXP + apix[mt_]];                                              \
        }                                                                                                         \
        d_[2 * MT + 0] = sm[G::IN_SLOTS + ((t_) * 4 + h) * G::WG + r];                                            \
        d_[2 * MT + 1] = sm[G::IN_SLOTS + ((t_) * 4 + 2 + h) * G::WG + r];                                        \
        d_[2 * MT + 2] = sm[G::IN_SLOTS + ((t_) * 4 + h) * G::WG + 32 + r];                                       \
        d_[2 * MT + 3] = sm[G::IN_SLOTS + ((t_) * 4 + 2 + h) * G::WG + 32 + r];                                   \
    } while (0)
                WS_FRAG_S(0, f[0]);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT + 4, 0);     // reads of step 0 lead; then R(s+1), M(s), ...
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (t + 1 < NT) WS_FRAG_S(t + 1, f[(t + 1) & 1]);
                    const uint4* c = f[t & 1];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        mma_bf16(c[2 * mt + 1], c[2 * MT + 0], acc[mt][0]); mma_bf16(c[2 * mt], c[2 * MT + 1], acc[mt][0]);
                        mma_bf16(c[2 * mt], c[2 * MT + 0], acc[mt][0]);
                        mma_bf16(c[2 * mt + 1], c[2 * MT + 2], acc[mt][1]); mma_bf16(c[2 * mt], c[2 * MT + 3], acc[mt][1]);
                        mma_bf16(c[2 * mt], c[2 * MT + 2], acc[mt][1]);
                    }
                    if (t + 1 < NT) sched_mfma_reads<6 * MT, 2 * MT + 4>();
                    else __builtin_amdgcn_sched_group_barrier(0x008, 6 * MT, 0);
                }
#undef WS_FRAG_S
            } else {
                constexpr int NSTEP = 2 * NT;
                constexpr int NMF = sizeof(T) == 2 ? 4 : 16;
                uint4 f[2][MT + 2];      // [buffer][a[mt] ..., b0, b1]
#define WS_FRAG(s_, d_)                                                                                           \
    do {                                                                                                          \
        const int t_ = (s_) >> 1, g_ = ((s_) & 1) * 2 + h;                                                        \
        const int ib_ = (t_ / 3) * HW_ + (t_ % 3);                                                                \
        _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) d_[mt_] = sm[ib_ + g_ * NPIXP + apix[mt_]];          \
        d_[MT] = sm[G::IN_SLOTS + (t_ * 4 + g_) * G::WG + r];                                                     \
        d_[MT + 1] = sm[G::IN_SLOTS + (t_ * 4 + g_) * G::WG + 32 + r];                                            \
    } while (0)
                WS_FRAG(0, f[0]);
                __builtin_amdgcn_sched_group_barrier(0x100, MT + 2, 0);     // reads of step 0 lead; then R(s+1), M(s), ...
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) {
                    if (st + 1 < NSTEP) WS_FRAG(st + 1, f[(st + 1) & 1]);
                    const uint4* c = f[st & 1];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        mma16<T>(c[mt], c[MT], acc[mt][0]);
                        mma16<T>(c[mt], c[MT + 1], acc[mt][1]);
                    }
                    if (st + 1 < NSTEP) sched_mfma_reads<NMF * MT / 2, MT + 2>();
                    else __builtin_amdgcn_sched_group_barrier(0x008, NMF * MT / 2, 0);
                }
#undef WS_FRAG
            }
            c1 = DIAG_T();
            __syncthreads();                               // hand stage ks&1 back to the producers
            d_cmp += c1 - c0; d_cbar += DIAG_T() - c1;
        }
        DIAG_ADD(4, d_cmp); DIAG_ADD(5, d_cbar);           // [4] consumer MFMA loop, [5] consumer at barrier
    }
    const unsigned long long e0 = DIAG_T();

    // ---------------------------------------------------------------------- epilogue (consumers work, everyone syncs)
    const bool cons = !producer;
    float bcol[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n0 + 32 * nt + r;
        bcol[nt] = (p.bias && n < p.Np) ? p.bias[n] : 0.f;
    }
    unsigned long long vmask = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = 32 * MT * cw + 32 * mt + acc_row(e, h);
            if (y0 + m / TW < p.H && x0 + m % TW < p.W) vmask |= 1ull << (mt * 16 + e);
        }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[mt][nt][e] + bcol[nt];
                if (p.relu) v = fmaxf(v, 0.f);
                acc[mt][nt][e] = v;
            }
    float* ebuf = reinterpret_cast<float*>(smem);     // the last loop barrier released both stages
    if (p.stats) {
        float* sbuf = ebuf + 4 * 32 * 68;
        if (cons) {
            float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float v = (vmask >> (mt * 16 + e)) & 1 ? acc[mt][nt][e] : 0.f;
                        s1[nt] += v;
                        s2[nt] += v * v;
                    }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                s1[nt] += __shfl_xor(s1[nt], 32);
                s2[nt] += __shfl_xor(s2[nt], 32);
            }
            if (h == 0) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    sbuf[(cw * 2 + 0) * 64 + 32 * nt + r] = s1[nt];
                    sbuf[(cw * 2 + 1) * 64 + 32 * nt + r] = s2[nt];
                }
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int k = tid >> 6, c = tid & 63;
            const float t = sbuf[(0 * 2 + k) * 64 + c] + sbuf[(1 * 2 + k) * 64 + c] + sbuf[(2 * 2 + k) * 64 + c] +
                            sbuf[(3 * 2 + k) * 64 + c];
            if (n0 + c < p.Np) p.stats[((size_t)tm * 2 + k) * p.Np + n0 + c] = t;      // partial row of this pixel tile
        }
    }
    T* out = (T*)p.y;
    float* wbuf = ebuf + cw * 32 * 68;
    float bs[5][8];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) bs[k][j] = 0.f;
    const bool do_bn = p.bn_y != nullptr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        __syncthreads();
        if (cons) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) wbuf[acc_row(e, h) * 68 + 32 * nt + r] = acc[mt][nt][e];
        }
        __syncthreads();
        if (cons) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = ps * 8 + (lane >> 3), cgp = lane & 7;
                const int m = 32 * MT * cw + 32 * mt + row;
                const int yy = y0 + m / TW, xx = x0 + m % TW;
                const int n = n0 + cgp * 8;
                if (yy < p.H && xx < p.W && n < p.Np) {
                    float v[8];
                    const float4 lo = *reinterpret_cast<const float4*>(wbuf + row * 68 + cgp * 8);
                    const float4 hi = *reinterpret_cast<const float4*>(wbuf + row * 68 + cgp * 8 + 4);
                    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
                    const long long pixo = ((long long)b * p.H + yy) * p.W + xx;
                    Vec8<T>::store(out + pixo * p.y_ldc + n, v);
                    if (do_bn) {
                        float yv[8];
                        Vec8<T>::load((const T*)p.bn_y + pixo * p.Np + n, yv);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float pos = yv[j] > 0.f ? 1.f : 0.f;
                            bs[0][j] += v[j]; bs[1][j] += v[j] * yv[j]; bs[2][j] += v[j] * pos;
                            bs[3][j] += pos; bs[4][j] += yv[j];
                        }
                    }
                }
            }
        }
    }
    if (do_bn) {
        if (cons) {
#pragma unroll
            for (int k = 0; k < 5; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float t = bs[k][j];
                    t += __shfl_xor(t, 8); t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
                    bs[k][j] = t;
                }
        }
        __syncthreads();
        if (cons && lane < 8) {
#pragma unroll
            for (int k = 0; k < 5; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) ebuf[(cw * 5 + k) * 64 + lane * 8 + j] = bs[k][j];
        }
        __syncthreads();
        for (int i = tid; i < 5 * 64; i += 512) {
            const int k = i >> 6, c = i & 63;
            const float t = ebuf[(0 * 5 + k) * 64 + c] + ebuf[(1 * 5 + k) * 64 + c] + ebuf[(2 * 5 + k) * 64 + c] +
                            ebuf[(3 * 5 + k) * 64 + c];
            if (n0 + c < p.Np) p.bn_sums[((size_t)tm * 5 + k) * p.Np + n0 + c] = t;
        }
    }
    if (wave == 0) DIAG_ADD(6, DIAG_T() - e0);             // [6] epilogue (one wave per workgroup)
    if (wave == 0) DIAG_ADD(7, 1);                         // [7] workgroups
}

int ws_rows(const IgemmParams& p, int mt) {
    const bool wide = p.W >= 32;
    if (!wide && mt > 2) mt = 2;
    const int TW = wide ? 32 : 16, TH = 128 * mt / TW;
    const long long tiles = (long long)((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH) * p.B;
    return tiles > 0x7fffffff ? -1 : (int)tiles;
}

template <typename T>
static int launch_ws_t(const IgemmParams& p, hipStream_t s, int mt) {
    const bool wide = p.W >= 32;
    if (!wide && mt > 2) mt = 2;             // 16-wide tiles: 256 pixels (16 x 16) or 128 (16 x 8)
    const int TW = wide ? 32 : 16, TH = 128 * mt / TW;
    const long long tiles = (long long)((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH) * p.B;
    const long long nblk = tiles * ((p.Np + 63) / 64);
    if (nblk <= 0 || nblk > 0x7fffffff) return clamd_fail("igemm_ws: grid out of range");
    if (!wide && mt == 1) hipLaunchKernelGGL((igemm_ws_kernel<T, 16, 1>), dim3((unsigned)nblk), dim3(512), 0, s, p);
    else if (!wide) hipLaunchKernelGGL((igemm_ws_kernel<T, 16, 2>), dim3((unsigned)nblk), dim3(512), 0, s, p);
    else if (mt == 1) hipLaunchKernelGGL((igemm_ws_kernel<T, 32, 1>), dim3((unsigned)nblk), dim3(512), 0, s, p);
    else if (mt == 4) hipLaunchKernelGGL((igemm_ws_kernel<T, 32, 4>), dim3((unsigned)nblk), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((igemm_ws_kernel<T, 32, 2>), dim3((unsigned)nblk), dim3(512), 0, s, p);
    return clamd_check_launch("igemm_ws");
}

}  // namespace clamd
#ifdef CLAMD_DIAG
extern "C" int clamd_debug_ws_diag(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(clamd::g_ws_diag), 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(clamd::g_ws_diag), z, 64) != hipSuccess) return -1; }
    return 0;
}
#endif
namespace clamd {

int launch_igemm_ws(const IgemmParams& p, int dtype, hipStream_t s, int mt) {
    if (dtype == CLAMD_BF16) return launch_ws_t<bf16_t>(p, s, mt);
    if (dtype == CLAMD_F32) return launch_ws_t<float>(p, s, mt);
    if (dtype == CLAMD_SPLIT) return launch_ws_t<split_t>(p, s, mt);
    return clamd_fail("igemm_ws: bad dtype");
}

}  // namespace clamd
