// Implicit-GEMM convolution kernels for gfx950 (SURVEY.md §8a rows A3, A7, A9 and their data-gradients in A11).
//
//   out[pixel m, n] = sum_{tap t} sum_{k} A_t[m, k] * Wp[t][n][k]        M = B*H*W pixels, fp32 accumulate
//
// One 256-thread workgroup (4 waves) owns a TH x TW = 256-pixel spatial tile x 64 output channels; each wave owns
// 64 pixels x 64 channels = 2x2 MFMA tiles of 32x32 (v_mfma_f32_32x32x16_bf16, or v_mfma_f32_32x32x2_f32 for
// the exact-fp32 path).  Per K-step the block stages through LDS
//   * the (TH+2)x(TW+2) input halo tile of one 64-byte channel chunk (32 bf16 / 16 f32 channels), re-used by all
//     nine taps (zero padding is materialised here, so the MFMA loop is branch-free), and
//   * the 9 x 64 x 64-byte filter slab of that chunk,
// both in a [16-byte group][row] layout so every ds_read_b128 of a fragment walks consecutive 16-B slots
// (conflict-free), with the row count padded to == 2 (mod 8) so the staging ds_write_b128 are conflict-free too.
// Global loads for K-step k+1 are issued into registers before the MFMA loop of K-step k (issue-early /
// write-late), two workgroups per CU overlap each other's barriers.
//
// A-operand gathers (MODE):  CONV3 3x3/s1/p1 halo tile;  PW 1x1;  UP2 = the data-gradient of ConvTranspose2d(k2,s2):
//                            K runs over (dy,dx,c) and pixel (y,x) reads input pixel (2y+dy, 2x+dx).
// Epilogues (EPI):  NHWC  (+bias, ReLU, per-channel sum / sum-of-squares for BatchNorm, store T),
//                   UP2   (ConvTranspose2d forward: column n = (dy,dx,co) is scattered to pixel (2y+dy,2x+dx)),
//                   NCHW  (logits head: fp32 NCHW, only the logical classes).
#include <string.h>
#include <stdio.h>
#include "common.hip.h"
#include "igemm_common.hip.h"
#include "clamd_internal.h"
#include "wino_common.hip.h"

namespace clamd {

template <typename T, int MODE, int EPI, int TW, int VAR>
__global__ void __launch_bounds__(256, 2) igemm_kernel(const IgemmParams p) {
    using G = Geo<MODE, TW>;
    constexpr int TH = G::TH, NT = G::NT, NIN = G::NIN, HW_ = G::HW_, NPIX = G::NPIX, NPIXP = G::NPIXP, NJ = G::NJ;
    constexpr int KC = DT<T>::KC, VEC = DT<T>::VEC;
    constexpr bool SPLIT = sizeof(T) == 4 && DT<T>::KC == 16 && !__is_same(T, float);
    __shared__ uint4 smem[G::SLOTS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // ---- which tile -------------------------------------------------------------------------------
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int ntm = tiles_x * tiles_y * p.B, ntn = (p.Np + 63) >> 6;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    if (p.m_fastest) { tm = bid % ntm; tn = bid / ntm; } else { tn = bid % ntn; tm = bid / ntn; }
    const int x0 = (tm % tiles_x) * TW, y0 = ((tm / tiles_x) % tiles_y) * TH, b = tm / (tiles_x * tiles_y);
    const int n0 = tn * 64;

    // ---- per-thread staging descriptors (constant over the K loop) ---------------------------------
    // Buffer descriptors: input = this image only (per-image byte offsets always fit 31 bits), filters = whole pack.
    constexpr int ESZ = sizeof(T);
    const int g4 = tid & 3;
    const unsigned img_elems = (MODE == MODE_UP2 ? 4u : 1u) * (unsigned)p.H * (unsigned)p.W * (unsigned)p.x_ldc;
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc((const char*)p.x + (size_t)b * img_elems * ESZ, img_elems * ESZ);
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(p.w, (unsigned)((MODE == MODE_CONV3 ? 9u : 1u) * p.Np * p.Kp * ESZ));
    // staging slot j of this thread; the ragged last pass wraps around and re-stages the first pixels (same data, same
    // LDS slot), so no store is conditional (a store guarded by "slot < NPIX" lets hipcc sink the load next to it)
    auto slot_pix = [&](int j) { const int pix = (tid >> 2) + 64 * j; return pix >= NPIX ? pix - NPIX : pix; };
    unsigned in_vo[NJ];   // byte offset of this thread's 16-B group at channel 0 inside the image, or BUF_OOB (-> zeros)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int pix = slot_pix(j);
        const int hy = pix / HW_, hx = pix - hy * HW_;
        unsigned off = BUF_OOB;
        {
            if constexpr (MODE == MODE_CONV3) {
                const int yy = y0 + hy - 1, xx = x0 + hx - 1;
                if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) off = ((yy * p.W + xx) * p.x_ldc + g4 * VEC) * ESZ;
            } else if constexpr (MODE == MODE_PW) {
                const int yy = y0 + hy, xx = x0 + hx;
                if (yy < p.H && xx < p.W) off = ((yy * p.W + xx) * p.x_ldc + g4 * VEC) * ESZ;
            } else {
                const int yy = y0 + hy, xx = x0 + hx;
                if (yy < p.H && xx < p.W) off = ((2 * yy * 2 * p.W + 2 * xx) * p.x_ldc + g4 * VEC) * ESZ;
            }
        }
        in_vo[j] = off;
    }
    const int wco = tid >> 2;
    // filters: CONV3 packs are K-chunk-major [chunk][tap][n][KC]; PW/UP2 packs are [n][Kp]
    const unsigned w_vo = n0 + wco < p.Np
        ? (unsigned)(((n0 + wco) * (MODE == MODE_CONV3 ? KC : p.Kp) + g4 * VEC) * ESZ) : BUF_OOB;
    const unsigned w_slab = (unsigned)(p.Np * KC * ESZ);     // CONV3: bytes of one [tap] slab of one K-chunk

    uint4 rin[NIN][NJ], rw[NT];
    const uint4 zero4_ = make_uint4(0u, 0u, 0u, 0u);

// Global -> register staging of K-step `ks` (macro, not a lambda: the register arrays must stay in VGPRs).
// One buffer_load_dwordx4 per 16 bytes; the only per-K-step address work is scalar (soffset).
#define IGEMM_GLOAD(ks_)                                                                                          \
    do {                                                                                                          \
        if constexpr (MODE == MODE_CONV3) {                                                                       \
            const unsigned so_ = (unsigned)((ks_) * KC * ESZ);                                                    \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                        \
                rin[0][j] = VAR == 4 ? zero4_ : buf_ld16(xrs, in_vo[j], so_);                                     \
            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                        \
                rw[t] = (VAR == 3 || VAR == 4) ? zero4_ : buf_ld16(wrs, w_vo, (unsigned)((ks_) * NT + t) * w_slab); \
        } else {                                                                                                  \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                      \
                const int k0_ = ((ks_) * NT + t) * KC;                                                            \
                const bool kok_ = k0_ < p.Kp;                      /* wave-uniform */                             \
                int koff_ = k0_;                                                                                  \
                if constexpr (MODE == MODE_UP2) {                                                                 \
                    const int q_ = k0_ / p.aux, c0_ = k0_ - q_ * p.aux;                                           \
                    koff_ = ((q_ >> 1) * 2 * p.W + (q_ & 1)) * p.x_ldc + c0_;                                     \
                }                                                                                                 \
                /* past-the-end K-steps load with every lane out of range (zeros, no traffic): a "load or zero" select */ \
                /* on a runtime condition makes hipcc branch around each load and drain vmcnt per element            */ \
                _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                    \
                    rin[t][j] = buf_ld16(xrs, kok_ ? in_vo[j] : BUF_OOB, (unsigned)(koff_ * ESZ));                \
                rw[t] = buf_ld16(wrs, kok_ ? w_vo : BUF_OOB, (unsigned)(k0_ * ESZ));                              \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)

// Registers -> LDS (unconditional: see slot_pix).
#define IGEMM_LDS_STORE()                                                                                         \
    do {                                                                                                          \
        _Pragma("unroll") for (int t = 0; t < NIN; ++t)                                                           \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                      \
                const int pix_ = slot_pix(j);                                                                     \
                /* split_t: the 64-byte K-chunk already is [hi 0-7][hi 8-15][lo 0-7][lo 8-15] = LDS slots 0..3 */ \
                smem[t * 4 * NPIXP + g4 * NPIXP + pix_] = rin[t][j];                                              \
            }                                                                                                     \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) smem[G::IN_SLOTS + (t * 4 + g4) * G::WG + wco] = rw[t];    \
    } while (0)

    // ---- fragment addresses ---------------------------------------------------------------------------
    int apix[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = 64 * wave + 32 * mt + r;
        apix[mt] = (m / TW) * HW_ + (m % TW);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = MODE == MODE_CONV3 ? p.Kp / KC : (p.Kp + KC * NT - 1) / (KC * NT);
    IGEMM_GLOAD(0);
    for (int ks = 0; ks < nk; ++ks) {
        if (ks) __syncthreads();          // every wave finished reading the previous K-step's tiles
        IGEMM_LDS_STORE();
        __syncthreads();
        if (ks + 1 < nk) IGEMM_GLOAD(ks + 1);   // in flight during the MFMA loop below
        if constexpr (SPLIT) {
            // one 16-channel k-step per tap: groups h / 2+h hold the hi / lo halves; 3 MFMAs per tile
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int in_base = MODE == MODE_CONV3 ? (t / 3) * HW_ + (t % 3) : t * 4 * NPIXP;
                const uint4 ah0 = smem[in_base + h * NPIXP + apix[0]], al0 = smem[in_base + (2 + h) * NPIXP + apix[0]];
                const uint4 ah1 = smem[in_base + h * NPIXP + apix[1]], al1 = smem[in_base + (2 + h) * NPIXP + apix[1]];
                const uint4 bh0 = smem[G::IN_SLOTS + (t * 4 + h) * G::WG + r], bl0 = smem[G::IN_SLOTS + (t * 4 + 2 + h) * G::WG + r];
                const uint4 bh1 = smem[G::IN_SLOTS + (t * 4 + h) * G::WG + 32 + r], bl1 = smem[G::IN_SLOTS + (t * 4 + 2 + h) * G::WG + 32 + r];
                mma_bf16(al0, bh0, acc[0][0]); mma_bf16(ah0, bl0, acc[0][0]); mma_bf16(ah0, bh0, acc[0][0]);
                mma_bf16(al0, bh1, acc[0][1]); mma_bf16(ah0, bl1, acc[0][1]); mma_bf16(ah0, bh1, acc[0][1]);
                mma_bf16(al1, bh0, acc[1][0]); mma_bf16(ah1, bl0, acc[1][0]); mma_bf16(ah1, bh0, acc[1][0]);
                mma_bf16(al1, bh1, acc[1][1]); mma_bf16(ah1, bl1, acc[1][1]); mma_bf16(ah1, bh1, acc[1][1]);
            }
        } else if constexpr (VAR == 0 || VAR >= 3) {   // 3..6 are timing ablations (tools/conv_ab.py)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int in_base = MODE == MODE_CONV3 ? (t / 3) * HW_ + (t % 3) : t * 4 * NPIXP;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int g = kk * 2 + h;
                    const uint4 a0 = smem[in_base + g * NPIXP + apix[0]];
                    const uint4 a1 = smem[in_base + g * NPIXP + apix[1]];
                    const uint4 b0 = smem[G::IN_SLOTS + (t * 4 + g) * G::WG + r];
                    const uint4 b1 = smem[G::IN_SLOTS + (t * 4 + g) * G::WG + 32 + r];
                    if constexpr (VAR == 5) {      // ablation: LDS reads kept alive, no MFMA
                        asm volatile("" :: "v"(a0.x), "v"(a1.x), "v"(b0.x), "v"(b1.x));
                    } else {
                        mma16<T>(a0, b0, acc[0][0]);
                        mma16<T>(a0, b1, acc[0][1]);
                        mma16<T>(a1, b0, acc[1][0]);
                        mma16<T>(a1, b1, acc[1][1]);
                    }
                }
            }
        } else {
            // fragments of step s+1 are read from LDS while the MFMAs of step s run (explicit double buffering)
            constexpr int NSTEP = 2 * NT;
            constexpr int NMF = sizeof(T) == 2 ? 4 : 16;
            uint4 fa0, fa1, fb0, fb1, na0, na1, nb0, nb1;
#define IGEMM_FRAG(s_, a0_, a1_, b0_, b1_)                                                                       \
    do {                                                                                                          \
        const int t_ = (s_) >> 1, g_ = ((s_) & 1) * 2 + h;                                                        \
        const int ib_ = MODE == MODE_CONV3 ? (t_ / 3) * HW_ + (t_ % 3) : t_ * 4 * NPIXP;                          \
        a0_ = smem[ib_ + g_ * NPIXP + apix[0]];                                                                   \
        a1_ = smem[ib_ + g_ * NPIXP + apix[1]];                                                                   \
        b0_ = smem[G::IN_SLOTS + (t_ * 4 + g_) * G::WG + r];                                                      \
        b1_ = smem[G::IN_SLOTS + (t_ * 4 + g_) * G::WG + 32 + r];                                                 \
    } while (0)
            IGEMM_FRAG(0, fa0, fa1, fb0, fb1);
            if constexpr (VAR >= 1) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // R(0) leads: R(s+1) precedes M(s)
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                if (st + 1 < NSTEP) IGEMM_FRAG(st + 1, na0, na1, nb0, nb1);
                if constexpr (VAR == 2) __builtin_amdgcn_s_setprio(1);
                mma16<T>(fa0, fb0, acc[0][0]);
                mma16<T>(fa0, fb1, acc[0][1]);
                mma16<T>(fa1, fb0, acc[1][0]);
                mma16<T>(fa1, fb1, acc[1][1]);
                if constexpr (VAR == 2) __builtin_amdgcn_s_setprio(0);
                if constexpr (VAR >= 1) {
                    if (st + 1 < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
                }
                fa0 = na0; fa1 = na1; fb0 = nb0; fb1 = nb1;
            }
#undef IGEMM_FRAG
        }
    }

    // ---- epilogue -----------------------------------------------------------------------------------
    if constexpr (VAR == 6) {      // ablation: no epilogue (accumulators kept alive, nothing stored)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(acc[i][j]));
        return;
    }
    // accumulator (mt, nt, reg): pixel m = 64*wave + 32*mt + acc_row(reg,h), channel n = n0 + 32*nt + r
    float bcol[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n0 + 32 * nt + r;
        const int bi = EPI == EPI_UP2 ? n % p.aux : n;
        bcol[nt] = (p.bias && n < p.Np) ? p.bias[bi] : 0.f;
    }
    unsigned vmask = 0;   // bit (mt*16 + reg): the pixel of that accumulator row lies inside the image
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = 64 * wave + 32 * mt + acc_row(e, h);
            if (y0 + m / TW < p.H && x0 + m % TW < p.W) vmask |= 1u << (mt * 16 + e);
        }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[mt][nt][e] + bcol[nt];
                if (p.relu) v = fmaxf(v, 0.f);
                acc[mt][nt][e] = v;
            }

    if constexpr (EPI == EPI_NCHW) {
        if (p.pred) {
            // arg-max fused into the head (trainer.py:279 `torch.max(outputs, 1)`): lane = class, register = pixel; a butterfly
            // over the 32 lanes of a half-wave on (value, class) pairs, ties to the LOWER class (first maximum, as torch).
            // Only the first 64-class slab takes part (num_classes <= 64 is checked by the launcher; tn == 0 here).
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v0 = (n0 + r < p.aux) ? acc[mt][0][e] : -__builtin_inff();
                    float v1 = (n0 + 32 + r < p.aux) ? acc[mt][1][e] : -__builtin_inff();
                    int i0 = n0 + r;
                    if (v1 > v0) { v0 = v1; i0 = n0 + 32 + r; }
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) {
                        const float ov = __shfl_xor(v0, o);
                        const int oi = __shfl_xor(i0, o);
                        if (ov > v0 || (ov == v0 && oi < i0)) { v0 = ov; i0 = oi; }
                    }
                    const int m = 64 * wave + 32 * mt + acc_row(e, h);
                    const int yy = y0 + m / TW, xx = x0 + m % TW;
                    if (r == ((mt * 16 + e) & 31) && yy < p.H && xx < p.W) p.pred[((long long)b * p.H + yy) * p.W + xx] = i0;
                }
            if (!p.y) return;
        }
        // logits: lane = class, 4 consecutive accumulator registers = 4 consecutive pixels along W
        float* out = (float*)p.y;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + 32 * nt + r;
            if (n >= p.aux) continue;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = 64 * wave + 32 * mt + 8 * q + 4 * h;
                    const int yy = y0 + m / TW, xx = x0 + m % TW;
                    if (yy >= p.H) continue;
                    float* dst = out + (((long long)b * p.aux + n) * p.H + yy) * p.W + xx;
                    if (xx + 3 < p.W && (p.W & 3) == 0) {
                        *reinterpret_cast<float4*>(dst) = make_float4(acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1],
                                                                      acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (xx + e < p.W) dst[e] = acc[mt][nt][4 * q + e];
                    }
                }
        }
        return;
    } else {
        __syncthreads();   // all waves are done with the staging tiles: LDS becomes the transposition buffer
        float* ebuf = reinterpret_cast<float*>(smem);
        if constexpr (EPI == EPI_NHWC) {
            if (p.stats) {
                float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
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
                // [wave][2][64] partials at the top of the buffer (beyond the 4*32*68 transposition area? no:
                // use a separate tail region that the transposition does not touch)
                float* sbuf = ebuf + 4 * 32 * 68;
                if (h == 0) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        sbuf[(wave * 2 + 0) * 64 + 32 * nt + r] = s1[nt];
                        sbuf[(wave * 2 + 1) * 64 + 32 * nt + r] = s2[nt];
                    }
                }
                __syncthreads();
                if (tid < 128) {
                    const int k = tid >> 6, c = tid & 63;
                    const float t = sbuf[(0 * 2 + k) * 64 + c] + sbuf[(1 * 2 + k) * 64 + c] +
                                    sbuf[(2 * 2 + k) * 64 + c] + sbuf[(3 * 2 + k) * 64 + c];
                    if (n0 + c < p.Np)
                        p.stats[((size_t)tm * 2 + k) * p.Np + n0 + c] = t;          // partial row of this pixel tile
                }
            }
        }
        T* out = (T*)p.y;
        float* wbuf = ebuf + wave * 32 * 68;
        float bs[5][8];
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) bs[k][j] = 0.f;
        const bool do_bn = EPI == EPI_NHWC && p.bn_y != nullptr;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            __syncthreads();
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) wbuf[acc_row(e, h) * 68 + 32 * nt + r] = acc[mt][nt][e];
            __syncthreads();
            // 8 lanes per pixel row (8 channels each), 8 rows per pass, 4 passes
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = ps * 8 + (lane >> 3), cgp = lane & 7;
                const int m = 64 * wave + 32 * mt + row;
                const int yy = y0 + m / TW, xx = x0 + m % TW;
                const int n = n0 + cgp * 8;
                if (yy < p.H && xx < p.W && n < p.Np) {
                    float v[8];
                    const float4 lo = *reinterpret_cast<const float4*>(wbuf + row * 68 + cgp * 8);
                    const float4 hi = *reinterpret_cast<const float4*>(wbuf + row * 68 + cgp * 8 + 4);
                    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
                    long long off;
                    if constexpr (EPI == EPI_NHWC) {
                        off = (((long long)b * p.H + yy) * p.W + xx) * p.y_ldc + n;
                    } else {   // EPI_UP2
                        const int q = n / p.aux, co = n - q * p.aux;
                        off = (((long long)b * 2 * p.H + 2 * yy + (q >> 1)) * 2 * p.W + 2 * xx + (q & 1)) * p.y_ldc + co;
                    }
                    Vec8<T>::store(out + off, v);
                    if constexpr (EPI == EPI_NHWC) {
                        if (do_bn) {
                            float yv[8];
                            Vec8<T>::load((const T*)p.bn_y + (((long long)b * p.H + yy) * p.W + xx) * p.Np + n, yv);
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
        if constexpr (EPI == EPI_NHWC) {
            if (do_bn) {
                // lanes sharing (lane & 7) own the same 8 channels: fold the 8 row-lanes, then the 4 waves through LDS
#pragma unroll
                for (int k = 0; k < 5; ++k)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float t = bs[k][j];
                        t += __shfl_xor(t, 8); t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
                        bs[k][j] = t;
                    }
                __syncthreads();                       // transposition buffer is free again
                if (lane < 8) {
#pragma unroll
                    for (int k = 0; k < 5; ++k)
#pragma unroll
                        for (int j = 0; j < 8; ++j) ebuf[(wave * 5 + k) * 64 + lane * 8 + j] = bs[k][j];
                }
                __syncthreads();
                for (int i = tid; i < 5 * 64; i += 256) {
                    const int k = i >> 6, c = i & 63;
                    const float t = ebuf[(0 * 5 + k) * 64 + c] + ebuf[(1 * 5 + k) * 64 + c] + ebuf[(2 * 5 + k) * 64 + c] +
                                    ebuf[(3 * 5 + k) * 64 + c];
                    if (n0 + c < p.Np)
                        p.bn_sums[((size_t)tm * 5 + k) * p.Np + n0 + c] = t;
                }
            }
        }
    }
}

// EPI_NHWC statistics scratch lives after the transposition area: make sure the static buffer covers it.
static_assert(Geo<MODE_CONV3, 32>::SLOTS * 16 >= (4 * 32 * 68 + 4 * 2 * 64) * 4, "LDS too small for epilogue");
static_assert(Geo<MODE_CONV3, 16>::SLOTS * 16 >= (4 * 32 * 68 + 4 * 2 * 64) * 4, "LDS too small for epilogue");
static_assert(Geo<MODE_PW, 32>::SLOTS * 16 >= (4 * 32 * 68 + 4 * 2 * 64) * 4, "LDS too small for epilogue");
static_assert(Geo<MODE_PW, 16>::SLOTS * 16 >= (4 * 32 * 68 + 4 * 2 * 64) * 4, "LDS too small for epilogue");

// pixel tiles of the baseline kernel (256 pixels each) = its partial statistics rows
static long long igemm_tiles(const IgemmParams& p) {
    const int TW = p.W >= 32 ? 32 : 16, TH = 256 / TW;
    return (long long)((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH) * p.B;
}

template <typename T, int MODE, int EPI>
static int launch_tw(const IgemmParams& p, hipStream_t s, int variant) {
    const bool wide = p.W >= 32;
    const int TW = wide ? 32 : 16, TH = 256 / TW;
    const long long tiles = (long long)((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH) * p.B;
    const long long nblk = tiles * ((p.Np + 63) / 64);
    if (nblk <= 0 || nblk > 0x7fffffff) return clamd_fail("igemm: grid out of range");
    const int var = (MODE == MODE_CONV3 && EPI == EPI_NHWC && !__is_same(T, split_t)) ? variant : 0;
#define IGEMM_LAUNCH(TW_, V_) hipLaunchKernelGGL((igemm_kernel<T, MODE, EPI, TW_, V_>), dim3((unsigned)nblk), dim3(256), 0, s, p)
    if constexpr (MODE == MODE_CONV3 && EPI == EPI_NHWC && !__is_same(T, split_t)) {
        if (wide) { if (var == 1) IGEMM_LAUNCH(32, 1); else if (var == 2) IGEMM_LAUNCH(32, 2);
#ifdef CLAMD_DIAG   // timing ablations (results are wrong on purpose): diagnostic build only
                    else if (var == 3) IGEMM_LAUNCH(32, 3); else if (var == 4) IGEMM_LAUNCH(32, 4);
                    else if (var == 5) IGEMM_LAUNCH(32, 5); else if (var == 6) IGEMM_LAUNCH(32, 6);
#endif
                    else IGEMM_LAUNCH(32, 0); }
        else { if (var == 1) IGEMM_LAUNCH(16, 1); else if (var == 2) IGEMM_LAUNCH(16, 2); else IGEMM_LAUNCH(16, 0); }
    } else {
        if (wide) IGEMM_LAUNCH(32, 0); else IGEMM_LAUNCH(16, 0);
    }
#undef IGEMM_LAUNCH
    (void)var;
    return clamd_check_launch("igemm");
}

template <int MODE, int EPI>
static int launch(const IgemmParams& p, int dtype, hipStream_t s, int variant = 0) {
    if (dtype == CLAMD_BF16) return launch_tw<bf16_t, MODE, EPI>(p, s, variant);
    if (dtype == CLAMD_F32) return launch_tw<float, MODE, EPI>(p, s, variant);
    if (dtype == CLAMD_SPLIT) return launch_tw<split_t, MODE, EPI>(p, s, variant);
    return clamd_fail("igemm: bad dtype");
}

// Which structure runs a 3x3 launch (measured choices, tools/conv_ab.py) and how many partial statistics rows it writes.
struct Conv3Plan { int kind; int mt; int rows; };      // kind 0 = baseline, 1 = producer/consumer, 2 = persistent
static Conv3Plan plan_conv3x3(const IgemmParams& p, int dtype, const clamd_tuning& tn) {
    if (tn.igemm_pws == 2 || (tn.igemm_pws == 1 && p.Kp <= 256)) {    // measured: faster up to 256 input channels
        const int gm = pws_rows(p, dtype, tn);
        if (gm > 0) return {2, 2, gm};                                  // -1: shape not supported there, fall through
    }
    int mt = 0;
    if (tn.igemm_ws == 1 || tn.igemm_ws == 3 || tn.igemm_ws == 4)      // forced: 256- / 512- / 128-pixel tiles
        mt = tn.igemm_ws == 1 ? 2 : tn.igemm_ws == 3 ? 4 : 1;
    else if (tn.igemm_ws == 2 && p.Kp >= 256) {                        // (exactly 256 only when the persistent kernel declined)
        // one workgroup per CU: take the 512-pixel tile only if it still gives every CU a workgroup
        const long long ntn = (p.Np + 63) / 64;
        const long long blocks4 = (long long)p.B * ((p.H + 15) / 16) * ((p.W + 31) / 32) * ntn;
        const long long blocks2 = (long long)p.B * (p.W >= 32 ? ((p.H + 7) / 8) * ((p.W + 31) / 32) : ((p.H + 15) / 16) * ((p.W + 15) / 16)) * ntn;
        const long long fill = 7LL * clamd_usable_cus(tn) / 8;           // 224 of 256 CUs; fewer when CUs are left to RCCL
        mt = (p.W >= 32 && blocks4 >= fill) ? 4 : blocks2 >= fill ? 2 : 1;
    }
    if (mt) return {1, mt, ws_rows(p, mt)};
    const long long t = igemm_tiles(p);
    return {0, 2, t > 0x7fffffff ? -1 : (int)t};
}

static int check_common(const IgemmParams& p, const char* who, int dtype, bool y_nhwc = true) {
    if (p.B <= 0 || p.H <= 0 || p.W <= 0) return clamd_fail("igemm: empty problem");
    if (int e = clamd_check_split(dtype, p.x, p.x_ldc)) return e;
    if (y_nhwc) if (int e = clamd_check_split(dtype, p.y, p.y_ldc)) return e;
    if (p.bn_y) if (int e = clamd_check_split(dtype, p.bn_y, p.Np)) return e;
    if (p.Kp % 32 || p.Np % 32 || p.x_ldc % 8 || p.y_ldc % 8) return clamd_fail("igemm: channel counts/pitches must be padded (K,N %32, ldc %8)");
    // 32-bit element offsets inside the kernel
    // buffer descriptors address ONE image with 32-bit byte offsets (OOB marker = 2^31)
    const long long img_bytes = (long long)p.H * p.W * p.x_ldc * (who[0] == 'u' ? 4 : 1) * 4;
    if (img_bytes >= (1ll << 31)) return clamd_fail("igemm: one image exceeds 2^31 bytes");
    if ((long long)p.H * p.W * p.y_ldc * (who[0] == 'u' ? 4 : 1) * 4 >= (1ll << 31)) return clamd_fail("igemm: one output image exceeds 2^31 bytes");
    if ((long long)9 * p.Np * p.Kp * 4 >= (1ll << 31)) return clamd_fail("igemm: packed filter exceeds 2^31 bytes");
    return 0;
}

}  // namespace clamd

using namespace clamd;

extern "C" {

int clamd_sizeof_tuning(void) { return (int)sizeof(clamd_tuning); }
void clamd_tuning_init(clamd_tuning* t) { if (t) *t = clamd_default_tuning(); }

int clamd_stat_rows(int op, int B, int H, int W, int Cin_p, int Cout_p, int dtype, int fused_bn, const clamd_tuning* tune) {
    if (B <= 0 || H <= 0 || W <= 0 || Cout_p <= 0) return clamd_fail("stat_rows: empty problem");
    if (int e = clamd_check_tuning(tune)) return e;
    const clamd_tuning& tn = clamd_tune(tune);
    IgemmParams p{};
    p.B = B; p.H = H; p.W = W; p.Kp = Cin_p; p.Np = Cout_p;
    p.bn_y = fused_bn ? (const void*)&p : nullptr;          // only tested against null by the planners
    long long rows = -1;
    switch (op) {
    case CLAMD_OP_CONV3X3: rows = plan_conv3x3(p, dtype, tn).rows; break;
    case CLAMD_OP_CONV3X3_WINOGRAD: rows = clamd_winograd_stat_rows(B, H, W, Cout_p, tn); break;
    case CLAMD_OP_CONV3X3_WINOGRAD24: rows = clamd_winograd24_stat_rows(B, H, W, Cout_p, tn); break;
    case CLAMD_OP_CONV3X3_WINOGRAD44: rows = clamd_winograd44_stat_rows(B, H, W, Cout_p, tn); break;
    case CLAMD_OP_CONV1X1:
    case CLAMD_OP_CONVT2X2_DGRAD: rows = igemm_tiles(p); break;
    case CLAMD_OP_BN_BWD_REDUCE: rows = clamd_bn_bwd_reduce_rows(B, H, W, Cout_p, Cin_p != 0, tn); break;
    default: return clamd_fail("stat_rows: unknown op");
    }
    if (rows <= 0 || rows > 0x7fffffff) return clamd_fail("stat_rows: out of range");
    return (int)rows;
}

static int check_rows(const IgemmParams& p, int have, int need, const char* who) {
    if ((p.stats || p.bn_sums) && have != need) {
        char msg[160];
        snprintf(msg, sizeof(msg), "%s: stat_rows = %d but this launch writes %d partial rows (size the buffer with clamd_stat_rows)", who, have, need);
        return clamd_fail(msg);
    }
    return 0;
}

int clamd_conv3x3(const void* x, int x_ldc, const void* w_packed, const float* bias, void* y, int y_ldc,
                  float* stats, const void* bn_y, float* bn_sums, int stat_rows, int B, int H, int W, int Cin_p, int Cout_p,
                  int relu, int m_fastest, int dtype, const clamd_tuning* tune, void* stream) {
    if (relu & ~3) return clamd_fail("conv3x3: bad relu flags");
    IgemmParams p{x, x_ldc, w_packed, bias, y, y_ldc, stats, B, H, W, Cin_p, Cout_p, relu & 1, 0, m_fastest, bn_y, bn_sums};
    p.bias_classes = (relu & CLAMD_BIAS_BORDER_CLASSES) ? 1 : 0;
    if (int e = check_common(p, "conv3x3", dtype)) return e;
    if (int e = clamd_check_tuning(tune)) return e;
    if ((bn_y == nullptr) != (bn_sums == nullptr)) return clamd_fail("conv3x3: bn_y and bn_sums go together");
    const clamd_tuning& tn = clamd_tune(tune);
    const Conv3Plan pl = plan_conv3x3(p, dtype, tn);
    if (pl.rows <= 0) return clamd_fail("conv3x3: grid out of range");
    if (p.bias_classes && (pl.kind != 2 || !bias || H < 2 || W < 2))
        return clamd_fail("conv3x3: the border-class bias table needs the persistent kernel (ask clamd_conv3x3_border_bias_ok), a table and H, W >= 2");
    if (int e = check_rows(p, stat_rows, pl.rows, "conv3x3")) return e;
    if (pl.kind == 2) return launch_igemm_pws(p, dtype, (hipStream_t)stream, tn);
    if (pl.kind == 1) return launch_igemm_ws(p, dtype, (hipStream_t)stream, pl.mt);
    return launch<MODE_CONV3, EPI_NHWC>(p, dtype, (hipStream_t)stream, tn.igemm_variant);
}

int clamd_conv3x3_border_bias_ok(int B, int H, int W, int Cin_p, int Cout_p, int dtype, const clamd_tuning* tune) {
    if (B <= 0 || H < 2 || W < 2 || clamd_check_tuning(tune)) return 0;
    IgemmParams p{nullptr, Cin_p, nullptr, nullptr, nullptr, Cout_p, nullptr, B, H, W, Cin_p, Cout_p, 1, 0, 0, nullptr, nullptr};
    return plan_conv3x3(p, dtype, clamd_tune(tune)).kind == 2 ? 1 : 0;
}

int clamd_conv3x3_bn_sums(int B, int H, int W, int Cin_p, int Cout_p, int dtype, const clamd_tuning* tune) {
    if (B <= 0 || H <= 0 || W <= 0 || clamd_check_tuning(tune)) return 0;
    IgemmParams p{nullptr, Cin_p, nullptr, nullptr, nullptr, Cout_p, nullptr, B, H, W, Cin_p, Cout_p, 0, 0, 0, nullptr, nullptr};
    p.bn_y = (const void*)&p;          // only tested against null by the planner
    const clamd_tuning& tn = clamd_tune(tune);
    return (dtype == CLAMD_BF16 && plan_conv3x3(p, dtype, tn).kind == 2) ? 2 : 5;
}

int clamd_conv1x1(const void* x, int x_ldc, const void* w_packed, const float* bias, void* y, int y_ldc,
                  float* stats, const void* bn_y, float* bn_sums, int stat_rows, int B, int H, int W, int Cin_p, int Cout_p,
                  int relu, int dtype, void* stream) {
    if (relu & ~1) return clamd_fail("conv1x1: relu must be 0 or 1");
    IgemmParams p{x, x_ldc, w_packed, bias, y, y_ldc, stats, B, H, W, Cin_p, Cout_p, relu, 0, 0, bn_y, bn_sums};
    if (int e = check_common(p, "conv1x1", dtype)) return e;
    if ((bn_y == nullptr) != (bn_sums == nullptr)) return clamd_fail("conv1x1: bn_y and bn_sums go together");
    if (int e = check_rows(p, stat_rows, (int)igemm_tiles(p), "conv1x1")) return e;
    return launch<MODE_PW, EPI_NHWC>(p, dtype, (hipStream_t)stream);
}

int clamd_conv1x1_logits(const void* x, int x_ldc, const void* w_packed, const float* bias, float* logits_nchw,
                         int B, int H, int W, int Cin_p, int Cout_p, int num_classes, int dtype, void* stream) {
    IgemmParams p{x, x_ldc, w_packed, bias, logits_nchw, 8, nullptr, B, H, W, Cin_p, Cout_p, 0, num_classes, 0, nullptr, nullptr};
    if (int e = check_common(p, "conv1x1_logits", dtype, false)) return e;
    if (num_classes > Cout_p) return clamd_fail("conv1x1_logits: num_classes > padded Cout");
    return launch<MODE_PW, EPI_NCHW>(p, dtype, (hipStream_t)stream);
}

int clamd_conv1x1_argmax(const void* x, int x_ldc, const void* w_packed, const float* bias, long long* pred, float* logits_nchw,
                         int B, int H, int W, int Cin_p, int Cout_p, int num_classes, int dtype, void* stream) {
    IgemmParams p{x, x_ldc, w_packed, bias, logits_nchw, 8, nullptr, B, H, W, Cin_p, Cout_p, 0, num_classes, 0, nullptr, nullptr, pred};
    if (int e = check_common(p, "conv1x1_argmax", dtype, false)) return e;
    if (!pred) return clamd_fail("conv1x1_argmax: pred is null");
    if (num_classes < 1 || num_classes > Cout_p || num_classes > 64) return clamd_fail("conv1x1_argmax: num_classes must be in [1, min(64, padded Cout)]");
    return launch<MODE_PW, EPI_NCHW>(p, dtype, (hipStream_t)stream);
}

int clamd_convT2x2_fwd(const void* x, int x_ldc, const void* w_packed, const float* bias, void* y, int y_ldc, int B,
                       int h, int w, int Cin_p, int Cout_p, int dtype, void* stream) {
    IgemmParams p{x, x_ldc, w_packed, bias, y, y_ldc, nullptr, B, h, w, Cin_p, 4 * Cout_p, 0, Cout_p, 0, nullptr, nullptr};
    if (int e = check_common(p, "convT_fwd", dtype)) return e;
    return launch<MODE_PW, EPI_UP2>(p, dtype, (hipStream_t)stream);
}

int clamd_convT2x2_dgrad(const void* gy, int gy_ldc, const void* w_packed, void* gx, int gx_ldc, const void* bn_y,
                         float* bn_sums, int stat_rows, int B, int h, int w, int Cin_p, int Cout_p, int dtype, void* stream) {
    IgemmParams p{gy, gy_ldc, w_packed, nullptr, gx, gx_ldc, nullptr, B, h, w, 4 * Cout_p, Cin_p, 0, Cout_p, 0, bn_y, bn_sums};
    if (int e = check_common(p, "up2_dgrad", dtype)) return e;
    if ((bn_y == nullptr) != (bn_sums == nullptr)) return clamd_fail("convT2x2_dgrad: bn_y and bn_sums go together");
    if (int e = check_rows(p, stat_rows, (int)igemm_tiles(p), "convT2x2_dgrad")) return e;
    return launch<MODE_UP2, EPI_NHWC>(p, dtype, (hipStream_t)stream);
}

}  // extern "C"
