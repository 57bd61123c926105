// Weight-gradient kernels for gfx950 (wgrad of the 3x3 convs, the 1x1 head and the
// ConvTranspose2d up-path).  All three are one GEMM whose contraction runs over PIXELS:
//
//   dW[t][r][c] = sum_{pixels p} A[p, r] * B[nbr_t(p), c]      r: channels of A, c: channels of B, t: tap
//
//   CONV3: A = g_z (d loss / d conv output), B = conv input u, nbr_t(p) = p + (dy-1, dx-1), zero outside the image
//   PW   : A = d logits,                     B = head input,   nbr(p) = p
//   UP2  : A = convT input,                  B = d convT output, nbr_t(y,x) = (2y+dy, 2x+dx)
//
// Both MFMA operands are therefore "transposed" (k = pixel is the strided dimension of an NHWC tensor).  bf16 uses
// ds_read_b64_tr_b16, which gathers 4 pixels x 16 channels per 16-lane group straight from a [pixel][channel]
// LDS image (pixel stride 192 B so the four rows of a group land on distinct bank quarters); f32 needs one element
// per lane (32x32x2 MFMA), read with conflict-free ds_read_b32.
//
// A 256-thread workgroup owns a 64(r) x 64(c) x taps output tile, wave (wr,wc) a 32x32 x taps slice held in up to
// 9 accumulators, and loops over its share of the pixel tiles (split-K across workgroups).  Partial tiles are
// written as fp32 slabs [split][tap][r][c] and summed by wgrad_reduce_kernel into the parameter's own layout
// ([Cout][Cin][3][3] / [Cin][Cout][2][2]) -- deterministic, no float atomics.
#include "common.hip.h"
#include "clamd_internal.h"
#include "wgrad_common.hip.h"

namespace clamd {


#ifdef CLAMD_DIAG
// diagnostic build only (python build.py --diag): per-role cycle shares of the pixel-tile loop, summed over workgroups
__device__ unsigned long long g_wg_diag[8];
#define WGD_T() __builtin_amdgcn_s_memtime()
#define WGD_ADD(i_, v_) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_wg_diag[i_], (unsigned long long)(v_)); } while (0)
#else
#define WGD_T() 0ull
#define WGD_ADD(i_, v_) do { } while (0)
#endif

// WS = false: 256 threads, every wave stages and multiplies, two workgroups per CU cover each other's stalls.
// WS = true : 512 threads, waves 4-7 (producers) stream pixel tile i+1 from global memory into LDS stage (i+1)&1 while
//             waves 0-3 (consumers, one per SIMD) multiply stage i&1; one barrier per pixel tile, one workgroup per CU
//             (two 64-KB stages), so half as many split-K slabs are written and reduced.  Same tile, LDS image,
//             fragment maps and summation order inside a slab as WS = false.
template <typename T, int MODE, int TW, bool WS>
__global__ void __launch_bounds__(WS ? 512 : 256, 2) wgrad_kernel(const WgradParams p) {
    using G = WGeo<T, MODE, TW>;
    constexpr int TH = G::TH, NT = G::NT, BW = G::BW, STRIDE = G::STRIDE, GPP = G::GPP, NJA = G::NJA, NJB = G::NJB;
    constexpr int VEC = DT<T>::VEC;
    constexpr bool SPLIT = G::SPLIT;
    constexpr int EPI_BYTES = 4 * 8 * 32 * NT * 4;        // epilogue: 8 rows x 32 cols x NT taps of fp32 per wave
    static_assert(EPI_BYTES <= G::BYTES, "epilogue staging must fit the pixel-tile image");
    static_assert((WS ? 2 : 1) * G::BYTES <= 160 * 1024, "LDS stages must fit one CU");
    __shared__ __attribute__((aligned(16))) char smem[(WS ? 2 : 1) * G::BYTES];
    constexpr int LO = (G::APIX + G::BPIX) * STRIDE;      // split: byte offset of the lo images (same layout as hi)

    const int tid = WS ? (threadIdx.x & 255) : threadIdx.x;         // index inside the role
    const int lane = tid & 63, wave = tid >> 6;
    const bool producer = WS && threadIdx.x >= 256;
    const int wr = wave >> 1, wc = wave & 1;

    const int rt = (p.Rp + 63) >> 6, ct = (p.Cp + 63) >> 6;
    int bid = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;     // an XCD gets a contiguous id range: neighbouring (tr, tc) tiles of ONE pixel range share its L2
    const int tr = bid % rt; bid /= rt;
    const int tc = bid % ct; bid /= ct;
    const int split = bid;
    const int r0 = tr * 64, c0 = tc * 64;

    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * p.B;
    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(ntiles, t_begin + p.tiles_per_split);

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    uint4 ra[NJA], rb[NJB];                  // staging registers (WS producers: a second set, ra2/rb2, below)

    // Per-thread staging descriptors, constant over the whole pixel loop: tile-relative byte offsets (the tile origin
    // goes into the scalar soffset of the buffer load) and the tile-relative coordinates needed for the edge tests.
    constexpr int ESZ = sizeof(T);
    constexpr int BSCALE = MODE == WG_UP2 ? 2 : 1;                  // B lives on the 2H x 2W grid for the convT gradient
    constexpr int HALO = MODE == WG_CONV3 ? 1 : 0;
    const unsigned a_img = (unsigned)p.H * p.W * p.a_ldc * ESZ;
    const unsigned b_img = (unsigned)(BSCALE * p.H) * (BSCALE * p.W) * p.b_ldc * ESZ;
    const unsigned b_shift = (unsigned)(HALO * (BSCALE * p.W + 1)) * p.b_ldc * ESZ;   // descriptor base sits one row + one pixel early
    // The split (bf16x3) variant is register-starved (144 accumulators + two fragment sets): it re-derives these few
    // integers per tile instead of keeping them (RECOMP), everything else precomputes them once.
    constexpr bool RECOMP = !WS && SPLIT && MODE == WG_CONV3 && TW == 32;
    unsigned a_vo[RECOMP ? 1 : NJA], b_vo[RECOMP ? 1 : NJB];
    unsigned b_hyx[RECOMP ? 1 : NJB];                               // (hy << 16) | hx: BW is not a power of two
#pragma unroll
    for (int j = 0; j < NJA; ++j) {
        const int i = wrap_idx(tid + 256 * j, G::APIX * GPP), pix = i / GPP, g = i % GPP;
        if constexpr (!RECOMP)
            a_vo[j] = (r0 + g * VEC < p.Rp) ? (unsigned)((((pix / TW) * p.W + pix % TW) * p.a_ldc + r0 + g * VEC) * ESZ) : BUF_OOB;
    }
#pragma unroll
    for (int j = 0; j < NJB; ++j) {
        const int i = wrap_idx(tid + 256 * j, G::BPIX * GPP), pix = i / GPP, g = i % GPP;
        const int hy = pix / BW, hx = pix % BW;
        if constexpr (!RECOMP) {
            b_hyx[j] = ((unsigned)hy << 16) | (unsigned)hx;
            b_vo[j] = (c0 + g * VEC < p.Cp)
                          ? (unsigned)(((hy * BSCALE * p.W + hx) * p.b_ldc + c0 + g * VEC) * ESZ) : BUF_OOB;
        }
    }

    const T* __restrict__ ag = (const T*)p.a;
    const T* __restrict__ bg = (const T*)p.b;
    auto gload = [&](int tile_, uint4 (&ra)[NJA], uint4 (&rb)[NJB]) {
        // tiles past the end (WS producers run a fixed load schedule) issue the same loads with every lane out of range
        const bool live = tile_ < t_end;
        const int tile = live ? tile_ : t_begin;
        // tile -> (image, origin): wave-uniform scalar arithmetic
        const int x0 = (tile % tiles_x) * TW, y0 = ((tile / tiles_x) % tiles_y) * TH, b = tile / (tiles_x * tiles_y);
        if constexpr (RECOMP) {
            // register-starved split variant: no persistent per-thread descriptors; predicated 64-bit loads whose
            // addresses are rebuilt per tile (the buffer-load form below kept 25 more VGPRs live and spilled)
#pragma unroll
            for (int j = 0; j < NJA; ++j) {
                const int i = wrap_idx(tid + 256 * j, G::APIX * GPP), pix = i / GPP, g = i % GPP;
                const int yy = y0 + pix / TW, xx = x0 + pix % TW;
                const bool ok = yy < p.H && xx < p.W && r0 + g * VEC < p.Rp;
                ra[j] = ldg16(ag + ((long long)(b * p.H + yy) * p.W + xx) * p.a_ldc + r0 + g * VEC, ok);
            }
#pragma unroll
            for (int j = 0; j < NJB; ++j) {
                const int i = wrap_idx(tid + 256 * j, G::BPIX * GPP), pix = i / GPP, g = i % GPP;
                const int hy = pix / BW, hx = pix % BW;
                const int yy = y0 + hy - 1, xx = x0 + hx - 1;
                const bool ok = c0 + g * VEC < p.Cp && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                rb[j] = ldg16(bg + ((long long)(b * p.H + yy) * p.W + xx) * p.b_ldc + c0 + g * VEC, ok);
            }
            return;
        }
        const __amdgpu_buffer_rsrc_t ars = make_rsrc((const char*)p.a + (size_t)b * a_img, a_img);
        const __amdgpu_buffer_rsrc_t brs = make_rsrc((const char*)p.b + (size_t)b * b_img - b_shift, b_img + b_shift);
        const unsigned a_so = (unsigned)((y0 * p.W + x0) * p.a_ldc * ESZ);
        const unsigned b_so = (unsigned)((BSCALE * y0 * BSCALE * p.W + BSCALE * x0) * p.b_ldc * ESZ);
#pragma unroll
        for (int j = 0; j < NJA; ++j) {
            const int i = wrap_idx(tid + 256 * j, G::APIX * GPP), pix = i / GPP, g = i % GPP;   // TW, GPP are powers of two: shifts
            const bool ok = live && y0 + pix / TW < p.H && x0 + pix % TW < p.W;
            (void)g;
            ra[j] = buf_ld16(ars, ok ? a_vo[j] : BUF_OOB, a_so);
        }
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
            const int yy = BSCALE * y0 + (int)(b_hyx[j] >> 16) - HALO, xx = BSCALE * x0 + (int)(b_hyx[j] & 0xffffu) - HALO;
            const bool ok = live && yy >= 0 && yy < BSCALE * p.H && xx >= 0 && xx < BSCALE * p.W;
            rb[j] = buf_ld16(brs, ok ? b_vo[j] : BUF_OOB, b_so);
        }
    };
    auto lds_store = [&](int stage, const uint4 (&ra)[NJA], const uint4 (&rb)[NJB]) {
        char* const sa = smem + stage * G::BYTES;
        char* const sb = sa + G::APIX * STRIDE;
#pragma unroll
        for (int j = 0; j < NJA; ++j) {
            const int i = wrap_idx(tid + 256 * j, G::APIX * GPP), pix = i / GPP, g = i % GPP;
#ifdef WG_ABLATE_STORE
            if (j > 0) { asm volatile("" :: "v"(ra[j].x), "v"(ra[j].y), "v"(ra[j].z), "v"(ra[j].w)); continue; }
#endif
            if constexpr (!SPLIT) {
                *reinterpret_cast<uint4*>(sa + pix * STRIDE + g * 16) = ra[j];
            } else {
                // piece g of the pixel's four [16 hi][16 lo] groups: pieces 0,1 of a group go to the hi image, 2,3 to the lo image
                *reinterpret_cast<uint4*>(sa + ((g & 2) ? LO : 0) + pix * STRIDE + (g >> 2) * 32 + (g & 1) * 16) = ra[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
            const int i = wrap_idx(tid + 256 * j, G::BPIX * GPP), pix = i / GPP, g = i % GPP;
#ifdef WG_ABLATE_STORE
            if (j > 0) { asm volatile("" :: "v"(rb[j].x), "v"(rb[j].y), "v"(rb[j].z), "v"(rb[j].w)); continue; }
#endif
            if constexpr (!SPLIT) {
                *reinterpret_cast<uint4*>(sb + pix * STRIDE + g * 16) = rb[j];
            } else {
                // piece g of the pixel's four [16 hi][16 lo] groups: pieces 0,1 of a group go to the hi image, 2,3 to the lo image
                *reinterpret_cast<uint4*>(sb + ((g & 2) ? LO : 0) + pix * STRIDE + (g >> 2) * 32 + (g & 1) * 16) = rb[j];
            }
        }
    };

    // B-tile pixel index of A-tile pixel (ty, tx) for tap t
    auto bpix = [&](int ty, int tx, int t) -> int {
        if constexpr (MODE == WG_CONV3) return (ty + t / 3) * BW + tx + t % 3;
        else if constexpr (MODE == WG_PW) return ty * BW + tx;
        else return (2 * ty + (t >> 1)) * BW + 2 * tx + (t & 1);
    };

    // the multiply of one staged pixel tile (all waves when !WS, consumer waves when WS)
    auto multiply = [&](int stage) {
        const char* const sa = smem + stage * G::BYTES;
        const char* const sb = sa + G::APIX * STRIDE;
        if constexpr (sizeof(T) == 2 || SPLIT) {
            // bf16 / split: k-group = 16 consecutive pixels of one tile row.  Transposed reads: lane -> (gq = lane>>4,
            // q = (lane>>2)&3, pp = lane&3); it supplies the address of pixel (8*(gq>>1) + q) [+4 for the second
            // read], channels 16*(gq&1) + 4*pp .. +3, and receives 4 pixels of channel 16*(gq&1) + (lane&15).
            const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
            const int kq = 8 * (gq >> 1) + q;
            const int a_ch = (32 * wr + 16 * (gq & 1) + 4 * pp) * 2;
            const int b_ch = (32 * wc + 16 * (gq & 1) + 4 * pp) * 2;
            if constexpr (WS && MODE == WG_CONV3) {
                // Consumer waves run one per SIMD, so nothing but this wave's own schedule hides the LDS latency: the
                // tile is walked in groups (16-pixel k-step u, tap row dy) = 3 MFMA products, and the fragment reads of
                // group g+D are issued before the MFMAs of group g (D = 2, split: 1; a ring of D+1 fragment sets, the
                // A fragment of a k-step shared by its three groups).  Prefetches past the last k-step read valid LDS
                // (the halo rows TH, TH+1 / the neighbouring image) and are never used.
                constexpr int XPR = TW / 16, NU = TH * XPR;               // k-steps per tile row / per tile
                constexpr int D = 1, RS = D + 1;
                static_assert(NU % 2 == 0 && 6 % RS == 0, "two k-steps (six groups) per loop iteration");
                constexpr int NRD = (SPLIT ? 2 : 1) * 6, NRA = (SPLIT ? 2 : 1) * 2, NMF = SPLIT ? 9 : 3;
                uint4 Ah[2], Al[2], Bh[RS][3], Bl[RS][3];
                const char* const ab = sa + kq * STRIDE + a_ch;
                const char* const bb = sb + kq * STRIDE + b_ch;
#define WG_LOADG(u_, dy_, slot_, par_)                                                                             \
    do {                                                                                                           \
        const int ty_ = (u_) / XPR, xs_ = ((u_) % XPR) * 16;                                                       \
        if ((dy_) == 0) {                                                                                          \
            const char* ap_ = ab + (ty_ * TW + xs_) * STRIDE;                                                      \
            const uint2 a0_ = ds_tr16(ap_), a1_ = ds_tr16(ap_ + 4 * STRIDE);                                       \
            Ah[par_] = make_uint4(a0_.x, a0_.y, a1_.x, a1_.y);                                                     \
            if constexpr (SPLIT) {                                                                                 \
                const uint2 l0_ = ds_tr16(ap_ + LO), l1_ = ds_tr16(ap_ + LO + 4 * STRIDE);                         \
                Al[par_] = make_uint4(l0_.x, l0_.y, l1_.x, l1_.y);                                                 \
            }                                                                                                      \
        }                                                                                                          \
        _Pragma("unroll") for (int dx_ = 0; dx_ < 3; ++dx_) {                                                      \
            const char* bp_ = bb + ((ty_ + (dy_)) * BW + xs_ + dx_) * STRIDE;                                      \
            const uint2 b0_ = ds_tr16(bp_), b1_ = ds_tr16(bp_ + 4 * STRIDE);                                       \
            Bh[slot_][dx_] = make_uint4(b0_.x, b0_.y, b1_.x, b1_.y);                                               \
            if constexpr (SPLIT) {                                                                                 \
                const uint2 m0_ = ds_tr16(bp_ + LO), m1_ = ds_tr16(bp_ + LO + 4 * STRIDE);                         \
                Bl[slot_][dx_] = make_uint4(m0_.x, m0_.y, m1_.x, m1_.y);                                           \
            }                                                                                                      \
        }                                                                                                          \
    } while (0)
#pragma unroll
                for (int g = 0; g < D; ++g) WG_LOADG(g / 3, g % 3, g % RS, (g / 3) & 1);
                __builtin_amdgcn_sched_group_barrier(0x100, NRA + D * NRD, 0);          // the prologue reads lead
#define WG_STEP(g_)                                                                                                \
    do {                                                                                                           \
        constexpr int gn_ = (g_) + D;                                              /* group to prefetch */         \
        WG_LOADG(ub + gn_ / 3, gn_ % 3, gn_ % RS, (gn_ / 3) & 1);                                                  \
        constexpr int par_s = ((g_) / 3) & 1, sl_ = (g_) % RS, dy_s = (g_) % 3;                                    \
        _Pragma("unroll") for (int dx = 0; dx < 3; ++dx) {                                                         \
            if constexpr (SPLIT) {                                                                                 \
                mma_bf16(Al[par_s], Bh[sl_][dx], acc[3 * dy_s + dx]);                                              \
                mma_bf16(Ah[par_s], Bl[sl_][dx], acc[3 * dy_s + dx]);                                              \
            }                                                                                                      \
            mma_bf16(Ah[par_s], Bh[sl_][dx], acc[3 * dy_s + dx]);                                                  \
        }                                                                                                          \
        /* issue order: the MFMAs of group g interleaved with the reads of group g+D, one tap at a time, so the */   \
        /* reads issue while an MFMA occupies the pipe instead of between two bursts                             */   \
        __builtin_amdgcn_sched_group_barrier(0x008, NMF / 3, 0);                                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, NRD / 3, 0);                                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, NMF / 3, 0);                                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, NRD / 3, 0);                                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, NMF / 3, 0);                                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, NRD / 3 + (gn_ % 3 == 0 ? NRA : 0), 0);                         \
    } while (0)
#pragma unroll 1
                for (int ub = 0; ub < NU; ub += 2) {
                    WG_STEP(0); WG_STEP(1); WG_STEP(2); WG_STEP(3); WG_STEP(4); WG_STEP(5);
                }
#undef WG_STEP
#undef WG_LOADG
            } else
#pragma unroll 1
            for (int ty = 0; ty < TH; ++ty)
#pragma unroll
                for (int xs = 0; xs < TW; xs += 16) {
                    const char* ap = sa + (ty * TW + xs + kq) * STRIDE + a_ch;
                    const uint2 alo = ds_tr16(ap), ahi = ds_tr16(ap + 4 * STRIDE);
                    const uint4 af = make_uint4(alo.x, alo.y, ahi.x, ahi.y);       // pixels 0-3 | 4-7 of this lane half
                    uint4 af_lo = af;
                    if constexpr (SPLIT) {
                        const uint2 l0 = ds_tr16(ap + LO), l1 = ds_tr16(ap + LO + 4 * STRIDE);
                        af_lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int step = MODE == WG_UP2 ? 2 : 1;
                        const char* bp = sb + (bpix(ty, xs, t) + step * kq) * STRIDE + b_ch;
                        const uint2 blo = ds_tr16(bp), bhi = ds_tr16(bp + 4 * step * STRIDE);
                        const uint4 bf = make_uint4(blo.x, blo.y, bhi.x, bhi.y);
                        if constexpr (SPLIT) {
                            const uint2 m0 = ds_tr16(bp + LO), m1 = ds_tr16(bp + LO + 4 * step * STRIDE);
                            const uint4 bf_lo = make_uint4(m0.x, m0.y, m1.x, m1.y);
                            mma_bf16(af_lo, bf, acc[t]);
                            mma_bf16(af, bf_lo, acc[t]);
                        }
                        mma_bf16(af, bf, acc[t]);
                    }
                }
        } else {
            // f32: one element per lane: A[channel lane&31][pixel lane>>5]
            const int i = lane & 31, kh = lane >> 5;
            const int step = MODE == WG_UP2 ? 2 : 1;
#pragma unroll
            for (int ty = 0; ty < TH; ++ty)
#pragma unroll 4
                for (int xs = 0; xs < TW; xs += 2) {
                    const float av = *reinterpret_cast<const float*>(sa + (ty * TW + xs + kh) * STRIDE + (32 * wr + i) * 4);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float bv = *reinterpret_cast<const float*>(sb + (bpix(ty, xs, t) + step * kh) * STRIDE + (32 * wc + i) * 4);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                    }
                }
        }
    };

    if constexpr (!WS) {
        if (t_begin < t_end) gload(t_begin, ra, rb);
        for (int tile = t_begin; tile < t_end; ++tile) {
            if (tile != t_begin) __syncthreads();
            lds_store(0, ra, rb);
            __syncthreads();
            if (tile + 1 < t_end) gload(tile + 1, ra, rb);
            multiply(0);
        }
    } else if (producer) {
        // Two register sets: the loads of tile i+2 are issued a whole period before they are stored, so their latency
        // is covered by the consumers' multiply of tile i instead of being waited for at the head of the next period.
        // The load schedule is unconditional (tiles past the end load nothing) so the compiler can count vmcnt.
        unsigned long long d0 = WGD_T(), d1, dw = 0, db = 0;
        (void)d1; (void)dw; (void)db;
        uint4 ra2[NJA], rb2[NJB];
        const int n = max(t_end - t_begin, 0);
        gload(t_begin, ra, rb);
        if (n > 0) lds_store(0, ra, rb);
        gload(t_begin + 1, ra, rb);
        gload(t_begin + 2, ra2, rb2);
        __syncthreads();                                   // stage 0 is ready
        WGD_ADD(0, WGD_T() - d0);                          // [0] producer prologue
        // Periods come in pairs with no exit between the halves (an odd tile count gets one idle period, consumers
        // included): with a mid-loop exit the structurised CFG has a first-half -> loop-header edge on which the OTHER
        // set's loads are the older ones, and the compiler then drains both sets before every store.
        for (int i = 0; i < n; i += 2) {
            d0 = WGD_T();
            if (i + 1 < n) lds_store(1, ra, rb);           // consumers are reading stage 0 (tile i)
            gload(t_begin + i + 3, ra, rb);
            d1 = WGD_T();
            __syncthreads();
            dw += d1 - d0; db += WGD_T() - d1;
            d0 = WGD_T();
            if (i + 2 < n) lds_store(0, ra2, rb2);         // consumers are reading stage 1 (tile i+1)
            gload(t_begin + i + 4, ra2, rb2);
            d1 = WGD_T();
            __syncthreads();
            dw += d1 - d0; db += WGD_T() - d1;
        }
        WGD_ADD(1, dw); WGD_ADD(4, db);                    // [1] producer wait+store+issue [4] producer at barrier
    } else {
        unsigned long long c0 = WGD_T(), c1, dm = 0, dc = 0;
        (void)c1; (void)dm; (void)dc;
#ifdef CLAMD_DIAG
        const unsigned long long k0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
        __syncthreads();                                   // stage 0 is ready
        const int n2 = max(t_end - t_begin, 0) + (max(t_end - t_begin, 0) & 1);        // periods: tiles rounded up to even
        for (int tile = t_begin; tile < t_begin + n2; ++tile) {
            c0 = WGD_T();
            if (tile < t_end) multiply((tile - t_begin) & 1);
            c1 = WGD_T();
            __syncthreads();                               // hand the stage back to the producers
            dm += c1 - c0; dc += WGD_T() - c1;
        }
        WGD_ADD(5, dm); WGD_ADD(6, dc);                    // [5] consumer multiply [6] consumer at barrier
#ifdef CLAMD_DIAG
        if (wave == 0) {                                   // [2] shader cycles, [3] 100-MHz ticks of the tile loop: the clock the chip held
            WGD_ADD(2, __builtin_amdgcn_s_memtime() - k0);
            WGD_ADD(3, __builtin_amdgcn_s_memrealtime() - rt0);
        }
#endif
        if (wave == 0) WGD_ADD(7, 1);                      // [7] workgroups
    }

    // ---- store the partial slab [split][r][c][t] (taps innermost = the parameter's own [Cout][Cin][3][3] /
    // [Cin][Cout][2][2] order).  Each wave transposes 8 rows x 32 cols x NT taps at a time through its private LDS
    // region (lane stride NT floats: odd or 4 -> at most 2-way conflicts) and stores 16-byte pieces of the
    // 32*NT-float contiguous row segments.
    if constexpr (!WS) __syncthreads();                    // WS: the last loop barrier already released both stages
    float* const wbuf = reinterpret_cast<float*>(smem) + wave * (8 * 32 * NT);
    const bool wave_in = r0 + 32 * wr < p.Rp && c0 + 32 * wc < p.Cp;      // Rp, Cp are multiples of 32: all or nothing
    float* const slab = p.partial + (((size_t)split * p.Rp + r0 + 32 * wr) * p.Cp + c0 + 32 * wc) * NT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j) __syncthreads();
        if (!producer) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int ee = 0; ee < 4; ++ee)             // acc_row(4j + ee, hh) = 8j + 4hh + ee
                    wbuf[(((lane >> 5) * 4 + ee) * 32 + (lane & 31)) * NT + t] = acc[t][4 * j + ee];
        }
        __syncthreads();
        if (!producer && wave_in) {
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int f = 4 * (k * 64 + lane);         // float index inside the 8 x (32*NT) staging block
                const int row8 = f / (32 * NT), rem = f % (32 * NT);
                *reinterpret_cast<float4*>(slab + ((size_t)(8 * j + row8) * p.Cp) * NT + rem) =
                    *reinterpret_cast<const float4*>(wbuf + f);
            }
        }
    }
}

// out[rl][cl][t] = sum_s partial[s][rp][cp][t]: a column sum over the split-K slabs whose element order already is the
// parameter's.  256 threads = LE lanes x KP split-phases; a lane owns 4 consecutive floats (16-byte loads), the phases
// stride over the slabs and are combined through LDS in a fixed order (deterministic).  Physical -> logical channel
// maps use the clamd_pack convention (two segments for concat inputs); when they are the identity the sum is stored
// with one 16-byte store.
struct ReduceParams {
    const float* partial; float* out;
    int nsplit, NT, Rp, Cp;
    int R, C;                 // logical sizes
    int r_seg0, r_seg0p;      // physical p < seg0p ? (p < seg0 ? p : pad) : seg0 + (p - seg0p)
    int c_seg0, c_seg0p;
    int identity;             // physical == logical for rows and columns
};

__device__ inline int wg_phys2log(int p, int seg0, int seg0p, int L) {
    if (p < seg0p) return p < seg0 ? p : -1;
    const int l = seg0 + (p - seg0p);
    return l < L ? l : -1;
}

template <int KP>
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const ReduceParams p) {
    SIDE_PRIO();
    constexpr int LE = 256 / KP;
    __shared__ float4 red[KP > 1 ? 256 : 1];
    const int le = threadIdx.x % LE, kp = threadIdx.x / LE;
    const long long E = (long long)p.NT * p.Rp * p.Cp;          // multiple of 4 (Cp % 32 == 0)
    for (long long base = (long long)blockIdx.x * (4 * LE); base < E; base += (long long)gridDim.x * (4 * LE)) {
        const long long e = base + 4 * le;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < E)
            for (int k = kp; k < p.nsplit; k += KP) {
                const float4 v = *reinterpret_cast<const float4*>(p.partial + (size_t)k * E + e);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        if constexpr (KP > 1) {
            red[kp * LE + le] = s;
            __syncthreads();
            if (kp == 0)
#pragma unroll 4
                for (int q = 1; q < KP; ++q) {
                    const float4 v = red[q * LE + le];
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
        }
        if (kp == 0 && e < E) {
            if (p.identity) {
                *reinterpret_cast<float4*>(p.out + e) = s;
            } else {
                const float out4[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const long long f = e + i;
                    const int t = (int)(f % p.NT), cp = (int)((f / p.NT) % p.Cp), rp = (int)(f / ((long long)p.NT * p.Cp));
                    const int rl = wg_phys2log(rp, p.r_seg0, p.r_seg0p, p.R);
                    const int cl = wg_phys2log(cp, p.c_seg0, p.c_seg0p, p.C);
                    if (rl >= 0 && cl >= 0) p.out[((long long)rl * p.C + cl) * p.NT + t] = out4[i];
                }
            }
        }
        if constexpr (KP > 1) __syncthreads();
    }
}

// Pixel-tile width.  The split (bf16x3) 3x3 variant always takes the 16-wide tile: its 6x18 halo needs 7 staging
// registers-quads instead of 9 and the kernel stays spill-free (the 32-wide variant spilled 36-100 bytes per lane).
// (clamd_tuning::wgrad_tw16 = 1: 16-wide tiles everywhere)
static inline int wgrad_tw(int W, int mode, bool split, int tw16) { return (W >= 32 && !(split && mode == WG_CONV3) && !tw16) ? 32 : 16; }

template <typename T, int MODE, bool WS>
static int launch_wg(const WgradParams& p, hipStream_t s, int grid, int tw16) {
    const dim3 blk(WS ? 512 : 256);
    constexpr bool NARROW_ONLY = __is_same(T, split_t) && MODE == WG_CONV3;      // wgrad_tw() never picks 32 there
    if constexpr (!NARROW_ONLY) {
        if (wgrad_tw(p.W, MODE, false, tw16) == 32) {
            hipLaunchKernelGGL((wgrad_kernel<T, MODE, 32, WS>), dim3(grid), blk, 0, s, p);
            return clamd_check_launch("wgrad");
        }
    }
    hipLaunchKernelGGL((wgrad_kernel<T, MODE, 16, WS>), dim3(grid), blk, 0, s, p);
    return clamd_check_launch("wgrad");
}

template <typename T>
static int launch_wg_mode(int mode, bool ws, const WgradParams& p, hipStream_t s, int grid, int tw16) {
    if (mode == WG_CONV3) return ws ? launch_wg<T, WG_CONV3, true>(p, s, grid, tw16) : launch_wg<T, WG_CONV3, false>(p, s, grid, tw16);
    if (mode == WG_PW) return launch_wg<T, WG_PW, false>(p, s, grid, tw16);
    return launch_wg<T, WG_UP2, false>(p, s, grid, tw16);
}

template <int KP>
static void launch_reduce(const ReduceParams& rp, hipStream_t s) {
    const long long n = (long long)rp.NT * rp.Rp * rp.Cp, per = 4 * (256 / KP);
    long long g = (n + per - 1) / per;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL((wgrad_reduce_kernel<KP>), dim3((unsigned)g), dim3(256), 0, s, rp);
}

}  // namespace clamd

using namespace clamd;

#ifdef CLAMD_DIAG
extern "C" int clamd_debug_wg_diag(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(clamd::g_wg_diag), 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(clamd::g_wg_diag), z, 64) != hipSuccess) return -1; }
    return 0;
}
#endif

// physical channel p maps to logical channel p for every p (no padding, segments butt-joined)
static inline int wg_identity(int L, int Lp, int seg0, int seg0p) { return L == Lp && (seg0 == seg0p || seg0p >= Lp); }

extern "C" {

size_t clamd_wgrad_workspace_bytes(int mode, int B, int H, int W, int Rp, int Cp, int dtype) {
    // upper bound used by callers to size the slab buffer: nsplit is capped at 512 blocks total (see below)
    const int NT = mode == WG_CONV3 ? 9 : (mode == WG_UP2 ? 4 : 1);
    const int rt = (Rp + 63) / 64, ct = (Cp + 63) / 64;
    int nsplit = 1024 / (rt * ct);     // upper bound over every value the tuning knob may take
    if (nsplit < 1) nsplit = 1;
    (void)B; (void)H; (void)W; (void)dtype;
    return (size_t)nsplit * NT * Rp * Cp * sizeof(float);
}

int clamd_wgrad(int mode, const void* a, int a_ldc, const void* b, int b_ldc, float* workspace, size_t ws_bytes,
                float* out, int B, int H, int W, int Rp, int Cp, int R, int C, int r_seg0, int r_seg0p, int c_seg0,
                int c_seg0p, int dtype, const clamd_tuning* tune, void* stream) {
    if (mode < 0 || mode > 2) return clamd_fail("wgrad: bad mode");
    if (int e = clamd_check_tuning(tune)) return e;
    const clamd_tuning& tn = clamd_tune(tune);
    if (Rp % 32 || Cp % 32 || a_ldc % 8 || b_ldc % 8) return clamd_fail("wgrad: channel counts/pitches must be padded");
    if (B <= 0 || H <= 0 || W <= 0) return clamd_fail("wgrad: empty problem");
    if (int e = clamd_check_split(dtype, a, a_ldc)) return e;
    if (int e = clamd_check_split(dtype, b, b_ldc)) return e;
    {   // buffer descriptors address one image with 32-bit byte offsets (OOB marker = 2^31)
        const long long sc = mode == WG_UP2 ? 4 : 1;
        if ((long long)H * W * a_ldc * 4 >= (1ll << 30) || sc * H * W * b_ldc * 4 >= (1ll << 30))
            return clamd_fail("wgrad: one image exceeds 2^30 bytes");
    }
    const int NT = mode == WG_CONV3 ? 9 : (mode == WG_UP2 ? 4 : 1);
    const int TW = wgrad_tw(W, mode, dtype == CLAMD_SPLIT, tn.wgrad_tw16);
    const int TH = (dtype == CLAMD_BF16 ? 128 : 64) / (mode == WG_UP2 ? 2 : 1) / TW;    // CLAMD_SPLIT tiles like fp32
    const int ntiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * B;
    const int rt = (Rp + 63) / 64, ct = (Cp + 63) / 64;
    // the producer/consumer kernel runs one workgroup per CU: half the slabs of the 2-per-CU kernel
    const bool ws = tn.wgrad_ws && mode == WG_CONV3;
    // split-K target: wgrad_blocks workgroups on a whole chip (512 = two per CU), scaled down with the CUs left to RCCL
    const int target = (int)((long long)tn.wgrad_blocks * clamd_usable_cus(tn) / clamd_num_cus());
    int nsplit = (ws ? target / 2 : target) / (rt * ct);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > ntiles) nsplit = ntiles;
    int per = (ntiles + nsplit - 1) / nsplit;
    nsplit = (ntiles + per - 1) / per;
    const size_t need = (size_t)nsplit * NT * Rp * Cp * sizeof(float);
    if (need > ws_bytes) return clamd_fail("wgrad: workspace too small");
    WgradParams p{a, a_ldc, b, b_ldc, workspace, B, H, W, Rp, Cp, nsplit, per, tn.wgrad_xcd};
    const int grid = rt * ct * nsplit;
    hipStream_t s = (hipStream_t)stream;
    int e;
    // LDS-DMA staging: +3..11 % except on the single-tile 64x64-channel layers (HBM-heavy, 6 % slower there); 2 = always
    if (dtype == CLAMD_BF16 && ws && (tn.wgrad_dma == 2 || (tn.wgrad_dma == 1 && rt * ct > 1))) e = launch_wgrad_dma(p, s, grid, TW);
    else if (dtype == CLAMD_BF16) e = launch_wg_mode<bf16_t>(mode, ws, p, s, grid, tn.wgrad_tw16);
    else if (dtype == CLAMD_F32) e = launch_wg_mode<float>(mode, ws, p, s, grid, tn.wgrad_tw16);
    else if (dtype == CLAMD_SPLIT) e = launch_wg_mode<split_t>(mode, ws, p, s, grid, tn.wgrad_tw16);
    else return clamd_fail("wgrad: bad dtype");
    if (e) return e;
    const int ident = wg_identity(R, Rp, r_seg0, r_seg0p) && wg_identity(C, Cp, c_seg0, c_seg0p);
    ReduceParams rp{workspace, out, nsplit, NT, Rp, Cp, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p, ident};
    const long long n = (long long)NT * Rp * Cp;
    if (nsplit <= 2) launch_reduce<1>(rp, s);
    else if (n <= 8192 && nsplit >= 64) launch_reduce<64>(rp, s);
    else if (nsplit >= 32) launch_reduce<16>(rp, s);
    else launch_reduce<4>(rp, s);
    return clamd_check_launch("wgrad_reduce");
}

}  // extern "C"
