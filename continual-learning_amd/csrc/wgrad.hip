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

namespace clamd {

enum { WG_CONV3 = 0, WG_PW = 1, WG_UP2 = 2 };

struct WgradParams {
    const void* a; int a_ldc;
    const void* b; int b_ldc;
    float* partial;        // [nsplit][NT][Rp][Cp]
    int B, H, W;           // pixel grid of A
    int Rp, Cp;            // physical channels of A / B
    int nsplit, tiles_per_split;
};

template <typename T, int MODE, int TW> struct WGeo {
    static constexpr bool SPLIT = __is_same(T, split_t);   // fp32 storage, bf16 hi/lo LDS images, 3 MFMAs per product
    static constexpr int TH = (sizeof(T) == 2 ? 128 : 64) / (MODE == 2 ? 2 : 1) / TW;   // pixel tile rows (UP2: B tile is 4x)
    static constexpr int NT = MODE == WG_CONV3 ? 9 : (MODE == WG_UP2 ? 4 : 1);
    static constexpr int BW = MODE == WG_CONV3 ? TW + 2 : (MODE == WG_UP2 ? 2 * TW : TW);
    static constexpr int BH = MODE == WG_CONV3 ? TH + 2 : (MODE == WG_UP2 ? 2 * TH : TH);
    static constexpr int STRIDE = (sizeof(T) == 2 || SPLIT) ? 192 : 256;        // bytes per pixel row (64 channels + pad)
    static constexpr int APIX = TH * TW, BPIX = BH * BW;
    static constexpr int GPP = 64 * sizeof(T) / 16;                             // 16-B groups per pixel (8 or 16)
    static constexpr int NJA = (APIX * GPP + 255) / 256, NJB = (BPIX * GPP + 255) / 256;
    static constexpr int BYTES = (APIX + BPIX) * STRIDE * (SPLIT ? 2 : 1);
};

__device__ inline uint2 ds_tr16(const char* lds_addr) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_addr));
    return __builtin_bit_cast(uint2, v);
}

template <typename T, int MODE, int TW>
__global__ void __launch_bounds__(256, 2) wgrad_kernel(const WgradParams p) {
    using G = WGeo<T, MODE, TW>;
    constexpr int TH = G::TH, NT = G::NT, BW = G::BW, STRIDE = G::STRIDE, GPP = G::GPP, NJA = G::NJA, NJB = G::NJB;
    constexpr int VEC = DT<T>::VEC;
    constexpr bool SPLIT = G::SPLIT;
    __shared__ __attribute__((aligned(16))) char smem[G::BYTES];
    char* const sa = smem;
    char* const sb = smem + G::APIX * STRIDE;
    constexpr int LO = (G::APIX + G::BPIX) * STRIDE;      // split: byte offset of the lo images (same layout as hi)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    const int rt = (p.Rp + 63) >> 6, ct = (p.Cp + 63) >> 6;
    int bid = blockIdx.x;
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

    uint4 ra[NJA], rb[NJB];

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
    constexpr bool RECOMP = SPLIT && MODE == WG_CONV3 && TW == 32;
    unsigned a_vo[RECOMP ? 1 : NJA], b_vo[RECOMP ? 1 : NJB];
    unsigned b_hyx[RECOMP ? 1 : NJB];                               // (hy << 16) | hx: BW is not a power of two
#pragma unroll
    for (int j = 0; j < NJA; ++j) {
        const int i = tid + 256 * j, pix = i / GPP, g = i % GPP;
        if constexpr (!RECOMP)
            a_vo[j] = (pix < G::APIX && r0 + g * VEC < p.Rp) ? (unsigned)((((pix / TW) * p.W + pix % TW) * p.a_ldc + r0 + g * VEC) * ESZ) : BUF_OOB;
    }
#pragma unroll
    for (int j = 0; j < NJB; ++j) {
        const int i = tid + 256 * j, pix = i / GPP, g = i % GPP;
        const int hy = pix / BW, hx = pix % BW;
        if constexpr (!RECOMP) {
            b_hyx[j] = ((unsigned)hy << 16) | (unsigned)hx;
            b_vo[j] = (pix < G::BPIX && c0 + g * VEC < p.Cp)
                          ? (unsigned)(((hy * BSCALE * p.W + hx) * p.b_ldc + c0 + g * VEC) * ESZ) : BUF_OOB;
        }
    }

    const T* __restrict__ ag = (const T*)p.a;
    const T* __restrict__ bg = (const T*)p.b;
    auto gload = [&](int tile) {
        // tile -> (image, origin): wave-uniform scalar arithmetic
        const int x0 = (tile % tiles_x) * TW, y0 = ((tile / tiles_x) % tiles_y) * TH, b = tile / (tiles_x * tiles_y);
        if constexpr (RECOMP) {
            // register-starved split variant: no persistent per-thread descriptors; predicated 64-bit loads whose
            // addresses are rebuilt per tile (the buffer-load form below kept 25 more VGPRs live and spilled)
#pragma unroll
            for (int j = 0; j < NJA; ++j) {
                const int i = tid + 256 * j, pix = i / GPP, g = i % GPP;
                const int yy = y0 + pix / TW, xx = x0 + pix % TW;
                const bool ok = pix < G::APIX && yy < p.H && xx < p.W && r0 + g * VEC < p.Rp;
                ra[j] = ldg16(ag + ((long long)(b * p.H + yy) * p.W + xx) * p.a_ldc + r0 + g * VEC, ok);
            }
#pragma unroll
            for (int j = 0; j < NJB; ++j) {
                const int i = tid + 256 * j, pix = i / GPP, g = i % GPP;
                const int hy = pix / BW, hx = pix % BW;
                const int yy = y0 + hy - 1, xx = x0 + hx - 1;
                const bool ok = pix < G::BPIX && c0 + g * VEC < p.Cp && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
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
            const int i = tid + 256 * j, pix = i / GPP, g = i % GPP;   // TW, GPP are powers of two: shifts
            const bool ok = y0 + pix / TW < p.H && x0 + pix % TW < p.W;
            (void)g;
            ra[j] = buf_ld16(ars, ok ? a_vo[j] : BUF_OOB, a_so);
        }
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
            const int yy = BSCALE * y0 + (int)(b_hyx[j] >> 16) - HALO, xx = BSCALE * x0 + (int)(b_hyx[j] & 0xffffu) - HALO;
            const bool ok = yy >= 0 && yy < BSCALE * p.H && xx >= 0 && xx < BSCALE * p.W;
            rb[j] = buf_ld16(brs, ok ? b_vo[j] : BUF_OOB, b_so);
        }
    };
    auto lds_store = [&]() {
#pragma unroll
        for (int j = 0; j < NJA; ++j) {
            const int i = tid + 256 * j, pix = i / GPP, g = i % GPP;
            const bool in = 256 * (j + 1) <= G::APIX * GPP || pix < G::APIX;      // only the last j can overrun the tile
            if constexpr (!SPLIT) {
                if (in) *reinterpret_cast<uint4*>(sa + pix * STRIDE + g * 16) = ra[j];
            } else if (in) {
                uint2 hi, lo;
                split4(ra[j], hi, lo);
                *reinterpret_cast<uint2*>(sa + pix * STRIDE + g * 8) = hi;
                *reinterpret_cast<uint2*>(sa + LO + pix * STRIDE + g * 8) = lo;
            }
        }
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
            const int i = tid + 256 * j, pix = i / GPP, g = i % GPP;
            const bool in = 256 * (j + 1) <= G::BPIX * GPP || pix < G::BPIX;
            if constexpr (!SPLIT) {
                if (in) *reinterpret_cast<uint4*>(sb + pix * STRIDE + g * 16) = rb[j];
            } else if (in) {
                uint2 hi, lo;
                split4(rb[j], hi, lo);
                *reinterpret_cast<uint2*>(sb + pix * STRIDE + g * 8) = hi;
                *reinterpret_cast<uint2*>(sb + LO + pix * STRIDE + g * 8) = lo;
            }
        }
    };

    // B-tile pixel index of A-tile pixel (ty, tx) for tap t
    auto bpix = [&](int ty, int tx, int t) -> int {
        if constexpr (MODE == WG_CONV3) return (ty + t / 3) * BW + tx + t % 3;
        else if constexpr (MODE == WG_PW) return ty * BW + tx;
        else return (2 * ty + (t >> 1)) * BW + 2 * tx + (t & 1);
    };

    if (t_begin < t_end) gload(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        if (tile != t_begin) __syncthreads();
        lds_store();
        __syncthreads();
        if (tile + 1 < t_end) gload(tile + 1);

        if constexpr (sizeof(T) == 2 || SPLIT) {
            // bf16 / split: k-group = 16 consecutive pixels of one tile row.  Transposed reads: lane -> (gq = lane>>4,
            // q = (lane>>2)&3, pp = lane&3); it supplies the address of pixel (8*(gq>>1) + q) [+4 for the second
            // read], channels 16*(gq&1) + 4*pp .. +3, and receives 4 pixels of channel 16*(gq&1) + (lane&15).
            const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
            const int kq = 8 * (gq >> 1) + q;
            const int a_ch = (32 * wr + 16 * (gq & 1) + 4 * pp) * 2;
            const int b_ch = (32 * wc + 16 * (gq & 1) + 4 * pp) * 2;
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
    }

    // ---- store the partial slab: row = r0 + 32*wr + acc_row, col = c0 + 32*wc + (lane & 31) ------------
    const int col = c0 + 32 * wc + (lane & 31), hh = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = r0 + 32 * wr + acc_row(e, hh);
            if (row < p.Rp && col < p.Cp)
                p.partial[(((size_t)split * NT + t) * p.Rp + row) * p.Cp + col] = acc[t][e];
        }
}

// out[rl][cl][t] = sum_s partial[s][t][rp][cp]: 64 consecutive (t,rp,cp) elements x 4 split-phases per block, so the
// slab reads are coalesced 256-B rows and the split loop is spread over the 4 waves; fixed summation order
// (deterministic).  Physical -> logical channel maps use the clamd_pack convention (two segments for concat inputs).
struct ReduceParams {
    const float* partial; float* out;
    int nsplit, NT, Rp, Cp;
    int R, C;                 // logical sizes
    int r_seg0, r_seg0p;      // physical p < seg0p ? (p < seg0 ? p : pad) : seg0 + (p - seg0p)
    int c_seg0, c_seg0p;
};

__device__ inline int wg_phys2log(int p, int seg0, int seg0p, int L) {
    if (p < seg0p) return p < seg0 ? p : -1;
    const int l = seg0 + (p - seg0p);
    return l < L ? l : -1;
}

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const ReduceParams p) {
    // 64 lanes x 4 consecutive (t,rp,cp) elements (16-byte loads, 1 KB per wave row) x 4 split-phases per block
    __shared__ float4 red[4][64];
    const int le = threadIdx.x & 63, kp = threadIdx.x >> 6;
    const long long E = (long long)p.NT * p.Rp * p.Cp;          // multiple of 4 (Cp % 32 == 0)
    for (long long base = (long long)blockIdx.x * 256; base < E; base += (long long)gridDim.x * 256) {
        const long long e = base + 4 * le;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < E)
            for (int k = kp; k < p.nsplit; k += 4) {
                const float4 v = *reinterpret_cast<const float4*>(p.partial + (size_t)k * E + e);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        red[kp][le] = s;
        __syncthreads();
        if (kp == 0 && e < E) {
            const float4 a = red[0][le], b = red[1][le], c = red[2][le], d = red[3][le];
            const float out4[4] = {(a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z),
                                   (a.w + b.w) + (c.w + d.w)};
            const int cp0 = (int)(e % p.Cp), rp = (int)((e / p.Cp) % p.Rp), t = (int)(e / ((long long)p.Cp * p.Rp));
            const int rl = wg_phys2log(rp, p.r_seg0, p.r_seg0p, p.R);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cl = wg_phys2log(cp0 + i, p.c_seg0, p.c_seg0p, p.C);
                if (rl >= 0 && cl >= 0) p.out[((long long)rl * p.C + cl) * p.NT + t] = out4[i];
            }
        }
        __syncthreads();
    }
}

// Pixel-tile width.  The split (bf16x3) 3x3 variant always takes the 16-wide tile: its 6x18 halo needs 7 staging
// registers-quads instead of 9 and the kernel stays spill-free (the 32-wide variant spilled 36-100 bytes per lane).
int g_wgrad_tw16 = 0;               // tuning knob: 1 = 16-wide tiles everywhere
static inline int wgrad_tw(int W, int mode, bool split) { return (W >= 32 && !(split && mode == WG_CONV3) && !g_wgrad_tw16) ? 32 : 16; }

int g_wgrad_target_blocks = 512;   // tuning knob (clamd_set_tuning "wgrad_blocks"): split-K until about this many workgroups

template <typename T, int MODE>
static int launch_wg(const WgradParams& p, hipStream_t s, int grid) {
    if (wgrad_tw(p.W, MODE, __is_same(T, split_t)) == 32) hipLaunchKernelGGL((wgrad_kernel<T, MODE, 32>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wgrad_kernel<T, MODE, 16>), dim3(grid), dim3(256), 0, s, p);
    return clamd_check_launch("wgrad");
}

}  // namespace clamd

using namespace clamd;

extern "C" {

size_t clamd_wgrad_workspace_bytes(int mode, int B, int H, int W, int Rp, int Cp, int dtype) {
    // upper bound used by callers to size the slab buffer: nsplit is capped at 512 blocks total (see below)
    const int NT = mode == WG_CONV3 ? 9 : (mode == WG_UP2 ? 4 : 1);
    const int rt = (Rp + 63) / 64, ct = (Cp + 63) / 64;
    int nsplit = 512 / (rt * ct);      // upper bound over every value the tuning knob may take
    if (nsplit < 1) nsplit = 1;
    (void)B; (void)H; (void)W; (void)dtype;
    return (size_t)nsplit * NT * Rp * Cp * sizeof(float);
}

int clamd_wgrad(int mode, const void* a, int a_ldc, const void* b, int b_ldc, float* workspace, size_t ws_bytes,
                float* out, int B, int H, int W, int Rp, int Cp, int R, int C, int r_seg0, int r_seg0p, int c_seg0,
                int c_seg0p, int dtype, void* stream) {
    if (mode < 0 || mode > 2) return clamd_fail("wgrad: bad mode");
    if (Rp % 32 || Cp % 32 || a_ldc % 8 || b_ldc % 8) return clamd_fail("wgrad: channel counts/pitches must be padded");
    if (B <= 0 || H <= 0 || W <= 0) return clamd_fail("wgrad: empty problem");
    {   // buffer descriptors address one image with 32-bit byte offsets (OOB marker = 2^31)
        const long long sc = mode == WG_UP2 ? 4 : 1;
        if ((long long)H * W * a_ldc * 4 >= (1ll << 30) || sc * H * W * b_ldc * 4 >= (1ll << 30))
            return clamd_fail("wgrad: one image exceeds 2^30 bytes");
    }
    const int NT = mode == WG_CONV3 ? 9 : (mode == WG_UP2 ? 4 : 1);
    const int TW = wgrad_tw(W, mode, dtype == CLAMD_SPLIT);
    const int TH = (dtype == CLAMD_BF16 ? 128 : 64) / (mode == WG_UP2 ? 2 : 1) / TW;    // CLAMD_SPLIT tiles like fp32
    const int ntiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * B;
    const int rt = (Rp + 63) / 64, ct = (Cp + 63) / 64;
    int nsplit = g_wgrad_target_blocks / (rt * ct);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > ntiles) nsplit = ntiles;
    int per = (ntiles + nsplit - 1) / nsplit;
    nsplit = (ntiles + per - 1) / per;
    const size_t need = (size_t)nsplit * NT * Rp * Cp * sizeof(float);
    if (need > ws_bytes) return clamd_fail("wgrad: workspace too small");
    WgradParams p{a, a_ldc, b, b_ldc, workspace, B, H, W, Rp, Cp, nsplit, per};
    const int grid = rt * ct * nsplit;
    hipStream_t s = (hipStream_t)stream;
    int e;
    if (dtype == CLAMD_BF16) {
        e = mode == WG_CONV3 ? launch_wg<bf16_t, WG_CONV3>(p, s, grid)
          : mode == WG_PW    ? launch_wg<bf16_t, WG_PW>(p, s, grid) : launch_wg<bf16_t, WG_UP2>(p, s, grid);
    } else if (dtype == CLAMD_F32) {
        e = mode == WG_CONV3 ? launch_wg<float, WG_CONV3>(p, s, grid)
          : mode == WG_PW    ? launch_wg<float, WG_PW>(p, s, grid) : launch_wg<float, WG_UP2>(p, s, grid);
    } else if (dtype == CLAMD_SPLIT) {
        e = mode == WG_CONV3 ? launch_wg<split_t, WG_CONV3>(p, s, grid)
          : mode == WG_PW    ? launch_wg<split_t, WG_PW>(p, s, grid) : launch_wg<split_t, WG_UP2>(p, s, grid);
    } else return clamd_fail("wgrad: bad dtype");
    if (e) return e;
    ReduceParams rp{workspace, out, nsplit, NT, Rp, Cp, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p};
    const long long n = (long long)NT * Rp * Cp;
    int g = (int)((n + 255) / 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(g), dim3(256), 0, s, rp);
    return clamd_check_launch("wgrad_reduce");
}

}  // extern "C"
