// Shared declarations of the implicit-GEMM kernels (igemm.hip: baseline structure; igemm_ws.hip: producer/consumer
// wave-specialised structure for the 3x3 convolutions).
#pragma once
#include "common.hip.h"
#include "clamd_internal.h"

namespace clamd {

enum { MODE_CONV3 = 0, MODE_PW = 1, MODE_UP2 = 2 };
enum { EPI_NHWC = 0, EPI_UP2 = 1, EPI_NCHW = 2 };

struct IgemmParams {
    const void* x; int x_ldc;
    const void* w;          // packed [taps][Np][Kp], K innermost
    const float* bias;      // indexed by n (NHWC/NCHW) or by n % aux (UP2 epilogue); may be null
    void* y; int y_ldc;
    float* stats;           // partial rows [row][2][Np] (one row per pixel tile or per workgroup, plain stores) or null
    int B, H, W;            // pixel grid of the GEMM rows
    int Kp, Np;
    int relu;
    int aux;                // EPI_UP2: convT Cout_p ; EPI_NCHW: logical classes ; MODE_UP2: channels per (dy,dx)
    int m_fastest;
    // EPI_NHWC only: when this launch produces the gradient g w.r.t. a BatchNorm output, accumulate the five
    // per-channel sums of the fused ReLU/BN backward (see bn_bwd_reduce_kernel) right here in the epilogue:
    // bn_y = that unit's saved post-ReLU activation [B,H,W,Np] (dense pitch Np), bn_sums = partial rows [row][5][Np].
    const void* bn_y;
    float* bn_sums;
    // EPI_NCHW only: arg-max over the logical classes per pixel (first maximum wins, as torch.max), int64 [B,H,W]; y may
    // then be null (eval forward, trainer.py:279: the logits are never materialised)
    long long* pred;
    // MODE_CONV3 forward launches behind a folded BatchNorm (bnfold.hip): bias is a [9][Np] table indexed by the border class of the pixel
    int bias_classes;
};

#ifndef IGEMM_PW_NT
#define IGEMM_PW_NT 2
#endif
template <int MODE, int TW> struct Geo {
    static constexpr int TH = 256 / TW;
    static constexpr int NT = MODE == MODE_CONV3 ? 9 : IGEMM_PW_NT; // filter slabs per staged K-step
    static constexpr int NIN = MODE == MODE_CONV3 ? 1 : NT;        // input slabs per staged K-step
    static constexpr int HW_ = MODE == MODE_CONV3 ? TW + 2 : TW;
    static constexpr int HH_ = MODE == MODE_CONV3 ? TH + 2 : TH;
    static constexpr int NPIX = HW_ * HH_;
    static constexpr int NPIXP = NPIX + ((10 - NPIX % 8) % 8);     // == 2 (mod 8)
    static constexpr int NJ = (NPIX * 4 + 255) / 256;              // 16-B input loads per thread per slab
    static constexpr int IN_SLOTS = NIN * 4 * NPIXP;
    static constexpr int WG = 66;                                  // padded channel rows per group, == 2 (mod 8)
    static constexpr int WT_SLOTS = NT * 4 * WG;
    static constexpr int EPI_SLOTS = 4 * 32 * 68 / 4;              // fp32 transposition buffer, 4 waves x [32][68]
    static constexpr int SLOTS = IN_SLOTS + WT_SLOTS > EPI_SLOTS ? IN_SLOTS + WT_SLOTS : EPI_SLOTS;
};


// Geometry of one LDS stage for a (128*MT)-pixel x 64-channel tile (MT = 32-pixel MFMA row tiles per consumer wave).
template <int TW, int MT> struct WsGeo {
    static constexpr int TH = 128 * MT / TW;
    static constexpr int NT = 9;
    static constexpr int HW_ = TW + 2, HH_ = TH + 2;
    static constexpr int NPIX = HW_ * HH_;
    static constexpr int NPIXP = NPIX + ((10 - NPIX % 8) % 8);     // == 2 (mod 8)
    static constexpr int NJ = (NPIX * 4 + 255) / 256;
    static constexpr int IN_SLOTS = 4 * NPIXP;
    static constexpr int WG = 66;
    static constexpr int WT_SLOTS = NT * 4 * WG;
};

// igemm_ws.hip: producer/consumer variant of the CONV3/NHWC kernel (same results)
int launch_igemm_ws(const IgemmParams& p, int dtype, hipStream_t s, int mt);   // mt: 32-pixel row tiles per consumer wave (1, 2 or 4)
int ws_rows(const IgemmParams& p, int mt);                                     // pixel tiles = partial statistics rows of that launch

// igemm_pws.hip: persistent producer/consumer variant for short-K layers (same activations; statistics rows are per
// workgroup instead of per tile); returns -1 without launching when the shape is not supported
int launch_igemm_pws(const IgemmParams& p, int dtype, hipStream_t s, const ::clamd_tuning& tn);
int pws_rows(const IgemmParams& p, int dtype, const ::clamd_tuning& tn);

}  // namespace clamd
