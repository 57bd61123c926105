// Shared declarations of the weight-gradient kernels (wgrad.hip, wgrad_dma.hip).
#pragma once
#include "common.hip.h"

namespace clamd {

enum { WG_CONV3 = 0, WG_PW = 1, WG_UP2 = 2 };

struct WgradParams {
    const void* a; int a_ldc;
    const void* b; int b_ldc;
    float* partial;        // [nsplit][NT][Rp][Cp]
    int B, H, W;           // pixel grid of A
    int Rp, Cp;            // physical channels of A / B
    int nsplit, tiles_per_split;
    int xcd;               // XCD-aware block order (tuning knob "wgrad_xcd")
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

// Staging slot of thread `i` in a tile image of `total` 16-byte pieces: the ragged last pass wraps around and re-stages the
// first pieces (same data to the same LDS address), so every load has an unconditional use -- a store guarded by
// "slot < total" lets the compiler sink the load next to it, behind a full s_waitcnt vmcnt(0).
__device__ inline int wrap_idx(int i, int total) { return i >= total ? i - total : i; }

__device__ inline uint2 ds_tr16(const char* lds_addr) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_addr));
    return __builtin_bit_cast(uint2, v);
}


// wgrad_dma.hip: bf16 3x3 producer/consumer kernel whose producers stage with LDS-DMA (buffer_load ... lds)
int launch_wgrad_dma(const WgradParams& p, hipStream_t s, int grid, int tw);

}  // namespace clamd
