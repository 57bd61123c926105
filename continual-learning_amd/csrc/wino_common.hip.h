// Shared declarations of the Winograd kernels (wino.hip: F(2x2,3x3); wino24.hip: F(2x4,3x3)).
#pragma once
#include "common.hip.h"
#include "clamd_internal.h"

namespace clamd {

struct WinoParams {
    const float* x; int x_ldc;
    const float* w;              // [Kp/8][16 | 24][Np][8]
    const float* bias;
    float* y; int y_ldc;
    float* stats;                // partial rows [pixel tile][2][Np] (plain stores) or null
    int B, H, W, Kp, Np, relu;
    int band;                    // output-channel slabs per band of the block order (see clamd_conv3x3_winograd)
    int nblk;                    // (pixel tile, slab) pairs; the grid is min(nblk, CUs) persistent workgroups
    // forward launches behind a folded BatchNorm (clamd_bn_fold_bias): bias is a [9][Np] table indexed by the border class of the output pixel
    int bias_classes = 0;
};

// One filter-transform job = one GEMM operand: dst[(k/8)*P + xi][n][k%8] = (G g G^T)[xi] (P = 16 or 24 planes) with
// g = w[n_l][k_l] (forward) or the tap-flipped w[k_l][n_l] (data gradient); physical -> logical channel maps as in
// clamd_pack (two segments for concat inputs, zero padding).  One thread per (n, k).
struct WinoPackJob {
    const float* w; float* dst;
    int Np, Kp, N, K;                 // physical / logical sizes of the GEMM's N (rows) and K
    int n_seg0, n_seg0p, k_seg0, k_seg0p;
    int dgrad;                        // 0: g = w[n][k], src [N][K][3][3]; 1: g = flip(w[k][n]), src [K][N][3][3]
    int block0;                       // first workgroup of this job
    const float* kscale;              // optional [Kp]: g is multiplied by kscale[physical k] before the transform (bnfold.hip)
};

__device__ inline int wn_phys2log(int p, int seg0, int seg0p, int L) {
    if (p < seg0p) return p < seg0 ? p : -1;
    const int l = seg0 + (p - seg0p);
    return l < L ? l : -1;
}

// block order of the forward kernels: output-channel slabs per band (HBM traffic model, wino.hip)
int wino_band(long long tiles, long long slabs, double x_elems, double f_elems, int forced);

// wino24.hip
long long clamd_winograd24_stat_rows(int B, int H, int W, int Cout_p, const clamd_tuning& tn);
int launch_wino24(WinoParams p, const clamd_tuning& tn, int stat_rows, hipStream_t stream);
// wino44g.hip
long long clamd_winograd44_stat_rows(int B, int H, int W, int Cout_p, const clamd_tuning& tn);
// wino24n.hip: the same launch as half-width workgroups (32 tiles x 32 channels, two per CU; clamd_tuning::wino_half)
long long clamd_winograd24_half_stat_rows(int B, int H, int W, int Cout_p, const clamd_tuning& tn);
int launch_wino24_half(WinoParams p, const clamd_tuning& tn, int stat_rows, hipStream_t stream);

}  // namespace clamd
