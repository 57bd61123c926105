// BatchNorm folded algebraically into the 3x3 convolution that consumes it (models/unet.py:13-18: Conv-ReLU-BN-Conv inside one block).
//
//   x = scale * r + shift          (r = the producer's saved post-ReLU activation, scale / shift from clamd_bn_finalize)
//   conv3x3(x, W)[co, y, x] = conv3x3(r, W * scale[ci])[co, y, x] + sum over the taps that do NOT read padding at (y, x) of T[co][tap],
//   T[co][tap] = sum_ci W[co][ci][tap] * shift[ci]
//
// nn.Conv2d pads x with zeros, i.e. AFTER the affine, so the shift term depends on which taps fall outside the image: nine border classes
// (border_class(), wino_common.hip.h).  The consumer's filters are packed with the per-input-channel scale (PackJob / WinoPackJob
// `kscale`), its epilogue takes the bias from a [9][Cout_p] table (relu flag CLAMD_BIAS_BORDER_CLASSES), and the normalised tensor x is
// never written: the clamd_bn_apply pass of the producer disappears from the forward pass (268 + 268 MB at 64 channels, 256 x 256, bs16).
//
// Backward: the data gradient is w.r.t. x and uses the unscaled filters -- unchanged.  The weight gradient wants x as its operand; on r it
// gives dWr, and   dW[co][ci][tap] = scale[ci] * dWr[co][ci][tap] + shift[ci] * S[tap][co],   S[tap][co] = sum of gz[., y, x, co] over the
// pixels whose tap reads inside the image = total - excluded row - excluded column + excluded corner.  The total is the conv-bias gradient
// clamd_bn_bwd_finalize has already written; rows / columns / corners are read from the border of gz here (fixed order: deterministic).
#include "common.hip.h"
#include "clamd_internal.h"

namespace clamd {

__global__ void __launch_bounds__(256) bn_fold_bias_kernel(const FoldBias f) {
    SIDE_PRIO();
    __shared__ float T[9];
    fold_bias_block(f, blockIdx.x, T);
}

// per image, edge (0 top row, 1 bottom row, 2 left column, 3 right column) and chunk of the edge: per-channel sum of gz.  256 threads =
// PL pixel lanes x CG channels (CG = min(Cp, 256), a power of two): a lane walks every PL-th pixel of its chunk, the lanes of a channel are
// added in lane order through LDS -- a fixed order.  B x 4 x FOLD_NCH workgroups: the pass is latency-bound, not bandwidth-bound.
constexpr int FOLD_NCH = 8;
template <typename T>
__global__ void __launch_bounds__(256) bn_fold_border_kernel(const T* __restrict__ gz, int ldc, float* __restrict__ part,
                                                            int H, int W, int Cp) {
    SIDE_PRIO();
    __shared__ float red[256];
    const int b = blockIdx.x >> 2, edge = blockIdx.x & 3, ch = blockIdx.y;
    const int n = edge < 2 ? W : H, per = (n + FOLD_NCH - 1) / FOLD_NCH;
    const int i0 = ch * per, i1 = i0 + per < n ? i0 + per : n;
    const size_t img = (size_t)b * H * W;
    const int CG = Cp < 256 ? Cp : 256, PL = 256 / CG;
    const int cl = threadIdx.x % CG, pl = threadIdx.x / CG;
    for (int c0 = 0; c0 < Cp; c0 += CG) {
        const int c = c0 + cl;
        float s = 0.f;
        for (int i = i0 + pl; i < i1; i += PL) {
            const size_t pix = edge == 0 ? (size_t)i : edge == 1 ? (size_t)(H - 1) * W + i : edge == 2 ? (size_t)i * W : (size_t)i * W + (W - 1);
            s += ld1<T>(gz + (img + pix) * ldc + c);
        }
        red[threadIdx.x] = s;
        __syncthreads();
        if (pl == 0) {
            for (int k = 1; k < PL; ++k) s += red[k * CG + cl];
            part[(((size_t)b * 4 + edge) * FOLD_NCH + ch) * Cp + c] = s;
        }
        __syncthreads();
    }
}

// one workgroup per logical output channel: S[9] from the edge partials, the four corner pixels and the total; then the whole [Cin][9] slab.
// Wave w adds edge w (images x chunks, lanes stride 64, then the xor butterfly), then corner w (TL TR BL BR): fixed orders.
template <typename T>
__global__ void __launch_bounds__(256) bn_fold_wgrad_kernel(const T* __restrict__ gz, int ldc, const float* __restrict__ part,
                                                           const float* __restrict__ total, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, float* __restrict__ dw,
                                                           int B, int H, int W, int Cp, int Cin) {
    __shared__ float q[8], S[9];
    const int co = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float e = 0.f, k = 0.f;
    for (int t = lane; t < B * FOLD_NCH; t += 64) {
        const int b = t / FOLD_NCH, ch = t - b * FOLD_NCH;
        e += part[(((size_t)b * 4 + wv) * FOLD_NCH + ch) * Cp + co];
    }
    for (int b = lane; b < B; b += 64) {
        const size_t pix = (size_t)b * H * W + (size_t)((wv >> 1) ? H - 1 : 0) * W + ((wv & 1) ? W - 1 : 0);
        k += ld1<T>(gz + pix * ldc + co);
    }
    e = wave_sum(e);
    k = wave_sum(k);
    if (lane == 0) { q[wv] = e; q[4 + wv] = k; }
    __syncthreads();
    if (threadIdx.x < 9) {
        const int ky = threadIdx.x / 3, kx = threadIdx.x % 3;
        float s = total[co];
        if (ky != 1) s -= q[ky == 0 ? 0 : 1];
        if (kx != 1) s -= q[kx == 0 ? 2 : 3];
        if (ky != 1 && kx != 1) s += q[4 + (ky == 2 ? 2 : 0) + (kx == 2 ? 1 : 0)];
        S[threadIdx.x] = s;
    }
    __syncthreads();
    float* d = dw + (size_t)co * Cin * 9;
    for (int i = threadIdx.x; i < Cin * 9; i += 256) {
        const int ci = i / 9, tap = i - 9 * ci;
        d[i] = fmaf(scale[ci], d[i], shift[ci] * S[tap]);
    }
}

// pointwise consumer (the 1x1 head): dW[co][ci] = scale[ci] * dW[co][ci] + shift[ci] * (sum of the gradient over all pixels)[co]
__global__ void __launch_bounds__(256) bn_fold_wgrad_pw_kernel(const float* __restrict__ total, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, float* __restrict__ dw, int Cout, int Cin) {
    SIDE_PRIO();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Cout * Cin) return;
    const int co = i / Cin, ci = i - co * Cin;
    dw[i] = fmaf(scale[ci], dw[i], shift[ci] * total[co]);
}

}  // namespace clamd

using namespace clamd;

extern "C" {

int clamd_bn_fold_bias(const float* w, const float* shift, const float* bias, float* table, int Cout, int Cin, int Cout_p, void* stream) {
    if (!w || !shift || !table || Cout <= 0 || Cin <= 0 || Cout_p < Cout) return clamd_fail("bn_fold_bias: bad arguments");
    hipLaunchKernelGGL(bn_fold_bias_kernel, dim3(Cout_p), dim3(256), 0, (hipStream_t)stream, FoldBias{w, shift, bias, table, Cout, Cin, Cout_p, 9});
    return clamd_check_launch("bn_fold_bias");
}

int clamd_bn_fold_pack(int form, const void* jobs_dev, int njobs, int total_blocks, int dtype, const float* w, int taps, const float* shift,
                       const float* bias, float* table, int Cout, int Cin, int Cout_p, void* stream) {
    if (!w || !shift || !table || Cout <= 0 || Cin <= 0 || Cout_p < Cout) return clamd_fail("bn_fold_pack: bad arguments");
    if (taps != 9 && !(taps == 1 && form == 0)) return clamd_fail("bn_fold_pack: taps must be 9 (3x3) or, with form 0, 1 (pointwise)");
    if (njobs <= 0 || total_blocks <= 0) return clamd_fail("bn_fold_pack: empty job table");
    const FoldBias f{w, shift, bias, table, Cout, Cin, Cout_p, taps};
    if (form == 0) return clamd_launch_pack(jobs_dev, njobs, total_blocks, dtype, &f, (hipStream_t)stream);
    if (form == 16) return clamd_launch_wino_pack(jobs_dev, njobs, total_blocks, &f, (hipStream_t)stream);
    if (form == 24) return clamd_launch_wino24_pack(jobs_dev, njobs, total_blocks, &f, (hipStream_t)stream);
    return clamd_fail("bn_fold_pack: form must be 0 (clamd_pack), 16 (clamd_wino_pack) or 24 (clamd_wino24_pack)");
}

int clamd_bn_fold_wgrad_pointwise(const float* sum_g, const float* scale, const float* shift, float* dw, int Cout, int Cin, void* stream) {
    if (!sum_g || !scale || !shift || !dw || Cout <= 0 || Cin <= 0) return clamd_fail("bn_fold_wgrad_pointwise: bad arguments");
    hipLaunchKernelGGL(bn_fold_wgrad_pw_kernel, dim3((Cout * Cin + 255) / 256), dim3(256), 0, (hipStream_t)stream, sum_g, scale, shift, dw, Cout, Cin);
    return clamd_check_launch("bn_fold_wgrad_pointwise");
}

size_t clamd_bn_fold_wgrad_workspace_bytes(int B, int Cout_p) { return (size_t)(B > 0 ? B : 0) * 4 * FOLD_NCH * (size_t)(Cout_p > 0 ? Cout_p : 0) * sizeof(float); }

int clamd_bn_fold_wgrad(const void* gz, int gz_ldc, const float* sum_gz, const float* scale, const float* shift, float* dw,
                        void* workspace, size_t ws_bytes, int B, int H, int W, int Cout_p, int Cout, int Cin, int dtype, void* stream) {
    if (!gz || !sum_gz || !scale || !shift || !dw || !workspace) return clamd_fail("bn_fold_wgrad: null argument");
    if (B <= 0 || H < 2 || W < 2 || Cout <= 0 || Cin <= 0 || Cout_p < Cout || gz_ldc < Cout_p) return clamd_fail("bn_fold_wgrad: bad sizes (H, W >= 2)");
    if (Cout_p < 256 ? 256 % Cout_p : Cout_p % 256) return clamd_fail("bn_fold_wgrad: Cout_p must divide 256 or be a multiple of it");
    if (ws_bytes < clamd_bn_fold_wgrad_workspace_bytes(B, Cout_p)) return clamd_fail("bn_fold_wgrad: workspace too small");
    if (int e = clamd_check_split(dtype, gz, gz_ldc)) return e;
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)workspace;
#define FOLD_LAUNCH(T_)                                                                                                              \
    do {                                                                                                                             \
        hipLaunchKernelGGL((bn_fold_border_kernel<T_>), dim3(4 * B, FOLD_NCH), dim3(256), 0, s, (const T_*)gz, gz_ldc, part, H, W, Cout_p);      \
        hipLaunchKernelGGL((bn_fold_wgrad_kernel<T_>), dim3(Cout), dim3(256), 0, s, (const T_*)gz, gz_ldc, part, sum_gz, scale, shift, \
                           dw, B, H, W, Cout_p, Cin);                                                                                \
    } while (0)
    if (dtype == CLAMD_F32) FOLD_LAUNCH(float);
    else if (dtype == CLAMD_BF16) FOLD_LAUNCH(bf16_t);
    else if (dtype == CLAMD_SPLIT) FOLD_LAUNCH(split_t);
    else return clamd_fail("bn_fold_wgrad: bad dtype");
#undef FOLD_LAUNCH
    return clamd_check_launch("bn_fold_wgrad");
}

}  // extern "C"
