// Winograd F(4x4,3x3) with PRE-TRANSFORMED operands for the wide 3x3 convolutions of the exact-fp32 path at 32x32 and 64x64
// (round 5; models/unet.py:28-33,50-55: enc3, enc4, dec2, dec3; loss.backward(), trainer.py:175).
//
// wino24g.hip runs these layers as F(2x4,3x3) -- 24 multiply-adds per 8 outputs = 3 per output -- with transform-free K loops
// that keep the fp32 MFMA pipe 0.89-0.93 busy: there is nothing left to gain there except FEWER multiply-adds.  The 2-D
// F(4x4,3x3) needs 36 per 16 outputs = 2.25 per output: 25 % fewer MFMAs, a transformed input of 2.25x instead of 3x the
// activation and the same filters' worth of arithmetic in the transforms, which are HBM-bound passes here, not loop work.
// fp32 error against an fp64 convolution: 1.9-3.4e-6 relative (F(2x4): 0.8-1.3e-6, direct fp32 sum: 0.6-1.2e-6; numpy model with
// these matrices) -- inside the 2e-5 bound every fp32 kernel of the library is held to.
//
//   * transforms: B6^T = [[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],
//     G6 = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],
//     A6^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]] in BOTH directions (the column forms of wino24.hip);
//   * wino44_xform_kernel: x [B,H,W,ldc] -> V [tile block][Kp/8][36 = 6i + j][2 = lane half][32 tiles][4 channels]; a tile block is
//     32 tiles of 4 x 4 outputs = 16 x 32 pixels (8 tiles across) or 32 x 16 (images narrower than 32);
//   * wino44_pack_kernel: U = G6 g G6^T as [Cin_p/8][36][Cout_p][8] (forward) / the tap-flipped transposed filters (data gradient);
//   * wino44g_kernel: 32 tiles x 64 output channels per workgroup as in wino24g_kernel, but SIX Winograd rows: 12 waves,
//     wave = (row i, 32-channel half nt), three waves per SIMD, 96 accumulators each.  Per 8-channel chunk a wave issues 24 MFMAs
//     and 12 buffer_load_dwordx4 whose destinations are the MFMA operand registers (one chunk ahead: the other two waves of the
//     SIMD cover the latency); no LDS, no barrier, no VALU in the loop; the load stream runs on into the next tile.  Epilogue:
//     A6^T in-lane (j -> q), then A6^T across the six rows through LDS, one 32-channel half at a time (110 KB);
//   * weight gradient: dU[p] = Yt[p]^T V[p] over the 36 planes with K = tiles by wino24g_wgrad_kernel (the plane count is a
//     parameter there); Yt = A6 dY A6^T by wino44g_wgrad_xform_kernel, dW = G6^T (sum of splits) G6 by wino44g_wgrad_reduce_kernel.
#include <string.h>
#include <algorithm>
#include "common.hip.h"
#include "clamd_internal.h"
#include "wino_common.hip.h"

namespace clamd {

#ifdef CLAMD_DIAG
// diagnostic build only (python build.py --diag; tools/w44_diag.py): cycles per phase of a tile, summed over workgroups (wave 0)
__device__ unsigned long long g_w44_diag[8];
#define W44_T() __builtin_amdgcn_s_memtime()
#define W44_ADD(i_, v_) do { if (threadIdx.x == 0) atomicAdd(&g_w44_diag[i_], (unsigned long long)(v_)); } while (0)
#else
#define W44_T() 0ull
#define W44_ADD(i_, v_) do { } while (0)
#endif

// B6^T of six values
__device__ inline void w44_bt6(const float (&t)[6], float (&o)[6]) {
    const float pq = fmaf(-4.f, t[2], t[4]), qq = fmaf(-4.f, t[1], t[3]);
    const float rr = t[4] - t[2], ss = t[3] - t[1];
    o[0] = fmaf(4.f, t[0], fmaf(-5.f, t[2], t[4]));
    o[1] = pq + qq; o[2] = pq - qq;
    o[3] = fmaf(2.f, ss, rr); o[4] = fmaf(-2.f, ss, rr);
    o[5] = fmaf(4.f, t[1], fmaf(-5.f, t[3], t[5]));
}
// A6 of four values (the transpose of A6^T): [[1,0,0,0],[1,1,1,1],[1,-1,1,-1],[1,2,4,8],[1,-2,4,-8],[0,0,0,1]]
__device__ inline void w44_a6(const float (&z)[4], float (&o)[6]) {
    const float sa = z[0] + z[2], sb = z[1] + z[3], sc = fmaf(4.f, z[2], z[0]), sd = fmaf(4.f, z[3], z[1]);
    o[0] = z[0]; o[1] = sa + sb; o[2] = sa - sb;
    o[3] = fmaf(2.f, sd, sc); o[4] = fmaf(-2.f, sd, sc); o[5] = z[3];
}
// G6^T of six values -> three taps
__device__ inline void w44_gt6(const float (&t)[6], float (&o)[3]) {
    const float s12 = t[1] + t[2], d12 = t[2] - t[1], s34 = t[3] + t[4], d34 = t[3] - t[4];
    o[0] = 0.25f * t[0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
    o[1] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
    o[2] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + t[5];
}

// tile-block geometry: TXN tiles across x 32 / TXN down, 4 x 4 output pixels each
static inline void w44_block(int W, int& ph, int& pw) { if (W >= 32) { ph = 16; pw = 32; } else { ph = 32; pw = 16; } }
static long long w44_blocks(int B, int H, int W) {
    int ph, pw;
    w44_block(W, ph, pw);
    return (long long)B * ((H + ph - 1) / ph) * ((W + pw - 1) / pw);
}

// ---------------------------------------------------------------------------------------------------------------------
// input transform (forward / data gradient)
// ---------------------------------------------------------------------------------------------------------------------
struct W44XformParams {
    const float* x; int x_ldc;
    const float* scale; const float* shift;      // optional per-channel affine applied on load (a BatchNorm folded into the transform)
    float* v;
    int B, H, W, Kp;
};

// One workgroup = HALF a tile block (16 tiles: the upper or the lower tile rows) x 32 channels.  The halo of the half block is
// staged in LDS with whole 128-byte lines per pixel (49 KB: three workgroups per CU), zero padding -- and the optional affine
// y * scale + shift, in-image pixels only -- materialised at staging (wino24_xform_kernel).  Wave w owns 8-channel chunk w; lane
// (r16 = tile, h = 4-channel half, rh = row half) reads the 6 x 6 patch of its tile and forms rows i = 3 rh .. 3 rh + 2 of
// V = B6^T d B6; a store instruction of a wave covers 2 planes x 2 halves x 256 contiguous bytes.
template <int TXN>
__global__ void __launch_bounds__(256) wino44_xform_kernel(const W44XformParams p) {
    PASS_PRIO();      // a pass of the critical chain beside the second stream's MFMA kernels (elementwise.hip)
    constexpr int TYN = 32 / TXN, PW = 4 * TXN, PH = 4 * TYN;
    constexpr int HPH = PH / 2;                                        // pixel rows of a half block
    constexpr int HW_ = PW + 2, HH_ = HPH + 2, PIX = HW_ * HH_;
    constexpr int PITCH = 9;
    constexpr int NJ = (PIX * 8 + 255) / 256;
    __shared__ uint4 sm[PIX * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nk = p.Kp >> 3;
    const int tiles_x = (p.W + PW - 1) / PW, tiles_y = (p.H + PH - 1) / PH;
    const int nth = 2 * tiles_x * tiles_y * p.B;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int kg = bid / nth, tmh = bid - kg * nth;
    const int tm = tmh >> 1, s = tmh & 1;
    const int x0 = (tm % tiles_x) * PW, y0 = ((tm / tiles_x) % tiles_y) * PH + s * HPH, b = tm / (tiles_x * tiles_y);
    const unsigned img = (unsigned)p.H * (unsigned)p.W * (unsigned)p.x_ldc * 4u;
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc((const char*)p.x + (size_t)b * img, img);
    {
        const int g = tid & 7;
        const int ch = kg * 32 + 4 * g;
        const bool chan_ok = ch < p.Kp;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.scale && chan_ok) { sc = *reinterpret_cast<const float4*>(p.scale + ch); sh = *reinterpret_cast<const float4*>(p.shift + ch); }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int pix = (tid >> 3) + 32 * j;
            if (pix < PIX) {
                const int hy = pix / HW_, hx = pix - hy * HW_;
                const int yy = y0 + hy - 1, xx = x0 + hx - 1;
                const bool ok = chan_ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                const uint4 u = buf_ld16(xrs, ok ? (unsigned)(((yy * p.W + xx) * p.x_ldc + ch) * 4) : BUF_OOB, 0u);
                float4 f = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
                if (p.scale) {
                    f.x = ok ? fmaf(f.x, sc.x, sh.x) : 0.f; f.y = ok ? fmaf(f.y, sc.y, sh.y) : 0.f;
                    f.z = ok ? fmaf(f.z, sc.z, sh.z) : 0.f; f.w = ok ? fmaf(f.w, sc.w, sh.w) : 0.f;
                }
                sm[pix * PITCH + g] = make_uint4(__float_as_uint(f.x), __float_as_uint(f.y), __float_as_uint(f.z), __float_as_uint(f.w));
            }
        }
    }
    __syncthreads();
    const int kc = kg * 4 + w;
    if (kc >= nk) return;                                              // whole wave; no barrier below
    const int r16 = lane & 15, h = (lane >> 4) & 1, rh = lane >> 5;
    const int ty = r16 / TXN, tx = r16 % TXN;                          // tile inside the half block
    const uint4* const src = sm + ((4 * ty) * HW_ + 4 * tx) * PITCH + 2 * w + h;
    float t[3][6][4];                                                  // rows i = 3 rh + i' of B6^T d, per column c and channel e
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        float d[6][4];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const uint4 u = src[(a * HW_ + c) * PITCH];
            d[a][0] = __uint_as_float(u.x); d[a][1] = __uint_as_float(u.y); d[a][2] = __uint_as_float(u.z); d[a][3] = __uint_as_float(u.w);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float col[6] = {d[0][e], d[1][e], d[2][e], d[3][e], d[4][e], d[5][e]};
            float o[6];
            w44_bt6(col, o);
            t[0][c][e] = rh ? o[3] : o[0]; t[1][c][e] = rh ? o[4] : o[1]; t[2][c][e] = rh ? o[5] : o[2];
        }
    }
    const int r = 16 * s + r16;                                        // tile inside the block
    float* const dst = p.v + (((size_t)tm * nk + kc) * 36 + 18 * rh) * 256 + h * 128 + r * 4;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float o[6][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float row[6] = {t[i][0][e], t[i][1][e], t[i][2][e], t[i][3][e], t[i][4][e], t[i][5][e]};
            float oo[6];
            w44_bt6(row, oo);
#pragma unroll
            for (int j = 0; j < 6; ++j) o[j][e] = oo[j];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j)
            *reinterpret_cast<float4*>(dst + (6 * i + j) * 256) = make_float4(o[j][0], o[j][1], o[j][2], o[j][3]);
    }
}

// ---- filter transform: dst[(k/8)*36 + 6i + j][n][k%8] = (G6 g G6^T)[i][j] (jobs as in wino.hip) ----------------------------------
__global__ void __launch_bounds__(256) wino44_pack_kernel(const WinoPackJob* __restrict__ jobs, int njobs) {
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
    // rows: G6 g
    float t[6][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a = g[0][c], b = g[1][c], cc = g[2][c];
        t[0][c] = 0.25f * a;
        t[1][c] = (-1.f / 6.f) * (a + b + cc);
        t[2][c] = (-1.f / 6.f) * (a - b + cc);
        t[3][c] = (1.f / 24.f) * a + (1.f / 12.f) * b + (1.f / 6.f) * cc;
        t[4][c] = (1.f / 24.f) * a - (1.f / 12.f) * b + (1.f / 6.f) * cc;
        t[5][c] = cc;
    }
    float* d = J.dst + ((size_t)kc * 36 * J.Np + n) * 8 + k8;
    const size_t xs = (size_t)J.Np * 8;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float a = t[i][0], b = t[i][1], c = t[i][2];
        d[(6 * i + 0) * xs] = 0.25f * a;
        d[(6 * i + 1) * xs] = (-1.f / 6.f) * (a + b + c);
        d[(6 * i + 2) * xs] = (-1.f / 6.f) * (a - b + c);
        d[(6 * i + 3) * xs] = (1.f / 24.f) * a + (1.f / 12.f) * b + (1.f / 6.f) * c;
        d[(6 * i + 4) * xs] = (1.f / 24.f) * a - (1.f / 12.f) * b + (1.f / 6.f) * c;
        d[(6 * i + 5) * xs] = c;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward / data-gradient kernel on the transformed input
// ---------------------------------------------------------------------------------------------------------------------
constexpr int W44_EXP = 36;                                   // row pitch (floats) of the epilogue exchange block

template <int TXN, bool RAGGED>
__global__ void __launch_bounds__(768, 1) wino44g_kernel(const WinoParams p) {
    constexpr int TYN = 32 / TXN;
    constexpr int PW = 4 * TXN, PH = 4 * TYN;
    constexpr int EXB = 6 * 4 * 32 * W44_EXP;                             // floats of the exchange block [row i][q][tile][EXP]
    static_assert(EXB * 4 + 4 * 512 * 16 <= 160 * 1024, "LDS budget");
    __shared__ uint4 smem[EXB * 4 / 16 + 4 * 512];                       // exchange block + the readers' running statistics

    // statistics rows: wino24g_kernel's scheme (per-workgroup rows on the persistent grid, registers across the tiles of one
    // output slab, one fold per slab); thread (k = tid >> 6 < 2, c = tid & 63) owns word (k, 64 slab + c) of this workgroup's row
    float* const rows_base = p.stats;
    const bool per_wg_rows = rows_base != nullptr && gridDim.x < (unsigned)p.nblk;
    if (per_wg_rows && threadIdx.x < 128)
        for (int n = threadIdx.x & 63; n < p.Np; n += 64) rows_base[((size_t)blockIdx.x * 2 + (threadIdx.x >> 6)) * p.Np + n] = 0.f;
    float racc = 0.f;
    int rslab = -1;
    auto flush_row = [&]() {          // threads < 128 only
        if (rslab >= 0 && rslab * 64 + (int)(threadIdx.x & 63) < p.Np) {
            float* dst = rows_base + ((size_t)blockIdx.x * 2 + (threadIdx.x >> 6)) * p.Np + rslab * 64 + (threadIdx.x & 63);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float old = __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst, old + racc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // reader threads (tid < 512: tile tl, 4-channel group ng, column pair qh) keep partial sums of their 4 + 4 channels across the tiles
    // of one output slab -- in LDS behind the exchange block, not in registers: a wave has 168 registers (three per SIMD), 144 of them
    // are accumulators and operand fragments in the K loop.  sacc[(nt * 2 + kind) * 512 + tid] = float4 of channels 4 ng ..
    float4* const sacc = reinterpret_cast<float4*>(smem) + EXB / 4;
    if (threadIdx.x < 512) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sacc[i * 512 + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int cur_tn = -1, cur_tm = 0;
    auto fold_stats = [&]() {         // every thread of the workgroup
        __syncthreads();              // the readers of the last tile's second half are done with the exchange block (no barrier behind that phase)
        float* sb = reinterpret_cast<float*>(smem);                            // [reader wave 0..7][2][64]
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (w < 8) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float4 s1 = sacc[(nt * 2 + 0) * 512 + threadIdx.x], s2 = sacc[(nt * 2 + 1) * 512 + threadIdx.x];
                sacc[(nt * 2 + 0) * 512 + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
                sacc[(nt * 2 + 1) * 512 + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
                const float a4[4] = {s1.x, s1.y, s1.z, s1.w}, q4[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float a = a4[c], q = q4[c];
                    a += __shfl_xor(a, 8); a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
                    q += __shfl_xor(q, 8); q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
                    if (lane < 8) { sb[(w * 2 + 0) * 64 + 32 * nt + 4 * lane + c] = a; sb[(w * 2 + 1) * 64 + 32 * nt + 4 * lane + c] = q; }
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int k = threadIdx.x >> 6, c = threadIdx.x & 63;
            float t = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) t += sb[(ww * 2 + k) * 64 + c];
            if (!per_wg_rows) {
                if (cur_tn * 64 + c < p.Np) rows_base[((size_t)cur_tm * 2 + k) * p.Np + cur_tn * 64 + c] = t;
            } else {
                rslab = cur_tn; racc = t;
                flush_row();
            }
        }
        __syncthreads();
        cur_tn = -1;
    };

    const int tiles_x = (p.W + PW - 1) / PW, tiles_y = (p.H + PH - 1) / PH;
    const int ntn = p.Np >> 6;                                              // Np % 64 == 0 (checked on entry)
    const int nk = p.Kp >> 3;                                               // >= 2 (checked on entry)
    const int per_band = (p.nblk / ntn) * p.band;
    const __amdgpu_buffer_rsrc_t vrs = make_rsrc(p.x, (unsigned)((size_t)(p.nblk / ntn) * nk * 36 * 1024));
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(p.w, (unsigned)(36u * p.Np * p.Kp * 4u));
    const __amdgpu_buffer_rsrc_t vrs_dead = make_rsrc(p.x, 0u), wrs_dead = make_rsrc(p.w, 0u);
    const unsigned ustride = (unsigned)p.Np * 32u;                          // bytes of one (chunk, plane) of the filters
    auto decode = [&](int v, int& tn, int& tm) {
        const int bid = xcd_remap(v, p.nblk);
        const int bnd = bid / per_band, rem = bid - bnd * per_band;
        tn = bnd * p.band + rem % p.band; tm = rem / p.band;
    };

    uint4 A[6], Bq[6];
    bool pre = false;
    for (int v = blockIdx.x; v < p.nblk; v += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                                        // per-tile re-derivation (hoisted constants would spill)
        const unsigned long long dt0 = W44_T(); (void)dt0;
        const int lane = tid & 63;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wi = wv % 6, nt_w = wv / 6;                                 // this wave: Winograd row i, 32-channel half
        const int r = lane & 31, h = lane >> 5;
        int tn, tm;
        decode(v, tn, tm);
        const int x0 = (tm % tiles_x) * PW, y0 = ((tm / tiles_x) % tiles_y) * PH, b = tm / (tiles_x * tiles_y);
        const int n0 = tn * 64;
        if (cur_tn >= 0 && (tn != cur_tn || !per_wg_rows)) fold_stats();

        const int vn = v + (int)gridDim.x;
        const bool has_next = vn < p.nblk;
        int tnn, tmn;
        decode(has_next ? vn : v, tnn, tmn);

        const unsigned a_vo = (unsigned)(h * 512 + r * 16);
        const unsigned b_vo = (unsigned)((r * 8 + 4 * h) * 4);
        const unsigned plane0 = (unsigned)(wi * 6);
        const unsigned vb_cur = (unsigned)tm * (unsigned)nk * 36u * 1024u, vb_nxt = (unsigned)tmn * (unsigned)nk * 36u * 1024u;
        const unsigned ub_cur = (unsigned)(n0 + 32 * nt_w) * 32u, ub_nxt = (unsigned)(tnn * 64 + 32 * nt_w) * 32u;
        auto load_j = [&](int j, __amdgpu_buffer_rsrc_t vr, __amdgpu_buffer_rsrc_t ur, unsigned vb, unsigned ub, int k) {
            const unsigned pl = (unsigned)k * 36u + plane0 + (unsigned)j;
            A[j] = buf_ld16(vr, a_vo, vb + pl * 1024u);
            Bq[j] = buf_ld16(ur, b_vo, ub + pl * ustride);
        };

        f32x16 acc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

        if (!pre) {                                                           // first tile of this workgroup
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                load_j(j, vrs, wrs, vb_cur, ub_cur, 0);
                asm volatile("" ::: "memory");                                 // ring order (see wino24g_wgrad_kernel)
            }
        }
        const unsigned long long dt1 = W44_T(); (void)dt1;
        for (int k = 0; k < nk - 1; ++k) {
            // the fragments of plane j are refetched (chunk k + 1) behind the MFMAs of plane j + 1: their registers are free by then
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                mma16<float>(A[j], Bq[j], acc[j]);
                if (j > 0) load_j(j - 1, vrs, wrs, vb_cur, ub_cur, k + 1);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                if (j > 0) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            }
            load_j(5, vrs, wrs, vb_cur, ub_cur, k + 1);
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) mma16<float>(A[j], Bq[j], acc[j]);        // last chunk: nothing to fetch yet (see the epilogue)

        const unsigned long long dt2 = W44_T(); (void)dt2;
        unsigned long long dph[5] = {0, 0, 0, 0, 0};
        // ---- epilogue: Y = A6^T M A6.  A6^T along the columns in-lane (j -> q), A6^T along the rows across the six waves of a
        // channel half through LDS; the two channel halves take the exchange block in turn --------------------------------------
        float* const ex = reinterpret_cast<float*>(smem);
        const int tl = (tid >> 3) & 31, ng = tid & 7, qh = tid >> 8;          // reader (tid < 512): tile, 4-channel group, column pair
        const float relu_lo = p.relu ? 0.f : -__builtin_inff();
        const bool plain = !p.relu && !p.bias && !p.stats;
        const int oty = tl / TXN, otx = tl % TXN;
        // the in-lane half of the output transform for EVERY wave first: 96 accumulators become 64 values, so that the waves of the second
        // channel half can read back the first half's block without spilling (with the accumulators alive across that phase every output store
        // was followed by a spill reload and `s_waitcnt vmcnt(0)`: the reader stood behind each of its eight stores -- 8.9 k cycles per phase)
        // The barrier that frees the exchange block for THIS tile's first write sits here, behind the K loop, not behind the previous tile's
        // last read-back: a wave that is done reading (or stands at the issue of its output stores: 131 KB per CU and tile, every CU at the same
        // moment -- the stores take 9 k cycles to issue, ablation builds -DW44_ABLATE_ST) goes straight on into the next tile's K loop, and the
        // store drain runs under the MFMAs of whichever waves of the SIMD are already there.  Waiting here costs nothing: the waves finish their K
        // loops a third of the phase apart (oldest first) and waited at the next barrier anyway.
        __syncthreads();
        float4 bias_a = make_float4(0.f, 0.f, 0.f, 0.f), bias_b = bias_a;      // the reader's bias of either channel half (4 ng .. 4 ng + 3)
        if (p.bias && tid < 512) {
            bias_a = *reinterpret_cast<const float4*>(p.bias + n0 + 4 * ng);
            bias_b = *reinterpret_cast<const float4*>(p.bias + n0 + 32 + 4 * ng);
        }
        float tq[4][16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float m0 = acc[0][e], m1 = acc[1][e], m2 = acc[2][e], m3 = acc[3][e], m4 = acc[4][e], m5 = acc[5][e];
            const float sa = m1 + m2, sb = m1 - m2, sc = m3 + m4, sd = m3 - m4;
            tq[0][e] = m0 + sa + sc;
            tq[1][e] = fmaf(2.f, sd, sb);
            tq[2][e] = fmaf(4.f, sc, sa);
            tq[3][e] = fmaf(8.f, sd, sb) + m5;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (nt_w == nt) {                                                  // wave-uniform
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = acc_row(e, h);
#pragma unroll
                    for (int q = 0; q < 4; ++q) ex[((wi * 4 + q) * 32 + row) * W44_EXP + r] = tq[q][e];
                }
            }
            __syncthreads();
            dph[2 * nt + 1] = W44_T();
            if (nt == 0) {
                // both bias vectors have arrived by now; taking them HERE keeps the second phase from waiting for them behind the first phase's
                // output stores (loads and stores share one in-order counter: `s_waitcnt vmcnt(12)` in phase 1 stood for the store drain, 6 k cycles)
                asm volatile("" : "+v"(bias_a.x), "+v"(bias_a.y), "+v"(bias_a.z), "+v"(bias_a.w), "+v"(bias_b.x), "+v"(bias_b.y), "+v"(bias_b.z), "+v"(bias_b.w));
            }
            if (tid < 512) {
                const int n = n0 + 32 * nt + 4 * ng;
                const float4 bias4 = nt ? bias_b : bias_a;                 // loaded behind the K loop: no memory latency inside the phase
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
                float4 outv[2][4];                                         // the phase's eight outputs: stored LAST, behind the LDS traffic
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const int q = 2 * qh + qq;
                    float4 R[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) R[i] = *reinterpret_cast<const float4*>(ex + ((i * 4 + q) * 32 + tl) * W44_EXP + 4 * ng);
                    float4 o[4];
#define W44_ROWS(m_)                                                                                                   \
    do {                                                                                                               \
        const float sa_ = R[1].m_ + R[2].m_, sb_ = R[1].m_ - R[2].m_, sc_ = R[3].m_ + R[4].m_, sd_ = R[3].m_ - R[4].m_; \
        o[0].m_ = R[0].m_ + sa_ + sc_;                                                                                 \
        o[1].m_ = fmaf(2.f, sd_, sb_);                                                                                 \
        o[2].m_ = fmaf(4.f, sc_, sa_);                                                                                 \
        o[3].m_ = fmaf(8.f, sd_, sb_) + R[5].m_;                                                                       \
    } while (0)
                    W44_ROWS(x); W44_ROWS(y); W44_ROWS(z); W44_ROWS(w);
#undef W44_ROWS
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) {
                        float4 oo = o[pp];
                        if (!plain) {
                            oo.x = fmaxf(oo.x + bias4.x, relu_lo); oo.y = fmaxf(oo.y + bias4.y, relu_lo);
                            oo.z = fmaxf(oo.z + bias4.z, relu_lo); oo.w = fmaxf(oo.w + bias4.w, relu_lo);
                            const int yy = y0 + 4 * oty + pp, xx = x0 + 4 * otx + q;
                            if (!RAGGED || (yy < p.H && xx < p.W)) {
                                s1[0] += oo.x; s1[1] += oo.y; s1[2] += oo.z; s1[3] += oo.w;
                                s2[0] = fmaf(oo.x, oo.x, s2[0]); s2[1] = fmaf(oo.y, oo.y, s2[1]);
                                s2[2] = fmaf(oo.z, oo.z, s2[2]); s2[3] = fmaf(oo.w, oo.w, s2[3]);
                            }
                        }
                        outv[qq][pp] = oo;
                    }
                }
                if (p.stats) {                                             // running sums of this thread: tile after tile, in a fixed order
                    float4 a = sacc[(nt * 2 + 0) * 512 + tid], q = sacc[(nt * 2 + 1) * 512 + tid];
                    a.x += s1[0]; a.y += s1[1]; a.z += s1[2]; a.w += s1[3];
                    q.x += s2[0]; q.y += s2[1]; q.z += s2[2]; q.w += s2[3];
                    sacc[(nt * 2 + 0) * 512 + tid] = a; sacc[(nt * 2 + 1) * 512 + tid] = q;
                }
                // one descriptor per image, the pixel part of the address from scalar strides: a store costs two VALU, not a 64-bit multiply-add chain
                const unsigned img_bytes = (unsigned)p.H * (unsigned)p.W * (unsigned)p.y_ldc * 4u;
                const __amdgpu_buffer_rsrc_t yrs = make_rsrc((const char*)p.y + (size_t)b * img_bytes, img_bytes);
                const unsigned row_b = (unsigned)p.W * (unsigned)p.y_ldc * 4u, pix_b = (unsigned)p.y_ldc * 4u;
                const unsigned vo0 = (unsigned)(y0 + 4 * oty) * row_b + (unsigned)(x0 + 4 * otx + 2 * qh) * pix_b + (unsigned)n * 4u;
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) {
                        const int yy = y0 + 4 * oty + pp, xx = x0 + 4 * otx + 2 * qh + qq;
                        const unsigned vo = (!RAGGED || (yy < p.H && xx < p.W)) ? vo0 + pp * row_b + qq * pix_b : BUF_OOB;
                        const float4 oo = outv[qq][pp];
#ifdef W44_ABLATE_ST            // timing ablation (variant builds only): no output stores in phase W44_ABLATE_ST - 1
                        if (nt != W44_ABLATE_ST - 1)
#endif
                        buf_st16(yrs, vo, 0u, make_uint4(__float_as_uint(oo.x), __float_as_uint(oo.y), __float_as_uint(oo.z), __float_as_uint(oo.w)));
                    }
            }
            if (nt == 1) {
                // chunk 0 of this workgroup's next tile: requested here -- BEHIND the read-back of the second half (in front of it, the 144 KB of requests of the twelve waves stood
                // in the memory pipeline ahead of the readers' stores: 9.5 k instead of 3.5 k cycles for the phase) --, where the accumulators of every wave are dead (144 of a wave's 168
                // registers are accumulators and fragments in the K loop; the readers below need 60), it lands under the read-back,
                // row transform and stores of the second channel half.  An empty descriptor behind the last tile: zeros, no traffic --
                // the registers are redefined on every path
                const __amdgpu_buffer_rsrc_t vr = has_next ? vrs : vrs_dead, ur = has_next ? wrs : wrs_dead;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    load_j(j, vr, ur, vb_nxt, ub_nxt, 0);
                    asm volatile("" ::: "memory");
                }
                pre = has_next;
            }
            if (nt == 0) __syncthreads();                                  // the exchange block is free for the second half (behind the second half: see above)
            dph[2 * nt + 2] = W44_T();
        }
        // [0] setup + first loads  [1] K loop  [2] write 0 + barrier  [3] read 0 + barrier  [4] write 1 + barrier  [5] read 1 + barrier  [6] tiles  [7] chunks
        W44_ADD(0, dt1 - dt0); W44_ADD(1, dt2 - dt1); W44_ADD(2, dph[1] - dt2); W44_ADD(3, dph[2] - dph[1]); W44_ADD(4, dph[3] - dph[2]);
        W44_ADD(5, dph[4] - dph[3]); W44_ADD(6, 1); W44_ADD(7, nk);
        (void)dph;
        if (rows_base) { cur_tn = tn; cur_tm = tm; }
    }
    if (cur_tn >= 0) fold_stats();
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient: gradient-side operand transform  Yt = A6 dY A6^T of every 4 x 4 gradient tile -> [36][Tp][Rp], tiles in the order of
// the forward image V (t = 32 * tile block + tile in block).  thread = (tile, 4-channel group), channel groups fastest.
// ---------------------------------------------------------------------------------------------------------------------
struct W44WgXformParams {
    const float* src; int ldc;       // gz [B,H,W,ldc]
    float* dst;                      // Yt [36][Tp][Rp]
    int B, H, W, Rp, Tp;
};

template <int TXN>
__global__ void __launch_bounds__(256) wino44g_wgrad_xform_kernel(const W44WgXformParams p) {
    SIDE_PRIO();
    constexpr int TYN = 32 / TXN, PW = 4 * TXN, PH = 4 * TYN;
    const int ng = p.Rp >> 2;
    const long long total = (long long)p.Tp * ng;
    const int tiles_x = (p.W + PW - 1) / PW, tiles_y = (p.H + PH - 1) / PH;
    const unsigned img = (unsigned)p.H * (unsigned)p.W * (unsigned)p.ldc * 4u;
    const size_t ps = (size_t)p.Tp * p.Rp;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int g = (int)(idx % ng);
        const int t = (int)(idx / ng);
        const int tm = t >> 5, r = t & 31;
        const int x0 = (tm % tiles_x) * PW + 4 * (r % TXN), y0 = ((tm / tiles_x) % tiles_y) * PH + 4 * (r / TXN), b = tm / (tiles_x * tiles_y);
        float* const dst = p.dst + (size_t)t * p.Rp + 4 * g;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc((const char*)p.src + (size_t)b * img, img);
        float z[6][4][4];                                              // rows transformed: [i][column q][channel]
        {
            float gq[4][4][4];                                         // [row][column][channel]
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int yy = y0 + a, xx = x0 + q;
                    const uint4 u = buf_ld16(rs, (b < p.B && yy < p.H && xx < p.W) ? (unsigned)(((yy * p.W + xx) * p.ldc + 4 * g) * 4) : BUF_OOB, 0u);
                    gq[a][q][0] = __uint_as_float(u.x); gq[a][q][1] = __uint_as_float(u.y);
                    gq[a][q][2] = __uint_as_float(u.z); gq[a][q][3] = __uint_as_float(u.w);
                }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float col[4] = {gq[0][q][e], gq[1][q][e], gq[2][q][e], gq[3][q][e]};
                    float o[6];
                    w44_a6(col, o);
#pragma unroll
                    for (int i = 0; i < 6; ++i) z[i][q][e] = o[i];
                }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float o[6][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float row[4] = {z[i][0][e], z[i][1][e], z[i][2][e], z[i][3][e]};
                float oo[6];
                w44_a6(row, oo);
#pragma unroll
                for (int j = 0; j < 6; ++j) o[j][e] = oo[j];
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<float4*>(dst + (6 * i + j) * ps) = make_float4(o[j][0], o[j][1], o[j][2], o[j][3]);
        }
    }
}

// out[rl][cl][3][3] = G6^T (sum_s dU_s) G6.  A thread owns four consecutive (r, c) pairs (16-byte loads of all 36 planes); the splits
// are dealt to PHS lane groups of a wave and combined with fixed shuffle steps: the summation order is fixed (deterministic).
struct W44GReduceParams {
    const float* partial; float* out;
    int nsplit, Rp, Cp, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p;
};

template <int PHS>
__global__ void __launch_bounds__(256) wino44g_wgrad_reduce_kernel(const W44GReduceParams p) {
    SIDE_PRIO();
    constexpr int QW = 64 / PHS;                                       // quads per wave
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = lane % QW, ph = lane / QW;
    const long long nquad = (long long)p.Rp * p.Cp / 4;
    const size_t plane_sz = (size_t)p.Rp * p.Cp, split_sz = plane_sz * 36;
    for (long long base = ((long long)blockIdx.x * 4 + wv) * QW; base < nquad; base += (long long)gridDim.x * 4 * QW) {
        const long long quad = base + q;
        const bool ok = quad < nquad;
        float4 s[36];
#pragma unroll
        for (int i = 0; i < 36; ++i) s[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok)
            for (int k = ph; k < p.nsplit; k += PHS) {
                const float* src = p.partial + (size_t)k * split_sz + (size_t)quad * 4;
#pragma unroll
                for (int i = 0; i < 36; ++i) {
                    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)i * plane_sz);
                    s[i].x += v.x; s[i].y += v.y; s[i].z += v.z; s[i].w += v.w;
                }
            }
        if constexpr (PHS > 1) {
#pragma unroll
            for (int i = 0; i < 36; ++i) {
#pragma unroll
                for (int o = QW; o < 64; o *= 2) {
                    s[i].x += __shfl_xor(s[i].x, o); s[i].y += __shfl_xor(s[i].y, o);
                    s[i].z += __shfl_xor(s[i].z, o); s[i].w += __shfl_xor(s[i].w, o);
                }
            }
        }
        if (ph == 0 && ok) {
            const int rp = (int)((quad * 4) / p.Cp), cp0 = (int)((quad * 4) % p.Cp);
            const int rl = wn_phys2log(rp, p.r_seg0, p.r_seg0p, p.R);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cl = wn_phys2log(cp0 + e, p.c_seg0, p.c_seg0p, p.C);
                if (rl < 0 || cl < 0) continue;
                float t[3][6];                                         // rows: G6^T U
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    float col[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const float4 v = s[6 * i + j];
                        col[i] = e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w));
                    }
                    float o3[3];
                    w44_gt6(col, o3);
                    t[0][j] = o3[0]; t[1][j] = o3[1]; t[2][j] = o3[2];
                }
                float* o = p.out + ((size_t)rl * p.C + cl) * 9;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    float o3[3];
                    w44_gt6(t[a], o3);
                    o[a * 3 + 0] = o3[0]; o[a * 3 + 1] = o3[1]; o[a * 3 + 2] = o3[2];
                }
            }
        }
    }
}

}  // namespace clamd

using namespace clamd;

namespace clamd {
// wino24g.hip: the batched plane GEMM dU[p] = Yt[p]^T V[p] (planes = 24 or 36) with its split-K plan
int w24g_wg_plan_planes(long long Tp, int Rp, int Cp, int planes, const clamd_tuning& tn, int* per_out);
long long w24g_wg_max_split(long long Tp, int Rp, int Cp, int planes);
int launch_w24g_wgrad_gemm(const float* yt, const float* v, float* partial, int Rp, int Cp, long long Tp, int nsplit, int per, int planes, hipStream_t s);
size_t w24g_sk_workspace_bytes(long long Tp, int Rp, int Cp, int planes);
size_t w24g_skw_workspace_bytes(long long Tp, int Rp, int Cp, int planes);
int launch_w24g_wgrad_skw(const float* yt, const float* v, float* workspace, size_t ws_bytes, float* out, long long Tp, int Rp, int Cp, int planes,
                          int R, int C, int r_seg0, int r_seg0p, int c_seg0, int c_seg0p, const clamd_tuning& tn, hipStream_t s);
int launch_w24g_wgrad_sk(const float* yt, const float* v, float* workspace, size_t ws_bytes, float* out, long long Tp, int Rp, int Cp, int planes,
                         int R, int C, int r_seg0, int r_seg0p, int c_seg0, int c_seg0p, const clamd_tuning& tn, hipStream_t s);
}  // namespace clamd

extern "C" {

#ifdef CLAMD_DIAG
int clamd_debug_w44_diag(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(clamd::g_w44_diag), 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(clamd::g_w44_diag), z, 64) != hipSuccess) return -1; }
    return 0;
}
#endif

int clamd_wino44_pack(const void* jobs_dev, int njobs, int total_blocks, void* stream) {
    if (njobs <= 0 || total_blocks <= 0) return clamd_fail("wino44_pack: empty job table");
    hipLaunchKernelGGL(clamd::wino44_pack_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const clamd::WinoPackJob*)jobs_dev, njobs);
    return clamd_check_launch("wino44_pack");
}

size_t clamd_winograd44_input_elems(int B, int H, int W, int Cp) {
    if (B <= 0 || H <= 0 || W <= 0 || Cp <= 0) return 0;
    return (size_t)w44_blocks(B, H, W) * (size_t)(Cp / 8) * 36 * 256;
}

int clamd_winograd44_transform_input(const float* x, int x_ldc, const float* scale, const float* shift, float* v, int B, int H, int W,
                                     int Cp, void* stream) {
    if ((scale == nullptr) != (shift == nullptr)) return clamd_fail("winograd44_transform_input: scale and shift go together");
    if (B <= 0 || H <= 0 || W <= 0) return clamd_fail("winograd44_transform_input: empty problem");
    if ((H & 3) || (W & 3)) return clamd_fail("winograd44_transform_input: H and W must be multiples of 4 (4x4 output tiles)");
    if (Cp % 8 || x_ldc % 4 || x_ldc < Cp) return clamd_fail("winograd44_transform_input: channel count / pitch must be padded");
    if ((long long)H * W * x_ldc * 4 >= (1ll << 31)) return clamd_fail("winograd44_transform_input: image exceeds 2^31 bytes");
    int ph, pw;
    w44_block(W, ph, pw);
    const long long nth = 2 * w44_blocks(B, H, W), nkg = (Cp / 8 + 3) / 4;
    if (nth * nkg > 0x7fffffff) return clamd_fail("winograd44_transform_input: grid out of range");
    W44XformParams p{x, x_ldc, scale, shift, v, B, H, W, Cp};
    if (pw == 32) hipLaunchKernelGGL(wino44_xform_kernel<8>, dim3((unsigned)(nth * nkg)), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(wino44_xform_kernel<4>, dim3((unsigned)(nth * nkg)), dim3(256), 0, (hipStream_t)stream, p);
    return clamd_check_launch("winograd44_transform_input");
}

}  // extern "C"

namespace clamd {
// rows of a launch: one per tile block (one workgroup per tile block and slab), or one per workgroup of the persistent grid
long long clamd_winograd44_stat_rows(int B, int H, int W, int Cout_p, const clamd_tuning& tn) {
    const long long tiles = w44_blocks(B, H, W), nblk = tiles * ((Cout_p + 63) / 64);
    return (tn.wino_persist && nblk > clamd_usable_cus(tn)) ? clamd_usable_cus(tn) : tiles;
}
}  // namespace clamd

extern "C" {

int clamd_conv3x3_winograd44_pre(const float* v, const float* w_wino, const float* bias, float* y, int y_ldc,
                                 float* stats, int stat_rows, int B, int H, int W, int Cin_p,
                                 int Cout_p, int relu, const clamd_tuning* tune, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return clamd_fail("conv3x3_winograd44_pre: empty problem");
    if ((H & 3) || (W & 3)) return clamd_fail("conv3x3_winograd44_pre: H and W must be multiples of 4 (4x4 output tiles)");
    if (Cin_p % 8 || Cin_p < 16 || Cout_p % 64 || y_ldc % 4) return clamd_fail("conv3x3_winograd44_pre: needs Cin_p % 8 == 0, Cin_p >= 16, Cout_p % 64 == 0");
    if (int e = clamd_check_tuning(tune)) return e;
    const clamd_tuning& tn = clamd_tune(tune);
    const long long tiles = w44_blocks(B, H, W), ntn = Cout_p / 64;
    if (tiles * ntn > 0x7fffffff) return clamd_fail("conv3x3_winograd44_pre: grid out of range");
    if ((unsigned long long)tiles * (Cin_p / 8) * 36 * 1024 >= (1ull << 32) || (long long)36 * Cout_p * Cin_p * 4 >= (1ll << 31))
        return clamd_fail("conv3x3_winograd44_pre: transformed input exceeds 2^32 bytes or filter 2^31 bytes");
    if (stats && stat_rows != clamd_winograd44_stat_rows(B, H, W, Cout_p, tn))
        return clamd_fail("conv3x3_winograd44_pre: stat_rows does not match clamd_stat_rows(CLAMD_OP_CONV3X3_WINOGRAD44, ...)");
    if (relu & ~1) return clamd_fail("conv3x3_winograd44_pre: relu must be 0 or 1 (no border-class bias here)");
    if ((long long)H * W * y_ldc * 4 >= (1ll << 31)) return clamd_fail("conv3x3_winograd44_pre: one output image exceeds 2^31 bytes");
    WinoParams p{v, 0, w_wino, bias, y, y_ldc, stats, B, H, W, Cin_p, Cout_p, relu, 1, 0};
    p.band = wino_band(tiles, ntn, 2.25 * B * H * W * Cin_p, 36.0 * Cin_p * Cout_p, tn.wino_band);
    p.nblk = (int)(tiles * ntn);
    const unsigned grid = tn.wino_persist ? (unsigned)std::min<long long>(p.nblk, clamd_usable_cus(tn)) : (unsigned)p.nblk;
    int ph, pw;
    w44_block(W, ph, pw);
    const bool ragged = (H % ph) != 0 || (W % pw) != 0;
    hipStream_t s = (hipStream_t)stream;
#define W44G_LAUNCH(TXN_, RG_) hipLaunchKernelGGL((wino44g_kernel<TXN_, RG_>), dim3(grid), dim3(768), 0, s, p)
    if (pw == 32) { if (ragged) W44G_LAUNCH(8, true); else W44G_LAUNCH(8, false); }
    else { if (ragged) W44G_LAUNCH(4, true); else W44G_LAUNCH(4, false); }
#undef W44G_LAUNCH
    return clamd_check_launch("conv3x3_winograd44_pre");
}

size_t clamd_wgrad_winograd44_pre_operand_elems(int B, int H, int W, int Rp) {
    if (B <= 0 || H <= 0 || W <= 0 || Rp <= 0) return 0;
    return (size_t)(36 * w44_blocks(B, H, W) * 32) * (size_t)Rp;
}

size_t clamd_wgrad_winograd44_pre_workspace_bytes(int B, int H, int W, int Rp, int Cp) {
    if (B <= 0 || H <= 0 || W <= 0 || Rp < 128 || Cp < 128) return 0;
    const long long Tp = w44_blocks(B, H, W) * 32;
    if ((Rp % 256) || (Cp % 256)) return w24g_skw_workspace_bytes(Tp, Rp, Cp, 36);      // multiples of 128: the wave-level plan
    return std::max((size_t)w24g_wg_max_split(Tp, Rp, Cp, 36) * 36 * Rp * Cp * sizeof(float), w24g_sk_workspace_bytes(Tp, Rp, Cp, 36));
}

static int w44_launch_yt(const float* gz, int gz_ldc, float* yt, int B, int H, int W, int Rp, long long Tp, hipStream_t s) {
    W44WgXformParams pa{gz, gz_ldc, yt, B, H, W, Rp, (int)Tp};
    const long long na = (Tp * (Rp / 4) + 255) / 256;
    const unsigned g = (unsigned)std::min<long long>(na, 1 << 20);
    if (W >= 32) hipLaunchKernelGGL(wino44g_wgrad_xform_kernel<8>, dim3(g), dim3(256), 0, s, pa);
    else hipLaunchKernelGGL(wino44g_wgrad_xform_kernel<4>, dim3(g), dim3(256), 0, s, pa);
    return clamd_check_launch("wgrad_winograd44_pre transform");
}

int clamd_wgrad_winograd44_pre(const float* gz, int gz_ldc, const float* v, float* yt, float* workspace, size_t ws_bytes, float* out,
                               int B, int H, int W, int Rp, int Cp, int R, int C, int r_seg0, int r_seg0p, int c_seg0, int c_seg0p,
                               const clamd_tuning* tune, void* stream) {
    if (int e = clamd_check_tuning(tune)) return e;
    if (B <= 0 || H <= 0 || W <= 0) return clamd_fail("wgrad_winograd44_pre: empty problem");
    if ((H & 3) || (W & 3)) return clamd_fail("wgrad_winograd44_pre: H and W must be multiples of 4");
    if (Rp % 128 || Cp % 128 || Rp <= 0 || Cp <= 0 || gz_ldc % 4) return clamd_fail("wgrad_winograd44_pre: needs Rp and Cp multiples of 128");
    const bool wave_level = (Rp % 256) || (Cp % 256);          // a wave owns a 128 x 128 block; 256 x 256 workgroup blocks where both counts allow
    if (gz && (long long)H * W * gz_ldc * 4 >= (1ll << 31)) return clamd_fail("wgrad_winograd44_pre: one image exceeds 2^31 bytes");
    const clamd_tuning& tn = clamd_tune(tune);
    const long long ntm = w44_blocks(B, H, W), Tp = ntm * 32;
    if (Tp * Rp * 4 >= (1ll << 32) || (unsigned long long)ntm * (Cp / 8) * 36 * 1024 >= (1ull << 32))
        return clamd_fail("wgrad_winograd44_pre: an operand exceeds 2^32 bytes");
    int per = 0;
    const int nsplit = wave_level ? 1 : w24g_wg_plan_planes(Tp, Rp, Cp, 36, tn, &per);
    // stream-K where the items alone nearly fill the chip (an item then gets 2-3 slots); with fewer, larger items the split-K plan's
    // whole rounds are as even and its slabs are fewer (tools/wino44g_ab.py: 1.10-1.21x faster from 96 items on, 0.78-0.95x below)
    const bool streamk = wave_level || tn.wgrad_streamk == 2 || (tn.wgrad_streamk == 1 && (long long)36 * (Rp / 256) * (Cp / 256) * 8 >= 3LL * clamd_usable_cus(tn));
    if (!streamk && (size_t)nsplit * 36 * Rp * Cp * sizeof(float) > ws_bytes) return clamd_fail("wgrad_winograd44_pre: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    if (gz)                                    // gz == NULL: yt already holds the transformed gradient (clamd_wgrad_winograd44_pre_transform)
        if (int e = w44_launch_yt(gz, gz_ldc, yt, B, H, W, Rp, Tp, s)) return e;
    if (wave_level)
        return launch_w24g_wgrad_skw(yt, v, workspace, ws_bytes, out, Tp, Rp, Cp, 36, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p, tn, s);
    if (streamk)
        return launch_w24g_wgrad_sk(yt, v, workspace, ws_bytes, out, Tp, Rp, Cp, 36, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p, tn, s);
    if (int e = launch_w24g_wgrad_gemm(yt, v, workspace, Rp, Cp, Tp, nsplit, per, 36, s)) return e;
    W44GReduceParams rp{workspace, out, nsplit, Rp, Cp, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p};
    const long long nquad = (long long)Rp * Cp / 4;
#define W44G_REDUCE(PHS_)                                                                                              \
    do {                                                                                                               \
        long long g = (nquad + 4 * (64 / PHS_) - 1) / (4 * (64 / PHS_));                                               \
        if (g > 8192) g = 8192;                                                                                        \
        hipLaunchKernelGGL(wino44g_wgrad_reduce_kernel<PHS_>, dim3((unsigned)g), dim3(256), 0, s, rp);                 \
    } while (0)
    if (nsplit >= 4 && nquad <= 65536) W44G_REDUCE(4);
    else if (nsplit >= 2 && nquad <= 131072) W44G_REDUCE(2);
    else W44G_REDUCE(1);
#undef W44G_REDUCE
    return clamd_check_launch("wgrad_winograd44_pre_reduce");
}

int clamd_wgrad_winograd44_pre_transform(const float* gz, int gz_ldc, float* yt, int B, int H, int W, int Rp, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || !gz || !yt) return clamd_fail("wgrad_winograd44_pre_transform: bad arguments");
    if ((H & 3) || (W & 3)) return clamd_fail("wgrad_winograd44_pre_transform: H and W must be multiples of 4");
    if (Rp % 4 || Rp <= 0 || gz_ldc % 4) return clamd_fail("wgrad_winograd44_pre_transform: channel count / pitch must be padded");
    const long long Tp = w44_blocks(B, H, W) * 32;
    if ((long long)H * W * gz_ldc * 4 >= (1ll << 31) || Tp * Rp * 4 >= (1ll << 32)) return clamd_fail("wgrad_winograd44_pre_transform: operand too large");
    return w44_launch_yt(gz, gz_ldc, yt, B, H, W, Rp, Tp, (hipStream_t)stream);
}

}  // extern "C"
