// Winograd F(2x2, 3x3) convolution for the exact-fp32 path on gfx950 (forward and data gradient of nn.Conv2d(k3,s1,p1),
// models/unet.py:13,16,...; loss.backward() trainer.py:175).
//
// The fp32 path is MFMA-bound (v_mfma_f32_32x32x2_f32 runs at the vector-FMA rate, the direct kernels sit at 90 % of
// it), so the only large lever left is doing fewer multiplies: with Y = A^T [ (G g G^T) .* (B^T d B) ] A a 2x2 output
// tile costs 16 multiply-adds per (Cin, Cout) pair instead of 36 -- 2.25x fewer MFMA cycles for the same result.  In
// fp32 the transforms cost nothing measurable in accuracy: relative error vs an fp64 reference 2.0e-7 .. 3.5e-7 against
// 1.7e-7 .. 2.3e-7 for the direct sum (tests hold both to the same 2e-5 bound).
//
//   * filters are transformed once per step by wino_pack_kernel: U[xi = 4i + j] = (G g G^T)[i][j], stored
//     [Cin_p/8][16][Cout_p][8] so the 16 x 64 x 8 slab of one K-chunk and one 64-channel output slab is staged with
//     coalesced 2-KB rows (the data gradient uses the tap-flipped, transposed filter);
//   * a 256-thread workgroup (ONE per CU: 256 accumulator registers per lane) owns 8 x 8 Winograd tiles = 16 x 16
//     output pixels x 64 output channels.  Per 8-channel K-chunk it stages the 18 x 18 input halo and the filter slab in
//     LDS (two stages: while chunk k multiplies from registers, the fragments of chunk k+1 are read and transformed,
//     chunk k+2 is written over chunk k's stage and the loads of chunk k+3 are in flight);
//   * wave w owns Winograd row i = w: it forms t[b] = sum_a B^T[w][a] d[a][b] from two halo rows (8 ds_read_b128) and
//     the four V[w][j] = sum_b B^T[j][b] t[b] in registers -- the input transform is never materialised -- and
//     accumulates M[w][j] (64 tiles x 64 channels each) with the same 4-MFMA-per-16-byte-group step as the direct kernels;
//   * epilogue: R[q] = sum_j A^T[q][j] M[w][j] locally, the sum over the four waves' rows through LDS, then bias, ReLU,
//     BatchNorm statistics and 16-byte stores.
#include <string.h>
#include "common.hip.h"
#include "clamd_internal.h"

namespace clamd {

struct WinoParams {
    const float* x; int x_ldc;
    const float* w;              // [Kp/8][16][Np][8]
    const float* bias;
    float* y; int y_ldc;
    float* stats;                // [STAT_REPLICAS][2][Np] or null
    int B, H, W, Kp, Np, relu;
};

constexpr int WN_HW = 18, WN_PIX = WN_HW * WN_HW;            // input halo of a 16x16 output tile
constexpr int WN_PIXP = 330;                                  // == 2 (mod 8): conflict-free staging stores
constexpr int WN_WG = 66;                                     // padded rows per (xi, group)
constexpr int WN_IN_SLOTS = 2 * WN_PIXP, WN_WT_SLOTS = 16 * 2 * WN_WG;
constexpr int WN_STAGE = WN_IN_SLOTS + WN_WT_SLOTS;           // 16-byte slots per stage (44 KB)
constexpr int WN_EXP = 36;                                    // row pitch (floats) of the epilogue exchange block

__global__ void __launch_bounds__(256, 1) wino_kernel(const WinoParams p) {
    constexpr int NST = 2;                     // LDS stages: chunk k+1 readable, chunk k+2 being written (chunk k is in registers)
    static_assert(4 * 2 * 32 * WN_EXP * 4 <= NST * WN_STAGE * 16 && NST * WN_STAGE * 16 <= 160 * 1024, "LDS budget");
    __shared__ uint4 smem[NST * WN_STAGE];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_x = (p.W + 15) >> 4, tiles_y = (p.H + 15) >> 4;
    const int ntn = (p.Np + 63) >> 6;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % ntn, tm = bid / ntn;
    const int x0 = (tm % tiles_x) * 16, y0 = ((tm / tiles_x) % tiles_y) * 16, b = tm / (tiles_x * tiles_y);
    const int n0 = tn * 64;
    const int nk = p.Kp >> 3;

    // ---- staging descriptors --------------------------------------------------------------------------------------
    const unsigned img = (unsigned)p.H * (unsigned)p.W * (unsigned)p.x_ldc * 4u;
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc((const char*)p.x + (size_t)b * img, img);
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(p.w, (unsigned)(16u * p.Np * p.Kp * 4u));
    unsigned in_vo[3];
    int in_slot[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int piece = tid + 256 * j;                           // (pixel, 16-byte group): 648 pieces, the last pass wraps
        if (piece >= 2 * WN_PIX) piece -= 2 * WN_PIX;
        const int g = piece & 1, pix = piece >> 1;
        const int hy = pix / WN_HW, hx = pix - hy * WN_HW;
        const int yy = y0 + hy - 1, xx = x0 + hx - 1;
        in_vo[j] = (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) ? (unsigned)(((yy * p.W + xx) * p.x_ldc + 4 * g) * 4) : BUF_OOB;
        in_slot[j] = g * WN_PIXP + pix;
    }
    // filter slab of one K-chunk: 16 xi x 64 rows x 2 groups = 2048 pieces, piece = tid + 256*j: xi = (tid >> 7) + 2j
    const int wg_ = tid & 1, wn_ = (tid >> 1) & 63, wxi0 = tid >> 7;
    const unsigned w_vo0 = n0 + wn_ < p.Np ? (unsigned)(((wxi0 * p.Np + n0 + wn_) * 8 + 4 * wg_) * 4) : BUF_OOB;
    const unsigned w_vstep = (unsigned)(2 * p.Np * 8 * 4);    // two xi further
    const unsigned w_chunk = (unsigned)(16 * p.Np * 8 * 4);   // bytes of one K-chunk
    const int w_slot0 = WN_IN_SLOTS + (wxi0 * 2 + wg_) * WN_WG + wn_;

    uint4 rin[3], rw[8];
    auto gload = [&](int k, bool live) {
        const unsigned so = (unsigned)(k * 8 * 4);
#pragma unroll
        for (int j = 0; j < 3; ++j) rin[j] = buf_ld16(xrs, live ? in_vo[j] : BUF_OOB, so);
        const unsigned wv = live ? w_vo0 : BUF_OOB;
#pragma unroll
        for (int j = 0; j < 8; ++j) rw[j] = buf_ld16(wrs, wv + j * w_vstep, (unsigned)k * w_chunk);
    };
    auto lds_store = [&](int st) {
        uint4* sm = smem + st * WN_STAGE;
#pragma unroll
        for (int j = 0; j < 3; ++j) sm[in_slot[j]] = rin[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) sm[w_slot0 + j * 4 * WN_WG] = rw[j];
    };

    // ---- fragment addressing: wave w = Winograd row i: t[b] = s1 * d[a1][b] + s2 * d[a2][b] ------------------------
    const int a1 = w == 0 ? 0 : 1, a2 = w == 3 ? 3 : 2;
    const float s1 = w == 2 ? -1.f : 1.f, s2 = (w == 0 || w == 3) ? -1.f : 1.f;
    int p1[2], p2[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = 32 * mt + r, ty = m >> 3, tx = m & 7;
        const int pb = h * WN_PIXP + (2 * ty) * WN_HW + 2 * tx;
        p1[mt] = pb + a1 * WN_HW;
        p2[mt] = pb + a2 * WN_HW;
    }
    const int wb = WN_IN_SLOTS + (4 * w * 2 + h) * WN_WG + r;            // + j*2*WG + 32*nt

    f32x16 acc[4][2][2];                                                  // [j][tile half][channel half]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i >> 1][i & 1][e] = 0.f;

    // fragments of one chunk: the input transform V[w][j] for both tile halves + the filter rows of this wave's 4 xi
    uint4 A[4][2], Bf[4][2];
    auto frags = [&](int st) {
        const uint4* sm = smem + st * WN_STAGE;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float4 t[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint4 u1 = sm[p1[mt] + c], u2 = sm[p2[mt] + c];
                t[c].x = fmaf(s2, __uint_as_float(u2.x), s1 * __uint_as_float(u1.x));
                t[c].y = fmaf(s2, __uint_as_float(u2.y), s1 * __uint_as_float(u1.y));
                t[c].z = fmaf(s2, __uint_as_float(u2.z), s1 * __uint_as_float(u1.z));
                t[c].w = fmaf(s2, __uint_as_float(u2.w), s1 * __uint_as_float(u1.w));
            }
#define WN_PK(v_) make_uint4(__float_as_uint((v_).x), __float_as_uint((v_).y), __float_as_uint((v_).z), __float_as_uint((v_).w))
            A[0][mt] = WN_PK(make_float4(t[0].x - t[2].x, t[0].y - t[2].y, t[0].z - t[2].z, t[0].w - t[2].w));
            A[1][mt] = WN_PK(make_float4(t[1].x + t[2].x, t[1].y + t[2].y, t[1].z + t[2].z, t[1].w + t[2].w));
            A[2][mt] = WN_PK(make_float4(t[2].x - t[1].x, t[2].y - t[1].y, t[2].z - t[1].z, t[2].w - t[1].w));
            A[3][mt] = WN_PK(make_float4(t[1].x - t[3].x, t[1].y - t[3].y, t[1].z - t[3].z, t[1].w - t[3].w));
#undef WN_PK
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) Bf[j][nt] = sm[wb + j * 2 * WN_WG + 32 * nt];
    };

    // Pipeline (one wave per SIMD, nothing else hides a stall): while the MFMAs of chunk k run from registers, the
    // fragments of chunk k+1 are read from LDS and transformed, the raw data of chunk k+2 (loaded one chunk ago) is
    // written to the third stage and the loads of chunk k+3 are issued.  One barrier per chunk.
    gload(0, true);
    lds_store(0);
    gload(1, 1 < nk);
    lds_store(1);
    gload(2, 2 < nk);
    __syncthreads();
    frags(0);
    __syncthreads();                                                      // stage 0 is free again
    for (int k = 0; k < nk; ++k) {
        uint4 Ac[4][2], Bc[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) { Ac[j][i] = A[j][i]; Bc[j][i] = Bf[j][i]; }
        // straight-line body (past the last chunk the reads hit valid LDS and the stores write zeros nobody reads), so
        // that the 24 fragment reads, ~130 transform VALU ops, 11 staging stores and 11 loads can be issued in the gaps
        // of the 64 MFMAs instead of in front of them
        frags((k + 1) % NST);                                             // visible since the last barrier
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) mma16<float>(Ac[j][mt], Bc[j][nt], acc[j][mt][nt]);
        lds_store(k % NST);                                               // chunk k+2 over chunk k's stage (read before the last barrier)
        gload(k + 3, k + 3 < nk);
        sched_mfma_slots<64, 24, 26, 37, 38, 49, 2>();
        __syncthreads();
    }

    // ---- epilogue: output transform Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]] -------------------------------------
    float* const ex = reinterpret_cast<float*>(smem);                     // [wave][q][32 tiles][WN_EXP]
    const int tl = tid >> 3, ng = tid & 7;                                // reader: tile inside the half, 4-channel group
    float st1[2][4], st2[2][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { st1[i >> 2][i & 3] = 0.f; st2[i >> 2][i & 3] = 0.f; }
    const float relu_lo = p.relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float m0 = acc[0][mt][nt][e], m1 = acc[1][mt][nt][e], m2 = acc[2][mt][nt][e], m3 = acc[3][mt][nt][e];
                ex[((w * 2 + 0) * 32 + acc_row(e, h)) * WN_EXP + r] = m0 + m1 + m2;
                ex[((w * 2 + 1) * 32 + acc_row(e, h)) * WN_EXP + r] = m1 - m2 - m3;
            }
            __syncthreads();
            float4 R[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) R[i][q] = *reinterpret_cast<const float4*>(ex + ((i * 2 + q) * 32 + tl) * WN_EXP + 4 * ng);
            const int m = 32 * mt + tl, ty = m >> 3, tx = m & 7;
            const int n = n0 + 32 * nt + 4 * ng;
            float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias && n < p.Np) bias4 = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float4 v;
                    if (pp == 0) {
                        v.x = R[0][q].x + R[1][q].x + R[2][q].x; v.y = R[0][q].y + R[1][q].y + R[2][q].y;
                        v.z = R[0][q].z + R[1][q].z + R[2][q].z; v.w = R[0][q].w + R[1][q].w + R[2][q].w;
                    } else {
                        v.x = R[1][q].x - R[2][q].x - R[3][q].x; v.y = R[1][q].y - R[2][q].y - R[3][q].y;
                        v.z = R[1][q].z - R[2][q].z - R[3][q].z; v.w = R[1][q].w - R[2][q].w - R[3][q].w;
                    }
                    v.x = fmaxf(v.x + bias4.x, relu_lo); v.y = fmaxf(v.y + bias4.y, relu_lo);
                    v.z = fmaxf(v.z + bias4.z, relu_lo); v.w = fmaxf(v.w + bias4.w, relu_lo);
                    const int yy = y0 + 2 * ty + pp, xx = x0 + 2 * tx + q;
                    if (yy < p.H && xx < p.W && n < p.Np) {
                        *reinterpret_cast<float4*>(p.y + (((size_t)b * p.H + yy) * p.W + xx) * p.y_ldc + n) = v;
                        st1[nt][0] += v.x; st1[nt][1] += v.y; st1[nt][2] += v.z; st1[nt][3] += v.w;
                        st2[nt][0] = fmaf(v.x, v.x, st2[nt][0]); st2[nt][1] = fmaf(v.y, v.y, st2[nt][1]);
                        st2[nt][2] = fmaf(v.z, v.z, st2[nt][2]); st2[nt][3] = fmaf(v.w, v.w, st2[nt][3]);
                    }
                }
            __syncthreads();
        }
    if (p.stats) {
        // threads with equal (tid & 7) own the same channels: fold the 8 tiles of a wave (lane bits 3-5), then the 4 waves
        float* sb = ex;                                                    // [wave][2][64]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = st1[nt][c], q = st2[nt][c];
                a += __shfl_xor(a, 8); a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
                q += __shfl_xor(q, 8); q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
                if (lane < 8) { sb[(w * 2 + 0) * 64 + 32 * nt + 4 * lane + c] = a; sb[(w * 2 + 1) * 64 + 32 * nt + 4 * lane + c] = q; }
            }
        __syncthreads();
        if (tid < 128) {
            const int k = tid >> 6, c = tid & 63;
            const float t = sb[(0 * 2 + k) * 64 + c] + sb[(1 * 2 + k) * 64 + c] + sb[(2 * 2 + k) * 64 + c] + sb[(3 * 2 + k) * 64 + c];
            if (n0 + c < p.Np) atomicAdd(p.stats + ((size_t)(blockIdx.x % STAT_REPLICAS) * 2 + k) * p.Np + n0 + c, t);
        }
    }
}

// ---- filter transform ---------------------------------------------------------------------------------------------
// One job = one GEMM operand: dst[(k/8)*16 + xi][n][k%8] = (G g G^T)[xi] with g = w[n_l][k_l] (forward) or the tap-flipped
// w[k_l][n_l] (data gradient); physical -> logical channel maps as in clamd_pack (two segments for concat inputs, zero
// padding).  One thread per (n, k): consecutive threads write consecutive floats of every xi row.
struct WinoPackJob {
    const float* w; float* dst;
    int Np, Kp, N, K;                 // physical / logical sizes of the GEMM's N (rows) and K
    int n_seg0, n_seg0p, k_seg0, k_seg0p;
    int dgrad;                        // 0: g = w[n][k], src [N][K][3][3]; 1: g = flip(w[k][n]), src [K][N][3][3]
    int block0;                       // first workgroup of this job
};

__device__ inline int wn_phys2log(int p, int seg0, int seg0p, int L) {
    if (p < seg0p) return p < seg0 ? p : -1;
    const int l = seg0 + (p - seg0p);
    return l < L ? l : -1;
}

__global__ void __launch_bounds__(256) wino_pack_kernel(const WinoPackJob* __restrict__ jobs, int njobs) {
    int ji = 0;
    while (ji + 1 < njobs && (int)blockIdx.x >= jobs[ji + 1].block0) ++ji;     // few jobs: linear search
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
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i / 3][i % 3] = J.dgrad ? s[8 - i] : s[i];
    }
    // U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    float t[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        t[0][c] = g[0][c];
        t[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
        t[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
        t[3][c] = g[2][c];
    }
    float* d = J.dst + ((size_t)kc * 16 * J.Np + n) * 8 + k8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float u0 = t[i][0], u1 = 0.5f * (t[i][0] + t[i][1] + t[i][2]), u2 = 0.5f * (t[i][0] - t[i][1] + t[i][2]), u3 = t[i][2];
        d[(size_t)(4 * i + 0) * J.Np * 8] = u0;
        d[(size_t)(4 * i + 1) * J.Np * 8] = u1;
        d[(size_t)(4 * i + 2) * J.Np * 8] = u2;
        d[(size_t)(4 * i + 3) * J.Np * 8] = u3;
    }
}

}  // namespace clamd

using namespace clamd;

extern "C" {

int clamd_sizeof_wino_pack_job(void) { return (int)sizeof(WinoPackJob); }

int clamd_wino_pack(const void* jobs_dev, int njobs, int total_blocks, void* stream) {
    if (njobs <= 0 || total_blocks <= 0) return clamd_fail("wino_pack: empty job table");
    hipLaunchKernelGGL(wino_pack_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const WinoPackJob*)jobs_dev, njobs);
    return clamd_check_launch("wino_pack");
}

int clamd_conv3x3_winograd(const float* x, int x_ldc, const float* w_wino, const float* bias, float* y, int y_ldc,
                           float* stats, int B, int H, int W, int Cin_p, int Cout_p, int relu, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return clamd_fail("conv3x3_winograd: empty problem");
    if ((H | W) & 1) return clamd_fail("conv3x3_winograd: H and W must be even (2x2 output tiles)");
    if (Cin_p % 32 || Cout_p % 32 || x_ldc % 8 || y_ldc % 8) return clamd_fail("conv3x3_winograd: channel counts/pitches must be padded");
    if ((long long)H * W * x_ldc * 4 >= (1ll << 31) || (long long)16 * Cout_p * Cin_p * 4 >= (1ll << 31))
        return clamd_fail("conv3x3_winograd: image or filter exceeds 2^31 bytes");
    WinoParams p{x, x_ldc, w_wino, bias, y, y_ldc, stats, B, H, W, Cin_p, Cout_p, relu};
    const long long nblk = (long long)B * ((H + 15) / 16) * ((W + 15) / 16) * ((Cout_p + 63) / 64);
    if (nblk > 0x7fffffff) return clamd_fail("conv3x3_winograd: grid out of range");
    hipLaunchKernelGGL(wino_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    return clamd_check_launch("conv3x3_winograd");
}

}  // extern "C"
