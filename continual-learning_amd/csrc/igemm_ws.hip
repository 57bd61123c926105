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

template <typename T, int TW>
__global__ void __launch_bounds__(512, 2) igemm_ws_kernel(const IgemmParams p) {
    using G = Geo<MODE_CONV3, TW>;
    constexpr int TH = G::TH, NT = G::NT, HW_ = G::HW_, NPIX = G::NPIX, NPIXP = G::NPIXP, NJ = G::NJ;
    constexpr int KC = DT<T>::KC, VEC = DT<T>::VEC;
    constexpr bool SPLIT = __is_same(T, split_t);
    constexpr int STAGE = G::IN_SLOTS + G::WT_SLOTS;
    static_assert(2 * STAGE * 16 >= (4 * 32 * 68 + 4 * 2 * 64) * 4, "LDS too small for the epilogue");
    __shared__ uint4 smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int ltid = tid & 255;                    // index inside the role (4 waves each)
    const int cw = wave & 3;                       // consumer wave index (M quarter of the tile)
    const int r = lane & 31, h = lane >> 5;
    const T* __restrict__ xg = (const T*)p.x;
    const T* __restrict__ wg = (const T*)p.w;

    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int ntm = tiles_x * tiles_y * p.B, ntn = (p.Np + 63) >> 6;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    if (p.m_fastest) { tm = bid % ntm; tn = bid / ntm; } else { tn = bid % ntn; tm = bid / ntn; }
    const int x0 = (tm % tiles_x) * TW, y0 = ((tm / tiles_x) % tiles_y) * TH, b = tm / (tiles_x * tiles_y);
    const int n0 = tn * 64;
    const int nk = p.Kp / KC;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (producer) {
        // ------------------------------------------------------------------ producers: global -> registers -> LDS
        const int g4 = ltid & 3;
        int in_off[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int pix = (ltid >> 2) + 64 * j;
            const int hy = pix / HW_, hx = pix - hy * HW_;
            const int yy = y0 + hy - 1, xx = x0 + hx - 1;
            in_off[j] = (pix < NPIX && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W)
                            ? ((b * p.H + yy) * p.W + xx) * p.x_ldc + g4 * VEC : -1;
        }
        const int wco = ltid >> 2;
        const bool w_ok = n0 + wco < p.Np;
        uint4 rin[NJ], rw[NT];

#define WS_GLOAD(ks_)                                                                                             \
    do {                                                                                                          \
        const int k0_ = (ks_) * KC;                                                                               \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) rin[j] = ldg16(xg + in_off[j] + k0_, in_off[j] >= 0);      \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                            \
            rw[t] = ldg16(wg + ((long long)((ks_) * NT + t) * p.Np + n0 + wco) * KC + g4 * VEC, w_ok);            \
    } while (0)
#define WS_STORE(st_)                                                                                             \
    do {                                                                                                          \
        uint4* sm_ = smem + (st_) * STAGE;                                                                        \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                          \
            const int pix_ = (ltid >> 2) + 64 * j;                                                                \
            if constexpr (!SPLIT) {                                                                               \
                if (pix_ < NPIX) sm_[g4 * NPIXP + pix_] = rin[j];                                                 \
            } else if (pix_ < NPIX) {                                                                             \
                uint2 hi_, lo_;                                                                                   \
                split4(rin[j], hi_, lo_);                                                                         \
                char* b_ = reinterpret_cast<char*>(sm_) + ((g4 >> 1) * NPIXP + pix_) * 16 + 8 * (g4 & 1);         \
                *reinterpret_cast<uint2*>(b_) = hi_;                                                              \
                *reinterpret_cast<uint2*>(b_ + 2 * NPIXP * 16) = lo_;                                             \
            }                                                                                                     \
        }                                                                                                         \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) sm_[G::IN_SLOTS + (t * 4 + g4) * G::WG + wco] = rw[t];     \
    } while (0)

        WS_GLOAD(0);
        WS_STORE(0);
        if (nk > 1) WS_GLOAD(1);
        __syncthreads();                                   // stage 0 is ready
        for (int ks = 0; ks < nk; ++ks) {
            if (ks + 1 < nk) {
                WS_STORE((ks + 1) & 1);                    // consumers are reading stage ks&1
                if (ks + 2 < nk) WS_GLOAD(ks + 2);         // lands during the next period
            }
            __syncthreads();
        }
#undef WS_GLOAD
#undef WS_STORE
    } else {
        // ------------------------------------------------------------------ consumers: LDS fragments -> MFMA
        int apix[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = 64 * cw + 32 * mt + r;
            apix[mt] = (m / TW) * HW_ + (m % TW);
        }
        __syncthreads();                                   // stage 0 is ready
        for (int ks = 0; ks < nk; ++ks) {
            const uint4* sm = smem + (ks & 1) * STAGE;
            if constexpr (SPLIT) {
                uint4 f[2][8];      // [buffer][ah0, al0, ah1, al1, bh0, bl0, bh1, bl1]
#define WS_FRAG_S(t_, d_)                                                                                         \
    do {                                                                                                          \
        const int ib_ = ((t_) / 3) * HW_ + ((t_) % 3);                                                            \
        d_[0] = sm[ib_ + h * NPIXP + apix[0]];       d_[1] = sm[ib_ + (2 + h) * NPIXP + apix[0]];                 \
        d_[2] = sm[ib_ + h * NPIXP + apix[1]];       d_[3] = sm[ib_ + (2 + h) * NPIXP + apix[1]];                 \
        d_[4] = sm[G::IN_SLOTS + ((t_) * 4 + h) * G::WG + r];                                                     \
        d_[5] = sm[G::IN_SLOTS + ((t_) * 4 + 2 + h) * G::WG + r];                                                 \
        d_[6] = sm[G::IN_SLOTS + ((t_) * 4 + h) * G::WG + 32 + r];                                                \
        d_[7] = sm[G::IN_SLOTS + ((t_) * 4 + 2 + h) * G::WG + 32 + r];                                            \
    } while (0)
                WS_FRAG_S(0, f[0]);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);     // reads of step 0 lead; then R(s+1), M(s), ...
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (t + 1 < NT) WS_FRAG_S(t + 1, f[(t + 1) & 1]);
                    const uint4* c = f[t & 1];
                    mma_bf16(c[1], c[4], acc[0][0]); mma_bf16(c[0], c[5], acc[0][0]); mma_bf16(c[0], c[4], acc[0][0]);
                    mma_bf16(c[1], c[6], acc[0][1]); mma_bf16(c[0], c[7], acc[0][1]); mma_bf16(c[0], c[6], acc[0][1]);
                    mma_bf16(c[3], c[4], acc[1][0]); mma_bf16(c[2], c[5], acc[1][0]); mma_bf16(c[2], c[4], acc[1][0]);
                    mma_bf16(c[3], c[6], acc[1][1]); mma_bf16(c[2], c[7], acc[1][1]); mma_bf16(c[2], c[6], acc[1][1]);
                    if (t + 1 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
                }
#undef WS_FRAG_S
            } else {
                constexpr int NSTEP = 2 * NT;
                constexpr int NMF = sizeof(T) == 2 ? 4 : 16;
                uint4 f[2][4];      // [buffer][a0, a1, b0, b1]
#define WS_FRAG(s_, d_)                                                                                           \
    do {                                                                                                          \
        const int t_ = (s_) >> 1, g_ = ((s_) & 1) * 2 + h;                                                        \
        const int ib_ = (t_ / 3) * HW_ + (t_ % 3);                                                                \
        d_[0] = sm[ib_ + g_ * NPIXP + apix[0]];                                                                   \
        d_[1] = sm[ib_ + g_ * NPIXP + apix[1]];                                                                   \
        d_[2] = sm[G::IN_SLOTS + (t_ * 4 + g_) * G::WG + r];                                                      \
        d_[3] = sm[G::IN_SLOTS + (t_ * 4 + g_) * G::WG + 32 + r];                                                 \
    } while (0)
                WS_FRAG(0, f[0]);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);     // reads of step 0 lead; then R(s+1), M(s), ...
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) {
                    if (st + 1 < NSTEP) WS_FRAG(st + 1, f[(st + 1) & 1]);
                    const uint4* c = f[st & 1];
                    mma16<T>(c[0], c[2], acc[0][0]);
                    mma16<T>(c[0], c[3], acc[0][1]);
                    mma16<T>(c[1], c[2], acc[1][0]);
                    mma16<T>(c[1], c[3], acc[1][1]);
                    if (st + 1 < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
                }
#undef WS_FRAG
            }
            __syncthreads();                               // hand stage ks&1 back to the producers
        }
    }

    // ---------------------------------------------------------------------- epilogue (consumers work, everyone syncs)
    const bool cons = !producer;
    float bcol[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n0 + 32 * nt + r;
        bcol[nt] = (p.bias && n < p.Np) ? p.bias[n] : 0.f;
    }
    unsigned vmask = 0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = 64 * cw + 32 * mt + acc_row(e, h);
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
    float* ebuf = reinterpret_cast<float*>(smem);     // the last loop barrier released both stages
    if (p.stats) {
        float* sbuf = ebuf + 4 * 32 * 68;
        if (cons) {
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
            if (n0 + c < p.Np) atomicAdd(p.stats + ((size_t)(blockIdx.x % STAT_REPLICAS) * 2 + k) * p.Np + n0 + c, t);
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
    for (int mt = 0; mt < 2; ++mt) {
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
                const int m = 64 * cw + 32 * mt + row;
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
            if (n0 + c < p.Np) atomicAdd(p.bn_sums + ((size_t)(blockIdx.x % STAT_REPLICAS) * 5 + k) * p.Np + n0 + c, t);
        }
    }
}

template <typename T>
static int launch_ws_t(const IgemmParams& p, hipStream_t s) {
    const bool wide = p.W >= 32;
    const int TW = wide ? 32 : 16, TH = 256 / TW;
    const long long tiles = (long long)((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH) * p.B;
    const long long nblk = tiles * ((p.Np + 63) / 64);
    if (nblk <= 0 || nblk > 0x7fffffff) return clamd_fail("igemm_ws: grid out of range");
    if (wide) hipLaunchKernelGGL((igemm_ws_kernel<T, 32>), dim3((unsigned)nblk), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((igemm_ws_kernel<T, 16>), dim3((unsigned)nblk), dim3(512), 0, s, p);
    return clamd_check_launch("igemm_ws");
}

int launch_igemm_ws(const IgemmParams& p, int dtype, hipStream_t s) {
    if (dtype == CLAMD_BF16) return launch_ws_t<bf16_t>(p, s);
    if (dtype == CLAMD_F32) return launch_ws_t<float>(p, s);
    if (dtype == CLAMD_SPLIT) return launch_ws_t<split_t>(p, s);
    return clamd_fail("igemm_ws: bad dtype");
}

}  // namespace clamd
