// Winograd F(2x4,3x3), HALF-WIDTH workgroups: 32 Winograd tiles x 32 output channels, TWO workgroups resident per CU
// (round 5; models/unet.py:13-18,50-55,66-71 -- the 64/128-channel layers of the exact-fp32 path).
//
// wino24_kernel's workgroup (32 tiles x 64 channels, 192 accumulators, 120 KB of LDS, one wave per SIMD) spends a third of a
// 64-channel tile in a prologue and an epilogue that nothing overlaps.  Here a workgroup owns half the output channels: 96
// accumulators, <= 256 registers, 74 KB of LDS -- so two of them share a CU (two waves per SIMD) and one's halo wait, exchange,
// read-back and stores run under the other's K loop.  The price: the in-lane input transform (72 VALU per 8-channel chunk) now
// feeds 24 MFMAs instead of 48, and the input halo is staged once per 32 instead of per 64 output channels.
//
// Same decomposition otherwise -- wave w = Winograd row i of the vertical F(2,3), six j of the horizontal F(4,3) in-lane, the
// filters of clamd_wino24_pack, the same formulas and MFMA chains: activations BIT-IDENTICAL to clamd_conv3x3_winograd24.  The K
// loop is NOT software-pipelined inside the wave (fragments of chunk k are read, transformed and multiplied in iteration k; the
// other workgroup's waves fill the gaps) and keeps one staging register set: chunk k+1 is stored to the other LDS stage behind
// the MFMAs of chunk k, chunk k+2 requested, one barrier per chunk.
#include <string.h>
#include <algorithm>
#include "common.hip.h"
#include "clamd_internal.h"
#include "wino_common.hip.h"

namespace clamd {

constexpr int W24N_WG = 34;                                    // padded rows per (xi, group): 32 + 2, == 2 (mod 8)
constexpr int W24N_WT_SLOTS = 24 * 2 * W24N_WG;
constexpr int W24N_EXP = 36;                                   // row pitch (floats) of the epilogue exchange block

template <int TXN, bool RAGGED, bool CLS>
__global__ void __launch_bounds__(256, 2) wino24n_kernel(const WinoParams p) {
    constexpr int TYN = 32 / TXN;
    constexpr int PW = 4 * TXN, PH = 2 * TYN;
    constexpr int HW_ = PW + 2, HH_ = PH + 2, PIX = HW_ * HH_;
    constexpr int PIXMAX = (HH_ - 1) * HW_ + (HW_ - 1) + ((HH_ - 1) >> 1) + 1;
    constexpr int PIXP = PIXMAX + ((10 - PIXMAX % 8) % 8);
    constexpr int IN_SLOTS = 2 * PIXP, STAGE = IN_SLOTS + W24N_WT_SLOTS;
    constexpr int NJI = (2 * PIX + 255) / 256;
    constexpr int NJW = 6;                                                // 24 xi x 32 rows x 2 groups / 256
    constexpr int EXB = 4 * 4 * 32 * W24N_EXP;                            // floats of the exchange block [wave][q][tile][EXP]
    constexpr int LDS = (2 * STAGE * 16 > EXB * 4 ? 2 * STAGE * 16 : EXB * 4) / 16;
    static_assert(LDS * 16 <= 80 * 1024, "LDS budget: two workgroups per CU");
    __shared__ uint4 smem[LDS];

    // statistics rows: wino24_kernel's scheme with 32-channel slabs (thread (k, c < 32) owns word (k, 32 slab + c) of this
    // workgroup's row: it zeroes the words it later accumulates into)
    const bool per_wg_rows = p.stats != nullptr && gridDim.x < (unsigned)p.nblk;
    if (per_wg_rows && threadIdx.x < 128 && (threadIdx.x & 63) < 32)
        for (int n = threadIdx.x & 31; n < p.Np; n += 32) p.stats[((size_t)blockIdx.x * 2 + (threadIdx.x >> 6)) * p.Np + n] = 0.f;
    float racc = 0.f;
    int rslab = -1;
    auto flush_row = [&]() {          // threads (k, c < 32) only
        if (rslab >= 0 && rslab * 32 + (int)(threadIdx.x & 63) < p.Np) {
            float* dst = p.stats + ((size_t)blockIdx.x * 2 + (threadIdx.x >> 6)) * p.Np + rslab * 32 + (threadIdx.x & 63);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float old = __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst, old + racc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    float st1[4], st2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { st1[i] = 0.f; st2[i] = 0.f; }
    int cur_tn = -1, cur_tm = 0;
    auto fold_stats = [&]() {
        float* sb = reinterpret_cast<float*>(smem);                            // [wave][2][32]
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = st1[c], q = st2[c];
            a += __shfl_xor(a, 8); a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
            q += __shfl_xor(q, 8); q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
            if (lane < 8) { sb[(w * 2 + 0) * 32 + 4 * lane + c] = a; sb[(w * 2 + 1) * 32 + 4 * lane + c] = q; }
            st1[c] = 0.f; st2[c] = 0.f;
        }
        __syncthreads();
        if (threadIdx.x < 128 && (threadIdx.x & 63) < 32) {
            const int k = threadIdx.x >> 6, c = threadIdx.x & 63;
            const float t = sb[(0 * 2 + k) * 32 + c] + sb[(1 * 2 + k) * 32 + c] + sb[(2 * 2 + k) * 32 + c] + sb[(3 * 2 + k) * 32 + c];
            if (!per_wg_rows) {
                if (cur_tn * 32 + c < p.Np) p.stats[((size_t)cur_tm * 2 + k) * p.Np + cur_tn * 32 + c] = t;   // row = pixel tile
            } else {
                rslab = cur_tn; racc = t;
                flush_row();
            }
        }
        __syncthreads();
        cur_tn = -1;
    };

    for (int v = blockIdx.x; v < p.nblk; v += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                                        // per-tile re-derivation (hoisted constants would spill)
        const int lane = tid & 63, w = tid >> 6;
        const int r = lane & 31, h = lane >> 5;
        const int tiles_x = (p.W + PW - 1) / PW, tiles_y = (p.H + PH - 1) / PH;
        const int ntn = (p.Np + 31) >> 5;
        const int bid = xcd_remap(v, p.nblk);
        const int per_band = (p.nblk / ntn) * p.band;
        const int bnd = bid / per_band, rem = bid - bnd * per_band;
        const int tn = bnd * p.band + rem % p.band, tm = rem / p.band;
        const int x0 = (tm % tiles_x) * PW, y0 = ((tm / tiles_x) % tiles_y) * PH, b = tm / (tiles_x * tiles_y);
        const int n0 = tn * 32;
        const int nk = p.Kp >> 3;
        if (cur_tn >= 0 && (tn != cur_tn || !per_wg_rows)) fold_stats();      // workgroup-uniform

        // ---- staging descriptors ------------------------------------------------------------------------------------------
        const unsigned img = (unsigned)p.H * (unsigned)p.W * (unsigned)p.x_ldc * 4u;
        const __amdgpu_buffer_rsrc_t xrs = make_rsrc((const char*)p.x + (size_t)b * img, img);
        const __amdgpu_buffer_rsrc_t wrs = make_rsrc(p.w, (unsigned)(24u * p.Np * p.Kp * 4u));
        const __amdgpu_buffer_rsrc_t xrs_dead = make_rsrc(p.x, 0u), wrs_dead = make_rsrc(p.w, 0u);
        unsigned in_vo[NJI];
        int in_slot[NJI];
#pragma unroll
        for (int j = 0; j < NJI; ++j) {
            int piece = tid + 256 * j;                           // (pixel, 16-byte group); the last pass wraps
            if (piece >= 2 * PIX) piece -= 2 * PIX;
            const int g = piece & 1, pix = piece >> 1;
            const int hy = pix / HW_, hx = pix - hy * HW_;
            const int yy = y0 + hy - 1, xx = x0 + hx - 1;
            in_vo[j] = (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) ? (unsigned)(((yy * p.W + xx) * p.x_ldc + 4 * g) * 4) : BUF_OOB;
            in_slot[j] = g * PIXP + hy * HW_ + hx + (hy >> 1);
        }
        // filter slab of one K-chunk: 24 xi x 32 rows x 2 groups = 1536 pieces, piece = tid + 256 j: xi = (tid >> 6) + 4 j
        const int wg_ = tid & 1, wn_ = (tid >> 1) & 31, wxi0 = tid >> 6;
        const unsigned w_vo0 = n0 + wn_ < p.Np ? (unsigned)(((wxi0 * p.Np + n0 + wn_) * 8 + 4 * wg_) * 4) : BUF_OOB;
        const unsigned w_vstep = (unsigned)(4 * p.Np * 8 * 4);    // four xi further
        const unsigned w_chunk = (unsigned)(24 * p.Np * 8 * 4);   // bytes of one K-chunk
        const int w_slot0 = IN_SLOTS + (wxi0 * 2 + wg_) * W24N_WG + wn_;

        uint4 rin[NJI], rw[NJW];
        auto gload = [&](int k, bool live) {
            const unsigned so = (unsigned)(k * 8 * 4);
            const __amdgpu_buffer_rsrc_t xr = live ? xrs : xrs_dead, wr = live ? wrs : wrs_dead;
#pragma unroll
            for (int j = 0; j < NJI; ++j) rin[j] = buf_ld16(xr, in_vo[j], so);
#pragma unroll
            for (int j = 0; j < NJW; ++j) rw[j] = buf_ld16(wr, w_vo0, (unsigned)k * w_chunk + j * w_vstep);
        };
        auto lds_store = [&](int st) {
            uint4* sm = smem + st * STAGE;
#pragma unroll
            for (int j = 0; j < NJI; ++j) sm[in_slot[j]] = rin[j];
#pragma unroll
            for (int j = 0; j < NJW; ++j) sm[w_slot0 + j * 8 * W24N_WG] = rw[j];
        };

        // ---- fragment addressing (wino24_kernel's): wave w = vertical Winograd row i ------------------------------------------
        const int a1 = w == 0 ? 0 : 1, a2 = w == 3 ? 3 : 2;
        const float s2 = w == 1 ? 1.f : -1.f;
        const int ty = r / TXN, tx = r % TXN;
        const int pb = h * PIXP + (2 * ty) * HW_ + 4 * tx + ty;
        const int p1 = pb + a1 * HW_ + (a1 >> 1), p2 = pb + a2 * HW_ + (a2 >> 1);
        const int wb = IN_SLOTS + (6 * w * 2 + h) * W24N_WG + r;             // + j * 2 * WG

        f32x16 acc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

        gload(0, true);
        lds_store(0);
        gload(1, 1 < nk);
        __syncthreads();
        for (int k = 0; k < nk; ++k) {
            uint4 A[6], Bf[6];
            {
                const uint4* sm = smem + (k & 1) * STAGE;
                float4 t[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const uint4 u1 = sm[p1 + c], u2 = sm[p2 + c];
                    t[c].x = fmaf(s2, __uint_as_float(u2.x), __uint_as_float(u1.x));
                    t[c].y = fmaf(s2, __uint_as_float(u2.y), __uint_as_float(u1.y));
                    t[c].z = fmaf(s2, __uint_as_float(u2.z), __uint_as_float(u1.z));
                    t[c].w = fmaf(s2, __uint_as_float(u2.w), __uint_as_float(u1.w));
                }
                // B6^T, wino24_kernel's formulas (bit-identical V)
#define W24N_COL(m_)                                                                                              \
    do {                                                                                                          \
        const float pq_ = fmaf(-4.f, t[2].m_, t[4].m_), qq_ = fmaf(-4.f, t[1].m_, t[3].m_);                       \
        const float rr_ = t[4].m_ - t[2].m_, ss_ = t[3].m_ - t[1].m_;                                             \
        o0.m_ = fmaf(4.f, t[0].m_, fmaf(-5.f, t[2].m_, t[4].m_));                                                 \
        o1.m_ = pq_ + qq_; o2.m_ = pq_ - qq_;                                                                     \
        o3.m_ = fmaf(2.f, ss_, rr_); o4.m_ = fmaf(-2.f, ss_, rr_);                                                \
        o5.m_ = fmaf(4.f, t[1].m_, fmaf(-5.f, t[3].m_, t[5].m_));                                                 \
    } while (0)
                float4 o0, o1, o2, o3, o4, o5;
                W24N_COL(x); W24N_COL(y); W24N_COL(z); W24N_COL(w);
#undef W24N_COL
#define W24N_PK(v_) make_uint4(__float_as_uint((v_).x), __float_as_uint((v_).y), __float_as_uint((v_).z), __float_as_uint((v_).w))
                A[0] = W24N_PK(o0); A[1] = W24N_PK(o1); A[2] = W24N_PK(o2); A[3] = W24N_PK(o3); A[4] = W24N_PK(o4); A[5] = W24N_PK(o5);
#undef W24N_PK
#pragma unroll
                for (int j = 0; j < 6; ++j) Bf[j] = sm[wb + j * 2 * W24N_WG];
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) mma16<float>(A[j], Bf[j], acc[j]);
            lds_store((k + 1) & 1);                                           // chunk k+1 (zeros past the end: never read)
            gload(k + 2, k + 2 < nk);
            __syncthreads();
        }

        // ---- epilogue: wino24_kernel's for one 32-channel block ----------------------------------------------------------------
        float* const ex = reinterpret_cast<float*>(smem);
        const int tl = tid >> 3, ng = tid & 7;                                // reader: tile, 4-channel group
        const float relu_lo = p.relu ? 0.f : -__builtin_inff();
        const bool plain = !p.relu && !p.bias && !p.stats;
        const bool edge_tile = CLS && (y0 == 0 || x0 == 0 || y0 + PH >= p.H || x0 + PW >= p.W);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float m0 = acc[0][e], m1 = acc[1][e], m2 = acc[2][e], m3 = acc[3][e], m4 = acc[4][e], m5 = acc[5][e];
            const float sa = m1 + m2, sb = m1 - m2, sc = m3 + m4, sd = m3 - m4;
            const int row = acc_row(e, h);
            ex[((w * 4 + 0) * 32 + row) * W24N_EXP + r] = m0 + sa + sc;
            ex[((w * 4 + 1) * 32 + row) * W24N_EXP + r] = fmaf(2.f, sd, sb);
            ex[((w * 4 + 2) * 32 + row) * W24N_EXP + r] = fmaf(4.f, sc, sa);
            ex[((w * 4 + 3) * 32 + row) * W24N_EXP + r] = fmaf(8.f, sd, sb) + m5;
        }
        __syncthreads();
        const int oty = tl / TXN, otx = tl % TXN;
        {
            const int n = n0 + 4 * ng;
            float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias && n < p.Np) bias4 = *reinterpret_cast<const float4*>(p.bias + (CLS ? 4 * p.Np : 0) + n);   // 4: interior
            float4 bm1 = bias4;
            if (edge_tile && n < p.Np) {
                const int ya = y0 + 2 * (tl / TXN);
                bias4 = *reinterpret_cast<const float4*>(p.bias + ((ya == 0 ? 0 : (ya == p.H - 1 ? 6 : 3)) + 1) * p.Np + n);
                bm1 = *reinterpret_cast<const float4*>(p.bias + ((ya + 1 == p.H - 1 ? 6 : 3) + 1) * p.Np + n);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 R[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) R[i] = *reinterpret_cast<const float4*>(ex + ((i * 4 + q) * 32 + tl) * W24N_EXP + 4 * ng);
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    float4 o;
                    if (pp == 0) {
                        o.x = R[0].x + R[1].x + R[2].x; o.y = R[0].y + R[1].y + R[2].y;
                        o.z = R[0].z + R[1].z + R[2].z; o.w = R[0].w + R[1].w + R[2].w;
                    } else {
                        o.x = R[1].x - R[2].x - R[3].x; o.y = R[1].y - R[2].y - R[3].y;
                        o.z = R[1].z - R[2].z - R[3].z; o.w = R[1].w - R[2].w - R[3].w;
                    }
                    const int yy = y0 + 2 * oty + pp, xx = x0 + 4 * otx + q;
                    if (!plain) {
                        float4 b4 = pp ? bm1 : bias4;
                        if (edge_tile && n < p.Np && (q == 0 || q == 3) && (xx == 0 || xx == p.W - 1))
                            b4 = *reinterpret_cast<const float4*>(p.bias + border_class(yy, xx, p.H, p.W) * p.Np + n);
                        o.x = fmaxf(o.x + b4.x, relu_lo); o.y = fmaxf(o.y + b4.y, relu_lo);
                        o.z = fmaxf(o.z + b4.z, relu_lo); o.w = fmaxf(o.w + b4.w, relu_lo);
                    }
                    if (!RAGGED || (yy < p.H && xx < p.W && n < p.Np)) {
                        *reinterpret_cast<float4*>(p.y + (((size_t)b * p.H + yy) * p.W + xx) * p.y_ldc + n) = o;
                        if (!plain) {
                            st1[0] += o.x; st1[1] += o.y; st1[2] += o.z; st1[3] += o.w;
                            st2[0] = fmaf(o.x, o.x, st2[0]); st2[1] = fmaf(o.y, o.y, st2[1]);
                            st2[2] = fmaf(o.z, o.z, st2[2]); st2[3] = fmaf(o.w, o.w, st2[3]);
                        }
                    }
                }
            }
        }
        if (p.stats) { cur_tn = tn; cur_tm = tm; }
        __syncthreads();                                                   // exchange / statistics blocks are free again
    }
    if (cur_tn >= 0) fold_stats();
}

static inline void w24n_tile(int W, int& ph, int& pw) { if (W >= 32) { ph = 8; pw = 32; } else { ph = 16; pw = 16; } }
static long long w24n_tiles(int B, int H, int W) {
    int ph, pw;
    w24n_tile(W, ph, pw);
    return (long long)B * ((H + ph - 1) / ph) * ((W + pw - 1) / pw);
}

// rows of a half-width launch: one per pixel tile (one workgroup per tile), or one per workgroup of the persistent grid (two per CU)
long long clamd_winograd24_half_stat_rows(int B, int H, int W, int Cout_p, const clamd_tuning& tn) {
    const long long tiles = w24n_tiles(B, H, W), nblk = tiles * ((Cout_p + 31) / 32);
    const long long cap = 2LL * clamd_usable_cus(tn);
    return (tn.wino_persist && nblk > cap) ? cap : tiles;
}

int launch_wino24_half(WinoParams p, const clamd_tuning& tn, int stat_rows, hipStream_t stream) {
    int ph, pw;
    w24n_tile(p.W, ph, pw);
    const long long ntn = (p.Np + 31) / 32, tiles = w24n_tiles(p.B, p.H, p.W);
    if (tiles * ntn > 0x7fffffff) return clamd_fail("conv3x3_winograd24 (half-width): grid out of range");
    if (p.stats && stat_rows != clamd_winograd24_half_stat_rows(p.B, p.H, p.W, p.Np, tn))
        return clamd_fail("conv3x3_winograd24 (half-width): stat_rows does not match clamd_stat_rows(CLAMD_OP_CONV3X3_WINOGRAD24, ...)");
    // block order: the band of the full-width launch (output-channel slabs of 64 whose workgroups share one XCD's L2), in half slabs
    p.band = 1;
    if (p.Np % 64 == 0)
        p.band = 2 * wino_band(tiles, p.Np / 64, (double)p.B * p.H * p.W * p.Kp, 24.0 * p.Kp * p.Np, tn.wino_band);
    p.nblk = (int)(tiles * ntn);
    const unsigned grid = tn.wino_persist ? (unsigned)std::min<long long>(p.nblk, 2LL * clamd_usable_cus(tn)) : (unsigned)p.nblk;
    const bool ragged = (p.H % ph) != 0 || (p.W % pw) != 0 || (p.Np % 32) != 0;
#define W24N_LAUNCH(TXN_, RG_)                                                                                          \
    do {                                                                                                               \
        if (p.bias_classes) hipLaunchKernelGGL((wino24n_kernel<TXN_, RG_, true>), dim3(grid), dim3(256), 0, stream, p);  \
        else hipLaunchKernelGGL((wino24n_kernel<TXN_, RG_, false>), dim3(grid), dim3(256), 0, stream, p);               \
    } while (0)
    if (pw == 32) { if (ragged) W24N_LAUNCH(8, true); else W24N_LAUNCH(8, false); }
    else { if (ragged) W24N_LAUNCH(4, true); else W24N_LAUNCH(4, false); }
#undef W24N_LAUNCH
    return clamd_check_launch("conv3x3_winograd(F(2x4), half-width)");
}

}  // namespace clamd
