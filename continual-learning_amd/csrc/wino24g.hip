// Winograd F(2x4,3x3) with PRE-TRANSFORMED operands for the wide (>= 256-channel) 3x3 convolutions of the exact-fp32 path
// (models/unet.py:28-33,50-55: enc3/enc4/dec1/dec2/dec3; loss.backward(), trainer.py:175).
//
// wino24.hip forms B^T d B inside the K loop of EVERY output-slab workgroup (8-16x per tile on the 512/1024-channel layers)
// and wino24_wgrad.hip transforms both operands in its loop: 2-3.5 VALU and 0.5-2.3 LDS instructions per fp32 MFMA, each of
// which costs matrix-pipe time on gfx950.  Here the transforms run ONCE per tensor in HBM-bound
// kernels and the MFMA kernels read their operands from memory in fragment layout:
//
//   * wino24_xform_kernel: x [B,H,W,ldc] -> V [tile block][Kp/8][24 = 6i + j][2 = lane half][32 tiles][4 channels], the image
//     wino24_kernel's in-lane transform produces (same formulas, same row-2 sign convention), so the filters of
//     clamd_wino24_pack serve both kernels and the results are BIT-IDENTICAL to clamd_conv3x3_winograd24;
//   * wino24g_kernel: wino24_kernel's tile / wave decomposition (wave w = Winograd row i, six j in-lane, 32 tiles x 64 output
//     channels per workgroup, same epilogue, same statistics rows), but the K loop is 48 MFMAs + 18 buffer_load_dwordx4
//     straight into the MFMA operand registers: no LDS, no barrier, no VALU.  Nothing is shared between the four waves of a
//     workgroup inside the loop (each row i has its own slice of V and of the filters), so there is nothing to stage.  The
//     load stream runs NSET chunks ahead and is continuous across the tiles of the persistent loop (the last chunks of a
//     tile fetch the first chunks of the next one, whose latency then hides under the epilogue);
//   * weight gradient: dU[p] = Yt[p]^T V[p] for the 24 planes as a batched GEMM with K = Winograd tiles.  One transform
//     kernel writes Yt = A4 dY A6^T [24][tiles][Rp] (tiles in V's order); the x side is the forward image V itself, read in
//     place.  wino24g_wgrad_kernel gives every WAVE a 128 x 128 block of one plane: a lane loads 16 bytes = 4 channels of
//     one tile of each operand and issues the 4 x 4 outer product as 16 MFMAs (k = 2 tiles), 256 accumulator registers,
//     2 loads per 16 MFMAs, no LDS, no barrier; the four waves of a workgroup take the 2 x 2 blocks of a 256 x 256 block so
//     that operand panels are shared in L1.  Split-K over tile ranges into fp32 slabs [split][24][Rp][Cp], fixed-order
//     reduce with G4^T . G6 (deterministic).
#include <string.h>
#include <algorithm>
#include "common.hip.h"
#include "clamd_internal.h"
#include "wino_common.hip.h"

namespace clamd {

// ---------------------------------------------------------------------------------------------------------------------
// input transform (forward / data gradient)
// ---------------------------------------------------------------------------------------------------------------------
struct W24XformParams {
    const float* x; int x_ldc;
    const float* scale; const float* shift;      // optional per-channel affine applied on load (a BatchNorm folded into the transform)
    float* v;
    int B, H, W, Kp;
};

// B6^T of one transformed row t[0..5] (float4 = 4 channels): the formulas of wino24_kernel's W24_COL, so that V is
// bit-identical to the in-kernel transform
#define W24G_COLS(t_, o_)                                                                                          \
    do {                                                                                                           \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                                         \
            const float t0_ = t_[0][e_], t1_ = t_[1][e_], t2_ = t_[2][e_], t3_ = t_[3][e_], t4_ = t_[4][e_], t5_ = t_[5][e_]; \
            const float pq_ = fmaf(-4.f, t2_, t4_), qq_ = fmaf(-4.f, t1_, t3_);                                    \
            const float rr_ = t4_ - t2_, ss_ = t3_ - t1_;                                                          \
            o_[0][e_] = fmaf(4.f, t0_, fmaf(-5.f, t2_, t4_));                                                      \
            o_[1][e_] = pq_ + qq_; o_[2][e_] = pq_ - qq_;                                                          \
            o_[3][e_] = fmaf(2.f, ss_, rr_); o_[4][e_] = fmaf(-2.f, ss_, rr_);                                     \
            o_[5][e_] = fmaf(4.f, t1_, fmaf(-5.f, t3_, t5_));                                                      \
        }                                                                                                          \
    } while (0)

// One workgroup = one tile block x 32 channels (four 8-channel chunks, one per wave).  The halo of the block is staged in LDS
// with whole 128-byte lines per pixel (8 lanes x 16 bytes: every global load instruction reads full lines; the overlap with the
// neighbouring blocks is served by L2), zero padding -- and the optional affine y * scale + shift, i.e. the BatchNorm in front
// of this convolution, applied to in-image pixels only: padding stays zero AFTER the affine, as nn.Conv2d pads (models/unet.py:15-16)
// -- materialised at staging.  Then lane (h = lane >> 5, r = lane & 31) of wave w reads the 4 x 6 patch of tile r, channels
// 4h..4h+3 of chunk w, and writes its 16 bytes of each of the 24 planes: every store instruction of a wave is one contiguous
// 1-KB unit [h][r][4], exactly what one buffer_load_dwordx4 of wino24g_kernel fetches.
template <int TXN>
__global__ void __launch_bounds__(256) wino24_xform_kernel(const W24XformParams p) {
    PASS_PRIO();      // a pass of the critical chain beside the second stream's MFMA kernels (elementwise.hip)
    constexpr int TYN = 32 / TXN, PW = 4 * TXN, PH = 2 * TYN;
    constexpr int HW_ = PW + 2, HH_ = PH + 2, PIX = HW_ * HH_;
    constexpr int PITCH = 9;                                           // 16-byte slots per pixel: 8 used + 1 (spreads the tiles over the banks)
    constexpr int NJ = (PIX * 8 + 255) / 256;
    __shared__ uint4 sm[PIX * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nk = p.Kp >> 3;
    const int tiles_x = (p.W + PW - 1) / PW, tiles_y = (p.H + PH - 1) / PH;
    const int ntm = tiles_x * tiles_y * p.B;
    // neighbouring tile blocks (shared halo rows / columns) get neighbouring ids inside one XCD's range
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int kg = bid / ntm, tm = bid - kg * ntm;
    const int x0 = (tm % tiles_x) * PW, y0 = ((tm / tiles_x) % tiles_y) * PH, b = tm / (tiles_x * tiles_y);
    const unsigned img = (unsigned)p.H * (unsigned)p.W * (unsigned)p.x_ldc * 4u;
    const __amdgpu_buffer_rsrc_t xrs = make_rsrc((const char*)p.x + (size_t)b * img, img);
    {
        const int g = tid & 7;                                         // this thread's 4-channel group of the 32 channels, fixed
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
    const int ty = r / TXN, tx = r % TXN;
    const uint4* const src = sm + ((2 * ty) * HW_ + 4 * tx) * PITCH + 2 * w + h;
    float4 d[4][6];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const uint4 u = src[(a * HW_ + c) * PITCH];
            d[a][c] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
        }
    float* const dst = p.v + (((size_t)tm * nk + kc) * 24) * 256 + h * 128 + r * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // row i of the vertical F(2,3): t = d[a1] + s2 * d[a2] (row 2 carries the opposite sign; it is in the packed filters)
        const int a1 = i == 0 ? 0 : 1, a2 = i == 3 ? 3 : 2;
        const float s2 = i == 1 ? 1.f : -1.f;
        float t[6][4], o[6][4];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            t[c][0] = fmaf(s2, d[a2][c].x, d[a1][c].x); t[c][1] = fmaf(s2, d[a2][c].y, d[a1][c].y);
            t[c][2] = fmaf(s2, d[a2][c].z, d[a1][c].z); t[c][3] = fmaf(s2, d[a2][c].w, d[a1][c].w);
        }
        W24G_COLS(t, o);
#pragma unroll
        for (int j = 0; j < 6; ++j)
            *reinterpret_cast<float4*>(dst + (6 * i + j) * 256) = make_float4(o[j][0], o[j][1], o[j][2], o[j][3]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward / data-gradient kernel on the transformed input
constexpr int W24G_EXP = 36;                                  // row pitch (floats) of the epilogue exchange block

template <int TXN, bool RAGGED, int NSET>
__global__ void __launch_bounds__(256, 1) wino24g_kernel(const WinoParams p) {
    constexpr int TYN = 32 / TXN;
    constexpr int PW = 4 * TXN, PH = 2 * TYN;
    constexpr int EXB = 4 * 4 * 32 * W24G_EXP;                            // floats of one nt exchange block [wave][q][tile][EXP]
    constexpr int LDS = 2 * EXB * 4 / 16;
    static_assert(LDS * 16 <= 160 * 1024, "LDS budget");
    __shared__ uint4 smem[LDS];

    // statistics rows: exactly wino24_kernel's scheme (per-workgroup rows on the persistent grid, registers across the tiles
    // of one output slab, one fold per slab) -- the two kernels are interchangeable for bn_finalize
    float* const rows_base = p.stats;
    constexpr int NKR = 2;
    const bool per_wg_rows = rows_base != nullptr && gridDim.x < (unsigned)p.nblk;
    if (per_wg_rows)
        for (int k = threadIdx.x >> 6; k < NKR; k += 4)
            for (int n = threadIdx.x & 63; n < p.Np; n += 64) rows_base[((size_t)blockIdx.x * NKR + k) * p.Np + n] = 0.f;
    float racc = 0.f;
    int rslab = -1;
    auto flush_row = [&]() {
        if (rslab >= 0 && rslab * 64 + (int)(threadIdx.x & 63) < p.Np) {
            float* dst = rows_base + ((size_t)blockIdx.x * NKR + (threadIdx.x >> 6)) * p.Np + rslab * 64 + (threadIdx.x & 63);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float old = __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst, old + racc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    float st1[2][4], st2[2][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { st1[i >> 2][i & 3] = 0.f; st2[i >> 2][i & 3] = 0.f; }
    int cur_tn = -1, cur_tm = 0;
    auto fold_stats = [&]() {
        float* sb = reinterpret_cast<float*>(smem);                            // [wave][2][64]
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = st1[nt][c], q = st2[nt][c];
                a += __shfl_xor(a, 8); a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
                q += __shfl_xor(q, 8); q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
                if (lane < 8) { sb[(w * 2 + 0) * 64 + 32 * nt + 4 * lane + c] = a; sb[(w * 2 + 1) * 64 + 32 * nt + 4 * lane + c] = q; }
                st1[nt][c] = 0.f; st2[nt][c] = 0.f;
            }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int k = threadIdx.x >> 6, c = threadIdx.x & 63;
            const float t = sb[(0 * 2 + k) * 64 + c] + sb[(1 * 2 + k) * 64 + c] + sb[(2 * 2 + k) * 64 + c] + sb[(3 * 2 + k) * 64 + c];
            if (!per_wg_rows) {
                if (cur_tn * 64 + c < p.Np) rows_base[((size_t)cur_tm * NKR + k) * p.Np + cur_tn * 64 + c] = t;
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
    const int nk = p.Kp >> 3;                                               // nk % NSET == 0, nk >= 2 NSET (checked on entry)
    const int per_band = (p.nblk / ntn) * p.band;
    // one descriptor each for all of V and all of the filters: everything tile-dependent is a scalar offset
    const __amdgpu_buffer_rsrc_t vrs = make_rsrc(p.x, (unsigned)((size_t)(p.nblk / ntn) * nk * 24 * 1024));
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(p.w, (unsigned)(24u * p.Np * p.Kp * 4u));
    const __amdgpu_buffer_rsrc_t vrs_dead = make_rsrc(p.x, 0u), wrs_dead = make_rsrc(p.w, 0u);
    const unsigned ustride = (unsigned)p.Np * 32u;                          // bytes of one (chunk, plane) of the filters
    auto decode = [&](int v, int& tn, int& tm) {
        const int bid = xcd_remap(v, p.nblk);
        const int bnd = bid / per_band, rem = bid - bnd * per_band;
        tn = bnd * p.band + rem % p.band; tm = rem / p.band;
    };

    uint4 A[NSET][6], Bq[NSET][6][2];
    bool pre = false;
    for (int v = blockIdx.x; v < p.nblk; v += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                                        // per-tile re-derivation (see wino.hip: hoisted constants spill)
        const int lane = tid & 63;
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int r = lane & 31, h = lane >> 5;
        int tn, tm;
        decode(v, tn, tm);
        const int x0 = (tm % tiles_x) * PW, y0 = ((tm / tiles_x) % tiles_y) * PH, b = tm / (tiles_x * tiles_y);
        const int n0 = tn * 64;
        if (cur_tn >= 0 && (tn != cur_tn || !per_wg_rows)) fold_stats();

        // the tile after this one (the load stream does not stop at a tile boundary)
        const int vn = v + (int)gridDim.x;
        const bool has_next = vn < p.nblk;
        int tnn, tmn;
        decode(has_next ? vn : v, tnn, tmn);

        const unsigned a_vo = (unsigned)(h * 512 + r * 16);
        const unsigned b_vo = (unsigned)((r * 8 + 4 * h) * 4);
        const unsigned plane0 = (unsigned)(w * 6);
        const unsigned vb_cur = (unsigned)tm * (unsigned)nk * 24u * 1024u, vb_nxt = (unsigned)tmn * (unsigned)nk * 24u * 1024u;
        const unsigned ub_cur = (unsigned)n0 * 32u, ub_nxt = (unsigned)tnn * 64u * 32u;

        // all of chunk k (of the tile whose bases are vb / ub) into register set s
        auto load_j = [&](int s, int j, __amdgpu_buffer_rsrc_t vr, __amdgpu_buffer_rsrc_t ur, unsigned vb, unsigned ub, int k) {
            const unsigned pl = (unsigned)k * 24u + plane0 + (unsigned)j;
            A[s][j] = buf_ld16(vr, a_vo, vb + pl * 1024u);
            Bq[s][j][0] = buf_ld16(ur, b_vo, ub + pl * ustride);
            Bq[s][j][1] = buf_ld16(ur, b_vo, ub + pl * ustride + 1024u);
        };

        f32x16 acc[6][2];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

        if (!pre) {                                                           // first tile of this workgroup
#pragma unroll
            for (int s = 0; s < NSET; ++s)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    load_j(s, j, vrs, wrs, vb_cur, ub_cur, s);
                    asm volatile("" ::: "memory");                             // ring order (see wino24g_wgrad_kernel)
                }
        }
        // chunks [0, nk - NSET): every set is refilled from THIS tile
        for (int k = 0; k < nk - NSET; k += NSET) {
#pragma unroll
            for (int s = 0; s < NSET; ++s)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    mma16<float>(A[s][j], Bq[s][j][0], acc[j][0]);
                    mma16<float>(A[s][j], Bq[s][j][1], acc[j][1]);
                    load_j(s, j, vrs, wrs, vb_cur, ub_cur, k + s + NSET);
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
                }
        }
        // last NSET chunks: the sets are refilled with chunks 0..NSET-1 of the next tile (an empty descriptor behind the last
        // tile: zeros, no traffic -- the registers are redefined on every path)
        {
            const __amdgpu_buffer_rsrc_t vr = has_next ? vrs : vrs_dead, ur = has_next ? wrs : wrs_dead;
#pragma unroll
            for (int s = 0; s < NSET; ++s)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    mma16<float>(A[s][j], Bq[s][j][0], acc[j][0]);
                    mma16<float>(A[s][j], Bq[s][j][1], acc[j][1]);
                    load_j(s, j, vr, ur, vb_nxt, ub_nxt, s);
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
                }
            pre = has_next;
        }

        // ---- epilogue: wino24_kernel's.  Y = A4^T M A6: A6^T in-lane (j -> q), A4^T across the four waves through LDS ----------
        const int tl = tid >> 3, ng = tid & 7;                                // reader: tile, 4-channel group
        float* const ex = reinterpret_cast<float*>(smem);                                // reader: tile, 4-channel group
        const float relu_lo = p.relu ? 0.f : -__builtin_inff();
        const bool plain = !p.relu && !p.bias && !p.stats;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float* const exb = ex + nt * EXB;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float m0 = acc[0][nt][e], m1 = acc[1][nt][e], m2 = acc[2][nt][e], m3 = acc[3][nt][e], m4 = acc[4][nt][e],
                            m5 = acc[5][nt][e];
                const float sa = m1 + m2, sb = m1 - m2, sc = m3 + m4, sd = m3 - m4;
                const int row = acc_row(e, h);
                exb[((w * 4 + 0) * 32 + row) * W24G_EXP + r] = m0 + sa + sc;
                exb[((w * 4 + 1) * 32 + row) * W24G_EXP + r] = fmaf(2.f, sd, sb);
                exb[((w * 4 + 2) * 32 + row) * W24G_EXP + r] = fmaf(4.f, sc, sa);
                exb[((w * 4 + 3) * 32 + row) * W24G_EXP + r] = fmaf(8.f, sd, sb) + m5;
            }
        }
        __syncthreads();
        const int oty = tl / TXN, otx = tl % TXN;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float* const exb = ex + nt * EXB;
            const int n = n0 + 32 * nt + 4 * ng;
            float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias && n < p.Np) bias4 = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 R[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) R[i] = *reinterpret_cast<const float4*>(exb + ((i * 4 + q) * 32 + tl) * W24G_EXP + 4 * ng);
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
                    if (!plain) {
                        o.x = fmaxf(o.x + bias4.x, relu_lo); o.y = fmaxf(o.y + bias4.y, relu_lo);
                        o.z = fmaxf(o.z + bias4.z, relu_lo); o.w = fmaxf(o.w + bias4.w, relu_lo);
                    }
                    const int yy = y0 + 2 * oty + pp, xx = x0 + 4 * otx + q;
                    if (!RAGGED || (yy < p.H && xx < p.W)) {
                        *reinterpret_cast<float4*>(p.y + (((size_t)b * p.H + yy) * p.W + xx) * p.y_ldc + n) = o;
                        if (!plain) {
                            st1[nt][0] += o.x; st1[nt][1] += o.y; st1[nt][2] += o.z; st1[nt][3] += o.w;
                            st2[nt][0] = fmaf(o.x, o.x, st2[nt][0]); st2[nt][1] = fmaf(o.y, o.y, st2[nt][1]);
                            st2[nt][2] = fmaf(o.z, o.z, st2[nt][2]); st2[nt][3] = fmaf(o.w, o.w, st2[nt][3]);
                        }
                    }
                }
            }
        }
        if (rows_base) { cur_tn = tn; cur_tm = tm; }
        __syncthreads();                                                   // exchange / statistics blocks are free again
    }
    if (cur_tn >= 0) fold_stats();
}

// ---------------------------------------------------------------------------------------------------------------------
// hybrid kernel for the narrow layers (64 / 128 channels, levels 0-1): the input transform stays in the kernel (3x the
// activation bytes through HBM would make these layers HBM-bound), but the FILTERS -- 48 of the 59 KB a workgroup moves per
// chunk in wino24_kernel: global -> registers -> LDS (12 ds_write_b128 per thread) -> fragments (12 ds_read_b128 per wave) --
// go straight into the MFMA operand registers as in wino24g_kernel: each wave needs only the slice of its own row i, nothing
// is shared.  Per chunk and wave: 48 MFMAs, 72 transform VALU, 12 LDS reads + 3 LDS writes + 15 buffer loads (wino24_kernel:
// 103 VALU incl. 25 register copies, 24 LDS reads, 15 LDS writes, 15 loads).  The chunk loop is unrolled by two: the transformed
// input alternates between two register sets (no copies), the filter sets are refilled two chunks ahead behind their MFMAs and the
// stream runs on into the next tile.  Same V formulas, same MFMA chains, same epilogue: bit-identical to wino24_kernel.
// ---------------------------------------------------------------------------------------------------------------------
template <int TXN, bool RAGGED, bool CLS = false>      // CLS: bias from a border-class table (WinoParams::bias_classes)
__global__ void __launch_bounds__(256, 1) wino24h_kernel(const WinoParams p) {
    constexpr int TYN = 32 / TXN;
    constexpr int PW = 4 * TXN, PH = 2 * TYN;
    constexpr int HW_ = PW + 2, HH_ = PH + 2, PIX = HW_ * HH_;            // input halo
    constexpr int PIXMAX = (HH_ - 1) * HW_ + (HW_ - 1) + ((HH_ - 1) >> 1) + 1;   // slots incl. the row skew (see wino24.hip)
    constexpr int PIXP = PIXMAX + ((10 - PIXMAX % 8) % 8);                // == 2 (mod 8): conflict-free staging stores
    constexpr int IN_SLOTS = 2 * PIXP;
    constexpr int NJI = (2 * PIX + 255) / 256;                            // input staging loads per thread
    constexpr int EXB = 4 * 4 * 32 * W24G_EXP;
    constexpr int LDS = 2 * EXB * 4 / 16;                                 // the exchange blocks alias the two input stages
    static_assert(2 * IN_SLOTS <= LDS && LDS * 16 <= 160 * 1024, "LDS budget");
    __shared__ uint4 smem[LDS];

    float* const rows_base = p.stats;
    constexpr int NKR = 2;
    const bool per_wg_rows = rows_base != nullptr && gridDim.x < (unsigned)p.nblk;
    if (per_wg_rows)
        for (int k = threadIdx.x >> 6; k < NKR; k += 4)
            for (int n = threadIdx.x & 63; n < p.Np; n += 64) rows_base[((size_t)blockIdx.x * NKR + k) * p.Np + n] = 0.f;
    float racc = 0.f;
    int rslab = -1;
    auto flush_row = [&]() {
        if (rslab >= 0 && rslab * 64 + (int)(threadIdx.x & 63) < p.Np) {
            float* dst = rows_base + ((size_t)blockIdx.x * NKR + (threadIdx.x >> 6)) * p.Np + rslab * 64 + (threadIdx.x & 63);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float old = __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst, old + racc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    float st1[2][4], st2[2][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { st1[i >> 2][i & 3] = 0.f; st2[i >> 2][i & 3] = 0.f; }
    int cur_tn = -1, cur_tm = 0;
    auto fold_stats = [&]() {
        float* sb = reinterpret_cast<float*>(smem);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = st1[nt][c], q = st2[nt][c];
                a += __shfl_xor(a, 8); a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
                q += __shfl_xor(q, 8); q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
                if (lane < 8) { sb[(w * 2 + 0) * 64 + 32 * nt + 4 * lane + c] = a; sb[(w * 2 + 1) * 64 + 32 * nt + 4 * lane + c] = q; }
                st1[nt][c] = 0.f; st2[nt][c] = 0.f;
            }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int k = threadIdx.x >> 6, c = threadIdx.x & 63;
            const float t = sb[(0 * 2 + k) * 64 + c] + sb[(1 * 2 + k) * 64 + c] + sb[(2 * 2 + k) * 64 + c] + sb[(3 * 2 + k) * 64 + c];
            if (!per_wg_rows) {
                if (cur_tn * 64 + c < p.Np) rows_base[((size_t)cur_tm * NKR + k) * p.Np + cur_tn * 64 + c] = t;
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
    const int nk = p.Kp >> 3;                                               // even, >= 4 (checked on entry)
    const int per_band = (p.nblk / ntn) * p.band;
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(p.w, (unsigned)(24u * p.Np * p.Kp * 4u));
    const __amdgpu_buffer_rsrc_t wrs_dead = make_rsrc(p.w, 0u);
    const unsigned img = (unsigned)p.H * (unsigned)p.W * (unsigned)p.x_ldc * 4u;
    const __amdgpu_buffer_rsrc_t xrs_dead = make_rsrc(p.x, 0u);
    const unsigned ustride = (unsigned)p.Np * 32u;
    auto decode = [&](int v, int& tn, int& tm) {
        const int bid = xcd_remap(v, p.nblk);
        const int bnd = bid / per_band, rem = bid - bnd * per_band;
        tn = bnd * p.band + rem % p.band; tm = rem / p.band;
    };

    uint4 Bq[6][2];                            // filter fragments of ONE chunk, refilled with the next chunk's behind their MFMAs (the
    //                                            filters of these narrow layers are <= 3 MB: L2 hits, one chunk = 3k cycles ahead is plenty)
    uint4 pi0[NJI], pi1[NJI];                  // input chunks 0 and 1 of the tile about to start (requested under the epilogue)
    bool pre = false;
    for (int v = blockIdx.x; v < p.nblk; v += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
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

        // ---- input staging (wino24_kernel's halo image) -------------------------------------------------------------------
        const __amdgpu_buffer_rsrc_t xrs = make_rsrc((const char*)p.x + (size_t)b * img, img);
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
        uint4 rin[NJI];
        auto gload_in = [&](int k, bool live, uint4 (&ri)[NJI]) {
            const __amdgpu_buffer_rsrc_t xr = live ? xrs : xrs_dead;
#pragma unroll
            for (int j = 0; j < NJI; ++j) ri[j] = buf_ld16(xr, in_vo[j], (unsigned)(k * 32));
        };
        auto lds_store_in = [&](int st, const uint4 (&ri)[NJI]) {
            uint4* sm = smem + st * IN_SLOTS;
#pragma unroll
            for (int j = 0; j < NJI; ++j) sm[in_slot[j]] = ri[j];
        };

        // ---- filters: straight into the fragment registers ------------------------------------------------------------------
        const unsigned b_vo = (unsigned)((r * 8 + 4 * h) * 4);
        const unsigned plane0 = (unsigned)(w * 6);
        const unsigned ub_cur = (unsigned)n0 * 32u, ub_nxt = (unsigned)tnn * 64u * 32u;
        auto load_b = [&](int j, __amdgpu_buffer_rsrc_t ur, unsigned ub, int k) {
            const unsigned pl = (unsigned)k * 24u + plane0 + (unsigned)j;
            Bq[j][0] = buf_ld16(ur, b_vo, ub + pl * ustride);
            Bq[j][1] = buf_ld16(ur, b_vo, ub + pl * ustride + 1024u);
        };

        // ---- fragments of the input: wave w = vertical Winograd row i (wino24_kernel) ----------------------------------------
        const int a1 = w == 0 ? 0 : 1, a2 = w == 3 ? 3 : 2;
        const float s2 = w == 1 ? 1.f : -1.f;
        const int ty = r / TXN, tx = r % TXN;
        const int pb = h * PIXP + (2 * ty) * HW_ + 4 * tx + ty;
        const int p1 = pb + a1 * HW_ + (a1 >> 1), p2 = pb + a2 * HW_ + (a2 >> 1);
        auto frags = [&](int st, uint4 (&A)[6]) {
            const uint4* sm = smem + st * IN_SLOTS;
            float t[6][4], o[6][4];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const uint4 u1 = sm[p1 + c], u2 = sm[p2 + c];
                t[c][0] = fmaf(s2, __uint_as_float(u2.x), __uint_as_float(u1.x));
                t[c][1] = fmaf(s2, __uint_as_float(u2.y), __uint_as_float(u1.y));
                t[c][2] = fmaf(s2, __uint_as_float(u2.z), __uint_as_float(u1.z));
                t[c][3] = fmaf(s2, __uint_as_float(u2.w), __uint_as_float(u1.w));
            }
            W24G_COLS(t, o);
#pragma unroll
            for (int j = 0; j < 6; ++j)
                A[j] = make_uint4(__float_as_uint(o[j][0]), __float_as_uint(o[j][1]), __float_as_uint(o[j][2]), __float_as_uint(o[j][3]));
        };

        f32x16 acc[6][2];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

        if (!pre) {                                                           // first tile of this workgroup
            gload_in(0, true, pi0);
            gload_in(1, true, pi1);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                load_b(j, wrs, ub_cur, 0);
                asm volatile("" ::: "memory");                                 // ring order (see wino24g_wgrad_kernel)
            }
        }
        gload_in(2, 2 < nk, rin);
        lds_store_in(0, pi0);
        lds_store_in(1, pi1);
        __syncthreads();
        uint4 A0[6], A1[6];
        frags(0, A0);
        __syncthreads();                                                      // stage 0 is free again

        // one chunk: 48 MFMAs on (Acur, Bq); the transformed input of chunk k+1 goes to Anext; chunk k+2 of the input is stored
        // over chunk k's stage, chunk k+3 requested; Bq[j] is refilled with chunk k+1 (or the next tile's chunk 0) behind its MFMAs
#define W24H_CHUNK(k_, Acur_, Anext_, ur_, ub_, kb_)                                                                   \
    do {                                                                                                               \
        frags(((k_) + 1) & 1, Anext_);                                        /* visible since the last barrier */      \
        _Pragma("unroll") for (int j_ = 0; j_ < 6; ++j_) {                                                             \
            mma16<float>(Acur_[j_], Bq[j_][0], acc[j_][0]);                                                            \
            mma16<float>(Acur_[j_], Bq[j_][1], acc[j_][1]);                                                            \
            load_b(j_, ur_, ub_, kb_);                                                                                 \
        }                                                                                                              \
        lds_store_in((k_) & 1, rin);                                          /* chunk k+2 over chunk k's stage */       \
        gload_in((k_) + 3, (k_) + 3 < nk, rin);                                                                        \
        sched_mfma_slots<48, 12, 20, 20 + NJI, 24, 24 + NJI, 2, 4, 1>();                                               \
        __syncthreads();                                                                                               \
    } while (0)
        for (int k = 0; k < nk - 2; k += 2) {
            W24H_CHUNK(k, A0, A1, wrs, ub_cur, k + 1);
            W24H_CHUNK(k + 1, A1, A0, wrs, ub_cur, k + 2);
        }
        {
            W24H_CHUNK(nk - 2, A0, A1, wrs, ub_cur, nk - 1);
            const __amdgpu_buffer_rsrc_t ur = has_next ? wrs : wrs_dead;
            W24H_CHUNK(nk - 1, A1, A0, ur, ub_nxt, 0);
            pre = has_next;
        }
#undef W24H_CHUNK
        {   // input chunks 0 and 1 of the next tile: requested here, they land under the epilogue (empty descriptor behind the
            // last tile: the registers are redefined on every path)
            const int x0n = (tmn % tiles_x) * PW, y0n = ((tmn / tiles_x) % tiles_y) * PH, bn = tmn / (tiles_x * tiles_y);
            const __amdgpu_buffer_rsrc_t xrn = has_next ? make_rsrc((const char*)p.x + (size_t)bn * img, img) : xrs_dead;
#pragma unroll
            for (int j = 0; j < NJI; ++j) {
                int piece = tid + 256 * j;
                if (piece >= 2 * PIX) piece -= 2 * PIX;
                const int g = piece & 1, pix = piece >> 1;
                const int hy = pix / HW_, hx = pix - hy * HW_;
                const int yy = y0n + hy - 1, xx = x0n + hx - 1;
                const unsigned vo = (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) ? (unsigned)(((yy * p.W + xx) * p.x_ldc + 4 * g) * 4) : BUF_OOB;
                pi0[j] = buf_ld16(xrn, vo, 0u);
                pi1[j] = buf_ld16(xrn, vo, 32u);
            }
        }

        // ---- epilogue: wino24_kernel's -----------------------------------------------------------------------------------------
        const int tl = tid >> 3, ng = tid & 7;                                // reader: tile, 4-channel group
        float* const ex = reinterpret_cast<float*>(smem);
        const float relu_lo = p.relu ? 0.f : -__builtin_inff();
        const bool plain = !p.relu && !p.bias && !p.stats;
        const bool edge_tile = CLS && (y0 == 0 || x0 == 0 || y0 + PH >= p.H || x0 + PW >= p.W);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float* const exb = ex + nt * EXB;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float m0 = acc[0][nt][e], m1 = acc[1][nt][e], m2 = acc[2][nt][e], m3 = acc[3][nt][e], m4 = acc[4][nt][e],
                            m5 = acc[5][nt][e];
                const float sa = m1 + m2, sb = m1 - m2, sc = m3 + m4, sd = m3 - m4;
                const int row = acc_row(e, h);
                exb[((w * 4 + 0) * 32 + row) * W24G_EXP + r] = m0 + sa + sc;
                exb[((w * 4 + 1) * 32 + row) * W24G_EXP + r] = fmaf(2.f, sd, sb);
                exb[((w * 4 + 2) * 32 + row) * W24G_EXP + r] = fmaf(4.f, sc, sa);
                exb[((w * 4 + 3) * 32 + row) * W24G_EXP + r] = fmaf(8.f, sd, sb) + m5;
            }
        }
        __syncthreads();
        const int oty = tl / TXN, otx = tl % TXN;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float* const exb = ex + nt * EXB;
            const int n = n0 + 32 * nt + 4 * ng;
            float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias && n < p.Np) bias4 = *reinterpret_cast<const float4*>(p.bias + (CLS ? 4 * p.Np : 0) + n);   // 4: interior
            float4 bm1 = bias4;                 // bias of the thread's two output rows (pp = 0: bias4, pp = 1: bm1), interior column class
            if (edge_tile && n < p.Np) {        // tile-uniform branch: a folded BatchNorm's shift term depends on which taps read padding
                const int ya = y0 + 2 * (tl / TXN);
                bias4 = *reinterpret_cast<const float4*>(p.bias + ((ya == 0 ? 0 : (ya == p.H - 1 ? 6 : 3)) + 1) * p.Np + n);
                bm1 = *reinterpret_cast<const float4*>(p.bias + ((ya + 1 == p.H - 1 ? 6 : 3) + 1) * p.Np + n);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 R[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) R[i] = *reinterpret_cast<const float4*>(exb + ((i * 4 + q) * 32 + tl) * W24G_EXP + 4 * ng);
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
                        if (edge_tile && n < p.Np && (q == 0 || q == 3) && (xx == 0 || xx == p.W - 1))      // first / last pixel of an image row
                            b4 = *reinterpret_cast<const float4*>(p.bias + border_class(yy, xx, p.H, p.W) * p.Np + n);
                        o.x = fmaxf(o.x + b4.x, relu_lo); o.y = fmaxf(o.y + b4.y, relu_lo);
                        o.z = fmaxf(o.z + b4.z, relu_lo); o.w = fmaxf(o.w + b4.w, relu_lo);
                    }
                    if (!RAGGED || (yy < p.H && xx < p.W)) {
                        *reinterpret_cast<float4*>(p.y + (((size_t)b * p.H + yy) * p.W + xx) * p.y_ldc + n) = o;
                        if (!plain) {
                            st1[nt][0] += o.x; st1[nt][1] += o.y; st1[nt][2] += o.z; st1[nt][3] += o.w;
                            st2[nt][0] = fmaf(o.x, o.x, st2[nt][0]); st2[nt][1] = fmaf(o.y, o.y, st2[nt][1]);
                            st2[nt][2] = fmaf(o.z, o.z, st2[nt][2]); st2[nt][3] = fmaf(o.w, o.w, st2[nt][3]);
                        }
                    }
                }
            }
        }
        if (rows_base) { cur_tn = tn; cur_tm = tm; }
        __syncthreads();                                                   // exchange / statistics blocks are free again
    }
    if (cur_tn >= 0) fold_stats();
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient: gradient-side operand transform
// ---------------------------------------------------------------------------------------------------------------------
struct W24WgXformParams {
    const float* src; int ldc;       // gz [B,H,W,ldc]
    float* dst;                      // Yt [24][Tp][Rp]
    int B, H, W, Rp, Tp;
};

// Yt = A4 dY A6^T of every 2 x 4 gradient tile, tiles in the order of the forward image V (t = 32 * tile block + tile in block),
// so that V itself is the other operand of the weight-gradient GEMM.  thread = (tile, 4-channel group), channel groups fastest:
// reads and writes are contiguous along the channels.  Tiles of a ragged block that lie outside the image read zeros.
template <int TXN>
__global__ void __launch_bounds__(256) wino24g_wgrad_xform_kernel(const W24WgXformParams p) {
    SIDE_PRIO();
    constexpr int TYN = 32 / TXN, PW = 4 * TXN, PH = 2 * TYN;
    const int ng = p.Rp >> 2;
    const long long total = (long long)p.Tp * ng;
    const int tiles_x = (p.W + PW - 1) / PW, tiles_y = (p.H + PH - 1) / PH;
    const unsigned img = (unsigned)p.H * (unsigned)p.W * (unsigned)p.ldc * 4u;
    const size_t ps = (size_t)p.Tp * p.Rp;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int g = (int)(idx % ng);
        const int t = (int)(idx / ng);
        const int tm = t >> 5, r = t & 31;
        const int x0 = (tm % tiles_x) * PW + 4 * (r % TXN), y0 = ((tm / tiles_x) % tiles_y) * PH + 2 * (r / TXN), b = tm / (tiles_x * tiles_y);
        float* const dst = p.dst + (size_t)t * p.Rp + 4 * g;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc((const char*)p.src + (size_t)b * img, img);
        float gq[2][4][4];                                             // [row][column][channel]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int yy = y0 + a, xx = x0 + q;
                const uint4 u = buf_ld16(rs, (yy < p.H && xx < p.W) ? (unsigned)(((yy * p.W + xx) * p.ldc + 4 * g) * 4) : BUF_OOB, 0u);
                gq[a][q][0] = __uint_as_float(u.x); gq[a][q][1] = __uint_as_float(u.y);
                gq[a][q][2] = __uint_as_float(u.z); gq[a][q][3] = __uint_as_float(u.w);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // A4 = [[1,0],[1,1],[1,-1],[0,-1]]
            float z[4][4], o[6][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    z[q][e] = i == 0 ? gq[0][q][e] : (i == 1 ? gq[0][q][e] + gq[1][q][e] : (i == 2 ? gq[0][q][e] - gq[1][q][e] : -gq[1][q][e]));
            // A6 = [[1,0,0,0],[1,1,1,1],[1,-1,1,-1],[1,2,4,8],[1,-2,4,-8],[0,0,0,1]]
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sa = z[0][e] + z[2][e], sb = z[1][e] + z[3][e], sc = fmaf(4.f, z[2][e], z[0][e]), sd = fmaf(4.f, z[3][e], z[1][e]);
                o[0][e] = z[0][e]; o[1][e] = sa + sb; o[2][e] = sa - sb;
                o[3][e] = fmaf(2.f, sd, sc); o[4][e] = fmaf(-2.f, sd, sc); o[5][e] = z[3][e];
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<float4*>(dst + (6 * i + j) * ps) = make_float4(o[j][0], o[j][1], o[j][2], o[j][3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient: batched GEMM over the 24 planes, K = tiles
// ---------------------------------------------------------------------------------------------------------------------
struct W24WgGemmParams {
    const float* yt;                      // [24][Tp][Rp]
    const float* v;                       // forward image of the convolution input: [Tp/32][Cp/8][24][2][32][4]
    float* partial;                       // [nsplit][npl][Rp][Cp]
    int Rp, Cp, Tp, nsplit, tiles_per_split;
    int npl;                              // planes: 24 (F(2x4)) or 36 (F(4x4), wino44g.hip: the same GEMM on its operands)
};

constexpr int W24G_D = 8;                 // k-steps (of two tiles) the load stream runs ahead

// Every WAVE owns a 128 x 128 block of one plane: lane (i = lane & 31, h = lane >> 5) loads 16 bytes = 4 channels of tile
// t + h of each operand and the wave issues the 4 x 4 outer product as 16 MFMAs (k = the two tiles): MFMA (m, n) accumulates rows
// r0 + 4i + m, columns c0 + 4j + n.  The x side comes straight from the forward image V: the 4 channels c0 + 4i.. of tile t are
// the 16 bytes at [t >> 5][(c0 + 4i) / 8][plane][i & 1][t & 31] -- one 128-byte line per lane pair, whose other seven tiles are
// the next three k-steps' (L1).  Two loads per 16 MFMAs, running W24G_D steps ahead; no LDS, no barrier, no VALU in the loop.
__global__ void __launch_bounds__(256, 1) wino24g_wgrad_kernel(const W24WgGemmParams p) {
    int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i32 = lane & 31, h = lane >> 5;
    const int rb_n = p.Rp >> 8, cb_n = p.Cp >> 8;
    // (row block, column block) fastest: the workgroups of one XCD's id range share the operand panels of a (plane, split)
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int cb = bid % cb_n; bid /= cb_n;
    const int rb = bid % rb_n; bid /= rb_n;
    const int split = bid % p.nsplit;
    const int pl = bid / p.nsplit;
    const int r0 = rb * 256 + (w >> 1) * 128, c0 = cb * 256 + (w & 1) * 128;
    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(p.Tp, t_begin + p.tiles_per_split);
    const int nsteps = max(t_end - t_begin, 0) >> 1;                          // Tp and tiles_per_split are even

    const int nk = p.Cp >> 3;
    const unsigned a_bytes = (unsigned)p.Tp * (unsigned)p.Rp * 4u;
    const __amdgpu_buffer_rsrc_t ars = make_rsrc((const char*)p.yt + (size_t)pl * a_bytes, a_bytes);
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(p.v, (unsigned)((size_t)(p.Tp >> 5) * nk * p.npl * 1024));
    const __amdgpu_buffer_rsrc_t ars_dead = make_rsrc(p.yt, 0u), brs_dead = make_rsrc(p.v, 0u);
    const unsigned a_vo = (unsigned)((h * p.Rp + r0 + 4 * i32) * 4);
    const unsigned b_vo = (unsigned)(((c0 >> 3) + (i32 >> 1)) * p.npl * 1024 + (i32 & 1) * 512 + h * 16);
    const unsigned a_step = (unsigned)p.Rp * 8u;                              // two tiles
    const unsigned a_so0 = (unsigned)t_begin * (unsigned)p.Rp * 4u;
    const unsigned b_blk = (unsigned)nk * (unsigned)p.npl * 1024u, b_pl = (unsigned)pl * 1024u;

    f32x16 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

    uint4 a[W24G_D], b[W24G_D];
    auto load = [&](int d, int s) {                 // step s (wave-uniform) into set d; past the end of the split: zeros, no traffic
        const bool live = s < nsteps;
        const unsigned t = (unsigned)(t_begin + 2 * s);
        a[d] = buf_ld16(live ? ars : ars_dead, a_vo, a_so0 + (unsigned)s * a_step);
        b[d] = buf_ld16(live ? brs : brs_dead, b_vo, (t >> 5) * b_blk + b_pl + (t & 31u) * 16u);
    };
#pragma unroll
    for (int d = 0; d < W24G_D; ++d) {
        load(d, d);
        // keeps the prologue loads in ring order: hipcc otherwise reorders them, the loop header then has to wait for the
        // YOUNGEST of them on the entry path, and the merged wait at the top of every iteration becomes vmcnt(0)
        asm volatile("" ::: "memory");
    }
    for (int s0 = 0; s0 < nsteps; s0 += W24G_D) {
#pragma unroll
        for (int d = 0; d < W24G_D; ++d) {
            const float am[4] = {__uint_as_float(a[d].x), __uint_as_float(a[d].y), __uint_as_float(a[d].z), __uint_as_float(a[d].w)};
            const float bn[4] = {__uint_as_float(b[d].x), __uint_as_float(b[d].y), __uint_as_float(b[d].z), __uint_as_float(b[d].w)};
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(am[m], bn[n], acc[m][n], 0, 0, 0);
            load(d, s0 + d + W24G_D);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        }
    }

    // slab [split][plane][Rp][Cp]: MFMA (m, n) holds rows r0 + 4 i + m, columns c0 + 4 j + n; lane j stores its four n
    float* const out = p.partial + (((size_t)split * p.npl + pl) * p.Rp) * (size_t)p.Cp;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = r0 + 4 * acc_row(e, h) + m;
            *reinterpret_cast<float4*>(out + (size_t)row * p.Cp + c0 + 4 * i32) = make_float4(acc[m][0][e], acc[m][1][e], acc[m][2][e], acc[m][3][e]);
        }
}

// out[rl][cl][3][3] = G4^T (sum_s dU_s) G6.  A thread owns four consecutive (r, c) pairs (16-byte loads of all 24 planes); the
// splits are dealt to PHS lane groups of a wave and combined with two fixed shuffle steps: the summation order is fixed
// (deterministic).  Plane row i = 2 carries the forward image's sign convention (row 2 of B4^T negated) and is flipped here.
struct W24GReduceParams {
    const float* partial; float* out;
    int nsplit, Rp, Cp, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p;
};

template <int PHS>
__global__ void __launch_bounds__(256) wino24g_wgrad_reduce_kernel(const W24GReduceParams p) {
    SIDE_PRIO();
    constexpr int QW = 64 / PHS;                                       // quads per wave
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = lane % QW, ph = lane / QW;
    const long long nquad = (long long)p.Rp * p.Cp / 4;
    const size_t plane_sz = (size_t)p.Rp * p.Cp, split_sz = plane_sz * 24;
    for (long long base = ((long long)blockIdx.x * 4 + wv) * QW; base < nquad; base += (long long)gridDim.x * 4 * QW) {
        const long long quad = base + q;
        const bool ok = quad < nquad;
        float4 s[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) s[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok)
            for (int k = ph; k < p.nsplit; k += PHS) {
                const float* src = p.partial + (size_t)k * split_sz + (size_t)quad * 4;
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)i * plane_sz);
                    s[i].x += v.x; s[i].y += v.y; s[i].z += v.z; s[i].w += v.w;
                }
            }
        if constexpr (PHS > 1) {
#pragma unroll
            for (int i = 0; i < 24; ++i) {
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
                float U[4][6];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const float4 v = s[6 * i + j];
                        const float x = e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w));
                        U[i][j] = i == 2 ? -x : x;
                    }
                // rows: t = G4^T U, G4 = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
                float t[3][6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    t[0][j] = U[0][j] + 0.5f * (U[1][j] + U[2][j]);
                    t[1][j] = 0.5f * (U[1][j] - U[2][j]);
                    t[2][j] = U[3][j] + 0.5f * (U[1][j] + U[2][j]);
                }
                float* o = p.out + ((size_t)rl * p.C + cl) * 9;
                // columns: dg = t G6, G6 = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float s12 = t[a][1] + t[a][2], d12 = t[a][2] - t[a][1], s34 = t[a][3] + t[a][4], d34 = t[a][3] - t[a][4];
                    o[a * 3 + 0] = 0.25f * t[a][0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
                    o[a * 3 + 1] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
                    o[a * 3 + 2] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + t[a][5];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same plane GEMM with the k-steps of ALL (plane, 256 x 256 block) items dealt evenly to the workgroups ("stream-K", round 5).
// The split-K plan above launches items x splits workgroups in whole rounds of the chip: 36 planes x 8 blocks = 288 items on 256 CUs
// are two rounds for 1.125 rounds of work, and every split adds a 256-KB slab to write and to re-read.  Here workgroup w owns the
// global k-step range [w Q, (w + 1) Q) of the items laid end to end (S steps each): it finishes the item it starts in, runs whole
// items, and stops inside one -- every workgroup does the same number of steps, an item gets floor(last / Q) - floor(first / Q) + 1
// partial slabs (slot = w - the first workgroup that touches it), and the reduce adds an item's slots in slot order: the map is
// static, so the result is bit-reproducible.
struct W24WgSkParams {
    const float* yt;                      // [npl][Tp][Rp]
    const float* v;                       // [Tp/32][Cp/8][npl][2][32][4]
    float* partial;                       // [slot][npl][Rp][Cp]
    int Rp, Cp, Tp, npl;
    int S, Q, nbp;                        // k-steps (two tiles) per item, per workgroup; items
};

__global__ void __launch_bounds__(256, 1) wino24g_wgrad_sk_kernel(const W24WgSkParams p) {
    int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i32 = lane & 31, h = lane >> 5;
    const int rb_n = p.Rp >> 8, cb_n = p.Cp >> 8;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const long long total = (long long)p.nbp * p.S;
    long long g = (long long)wg * p.Q;
    const long long g_end = min(g + (long long)p.Q, total);
    const int nk = p.Cp >> 3;
    const unsigned a_bytes = (unsigned)p.Tp * (unsigned)p.Rp * 4u;
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(p.v, (unsigned)((size_t)(p.Tp >> 5) * nk * p.npl * 1024));
    const __amdgpu_buffer_rsrc_t ars_dead = make_rsrc(p.yt, 0u), brs_dead = make_rsrc(p.v, 0u);
    const unsigned a_step = (unsigned)p.Rp * 8u;                              // two tiles
    const unsigned b_blk = (unsigned)nk * (unsigned)p.npl * 1024u;
    while (g < g_end) {                                                       // workgroup-uniform
        const int b = (int)(g / p.S);
        const int kb = (int)(g - (long long)b * p.S);
        const int ke = (int)min((long long)p.S, (long long)kb + (g_end - g));
        // (row block, column block) fastest: neighbouring workgroups share operand panels of one plane
        const int cb = b % cb_n, rb = (b / cb_n) % rb_n, pl = b / (cb_n * rb_n);
        const int r0 = rb * 256 + (w >> 1) * 128, c0 = cb * 256 + (w & 1) * 128;
        const __amdgpu_buffer_rsrc_t ars = make_rsrc((const char*)p.yt + (size_t)pl * a_bytes, a_bytes);
        const unsigned a_vo = (unsigned)((h * p.Rp + r0 + 4 * i32) * 4);
        const unsigned b_vo = (unsigned)(((c0 >> 3) + (i32 >> 1)) * p.npl * 1024 + (i32 & 1) * 512 + h * 16);
        const unsigned b_pl = (unsigned)pl * 1024u;

        f32x16 acc[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

        uint4 a[W24G_D], bq[W24G_D];
        auto load = [&](int d, int s) {                 // absolute step s of this item into set d; past the end of the range: zeros, no traffic
            const bool live = s < ke;
            const unsigned t = (unsigned)(2 * s);
            a[d] = buf_ld16(live ? ars : ars_dead, a_vo, (unsigned)s * a_step);
            bq[d] = buf_ld16(live ? brs : brs_dead, b_vo, (t >> 5) * b_blk + b_pl + (t & 31u) * 16u);
        };
#pragma unroll
        for (int d = 0; d < W24G_D; ++d) {
            load(d, kb + d);
            asm volatile("" ::: "memory");                                   // ring order (see wino24g_wgrad_kernel)
        }
        for (int s0 = kb; s0 < ke; s0 += W24G_D) {
#pragma unroll
            for (int d = 0; d < W24G_D; ++d) {
                const float am[4] = {__uint_as_float(a[d].x), __uint_as_float(a[d].y), __uint_as_float(a[d].z), __uint_as_float(a[d].w)};
                const float bn[4] = {__uint_as_float(bq[d].x), __uint_as_float(bq[d].y), __uint_as_float(bq[d].z), __uint_as_float(bq[d].w)};
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(am[m], bn[n], acc[m][n], 0, 0, 0);
                load(d, s0 + d + W24G_D);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            }
        }
        const int slot = wg - (int)(((long long)b * p.S) / p.Q);
        float* const out = p.partial + (((size_t)slot * p.npl + pl) * p.Rp) * (size_t)p.Cp;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = r0 + 4 * acc_row(e, h) + m;
                *reinterpret_cast<float4*>(out + (size_t)row * p.Cp + c0 + 4 * i32) = make_float4(acc[m][0][e], acc[m][1][e], acc[m][2][e], acc[m][3][e]);
            }
        g += ke - kb;
    }
}

// The same stream-K GEMM with the WAVE as the unit of work, for channel counts that are multiples of 128 but not of 256 (round 5: the
// 128- / 256-channel layers at 128 x 128 and 64 x 64 -- dec4, enc2.block.4, enc3.block.1 -- whose weight gradients ran the in-kernel-transform
// kernel at 0.42-0.50 of the pipe).  An item is a (plane, 128 x 128 block) wave tile; wave (4 wg + w) owns the global k-step range
// [(4 wg + w) Q, ...) of the items laid end to end.  Nothing is shared between the waves of a workgroup (there never was a barrier in this
// kernel), a wave's two loads per 16 MFMAs are 8 B per cycle and CU: panel sharing in L1 was never what it lived on.
__global__ void __launch_bounds__(256, 1) wino24g_wgrad_skw_kernel(const W24WgSkParams p) {
    int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i32 = lane & 31, h = lane >> 5;
    const int rb_n = p.Rp >> 7, cb_n = p.Cp >> 7;
    const int wvg = xcd_remap(blockIdx.x, gridDim.x) * 4 + w;
    const long long total = (long long)p.nbp * p.S;
    long long g = (long long)wvg * p.Q;
    const long long g_end = min(g + (long long)p.Q, total);
    const int nk = p.Cp >> 3;
    const unsigned a_bytes = (unsigned)p.Tp * (unsigned)p.Rp * 4u;
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(p.v, (unsigned)((size_t)(p.Tp >> 5) * nk * p.npl * 1024));
    const __amdgpu_buffer_rsrc_t ars_dead = make_rsrc(p.yt, 0u), brs_dead = make_rsrc(p.v, 0u);
    const unsigned a_step = (unsigned)p.Rp * 8u;                              // two tiles
    const unsigned b_blk = (unsigned)nk * (unsigned)p.npl * 1024u;
    while (g < g_end) {                                                       // wave-uniform
        const int b = (int)(g / p.S);
        const int kb = (int)(g - (long long)b * p.S);
        const int ke = (int)min((long long)p.S, (long long)kb + (g_end - g));
        const int cb = b % cb_n, rb = (b / cb_n) % rb_n, pl = b / (cb_n * rb_n);
        const int r0 = rb * 128, c0 = cb * 128;
        const __amdgpu_buffer_rsrc_t ars = make_rsrc((const char*)p.yt + (size_t)pl * a_bytes, a_bytes);
        const unsigned a_vo = (unsigned)((h * p.Rp + r0 + 4 * i32) * 4);
        const unsigned b_vo = (unsigned)(((c0 >> 3) + (i32 >> 1)) * p.npl * 1024 + (i32 & 1) * 512 + h * 16);
        const unsigned b_pl = (unsigned)pl * 1024u;

        f32x16 acc[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

        uint4 a[W24G_D], bq[W24G_D];
        auto load = [&](int d, int s) {
            const bool live = s < ke;
            const unsigned t = (unsigned)(2 * s);
            a[d] = buf_ld16(live ? ars : ars_dead, a_vo, (unsigned)s * a_step);
            bq[d] = buf_ld16(live ? brs : brs_dead, b_vo, (t >> 5) * b_blk + b_pl + (t & 31u) * 16u);
        };
#pragma unroll
        for (int d = 0; d < W24G_D; ++d) {
            load(d, kb + d);
            asm volatile("" ::: "memory");
        }
        for (int s0 = kb; s0 < ke; s0 += W24G_D) {
#pragma unroll
            for (int d = 0; d < W24G_D; ++d) {
                const float am[4] = {__uint_as_float(a[d].x), __uint_as_float(a[d].y), __uint_as_float(a[d].z), __uint_as_float(a[d].w)};
                const float bn[4] = {__uint_as_float(bq[d].x), __uint_as_float(bq[d].y), __uint_as_float(bq[d].z), __uint_as_float(bq[d].w)};
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(am[m], bn[n], acc[m][n], 0, 0, 0);
                load(d, s0 + d + W24G_D);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            }
        }
        const int slot = wvg - (int)(((long long)b * p.S) / p.Q);
        float* const out = p.partial + (((size_t)slot * p.npl + pl) * p.Rp) * (size_t)p.Cp;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = r0 + 4 * acc_row(e, h) + m;
                *reinterpret_cast<float4*>(out + (size_t)row * p.Cp + c0 + 4 * i32) = make_float4(acc[m][0][e], acc[m][1][e], acc[m][2][e], acc[m][3][e]);
            }
        g += ke - kb;
    }
}

// first reduce stage of the wave-level plan (an item there has tens of slots): thread = (plane, four consecutive (r, c) pairs) adds the
// item's slots in slot order into reduced[plane][Rp][Cp]; wino_sk_reduce_kernel (S = Q = 1: one slot per item) then applies G^T . G
struct W24SkwSumParams {
    const float* partial; float* reduced;
    int Rp, Cp, npl, S, Q;
};

__global__ void __launch_bounds__(256) wino_skw_sum_kernel(const W24SkwSumParams p) {
    SIDE_PRIO();
    const long long nquad = (long long)p.Rp * p.Cp / 4;
    const size_t plane_sz = (size_t)p.Rp * p.Cp, slot_sz = plane_sz * p.npl;
    const int rb_n = p.Rp >> 7, cb_n = p.Cp >> 7;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < nquad * p.npl; idx += (long long)gridDim.x * 256) {
        const int pl = (int)(idx / nquad);
        const long long quad = idx - (long long)pl * nquad;
        const int rp = (int)((quad * 4) / p.Cp), cp0 = (int)((quad * 4) % p.Cp);
        const long long b = ((long long)pl * rb_n + (rp >> 7)) * cb_n + (cp0 >> 7);
        const int first = (int)((b * p.S) / p.Q), last = (int)(((b + 1) * p.S - 1) / p.Q);
        const float* src = p.partial + (size_t)pl * plane_sz + (size_t)quad * 4;
        float4 a = *reinterpret_cast<const float4*>(src);
        for (int k = 1; k <= last - first; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)k * slot_sz);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        *reinterpret_cast<float4*>(p.reduced + (size_t)pl * plane_sz + (size_t)quad * 4) = a;
    }
}

// out[rl][cl][3][3] = G^T (sum of an item's slots, in slot order) G for the stream-K slabs: NPL = 24: G4^T . G6 with the sign of plane row
// 2 (wino24g_wgrad_reduce_kernel), NPL = 36: G6^T . G6.  A thread owns four consecutive (r, c) pairs of all NPL planes.
struct W24SkReduceParams {
    const float* partial; float* out;
    int Rp, Cp, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p;
    int S, Q;
};

__device__ inline void wsk_gt6(const float (&t)[6], float (&o)[3]) {
    const float s12 = t[1] + t[2], d12 = t[2] - t[1], s34 = t[3] + t[4], d34 = t[3] - t[4];
    o[0] = 0.25f * t[0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
    o[1] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
    o[2] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + t[5];
}

template <int NPL>
__global__ void __launch_bounds__(256) wino_sk_reduce_kernel(const W24SkReduceParams p) {
    SIDE_PRIO();
    constexpr int NI = NPL / 6;
    const long long nquad = (long long)p.Rp * p.Cp / 4;
    const size_t plane_sz = (size_t)p.Rp * p.Cp, slot_sz = plane_sz * NPL;
    const int rb_n = p.Rp >> 8, cb_n = p.Cp >> 8;
    for (long long quad = (long long)blockIdx.x * 256 + threadIdx.x; quad < nquad; quad += (long long)gridDim.x * 256) {
        const int rp = (int)((quad * 4) / p.Cp), cp0 = (int)((quad * 4) % p.Cp);
        const int blk = (rp >> 8) * cb_n + (cp0 >> 8);
        float4 s[NPL];
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const long long b = (long long)i * rb_n * cb_n + blk;
            const int first = (int)((b * p.S) / p.Q), last = (int)(((b + 1) * p.S - 1) / p.Q);
            const float* src = p.partial + (size_t)i * plane_sz + (size_t)quad * 4;
            float4 a = *reinterpret_cast<const float4*>(src);
            for (int k = 1; k <= last - first; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(src + (size_t)k * slot_sz);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            s[i] = a;
        }
        const int rl = wn_phys2log(rp, p.r_seg0, p.r_seg0p, p.R);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cl = wn_phys2log(cp0 + e, p.c_seg0, p.c_seg0p, p.C);
            if (rl < 0 || cl < 0) continue;
            float t[3][6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float col[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const float4 v = s[6 * i + j];
                    col[i] = e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w));
                }
                if constexpr (NPL == 24) {          // rows: G4^T U, plane row 2 carries the forward image's sign convention
                    const float u2 = -col[2];
                    t[0][j] = col[0] + 0.5f * (col[1] + u2);
                    t[1][j] = 0.5f * (col[1] - u2);
                    t[2][j] = col[3] + 0.5f * (col[1] + u2);
                } else {
                    float o3[3];
                    const float c6[6] = {col[0], col[1], col[2], col[3], col[4], col[5]};
                    wsk_gt6(c6, o3);
                    t[0][j] = o3[0]; t[1][j] = o3[1]; t[2][j] = o3[2];
                }
            }
            float* o = p.out + ((size_t)rl * p.C + cl) * 9;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float o3[3];
                wsk_gt6(t[a], o3);
                o[a * 3 + 0] = o3[0]; o[a * 3 + 1] = o3[1]; o[a * 3 + 2] = o3[2];
            }
        }
    }
}

// workgroup tile of the forward kernels: 8 x 32 pixels, or 16 x 16 for images narrower than 32 (as wino24.hip)
static inline void w24g_tile(int W, int& ph, int& pw) { if (W >= 32) { ph = 8; pw = 32; } else { ph = 16; pw = 16; } }
static long long w24g_tiles(int B, int H, int W) {
    int ph, pw;
    w24g_tile(W, ph, pw);
    return (long long)B * ((H + ph - 1) / ph) * ((W + pw - 1) / pw);
}

}  // namespace clamd

using namespace clamd;

// Split-K plan of the weight-gradient GEMM: workgroups = 24 planes x (Rp/256) x (Cp/256) x nsplit.  A workgroup costs
// (tiles / 2) k-steps of 16 fp32 MFMAs (1024 cycles) plus ~28k cycles of prologue latency and slab stores (256 KB per
// workgroup; measured, tools/wino24g_ab.py), the chip runs ceil(workgroups / CUs) rounds of them, and the reduce pass reads
// every slab once: pick the split count with the smallest modelled time.  Tiles per split: a multiple of 2 W24G_D (whole
// passes of the prefetch ring).
namespace clamd {
long long w24g_wg_max_split(long long Tp, int Rp, int Cp, int planes) {
    const long long nb = (long long)planes * (Rp / 256) * (Cp / 256);
    return std::max<long long>(1, std::min<long long>((3LL * clamd_num_cus()) / nb, Tp / (2 * W24G_D)));
}
int w24g_wg_plan_planes(long long Tp, int Rp, int Cp, int planes, const clamd_tuning& tn, int* per_out) {
    const int quant = 2 * W24G_D;
    const int nb = planes * (Rp / 256) * (Cp / 256);
    const int cus = clamd_usable_cus(tn);
    const long long max_split = std::max<long long>(1, std::min<long long>((3LL * clamd_num_cus()) / nb, Tp / quant));
    double best = 0;
    long long best_ns = 1, best_per = Tp;
    for (long long ns = 1; ns <= max_split; ++ns) {
        long long per = (Tp + ns - 1) / ns;
        per = (per + quant - 1) / quant * quant;
        const long long n = (Tp + per - 1) / per;                           // splits that are not empty
        if (n != ns) continue;
        const double rounds = (double)((nb * n + cus - 1) / cus);
        const double cycles = rounds * (per / 2 * 1024.0 + 28000.0);
        const double t = cycles / 2.3e9 + (double)nb * n * 262144.0 / 4.5e12;
        if (best == 0 || t < best) { best = t; best_ns = n; best_per = per; }
    }
    if (per_out) *per_out = (int)best_per;
    return (int)best_ns;
}
int launch_w24g_wgrad_gemm(const float* yt, const float* v, float* partial, int Rp, int Cp, long long Tp, int nsplit, int per, int planes, hipStream_t s) {
    W24WgGemmParams p{yt, v, partial, Rp, Cp, (int)Tp, nsplit, per, planes};
    hipLaunchKernelGGL(wino24g_wgrad_kernel, dim3((unsigned)(planes * nsplit * (Rp / 256) * (Cp / 256))), dim3(256), 0, s, p);
    return clamd_check_launch("wgrad_winograd_pre (plane GEMM)");
}
// stream-K plan: k-steps per workgroup (a multiple of the prefetch ring) for `cus` workgroups; slots an item can get
void w24g_sk_plan(long long Tp, int Rp, int Cp, int planes, int cus, int* S, int* Q, int* nbp, int* nwg, int* max_slots) {
    *S = (int)(Tp / 2);
    *nbp = planes * (Rp / 256) * (Cp / 256);
    const long long total = (long long)*nbp * *S;
    long long q = (total + cus - 1) / cus;
    q = (q + W24G_D - 1) / W24G_D * W24G_D;
    *Q = (int)q;
    *nwg = (int)((total + q - 1) / q);
    *max_slots = (int)((*S + q - 1) / q) + 1;
}
size_t w24g_sk_workspace_bytes(long long Tp, int Rp, int Cp, int planes) {
    // the slot count is largest on the whole chip (smallest Q): size for that; a reserve (fewer workgroups) needs no more
    int S, Q, nbp, nwg, ms;
    w24g_sk_plan(Tp, Rp, Cp, planes, clamd_num_cus(), &S, &Q, &nbp, &nwg, &ms);
    return (size_t)ms * planes * Rp * Cp * sizeof(float);
}
int launch_w24g_wgrad_sk(const float* yt, const float* v, float* workspace, size_t ws_bytes, float* out, long long Tp, int Rp, int Cp, int planes,
                         int R, int C, int r_seg0, int r_seg0p, int c_seg0, int c_seg0p, const clamd_tuning& tn, hipStream_t s) {
    int S, Q, nbp, nwg, ms;
    w24g_sk_plan(Tp, Rp, Cp, planes, clamd_usable_cus(tn), &S, &Q, &nbp, &nwg, &ms);
    if ((size_t)ms * planes * Rp * Cp * sizeof(float) > ws_bytes) return clamd_fail("wgrad_winograd_pre: workspace too small");
    W24WgSkParams p{yt, v, workspace, Rp, Cp, (int)Tp, planes, S, Q, nbp};
    hipLaunchKernelGGL(wino24g_wgrad_sk_kernel, dim3((unsigned)nwg), dim3(256), 0, s, p);
    if (int e = clamd_check_launch("wgrad_winograd_pre (stream-K plane GEMM)")) return e;
    W24SkReduceParams rp{workspace, out, Rp, Cp, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p, S, Q};
    const long long nquad = (long long)Rp * Cp / 4;
    const unsigned g = (unsigned)std::min<long long>((nquad + 255) / 256, 8192);
    if (planes == 24) hipLaunchKernelGGL(wino_sk_reduce_kernel<24>, dim3(g), dim3(256), 0, s, rp);
    else hipLaunchKernelGGL(wino_sk_reduce_kernel<36>, dim3(g), dim3(256), 0, s, rp);
    return clamd_check_launch("wgrad_winograd_pre (stream-K reduce)");
}
// wave-level plan (channel counts that are multiples of 128): items = planes x (Rp/128) x (Cp/128) wave tiles, four waves per CU
static void w24g_skw_plan(long long Tp, int Rp, int Cp, int planes, int cus, int* S, int* Q, int* nitems, int* nwg, int* max_slots) {
    *S = (int)(Tp / 2);
    *nitems = planes * (Rp / 128) * (Cp / 128);
    const long long total = (long long)*nitems * *S, waves = 4LL * cus;
    long long q = (total + waves - 1) / waves;
    q = (q + W24G_D - 1) / W24G_D * W24G_D;
    *Q = (int)q;
    *nwg = (int)((((total + q - 1) / q) + 3) / 4);
    *max_slots = (int)((*S + q - 1) / q) + 1;
}
size_t w24g_skw_workspace_bytes(long long Tp, int Rp, int Cp, int planes) {
    int S, Q, ni, nwg, ms;
    w24g_skw_plan(Tp, Rp, Cp, planes, clamd_num_cus(), &S, &Q, &ni, &nwg, &ms);
    return (size_t)(ms + 1) * planes * Rp * Cp * sizeof(float);            // the slots + the reduced planes
}
int launch_w24g_wgrad_skw(const float* yt, const float* v, float* workspace, size_t ws_bytes, float* out, long long Tp, int Rp, int Cp, int planes,
                          int R, int C, int r_seg0, int r_seg0p, int c_seg0, int c_seg0p, const clamd_tuning& tn, hipStream_t s) {
    int S, Q, ni, nwg, ms;
    w24g_skw_plan(Tp, Rp, Cp, planes, clamd_usable_cus(tn), &S, &Q, &ni, &nwg, &ms);
    const size_t slot_floats = (size_t)planes * Rp * Cp;
    if ((size_t)(ms + 1) * slot_floats * sizeof(float) > ws_bytes) return clamd_fail("wgrad_winograd_pre: workspace too small");
    W24WgSkParams p{yt, v, workspace, Rp, Cp, (int)Tp, planes, S, Q, ni};
    hipLaunchKernelGGL(wino24g_wgrad_skw_kernel, dim3((unsigned)nwg), dim3(256), 0, s, p);
    if (int e = clamd_check_launch("wgrad_winograd_pre (wave-level stream-K plane GEMM)")) return e;
    float* reduced = workspace + (size_t)ms * slot_floats;
    W24SkwSumParams sp{workspace, reduced, Rp, Cp, planes, S, Q};
    const long long nthr = (long long)Rp * Cp / 4 * planes;
    hipLaunchKernelGGL(wino_skw_sum_kernel, dim3((unsigned)std::min<long long>((nthr + 255) / 256, 16384)), dim3(256), 0, s, sp);
    if (int e = clamd_check_launch("wgrad_winograd_pre (slot sum)")) return e;
    W24SkReduceParams rp{reduced, out, Rp, Cp, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p, 1, 1};
    const long long nquad = (long long)Rp * Cp / 4;
    const unsigned g = (unsigned)std::min<long long>((nquad + 255) / 256, 8192);
    if (planes == 24) hipLaunchKernelGGL(wino_sk_reduce_kernel<24>, dim3(g), dim3(256), 0, s, rp);
    else hipLaunchKernelGGL(wino_sk_reduce_kernel<36>, dim3(g), dim3(256), 0, s, rp);
    return clamd_check_launch("wgrad_winograd_pre (G^T . G)");
}
}  // namespace clamd
static int w24g_wg_plan(long long Tp, int Rp, int Cp, const clamd_tuning& tn, int* per_out) { return clamd::w24g_wg_plan_planes(Tp, Rp, Cp, 24, tn, per_out); }

extern "C" {

size_t clamd_winograd24_input_elems(int B, int H, int W, int Cp) {
    if (B <= 0 || H <= 0 || W <= 0 || Cp <= 0) return 0;
    return (size_t)w24g_tiles(B, H, W) * (size_t)(Cp / 8) * 24 * 256;
}

int clamd_winograd24_transform_input(const float* x, int x_ldc, const float* scale, const float* shift, float* v, int B, int H, int W,
                                     int Cp, void* stream) {
    if ((scale == nullptr) != (shift == nullptr)) return clamd_fail("winograd24_transform_input: scale and shift go together");
    if (B <= 0 || H <= 0 || W <= 0) return clamd_fail("winograd24_transform_input: empty problem");
    if ((H & 1) || (W & 3)) return clamd_fail("winograd24_transform_input: H must be even and W a multiple of 4 (2x4 output tiles)");
    if (Cp % 8 || x_ldc % 4 || x_ldc < Cp) return clamd_fail("winograd24_transform_input: channel count / pitch must be padded");
    if ((long long)H * W * x_ldc * 4 >= (1ll << 31)) return clamd_fail("winograd24_transform_input: image exceeds 2^31 bytes");
    int ph, pw;
    w24g_tile(W, ph, pw);
    const long long ntm = w24g_tiles(B, H, W), nkg = (Cp / 8 + 3) / 4;
    if (ntm * nkg > 0x7fffffff) return clamd_fail("winograd24_transform_input: grid out of range");
    W24XformParams p{x, x_ldc, scale, shift, v, B, H, W, Cp};
    if (pw == 32) hipLaunchKernelGGL(wino24_xform_kernel<8>, dim3((unsigned)(ntm * nkg)), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(wino24_xform_kernel<4>, dim3((unsigned)(ntm * nkg)), dim3(256), 0, (hipStream_t)stream, p);
    return clamd_check_launch("winograd24_transform_input");
}

int clamd_conv3x3_winograd24_pre(const float* v, const float* w_wino, const float* bias, float* y, int y_ldc,
                                 float* stats, int stat_rows, int B, int H, int W, int Cin_p,
                                 int Cout_p, int relu, const clamd_tuning* tune, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return clamd_fail("conv3x3_winograd24_pre: empty problem");
    if ((H & 1) || (W & 3)) return clamd_fail("conv3x3_winograd24_pre: H must be even and W a multiple of 4 (2x4 output tiles)");
    if (Cin_p % 32 || Cin_p < 64 || Cout_p % 64 || y_ldc % 8) return clamd_fail("conv3x3_winograd24_pre: needs Cin_p % 32 == 0, Cin_p >= 64, Cout_p % 64 == 0");
    if (int e = clamd_check_tuning(tune)) return e;
    const clamd_tuning& tn = clamd_tune(tune);
    const long long tiles = w24g_tiles(B, H, W), ntn = Cout_p / 64;
    if (tiles * ntn > 0x7fffffff) return clamd_fail("conv3x3_winograd24_pre: grid out of range");
    if ((unsigned long long)tiles * (Cin_p / 8) * 24 * 1024 >= (1ull << 32) || (long long)24 * Cout_p * Cin_p * 4 >= (1ll << 31))
        return clamd_fail("conv3x3_winograd24_pre: transformed input exceeds 2^32 bytes or filter 2^31 bytes");
    if (stats && stat_rows != clamd_winograd24_stat_rows(B, H, W, Cout_p, tn))
        return clamd_fail("conv3x3_winograd24_pre: stat_rows does not match clamd_stat_rows(CLAMD_OP_CONV3X3_WINOGRAD24, ...)");
    if (relu & ~1) return clamd_fail("conv3x3_winograd24_pre: relu must be 0 or 1 (no border-class bias here)");
    WinoParams p{v, 0, w_wino, bias, y, y_ldc, stats, B, H, W, Cin_p, Cout_p, relu, 1, 0};
    p.band = wino_band(tiles, ntn, 3.0 * B * H * W * Cin_p, 24.0 * Cin_p * Cout_p, tn.wino_band);
    p.nblk = (int)(tiles * ntn);
    const unsigned grid = tn.wino_persist ? (unsigned)std::min<long long>(p.nblk, clamd_usable_cus(tn)) : (unsigned)p.nblk;
    int ph, pw;
    w24g_tile(W, ph, pw);
    const bool ragged = (H % ph) != 0 || (W % pw) != 0;
    hipStream_t s = (hipStream_t)stream;
#define W24G_LAUNCH(TXN_, RG_) hipLaunchKernelGGL((wino24g_kernel<TXN_, RG_, 2>), dim3(grid), dim3(256), 0, s, p)
    if (pw == 32) { if (ragged) W24G_LAUNCH(8, true); else W24G_LAUNCH(8, false); }
    else { if (ragged) W24G_LAUNCH(4, true); else W24G_LAUNCH(4, false); }
#undef W24G_LAUNCH
    return clamd_check_launch("conv3x3_winograd24_pre");
}

int clamd_conv3x3_winograd24_direct_filters(const float* x, int x_ldc, const float* w_wino, const float* bias, float* y, int y_ldc,
                                            float* stats, int stat_rows, int B, int H, int W,
                                            int Cin_p, int Cout_p, int relu, const clamd_tuning* tune, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return clamd_fail("conv3x3_winograd24_direct_filters: empty problem");
    if ((H & 1) || (W & 3)) return clamd_fail("conv3x3_winograd24_direct_filters: H must be even and W a multiple of 4 (2x4 output tiles)");
    if (Cin_p % 32 || Cout_p % 64 || x_ldc % 8 || y_ldc % 8) return clamd_fail("conv3x3_winograd24_direct_filters: needs Cin_p % 32 == 0, Cout_p % 64 == 0");
    if ((long long)H * W * x_ldc * 4 >= (1ll << 31) || (long long)24 * Cout_p * Cin_p * 4 >= (1ll << 31))
        return clamd_fail("conv3x3_winograd24_direct_filters: image or filter exceeds 2^31 bytes");
    if (int e = clamd_check_tuning(tune)) return e;
    const clamd_tuning& tn = clamd_tune(tune);
    const long long tiles = w24g_tiles(B, H, W), ntn = Cout_p / 64;
    if (tiles * ntn > 0x7fffffff) return clamd_fail("conv3x3_winograd24_direct_filters: grid out of range");
    if (stats && stat_rows != clamd_winograd24_stat_rows(B, H, W, Cout_p, tn))
        return clamd_fail("conv3x3_winograd24_direct_filters: stat_rows does not match clamd_stat_rows(CLAMD_OP_CONV3X3_WINOGRAD24, ...)");
    if ((relu & ~3) || ((relu & CLAMD_BIAS_BORDER_CLASSES) && !bias))
        return clamd_fail("conv3x3_winograd24_direct_filters: bad relu flags (bit 1 needs the [9][Cout_p] bias table)");
    WinoParams p{x, x_ldc, w_wino, bias, y, y_ldc, stats, B, H, W, Cin_p, Cout_p, relu & 1, 1, 0};
    p.bias_classes = (relu & CLAMD_BIAS_BORDER_CLASSES) ? 1 : 0;
    p.band = wino_band(tiles, ntn, (double)B * H * W * Cin_p, 24.0 * Cin_p * Cout_p, tn.wino_band);
    p.nblk = (int)(tiles * ntn);
    const unsigned grid = tn.wino_persist ? (unsigned)std::min<long long>(p.nblk, clamd_usable_cus(tn)) : (unsigned)p.nblk;
    int ph, pw;
    w24g_tile(W, ph, pw);
    const bool ragged = (H % ph) != 0 || (W % pw) != 0;
    hipStream_t s = (hipStream_t)stream;
#define W24H_LAUNCH(TXN_, RG_)                                                                                         \
    do {                                                                                                               \
        if (p.bias_classes) hipLaunchKernelGGL((wino24h_kernel<TXN_, RG_, true>), dim3(grid), dim3(256), 0, s, p);     \
        else hipLaunchKernelGGL((wino24h_kernel<TXN_, RG_, false>), dim3(grid), dim3(256), 0, s, p);                   \
    } while (0)
    if (pw == 32) { if (ragged) W24H_LAUNCH(8, true); else W24H_LAUNCH(8, false); }
    else { if (ragged) W24H_LAUNCH(4, true); else W24H_LAUNCH(4, false); }
#undef W24H_LAUNCH
    return clamd_check_launch("conv3x3_winograd24_direct_filters");
}

size_t clamd_wgrad_winograd24_pre_operand_elems(int B, int H, int W, int Rp) {
    if (B <= 0 || H <= 0 || W <= 0 || Rp <= 0) return 0;
    return (size_t)(24 * w24g_tiles(B, H, W) * 32) * (size_t)Rp;
}

size_t clamd_wgrad_winograd24_pre_workspace_bytes(int B, int H, int W, int Rp, int Cp) {
    if (B <= 0 || H <= 0 || W <= 0 || Rp < 128 || Cp < 128) return 0;
    const long long Tp = w24g_tiles(B, H, W) * 32;
    if ((Rp % 256) || (Cp % 256)) return clamd::w24g_skw_workspace_bytes(Tp, Rp, Cp, 24);      // multiples of 128: the wave-level plan
    const long long nb = 24LL * (Rp / 256) * (Cp / 256);
    const long long max_split = std::max<long long>(1, std::min<long long>((3LL * clamd_num_cus()) / nb, Tp / (2 * W24G_D)));
    return std::max((size_t)max_split * 24 * Rp * Cp * sizeof(float), clamd::w24g_sk_workspace_bytes(Tp, Rp, Cp, 24));
}

int clamd_wgrad_winograd24_pre(const float* gz, int gz_ldc, const float* v, float* yt, float* workspace, size_t ws_bytes, float* out,
                               int B, int H, int W, int Rp, int Cp, int R, int C, int r_seg0, int r_seg0p, int c_seg0, int c_seg0p,
                               const clamd_tuning* tune, void* stream) {
    if (int e = clamd_check_tuning(tune)) return e;
    if (B <= 0 || H <= 0 || W <= 0) return clamd_fail("wgrad_winograd24_pre: empty problem");
    if ((H & 1) || (W & 3)) return clamd_fail("wgrad_winograd24_pre: H must be even and W a multiple of 4");
    if (Rp % 128 || Cp % 128 || Rp <= 0 || Cp <= 0 || gz_ldc % 4) return clamd_fail("wgrad_winograd24_pre: needs Rp and Cp multiples of 128");
    const bool wave_level = (Rp % 256) || (Cp % 256);          // a wave owns a 128 x 128 block; 256 x 256 workgroup blocks where both counts allow
    if (gz && (long long)H * W * gz_ldc * 4 >= (1ll << 31)) return clamd_fail("wgrad_winograd24_pre: one image exceeds 2^31 bytes");
    const clamd_tuning& tn = clamd_tune(tune);
    const long long ntm = w24g_tiles(B, H, W), Tp = ntm * 32;
    if (Tp * Rp * 4 >= (1ll << 32) || (unsigned long long)ntm * (Cp / 8) * 24 * 1024 >= (1ull << 32))
        return clamd_fail("wgrad_winograd24_pre: an operand exceeds 2^32 bytes");
    int per = 0;
    const int nsplit = wave_level ? 1 : w24g_wg_plan(Tp, Rp, Cp, tn, &per);
    // stream-K where the items alone nearly fill the chip (an item then gets 2-3 slots); with fewer, larger items the split-K plan's
    // whole rounds are as even and its slabs are fewer (tools/wino44g_ab.py: 1.10-1.21x faster from 96 items on, 0.78-0.95x below)
    const bool streamk = wave_level || tn.wgrad_streamk == 2 || (tn.wgrad_streamk == 1 && (long long)24 * (Rp / 256) * (Cp / 256) * 8 >= 3LL * clamd_usable_cus(tn));
    if (!streamk && (size_t)nsplit * 24 * Rp * Cp * sizeof(float) > ws_bytes) return clamd_fail("wgrad_winograd24_pre: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    if (gz) {                                  // gz == NULL: yt already holds the transformed gradient (clamd_wgrad_winograd24_pre_transform)
        W24WgXformParams pa{gz, gz_ldc, yt, B, H, W, Rp, (int)Tp};
        const long long na = (Tp * (Rp / 4) + 255) / 256;
        const unsigned g = (unsigned)std::min<long long>(na, 1 << 20);
        if (W >= 32) hipLaunchKernelGGL(wino24g_wgrad_xform_kernel<8>, dim3(g), dim3(256), 0, s, pa);
        else hipLaunchKernelGGL(wino24g_wgrad_xform_kernel<4>, dim3(g), dim3(256), 0, s, pa);
        if (int e = clamd_check_launch("wgrad_winograd24_pre transform")) return e;
    }
    if (wave_level)
        return launch_w24g_wgrad_skw(yt, v, workspace, ws_bytes, out, Tp, Rp, Cp, 24, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p, tn, s);
    if (streamk)
        return launch_w24g_wgrad_sk(yt, v, workspace, ws_bytes, out, Tp, Rp, Cp, 24, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p, tn, s);
    if (int e = launch_w24g_wgrad_gemm(yt, v, workspace, Rp, Cp, Tp, nsplit, per, 24, s)) return e;
    W24GReduceParams rp{workspace, out, nsplit, Rp, Cp, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p};
    const long long nquad = (long long)Rp * Cp / 4;
#define W24G_REDUCE(PHS_)                                                                                              \
    do {                                                                                                               \
        long long g = (nquad + 4 * (64 / PHS_) - 1) / (4 * (64 / PHS_));                                               \
        if (g > 8192) g = 8192;                                                                                        \
        hipLaunchKernelGGL(wino24g_wgrad_reduce_kernel<PHS_>, dim3((unsigned)g), dim3(256), 0, s, rp);                 \
    } while (0)
    if (nsplit >= 4 && nquad <= 65536) W24G_REDUCE(4);
    else if (nsplit >= 2 && nquad <= 131072) W24G_REDUCE(2);
    else W24G_REDUCE(1);
#undef W24G_REDUCE
    return clamd_check_launch("wgrad_winograd24_pre_reduce");
}

// The gradient-side transform alone (A4 dY A6^T of gz into yt, tiles in the order of the forward image): a caller with several
// streams can run it beside another launch's GEMM and then call clamd_wgrad_winograd24_pre with gz == NULL.
int clamd_wgrad_winograd24_pre_transform(const float* gz, int gz_ldc, float* yt, int B, int H, int W, int Rp, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || !gz || !yt) return clamd_fail("wgrad_winograd24_pre_transform: bad arguments");
    if ((H & 1) || (W & 3)) return clamd_fail("wgrad_winograd24_pre_transform: H must be even and W a multiple of 4");
    if (Rp % 4 || Rp <= 0 || gz_ldc % 4) return clamd_fail("wgrad_winograd24_pre_transform: channel count / pitch must be padded");
    const long long Tp = w24g_tiles(B, H, W) * 32;
    if ((long long)H * W * gz_ldc * 4 >= (1ll << 31) || Tp * Rp * 4 >= (1ll << 32)) return clamd_fail("wgrad_winograd24_pre_transform: operand too large");
    W24WgXformParams pa{gz, gz_ldc, yt, B, H, W, Rp, (int)Tp};
    const long long na = (Tp * (Rp / 4) + 255) / 256;
    const unsigned g = (unsigned)std::min<long long>(na, 1 << 20);
    if (W >= 32) hipLaunchKernelGGL(wino24g_wgrad_xform_kernel<8>, dim3(g), dim3(256), 0, (hipStream_t)stream, pa);
    else hipLaunchKernelGGL(wino24g_wgrad_xform_kernel<4>, dim3(g), dim3(256), 0, (hipStream_t)stream, pa);
    return clamd_check_launch("wgrad_winograd24_pre_transform");
}

}  // extern "C"
