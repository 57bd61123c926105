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
#include <algorithm>
#include "common.hip.h"
#include "clamd_internal.h"
#include "wino_common.hip.h"

// K-loop issue pattern (sched_mfma_slots): RPS LDS reads per slot in slots [0, R1), NV VALU ops per slot from slot V0.
// Measured alternatives (build with -DWN_RPS=.. etc.): reads two per slot and VALU from slot 12 (no s_waitcnt lgkmcnt(0)
// behind the reads any more) 209 vs 217 TF/s aggregate; NV = 4: 203; V0 = 16: 201.
#ifndef WN_RPS
#define WN_RPS 1
#define WN_R1 24
#define WN_V0 0
#define WN_NV 2
#endif
namespace clamd {

constexpr int WN_HW = 18;                                     // input halo width of a 16-pixel-wide output tile
constexpr int WN_WG = 66;                                     // padded rows per (xi, group)
constexpr int WN_WT_SLOTS = 16 * 2 * WN_WG;
constexpr int WN_EXP = 36;                                    // row pitch (floats) of the epilogue exchange block

// MT = 32-tile halves per workgroup: 2 -> 8x8 Winograd tiles = 16x16 output pixels; 1 -> 4x8 tiles = 8x16 pixels (used when
// the 16x16 grid would leave CUs without a workgroup)
// RAGGED = false: H, W multiples of the tile and Cout_p a multiple of 64 -- the 16 output stores of a tile and their
// statistics need no range predicate.
template <int MT, bool RAGGED>
__global__ void __launch_bounds__(256, 1) wino_kernel(const WinoParams p) {
    constexpr int WN_HH = 8 * MT + 2, WN_PIX = WN_HW * WN_HH;             // input halo of the output tile
    constexpr int WN_PIXP = WN_PIX + ((10 - WN_PIX % 8) % 8);             // == 2 (mod 8): conflict-free staging stores
    constexpr int WN_IN_SLOTS = 2 * WN_PIXP, WN_STAGE = WN_IN_SLOTS + WN_WT_SLOTS;
    constexpr int NJI = (2 * WN_PIX + 255) / 256;                         // input staging loads per thread
    constexpr int NST = 2;                     // LDS stages: chunk k+1 readable, chunk k+2 being written (chunk k is in registers)
    constexpr int WN_EXB = 4 * 2 * 32 * WN_EXP;                           // floats of one (mt, nt) exchange block
    constexpr int WN_LDS = (NST * WN_STAGE * 16 > 2 * MT * WN_EXB * 4 ? NST * WN_STAGE * 16 : 2 * MT * WN_EXB * 4) / 16;
    static_assert(WN_LDS * 16 <= 160 * 1024, "LDS budget");
    __shared__ uint4 smem[WN_LDS];
    // Statistics rows: one per pixel tile when every tile has its own workgroup; one per WORKGROUP of a persistent grid,
    // zeroed here and accumulated tile after tile by the same thread with plain loads and stores (see wino24.hip).
    const bool per_wg_rows = p.stats != nullptr && gridDim.x < (unsigned)p.nblk;
    if (per_wg_rows && threadIdx.x < 128)        // thread (k, c) zeroes exactly the words it later accumulates into
        for (int n = threadIdx.x & 63; n < p.Np; n += 64) p.stats[((size_t)blockIdx.x * 2 + (threadIdx.x >> 6)) * p.Np + n] = 0.f;
    float racc = 0.f;                            // running sum of this thread's (kind, channel) over the tiles of one slab
    int rslab = -1;
    auto flush_row = [&]() {                     // threads < 128 only; wave-uniform call sites (see wino24.hip)
        if (rslab >= 0 && rslab * 64 + (int)(threadIdx.x & 63) < p.Np) {
            float* dst = p.stats + ((size_t)blockIdx.x * 2 + (threadIdx.x >> 6)) * p.Np + rslab * 64 + (threadIdx.x & 63);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float old = __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst, old + racc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // Persistent: workgroup g walks the tiles g, g + grid, ... (same XCD every round).  The output stores and the
    // statistics row of a tile drain while the next tile is loaded and multiplied; a workgroup per tile instead waits for
    // its last store to be acknowledged before the CU can start the next one (measured: 3 us of 30 per 64-channel tile).
    for (int v = blockIdx.x; v < p.nblk; v += gridDim.x) {
        // everything below is re-derived per tile from an opaque copy of the thread id: hoisted out of the tile loop the
        // per-lane constants would stay live through the MFMA loop and spill (344 bytes per lane, measured)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, w = tid >> 6;
        const int r = lane & 31, h = lane >> 5;
        const int tiles_x = (p.W + 15) >> 4, tiles_y = (p.H + 8 * MT - 1) / (8 * MT);
        const int ntn = (p.Np + 63) >> 6;
        const int bid = xcd_remap(v, p.nblk);
        // block order [band of p.band slabs][pixel tile][slab in band]: the 32 workgroups resident on one XCD (contiguous
        // ids after xcd_remap) cover (32 / band) pixel tiles x band slabs and share those input tiles and filter slabs in L2
        const int per_band = (p.nblk / ntn) * p.band;
        const int bnd = bid / per_band, rem = bid - bnd * per_band;
        const int tn = bnd * p.band + rem % p.band, tm = rem / p.band;
        const int x0 = (tm % tiles_x) * 16, y0 = ((tm / tiles_x) % tiles_y) * (8 * MT), b = tm / (tiles_x * tiles_y);
        const int n0 = tn * 64;
        const int nk = p.Kp >> 3;

        // ---- staging descriptors --------------------------------------------------------------------------------------
        const unsigned img = (unsigned)p.H * (unsigned)p.W * (unsigned)p.x_ldc * 4u;
#ifdef WN_ABLATE_SAMETILE   // measurement only: every workgroup stages the same input tile (all loads hit the caches)
        const __amdgpu_buffer_rsrc_t xrs = make_rsrc((const char*)p.x, img);
#else
        const __amdgpu_buffer_rsrc_t xrs = make_rsrc((const char*)p.x + (size_t)b * img, img);
#endif
        const __amdgpu_buffer_rsrc_t wrs = make_rsrc(p.w, (unsigned)(16u * p.Np * p.Kp * 4u));
        const __amdgpu_buffer_rsrc_t xrs_dead = make_rsrc(p.x, 0u), wrs_dead = make_rsrc(p.w, 0u);
        unsigned in_vo[NJI];
        int in_slot[NJI];
    #pragma unroll
        for (int j = 0; j < NJI; ++j) {
            int piece = tid + 256 * j;                           // (pixel, 16-byte group); the last pass wraps
            if (piece >= 2 * WN_PIX) piece -= 2 * WN_PIX;
            const int g = piece & 1, pix = piece >> 1;
            const int hy = pix / WN_HW, hx = pix - hy * WN_HW;
#ifdef WN_ABLATE_SAMETILE
            const int yy = hy - 1 + 16, xx = hx - 1 + 16;
#else
            const int yy = y0 + hy - 1, xx = x0 + hx - 1;
#endif
            in_vo[j] = (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) ? (unsigned)(((yy * p.W + xx) * p.x_ldc + 4 * g) * 4) : BUF_OOB;
            in_slot[j] = g * WN_PIXP + pix;
        }
        // filter slab of one K-chunk: 16 xi x 64 rows x 2 groups = 2048 pieces, piece = tid + 256*j: xi = (tid >> 7) + 2j
        const int wg_ = tid & 1, wn_ = (tid >> 1) & 63, wxi0 = tid >> 7;
        const unsigned w_vo0 = n0 + wn_ < p.Np ? (unsigned)(((wxi0 * p.Np + n0 + wn_) * 8 + 4 * wg_) * 4) : BUF_OOB;
        const unsigned w_vstep = (unsigned)(2 * p.Np * 8 * 4);    // two xi further
        const unsigned w_chunk = (unsigned)(16 * p.Np * 8 * 4);   // bytes of one K-chunk
        const int w_slot0 = WN_IN_SLOTS + (wxi0 * 2 + wg_) * WN_WG + wn_;

        uint4 rin[NJI], rw[8];
        auto gload_to = [&](int k, bool live, uint4 (&ri)[NJI], uint4 (&rww)[8]) {
            const unsigned so = (unsigned)(k * 8 * 4);
            // loads past the last chunk use EMPTY descriptors (every lane out of range, no traffic): a scalar select
            // instead of a per-lane one on each offset
            const __amdgpu_buffer_rsrc_t xr = live ? xrs : xrs_dead, wr = live ? wrs : wrs_dead;
    #pragma unroll
            for (int j = 0; j < NJI; ++j) ri[j] = buf_ld16(xr, in_vo[j], so);
    #pragma unroll
            for (int j = 0; j < 8; ++j) rww[j] = buf_ld16(wr, w_vo0, (unsigned)k * w_chunk + j * w_vstep);   // j-step in the SCALAR offset: no VALU add per load
        };
        auto lds_store_from = [&](int st, const uint4 (&ri)[NJI], const uint4 (&rww)[8]) {
            uint4* sm = smem + st * WN_STAGE;
    #pragma unroll
            for (int j = 0; j < NJI; ++j) sm[in_slot[j]] = ri[j];
    #pragma unroll
            for (int j = 0; j < 8; ++j) sm[w_slot0 + j * 4 * WN_WG] = rww[j];
        };
        auto gload = [&](int k, bool live) { gload_to(k, live, rin, rw); };
        auto lds_store = [&](int st) { lds_store_from(st, rin, rw); };

        // ---- fragment addressing: wave w = Winograd row i: t[b] = d[a1][b] + s2 * d[a2][b] -----------------------------
        const int a1 = w == 0 ? 0 : 1, a2 = w == 3 ? 3 : 2;
        // B^T rows: d0 - d2 | d1 + d2 | d2 - d1 | d1 - d3.  Row 2 is formed as d1 - d2 = -(d2 - d1): its sign is folded into
        // the transformed filters (wino_pack_kernel negates U[2][j]), so every row is d[a1] + s2 * d[a2] -- one fma per
        // component with a wave-uniform coefficient instead of a multiply and an fma (32 of 134 VALU ops per chunk).
        const float s2 = w == 1 ? 1.f : -1.f;
        int p1[MT], p2[MT];
    #pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = 32 * mt + r, ty = m >> 3, tx = m & 7;
            const int pb = h * WN_PIXP + (2 * ty) * WN_HW + 2 * tx;
            p1[mt] = pb + a1 * WN_HW;
            p2[mt] = pb + a2 * WN_HW;
        }
        const int wb = WN_IN_SLOTS + (4 * w * 2 + h) * WN_WG + r;            // + j*2*WG + 32*nt

        f32x16 acc[4][MT][2];                                                 // [j][tile half][channel half]
    #pragma unroll
        for (int j = 0; j < 4; ++j)
    #pragma unroll
            for (int i = 0; i < 2 * MT; ++i)
    #pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][i >> 1][i & 1][e] = 0.f;

        // fragments of one chunk: the input transform V[w][j] for both tile halves + the filter rows of this wave's 4 xi
        uint4 A[4][MT], Bf[4][2];
        auto frags = [&](int st) {
            const uint4* sm = smem + st * WN_STAGE;
    #pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float4 t[4];
    #pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint4 u1 = sm[p1[mt] + c], u2 = sm[p2[mt] + c];
                    t[c].x = fmaf(s2, __uint_as_float(u2.x), __uint_as_float(u1.x));   // ONE op per component: row 2's
                    t[c].y = fmaf(s2, __uint_as_float(u2.y), __uint_as_float(u1.y));   // overall sign lives in the
                    t[c].z = fmaf(s2, __uint_as_float(u2.z), __uint_as_float(u1.z));   // packed filters (see below)
                    t[c].w = fmaf(s2, __uint_as_float(u2.w), __uint_as_float(u1.w));
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
        {   // prologue: the first three chunks are requested back to back (one memory latency, not three)
            uint4 ri0[NJI], rw0[8], ri1[NJI], rw1[8];
            gload_to(0, true, ri0, rw0);
            gload_to(1, 1 < nk, ri1, rw1);
            gload(2, 2 < nk);
            lds_store_from(0, ri0, rw0);
            lds_store_from(1, ri1, rw1);
        }
        __syncthreads();
        frags(0);
        __syncthreads();                                                      // stage 0 is free again
        for (int k = 0; k < nk; ++k) {
            uint4 Ac[4][MT], Bc[4][2];
    #pragma unroll
            for (int j = 0; j < 4; ++j) {
    #pragma unroll
                for (int i = 0; i < MT; ++i) Ac[j][i] = A[j][i];
    #pragma unroll
                for (int i = 0; i < 2; ++i) Bc[j][i] = Bf[j][i];
            }
            // straight-line body (past the last chunk the reads hit valid LDS and the stores write zeros nobody reads), so
            // that the 24 fragment reads, ~130 transform VALU ops, 11 staging stores and 11 loads can be issued in the gaps
            // of the 64 MFMAs instead of in front of them
            frags((k + 1) % NST);                                             // visible since the last barrier
    #pragma unroll
            for (int j = 0; j < 4; ++j)
    #pragma unroll
                for (int mt = 0; mt < MT; ++mt)
    #pragma unroll
                    for (int nt = 0; nt < 2; ++nt) mma16<float>(Ac[j][mt], Bc[j][nt], acc[j][mt][nt]);
            lds_store(k % NST);                                               // chunk k+2 over chunk k's stage (read before the last barrier)
            gload(k + 3, k + 3 < nk);
            if constexpr (MT == 2) sched_mfma_slots<64, WN_R1, 26, 37, 38, 49, WN_NV, WN_V0, WN_RPS>();
            else sched_mfma_slots<32, 16, 17, 27, 21, 31, 2>();
            __syncthreads();
        }

    #ifdef WN_ABLATE_EPI   // measurement only: no output transform / stores (keeps the accumulators alive)
        {
            float t = 0.f;
    #pragma unroll
            for (int j = 0; j < 4; ++j)
    #pragma unroll
                for (int mt = 0; mt < MT; ++mt)
    #pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
    #pragma unroll
                        for (int e = 0; e < 16; ++e) t += acc[j][mt][nt][e];
            if (t == 12345.678f) p.y[0] = t;
            return;
        }
    #endif
        // ---- epilogue: output transform Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]] -------------------------------------
        // All 2*MT (mt, nt) blocks are exchanged at once ([block][wave][q][32 tiles][WN_EXP], 144 KB for MT = 2): one
        // barrier per tile instead of two per block, and the 128 scalar LDS writes of a lane go out back to back (the
        // per-block version cost 21 % of the 64 -> 64 @ 256^2 launch, measured by ablation).
        float* const ex = reinterpret_cast<float*>(smem);
        const int tl = tid >> 3, ng = tid & 7;                                // reader: tile inside the half, 4-channel group
        float st1[2][4], st2[2][4];
    #pragma unroll
        for (int i = 0; i < 8; ++i) { st1[i >> 2][i & 3] = 0.f; st2[i >> 2][i & 3] = 0.f; }
        const float relu_lo = p.relu ? 0.f : -__builtin_inff();
    #pragma unroll
        for (int mt = 0; mt < MT; ++mt)
    #pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float* const exb = ex + (mt * 2 + nt) * WN_EXB;
    #pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float m0 = acc[0][mt][nt][e], m1 = acc[1][mt][nt][e], m2 = acc[2][mt][nt][e], m3 = acc[3][mt][nt][e];
                    exb[((w * 2 + 0) * 32 + acc_row(e, h)) * WN_EXP + r] = m0 + m1 + m2;
                    exb[((w * 2 + 1) * 32 + acc_row(e, h)) * WN_EXP + r] = m1 - m2 - m3;
                }
            }
        __syncthreads();
    #pragma unroll
        for (int mt = 0; mt < MT; ++mt)
    #pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float* const exb = ex + (mt * 2 + nt) * WN_EXB;
                float4 R[4][2];
    #pragma unroll
                for (int i = 0; i < 4; ++i)
    #pragma unroll
                    for (int q = 0; q < 2; ++q) R[i][q] = *reinterpret_cast<const float4*>(exb + ((i * 2 + q) * 32 + tl) * WN_EXP + 4 * ng);
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
    #ifndef WN_ABLATE_ST
                            *reinterpret_cast<float4*>(p.y + (((size_t)b * p.H + yy) * p.W + xx) * p.y_ldc + n) = v;
    #endif
                            st1[nt][0] += v.x; st1[nt][1] += v.y; st1[nt][2] += v.z; st1[nt][3] += v.w;
                            st2[nt][0] = fmaf(v.x, v.x, st2[nt][0]); st2[nt][1] = fmaf(v.y, v.y, st2[nt][1]);
                            st2[nt][2] = fmaf(v.z, v.z, st2[nt][2]); st2[nt][3] = fmaf(v.w, v.w, st2[nt][3]);
                        }
                    }
            }
        if (p.stats) {
            // threads with equal (tid & 7) own the same channels: fold the 8 tiles of a wave (lane bits 3-5), then the 4 waves
            __syncthreads();                                                   // every wave has read its exchange blocks
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
                if (!per_wg_rows) {
                    if (n0 + c < p.Np) p.stats[((size_t)tm * 2 + k) * p.Np + n0 + c] = t;          // row = pixel tile
                } else {
                    if (tn != rslab) { flush_row(); rslab = tn; racc = 0.f; }                      // wave-uniform
                    racc += t;
                }
            }
        }

        __syncthreads();                                                   // exchange / statistics blocks are free again
    }
    if (per_wg_rows && threadIdx.x < 128) flush_row();
}

// ---- filter transform ---------------------------------------------------------------------------------------------
// One job = one GEMM operand: dst[(k/8)*16 + xi][n][k%8] = (G g G^T)[xi] with g = w[n_l][k_l] (forward) or the tap-flipped
// w[k_l][n_l] (data gradient); physical -> logical channel maps as in clamd_pack (two segments for concat inputs, zero
// padding).  One thread per (n, k): consecutive threads write consecutive floats of every xi row.
__global__ void __launch_bounds__(256) wino_pack_kernel(const WinoPackJob* __restrict__ jobs, int njobs, int nblocks, const FoldBias fold) {
    if ((int)blockIdx.x >= nblocks) {       // appended blocks: the border-class bias table of a folded BatchNorm (common.hip.h)
        __shared__ float T[9];
        fold_bias_block(fold, (int)blockIdx.x - nblocks, T);
        return;
    }
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
        const float ks = J.kscale ? J.kscale[k] : 1.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i / 3][i % 3] = (J.dgrad ? s[8 - i] : s[i]) * ks;
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
        const float sg = i == 2 ? -0.5f : 0.5f;          // row 2 carries the sign of the kernel's input transform (see wino_kernel)
        const float u0 = 2.f * sg * t[i][0], u1 = sg * (t[i][0] + t[i][1] + t[i][2]), u2 = sg * (t[i][0] - t[i][1] + t[i][2]), u3 = 2.f * sg * t[i][2];
        d[(size_t)(4 * i + 0) * J.Np * 8] = u0;
        d[(size_t)(4 * i + 1) * J.Np * 8] = u1;
        d[(size_t)(4 * i + 2) * J.Np * 8] = u2;
        d[(size_t)(4 * i + 3) * J.Np * 8] = u3;
    }
}

}  // namespace clamd

namespace clamd {

// HBM traffic model of one launch: an XCD holds 32 workgroups at a time = a pixel tiles x b slabs (a * b = 32); every
// such group fetches its a input tiles and b filter slabs once, so bytes ~ X * (slabs / b) + F * (tiles / a).  Measured
// with b = all slabs (slab-fastest order): 553 MB for 1024 -> 1024 @ 16^2, where activations + filters are 88 MB.
int wino_band(long long tiles, long long slabs, double x_elems, double f_elems, int forced) {
    int best = 1;
    double best_cost = 0;
    for (int b = 1; b <= 32 && b <= slabs; b *= 2) {
        if (slabs % b) break;
        const double a = 32.0 / b < (double)tiles ? 32.0 / b : (double)tiles;
        const double cost = x_elems * ((double)slabs / b) + f_elems * ((double)tiles / a);
        if (b == 1 || cost < best_cost) { best = b; best_cost = cost; }
    }
    if (forced > 0) { best = 1; while (best * 2 <= forced && slabs % (best * 2) == 0) best *= 2; }   // "wino_band": rounded down to a divisor
    return best;
}

}  // namespace clamd

using namespace clamd;

// 16x16-pixel tiles unless that grid would leave a quarter of the CUs without a workgroup
static bool wino_mt2(int B, int H, int W, int Cout_p, const clamd_tuning& tn) {
    const long long ntn = (Cout_p + 63) / 64, nblk2 = (long long)B * ((H + 15) / 16) * ((W + 15) / 16) * ntn;
    const long long nblk1 = (long long)B * ((H + 7) / 8) * ((W + 15) / 16) * ntn;
    return tn.wino_mt ? tn.wino_mt == 2 : (nblk2 >= 192 || nblk1 == nblk2);
}
static long long wino_tiles(int B, int H, int W, int Cout_p, const clamd_tuning& tn) {
    return (long long)B * ((H + (wino_mt2(B, H, W, Cout_p, tn) ? 15 : 7)) / (wino_mt2(B, H, W, Cout_p, tn) ? 16 : 8)) * ((W + 15) / 16);
}
// rows of a launch: one per pixel tile (one workgroup per tile), or one per workgroup of the persistent grid
long long clamd_winograd_stat_rows(int B, int H, int W, int Cout_p, const clamd_tuning& tn) {
    const long long tiles = wino_tiles(B, H, W, Cout_p, tn), nblk = tiles * ((Cout_p + 63) / 64);
    return (tn.wino_persist && nblk > clamd_usable_cus(tn)) ? clamd_usable_cus(tn) : tiles;
}

int clamd_launch_wino_pack(const void* jobs_dev, int njobs, int total_blocks, const clamd::FoldBias* fold, hipStream_t stream) {
    if (njobs <= 0 || total_blocks <= 0) return clamd_fail("wino_pack: empty job table");
    const clamd::FoldBias f = fold ? *fold : clamd::FoldBias{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 9};
    hipLaunchKernelGGL(clamd::wino_pack_kernel, dim3(total_blocks + (fold ? fold->Cout_p : 0)), dim3(256), 0, stream, (const clamd::WinoPackJob*)jobs_dev, njobs,
                       total_blocks, f);
    return clamd_check_launch("wino_pack");
}

extern "C" {

int clamd_sizeof_wino_pack_job(void) { return (int)sizeof(WinoPackJob); }

int clamd_wino_pack(const void* jobs_dev, int njobs, int total_blocks, void* stream) {
    return clamd_launch_wino_pack(jobs_dev, njobs, total_blocks, nullptr, (hipStream_t)stream);
}

int clamd_conv3x3_winograd(const float* x, int x_ldc, const float* w_wino, const float* bias, float* y, int y_ldc,
                           float* stats, int stat_rows, int B, int H, int W, int Cin_p, int Cout_p, int relu,
                           const clamd_tuning* tune, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return clamd_fail("conv3x3_winograd: empty problem");
    if ((H | W) & 1) return clamd_fail("conv3x3_winograd: H and W must be even (2x2 output tiles)");
    if (Cin_p % 32 || Cout_p % 32 || x_ldc % 8 || y_ldc % 8) return clamd_fail("conv3x3_winograd: channel counts/pitches must be padded");
    if ((long long)H * W * x_ldc * 4 >= (1ll << 31) || (long long)16 * Cout_p * Cin_p * 4 >= (1ll << 31))
        return clamd_fail("conv3x3_winograd: image or filter exceeds 2^31 bytes");
    if (int e = clamd_check_tuning(tune)) return e;
    const clamd_tuning& tn = clamd_tune(tune);
    if (relu & ~1) return clamd_fail("conv3x3_winograd: relu must be 0 or 1 (no border-class bias here)");
    WinoParams p{x, x_ldc, w_wino, bias, y, y_ldc, stats, B, H, W, Cin_p, Cout_p, relu, 1, 0};
    const long long ntn = (Cout_p + 63) / 64, nblk2 = (long long)B * ((H + 15) / 16) * ((W + 15) / 16) * ntn;
    const long long nblk1 = (long long)B * ((H + 7) / 8) * ((W + 15) / 16) * ntn;
    if (nblk1 > 0x7fffffff) return clamd_fail("conv3x3_winograd: grid out of range");
    const bool mt2 = wino_mt2(B, H, W, Cout_p, tn);
    p.band = wino_band((mt2 ? nblk2 : nblk1) / ntn, ntn, (double)B * H * W * Cin_p, 16.0 * Cin_p * Cout_p, tn.wino_band);
    p.nblk = (int)(mt2 ? nblk2 : nblk1);
    if (stats && stat_rows != clamd_winograd_stat_rows(B, H, W, Cout_p, tn))
        return clamd_fail("conv3x3_winograd: stat_rows does not match clamd_stat_rows(CLAMD_OP_CONV3X3_WINOGRAD, ...)");
    const unsigned grid = tn.wino_persist ? (unsigned)std::min<long long>(p.nblk, clamd_usable_cus(tn)) : (unsigned)p.nblk;
    const bool ragged = (H % (mt2 ? 16 : 8)) != 0 || (W % 16) != 0 || (Cout_p % 64) != 0;
#define WN_LAUNCH(MT_, RG_) hipLaunchKernelGGL((wino_kernel<MT_, RG_>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p)
    if (mt2) { if (ragged) WN_LAUNCH(2, true); else WN_LAUNCH(2, false); }
    else { if (ragged) WN_LAUNCH(1, true); else WN_LAUNCH(1, false); }
#undef WN_LAUNCH
    return clamd_check_launch("conv3x3_winograd");
}

}  // extern "C"

namespace clamd {

// =====================================================================================================================
// Winograd weight gradient (fp32): dU[xi][r][c] = sum_tiles (A dY A^T)[xi][tile][r] * (B^T d B)[xi][tile][c], then
// dg = G^T dU G.  Same 2.25x saving as the forward: 16 instead of 36 multiply-adds per 2x2 tile and channel pair.
//
// One 256-thread workgroup per CU owns a 64 (r: channels of gz) x 64 (c: channels of x) x 16 (xi) block of dU for a range of
// pixel tiles (split-K, fp32 slabs, deterministic reduce as in wgrad.hip).  A pixel tile = 4 x 8 Winograd tiles = 8 x 16
// output pixels: gz tile (128 px) and x halo (10 x 18 px) are staged as [pixel][64 channels] fp32 images (two stages); the
// contraction index of the MFMA is the tile, so every operand is ONE float per lane (ds_read_b32, lanes = channels:
// conflict-free) and both transforms happen in registers: wave w owns Winograd row i = w,
//   A side: rc[q] = sum_p A[w][p] gz[p][q],      Yt[w][j] = sum_q A[j][q] rc[q]        (A = [[1,0],[1,1],[1,-1],[0,-1]])
//   B side: t[b]  = sum_a B^T[w][a] d[a][b],     V[w][j]  = sum_b B^T[j][b] t[b]
// and accumulates 4 (j) x 2 x 2 MFMA tiles = 256 accumulator registers.  Slabs are [split][i][r][c][j] (each wave stores
// its own plane with contiguous 16-byte pieces); wino_wgrad_reduce_kernel sums the splits and applies G^T . G.
#ifdef CLAMD_DIAG
__device__ unsigned long long g_ww_diag[4];      // diagnostic build only
#endif

struct WinoWgradParams {
    const float* a; int a_ldc;       // gz  [B,H,W,a_ldc]
    const float* b; int b_ldc;       // x   [B,H,W,b_ldc]
    float* partial;                  // [nsplit][4][Rp][Cp][4]
    int B, H, W, Rp, Cp, nsplit, tiles_per_split;
};

constexpr int WW_TY = 4, WW_TX = 8;                                   // Winograd tiles per pixel tile
constexpr int WW_APIX = (2 * WW_TY) * (2 * WW_TX);                    // 128 gz pixels
constexpr int WW_BW = 2 * WW_TX + 2, WW_BH = 2 * WW_TY + 2, WW_BPIX = WW_BW * WW_BH;   // 18 x 10 = 180 halo pixels
constexpr int WW_PS = 256;                                            // bytes per pixel (64 fp32 channels)
constexpr int WW_ABYTES = WW_APIX * WW_PS, WW_STAGE = (WW_APIX + WW_BPIX) * WW_PS;   // 32 KB + 45 KB
constexpr int WW_NJA = WW_APIX * 16 / 256, WW_NJB = (WW_BPIX * 16 + 255) / 256;      // 8 + 12 staging loads per thread

// RAGGED = false (H % 8 == 0 and W % 16 == 0: every pixel tile lies inside the image): the 20 per-lane staging offsets
// are computed ONCE; per tile only the image-border lanes of the halo are switched off (four per-lane bit masks x four
// wave-uniform tile flags) -- address and edge arithmetic was ~170 of the ~550 VALU instructions per tile.
template <bool RAGGED>
__global__ void __launch_bounds__(256, 1) wino_wgrad_kernel(const WinoWgradParams p) {
    static_assert(2 * WW_STAGE <= 160 * 1024, "two stages must fit one CU");
    __shared__ __attribute__((aligned(16))) char smem[2 * WW_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, kh = lane >> 5;

    const int rt = (p.Rp + 63) >> 6, ct = (p.Cp + 63) >> 6;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tr = bid % rt; bid /= rt;
    const int tc = bid % ct; bid /= ct;
    const int split = bid;
    const int r0 = tr * 64, c0 = tc * 64;
    const int tiles_x = (p.W + 2 * WW_TX - 1) / (2 * WW_TX), tiles_y = (p.H + 2 * WW_TY - 1) / (2 * WW_TY);
    const int ntiles = tiles_x * tiles_y * p.B;
    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(ntiles, t_begin + p.tiles_per_split);
    const int n = max(t_end - t_begin, 0);

    const unsigned a_img = (unsigned)p.H * p.W * p.a_ldc * 4u, b_img = (unsigned)p.H * p.W * p.b_ldc * 4u;
    const unsigned b_shift = (unsigned)(p.W + 1) * p.b_ldc * 4u;      // descriptor base sits one row + one pixel early
    uint4 ra[WW_NJA], rb[WW_NJB];
    // tile-invariant staging offsets (used when !RAGGED) and the border masks of the halo pieces: bit j of mT / mB / mL /
    // mR = piece j of this lane is in the first / last halo row / column
    unsigned a_vo[RAGGED ? 1 : WW_NJA], b_vo[RAGGED ? 1 : WW_NJB];
    unsigned mT = 0, mB = 0, mL = 0, mR = 0;
    if constexpr (!RAGGED) {
        const int g = tid & 15;
#pragma unroll
        for (int j = 0; j < WW_NJA; ++j) {
            const int pix = (tid >> 4) + 16 * j, py = pix / (2 * WW_TX), px = pix % (2 * WW_TX);
            a_vo[j] = r0 + 4 * g < p.Rp ? (unsigned)(((py * p.W + px) * p.a_ldc + r0 + 4 * g) * 4) : BUF_OOB;
        }
#pragma unroll
        for (int j = 0; j < WW_NJB; ++j) {
            int pix = (tid >> 4) + 16 * j;
            if (pix >= WW_BPIX) pix -= WW_BPIX;
            const int hy = pix / WW_BW, hx = pix % WW_BW;
            b_vo[j] = c0 + 4 * g < p.Cp ? (unsigned)(((hy * p.W + hx) * p.b_ldc + c0 + 4 * g) * 4) : BUF_OOB;
            mT |= (hy == 0 ? 1u : 0u) << j; mB |= (hy == WW_BH - 1 ? 1u : 0u) << j;
            mL |= (hx == 0 ? 1u : 0u) << j; mR |= (hx == WW_BW - 1 ? 1u : 0u) << j;
        }
    }
    auto gload = [&](int tile, bool live) {                            // piece = tid + 256*j: 16 lanes = the 256 bytes of one pixel
        const int x0 = (tile % tiles_x) * (2 * WW_TX), y0 = ((tile / tiles_x) % tiles_y) * (2 * WW_TY), b = live ? tile / (tiles_x * tiles_y) : 0;
        const __amdgpu_buffer_rsrc_t ars = make_rsrc((const char*)p.a + (size_t)b * a_img, a_img);
        const __amdgpu_buffer_rsrc_t brs = make_rsrc((const char*)p.b + (size_t)b * b_img - b_shift, b_img + b_shift);
        const unsigned a_so = (unsigned)((y0 * p.W + x0) * p.a_ldc) * 4u, b_so = (unsigned)((y0 * p.W + x0) * p.b_ldc) * 4u;
        if constexpr (!RAGGED) {
            // a dead load (past the last tile of this split) gets an EMPTY descriptor: every lane out of range, no traffic
            const __amdgpu_buffer_rsrc_t ars2 = make_rsrc((const char*)p.a + (size_t)b * a_img, live ? a_img : 0u);
            const __amdgpu_buffer_rsrc_t brs2 = make_rsrc((const char*)p.b + (size_t)b * b_img - b_shift, live ? b_img + b_shift : 0u);
            const unsigned em = (y0 == 0 ? mT : 0u) | (y0 + 2 * WW_TY == p.H ? mB : 0u) | (x0 == 0 ? mL : 0u) | (x0 + 2 * WW_TX == p.W ? mR : 0u);
#pragma unroll
            for (int j = 0; j < WW_NJA; ++j) ra[j] = buf_ld16(ars2, a_vo[j], a_so);
#pragma unroll
            for (int j = 0; j < WW_NJB; ++j) rb[j] = buf_ld16(brs2, (em >> j) & 1u ? BUF_OOB : b_vo[j], b_so);
            return;
        }
        const int g = tid & 15;
#pragma unroll
        for (int j = 0; j < WW_NJA; ++j) {
            const int pix = (tid >> 4) + 16 * j, py = pix / (2 * WW_TX), px = pix % (2 * WW_TX);
            const bool ok = live && y0 + py < p.H && x0 + px < p.W && r0 + 4 * g < p.Rp;
            ra[j] = buf_ld16(ars, ok ? (unsigned)(((py * p.W + px) * p.a_ldc + r0 + 4 * g) * 4) : BUF_OOB, a_so);
        }
#pragma unroll
        for (int j = 0; j < WW_NJB; ++j) {
            int pix = (tid >> 4) + 16 * j;
            if (pix >= WW_BPIX) pix -= WW_BPIX;                        // the ragged last pass re-stages the first pixels
            const int hy = pix / WW_BW, hx = pix % WW_BW;
            const int yy = y0 + hy - 1, xx = x0 + hx - 1;
            const bool ok = live && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W && c0 + 4 * g < p.Cp;
            rb[j] = buf_ld16(brs, ok ? (unsigned)(((hy * p.W + hx) * p.b_ldc + c0 + 4 * g) * 4) : BUF_OOB, b_so);
        }
    };
    auto lds_store = [&](int st) {
        char* sa = smem + st * WW_STAGE;
        char* sb = sa + WW_ABYTES;
        const int g = tid & 15;
#pragma unroll
        for (int j = 0; j < WW_NJA; ++j) *reinterpret_cast<uint4*>(sa + ((tid >> 4) + 16 * j) * WW_PS + 16 * g) = ra[j];
#pragma unroll
        for (int j = 0; j < WW_NJB; ++j) {
            int pix = (tid >> 4) + 16 * j;
            if (pix >= WW_BPIX) pix -= WW_BPIX;
            *reinterpret_cast<uint4*>(sb + pix * WW_PS + 16 * g) = rb[j];
        }
    };

    // wave row i = w: coefficients of the two transforms
    // A rows (1,0), (1,1), (1,-1), (0,-1): rc[q] = gA[q] + c1f * g1[q] with gA = g0 (rows 0-2) or g1 (row 3, c1f = 0: the
    // row comes out as +g1 and the reduce flips plane 3), one fma per element with a per-wave LDS row instead of mul + fma
    const float c1f = w == 1 ? 1.f : (w == 2 ? -1.f : 0.f);
    const int a_rowA = w == 3 ? 2 * WW_TX * WW_PS : 0;                  // byte offset of the row that supplies gA
    const int a1 = w == 0 ? 0 : 1, a2 = w == 3 ? 3 : 2;                                          // B^T[w]: rows a1, a2
    // as in wino_kernel: every B^T row is d[a1] + s2 * d[a2]; row 2 comes out negated and wino_wgrad_reduce_kernel flips
    // plane 2 back (one fma per element instead of a multiply and an fma)
    const float s2 = w == 1 ? 1.f : -1.f;
    // lane part of every fragment address: tile parity kh -> 2 pixels to the right, channel r (+32 for the second half)
    const int a_lane = (2 * kh) * WW_PS + r * 4;
    const int b_lane1 = WW_ABYTES + (a1 * WW_BW + 2 * kh) * WW_PS + r * 4, b_lane2 = WW_ABYTES + (a2 * WW_BW + 2 * kh) * WW_PS + r * 4;

    f32x16 acc[4][2][2];                                               // [j][r half][c half]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i >> 1][i & 1][e] = 0.f;

    gload(t_begin, n > 0);
    if (n > 0) lds_store(0);
    __syncthreads();
#ifdef CLAMD_DIAG
    const unsigned long long dg0 = __builtin_amdgcn_s_memtime(), dr0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int it = 0; it < n; ++it) {
#ifndef WW_ABLATE_STAGING
        gload(t_begin + it + 1, it + 1 < n);                           // in flight under this tile's 256 MFMAs
        const char* sm = smem + (it & 1) * WW_STAGE;
#else
        const char* sm = smem;                                         // timing ablation: every tile multiplies stage 0
#endif
        // One MFMA k-step = tiles 2s, 2s+1 (neighbours in x); this lane's tile = (ty, tx2 + kh).  Software pipeline, one
        // wave per SIMD: the 24 raw reads of step s+1 are issued before the 16 MFMAs of step s and transformed between its
        // two halves, so no read is waited for with the matrix pipe empty (without this: 60 % MFMA utilisation).
        float raw[2][2][6];                                            // [set][channel half][g00 g01 g10 g11 | .. see below]
        float rawb[2][2][8];                                           // [set][channel half][row a1: 4 columns, row a2: 4 columns]
        float Yt[2][4][2], V[2][4][2];
#define WW_LOAD(s_, z_)                                                                                            \
    do {                                                                                                           \
        constexpr int ty_ = (s_) / (WW_TX / 2), tx2_ = 2 * ((s_) % (WW_TX / 2));                                   \
        constexpr int apix_ = (2 * ty_) * (2 * WW_TX) + 2 * tx2_, bpix_ = (2 * ty_) * WW_BW + 2 * tx2_;            \
        _Pragma("unroll") for (int hf_ = 0; hf_ < 2; ++hf_) {                                                      \
            const char* ap_ = sm + a_lane + apix_ * WW_PS + 128 * hf_;                                             \
            raw[z_][hf_][0] = *reinterpret_cast<const float*>(ap_ + a_rowA);                                       \
            raw[z_][hf_][1] = *reinterpret_cast<const float*>(ap_ + a_rowA + WW_PS);                               \
            raw[z_][hf_][2] = *reinterpret_cast<const float*>(ap_ + 2 * WW_TX * WW_PS);                            \
            raw[z_][hf_][3] = *reinterpret_cast<const float*>(ap_ + (2 * WW_TX + 1) * WW_PS);                      \
            const char* bp1_ = sm + b_lane1 + bpix_ * WW_PS + 128 * hf_;                                           \
            const char* bp2_ = sm + b_lane2 + bpix_ * WW_PS + 128 * hf_;                                           \
            _Pragma("unroll") for (int bc_ = 0; bc_ < 4; ++bc_) {                                                  \
                rawb[z_][hf_][bc_] = *reinterpret_cast<const float*>(bp1_ + bc_ * WW_PS);                          \
                rawb[z_][hf_][4 + bc_] = *reinterpret_cast<const float*>(bp2_ + bc_ * WW_PS);                      \
            }                                                                                                      \
        }                                                                                                          \
    } while (0)
#define WW_TRANSFORM(z_)                                                                                           \
    do {                                                                                                           \
        _Pragma("unroll") for (int hf_ = 0; hf_ < 2; ++hf_) {                                                      \
            const float rc0_ = fmaf(c1f, raw[z_][hf_][2], raw[z_][hf_][0]);                                        \
            const float rc1_ = fmaf(c1f, raw[z_][hf_][3], raw[z_][hf_][1]);                                        \
            Yt[z_][0][hf_] = rc0_; Yt[z_][1][hf_] = rc0_ + rc1_; Yt[z_][2][hf_] = rc0_ - rc1_; Yt[z_][3][hf_] = rc1_;  /* column 3: sign in the reduce */ \
            float t_[4];                                                                                           \
            _Pragma("unroll") for (int bc_ = 0; bc_ < 4; ++bc_) t_[bc_] = fmaf(s2, rawb[z_][hf_][4 + bc_], rawb[z_][hf_][bc_]); \
            V[z_][0][hf_] = t_[0] - t_[2]; V[z_][1][hf_] = t_[1] + t_[2]; V[z_][2][hf_] = t_[2] - t_[1]; V[z_][3][hf_] = t_[1] - t_[3]; \
        }                                                                                                          \
    } while (0)
#define WW_MMA(z_, j0_, j1_)                                                                                       \
    do {                                                                                                           \
        _Pragma("unroll") for (int j_ = (j0_); j_ < (j1_); ++j_)                                                   \
            _Pragma("unroll") for (int rh_ = 0; rh_ < 2; ++rh_)                                                    \
                _Pragma("unroll") for (int ch_ = 0; ch_ < 2; ++ch_)                                                \
                    acc[j_][rh_][ch_] = __builtin_amdgcn_mfma_f32_32x32x2f32(Yt[z_][j_][rh_], V[z_][j_][ch_], acc[j_][rh_][ch_], 0, 0, 0); \
    } while (0)
#define WW_STEP(s_)                                                                                                \
    do {                                                                                                           \
        if constexpr ((s_) + 1 < WW_TY * WW_TX / 2) WW_LOAD((s_) + 1, ((s_) + 1) & 1);                             \
        __builtin_amdgcn_sched_barrier(0);                 /* keep the reads in front of these 8 MFMAs ... */      \
        WW_MMA((s_) & 1, 0, 2);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                 /* ... and their consumers behind them */               \
        if constexpr ((s_) + 1 < WW_TY * WW_TX / 2) WW_TRANSFORM(((s_) + 1) & 1);                                  \
        WW_MMA((s_) & 1, 2, 4);                                                                                    \
        sched_mfma_slots<8, 0, 0, 0, 0, 0, 6>();           /* the ~42 transform VALU ops in the gaps of these 8 MFMAs */ \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    } while (0)
        WW_LOAD(0, 0);
        WW_TRANSFORM(0);
        WW_STEP(0); WW_STEP(1); WW_STEP(2); WW_STEP(3); WW_STEP(4); WW_STEP(5); WW_STEP(6); WW_STEP(7);
        WW_STEP(8); WW_STEP(9); WW_STEP(10); WW_STEP(11); WW_STEP(12); WW_STEP(13); WW_STEP(14); WW_STEP(15);
        static_assert(WW_TY * WW_TX / 2 == 16, "16 k-steps per pixel tile");
#undef WW_STEP
#undef WW_MMA
#undef WW_TRANSFORM
#undef WW_LOAD
#ifndef WW_ABLATE_STAGING
        if (it + 1 < n) lds_store((it + 1) & 1);                       // that stage was released by the last barrier
        __syncthreads();
#endif
    }

#ifdef CLAMD_DIAG
    if (lane == 0 && w == 0) {      // [0] shader cycles, [1] 100-MHz ticks of the tile loop, [2] tiles, [3] workgroups
        atomicAdd(&g_ww_diag[0], __builtin_amdgcn_s_memtime() - dg0);
        atomicAdd(&g_ww_diag[1], __builtin_amdgcn_s_memrealtime() - dr0);
        atomicAdd(&g_ww_diag[2], (unsigned long long)n);
        atomicAdd(&g_ww_diag[3], 1ull);
    }
#endif
    // ---- slab: plane i = w, [r][c][j]: lane (col c = 32*ch + r, rows acc_row(e, kh) + 32*rh) stores 16 bytes (j = 0..3)
    float* const plane = p.partial + (((size_t)split * 4 + w) * p.Rp + r0) * (size_t)p.Cp * 4;
#pragma unroll
    for (int rh = 0; rh < 2; ++rh)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int col = c0 + 32 * ch + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 32 * rh + acc_row(e, kh);
                if (r0 + row < p.Rp && col < p.Cp)
                    *reinterpret_cast<float4*>(plane + ((size_t)row * p.Cp + col) * 4) =
                        make_float4(acc[0][rh][ch][e], acc[1][rh][ch][e], acc[2][rh][ch][e], acc[3][rh][ch][e]);
            }
        }
}

// out[rl][cl][3][3] = G^T (sum_s dU_s) G.  256 threads = 16 (r,c) pairs x 4 planes x 4 split-phases.
struct WinoReduceParams {
    const float* partial; float* out;
    int nsplit, Rp, Cp, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p;
};

__global__ void __launch_bounds__(256) wino_wgrad_reduce_kernel(const WinoReduceParams p) {
    SIDE_PRIO();
    __shared__ float4 red[4][4][16];                                   // [phase][plane][pair]
    const int pr = threadIdx.x & 15, pl = (threadIdx.x >> 4) & 3, ph = threadIdx.x >> 6;
    const long long npair = (long long)p.Rp * p.Cp;
    const size_t plane_sz = (size_t)npair * 4, split_sz = plane_sz * 4;
    for (long long base = (long long)blockIdx.x * 16; base < npair; base += (long long)gridDim.x * 16) {
        const long long e = base + pr;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < npair)
            for (int k = ph; k < p.nsplit; k += 4) {
                const float4 v = *reinterpret_cast<const float4*>(p.partial + (size_t)k * split_sz + pl * plane_sz + (size_t)e * 4);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        red[ph][pl][pr] = s;
        __syncthreads();
        if (threadIdx.x < 16 && e < npair) {
            float U[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 a = red[0][i][pr], b = red[1][i][pr], c = red[2][i][pr], d = red[3][i][pr];
                U[i][0] = (a.x + b.x) + (c.x + d.x); U[i][1] = (a.y + b.y) + (c.y + d.y);
                U[i][2] = (a.z + b.z) + (c.z + d.z); U[i][3] = (a.w + b.w) + (c.w + d.w);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {       // signs left out of wino_wgrad_kernel's transforms: planes 2, 3 and column 3
                U[2][j] = -U[2][j];
                U[3][j] = -U[3][j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) U[i][3] = -U[i][3];
            // dg = G^T U G, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
            float t[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[0][j] = U[0][j] + 0.5f * (U[1][j] + U[2][j]);
                t[1][j] = 0.5f * (U[1][j] - U[2][j]);
                t[2][j] = U[3][j] + 0.5f * (U[1][j] + U[2][j]);
            }
            const int cp = (int)(e % p.Cp), rp = (int)(e / p.Cp);
            const int rl = wn_phys2log(rp, p.r_seg0, p.r_seg0p, p.R), cl = wn_phys2log(cp, p.c_seg0, p.c_seg0p, p.C);
            if (rl >= 0 && cl >= 0) {
                float* o = p.out + ((size_t)rl * p.C + cl) * 9;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    o[a * 3 + 0] = t[a][0] + 0.5f * (t[a][1] + t[a][2]);
                    o[a * 3 + 1] = 0.5f * (t[a][1] - t[a][2]);
                    o[a * 3 + 2] = t[a][3] + 0.5f * (t[a][1] + t[a][2]);
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace clamd

extern "C" {

#ifdef CLAMD_DIAG
int clamd_debug_ww_diag(unsigned long long* out4, int reset) {
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(clamd::g_ww_diag), 32) != hipSuccess) return -1;
    if (reset) { unsigned long long z[4] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(clamd::g_ww_diag), z, 32) != hipSuccess) return -1; }
    return 0;
}
#endif

size_t clamd_wgrad_winograd_workspace_bytes(int Rp, int Cp) {
    int nsplit = 512 / (((Rp + 63) / 64) * ((Cp + 63) / 64));      // upper bound over every value of wgrad_blocks (<= 1024)
    if (nsplit < 1) nsplit = 1;
    return (size_t)nsplit * 16 * Rp * Cp * sizeof(float);
}

int clamd_wgrad_winograd(const float* gz, int gz_ldc, const float* x, int x_ldc, float* workspace, size_t ws_bytes, float* out,
                         int B, int H, int W, int Rp, int Cp, int R, int C, int r_seg0, int r_seg0p, int c_seg0, int c_seg0p,
                         const clamd_tuning* tune, void* stream) {
    using namespace clamd;
    if (int e = clamd_check_tuning(tune)) return e;
    if (B <= 0 || H <= 0 || W <= 0) return clamd_fail("wgrad_winograd: empty problem");
    if ((H | W) & 1) return clamd_fail("wgrad_winograd: H and W must be even");
    if (Rp % 32 || Cp % 32 || gz_ldc % 8 || x_ldc % 8) return clamd_fail("wgrad_winograd: channel counts/pitches must be padded");
    if ((long long)H * W * gz_ldc * 4 >= (1ll << 30) || (long long)H * W * x_ldc * 4 >= (1ll << 30)) return clamd_fail("wgrad_winograd: one image exceeds 2^30 bytes");
    const int rt = (Rp + 63) / 64, ct = (Cp + 63) / 64;
    const int ntiles = ((W + 2 * WW_TX - 1) / (2 * WW_TX)) * ((H + 2 * WW_TY - 1) / (2 * WW_TY)) * B;
    // one workgroup per CU the grid may occupy by default (wgrad_blocks = 512 counts two per CU, as for the direct kernels);
    // a larger target trades split-K slab traffic for smaller work items (robust when RCCL channels hold CUs)
    const clamd_tuning& tn = clamd_tune(tune);
    int nsplit = (int)((long long)(tn.wgrad_blocks / 2) * clamd_usable_cus(tn) / clamd_num_cus()) / (rt * ct);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > ntiles) nsplit = ntiles;
    const int per = (ntiles + nsplit - 1) / nsplit;
    nsplit = (ntiles + per - 1) / per;
    if ((size_t)nsplit * 16 * Rp * Cp * sizeof(float) > ws_bytes) return clamd_fail("wgrad_winograd: workspace too small");
    WinoWgradParams p{gz, gz_ldc, x, x_ldc, workspace, B, H, W, Rp, Cp, nsplit, per};
    hipStream_t s = (hipStream_t)stream;
    if (H % (2 * WW_TY) || W % (2 * WW_TX)) hipLaunchKernelGGL(wino_wgrad_kernel<true>, dim3(rt * ct * nsplit), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(wino_wgrad_kernel<false>, dim3(rt * ct * nsplit), dim3(256), 0, s, p);
    if (int e = clamd_check_launch("wgrad_winograd")) return e;
    WinoReduceParams rp{workspace, out, nsplit, Rp, Cp, R, C, r_seg0, r_seg0p, c_seg0, c_seg0p};
    long long g = ((long long)Rp * Cp + 15) / 16;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, rp);
    return clamd_check_launch("wgrad_winograd_reduce");
}

}  // extern "C"
