// Winograd F(2x4, 3x3) convolution for the exact-fp32 path on gfx950: F(2,3) down the rows, F(4,3) along the columns.
//
// wino.hip (F(2x2,3x3)) executes 16 multiply-adds per 2x2 outputs = 4 per output; a 2x4 output tile with a 4x6 input
// patch costs 24 per 8 outputs = 3 per output -- 1.33x fewer fp32 MFMAs again (3x fewer than the direct sum).  The 2-D
// F(4x4) form (2.25 per output) does not map onto four waves (6 Winograd rows) and quadruples the output exchange; this
// hybrid keeps wino.hip's structure exactly -- wave w owns Winograd ROW i = w of the vertical F(2,3), the SIX horizontal
// positions j of F(4,3) live in one lane -- so the cross-wave exchange per output pixel is unchanged and only the in-lane
// transforms grow (1.5 instead of 1.0 transform VALU ops per MFMA).  fp32 error against fp64: 1e-6 (F(2x2): 3e-7, direct
// sum 2e-7); the kernels are held to the same 2e-5 test bound.
//
//   * filters: U[xi = 6i + j] = (G4 g G6^T)[i][j], stored [Cin_p/8][24][Cout_p][8] by wino24_pack_kernel (row 2 negated: the
//     sign of the kernel's row transform, as in wino.hip);
//   * workgroup = 256 threads, ONE per CU, 32 Winograd tiles x 64 output channels: 4 x 8 tiles = 8 x 32 output pixels
//     (TXN = 8) or 8 x 4 tiles = 16 x 16 pixels (TXN = 4, images narrower than 32); 192 accumulator registers;
//   * per 8-channel K-chunk: 10 x 34 (18 x 18) input halo + the 24 x 64 x 8 filter slab in LDS, two stages, the same
//     software pipeline as wino.hip (fragments of chunk k+1 read and transformed, chunk k+2 stored, chunk k+3 loaded in
//     the gaps of the 48 MFMAs of chunk k);
//   * halo image: slot(hy, hx) = hy * HW + hx + (hy >> 1).  A lane's tile origin is (2 ty, 4 tx): without the skew the 16
//     lanes of a ds_read_b128 group would share four 16-byte bank columns (4-way conflict); with it tile rows are an odd
//     number of slots apart and the 16 lanes hit 16 different columns.
#include <string.h>
#include <algorithm>
#include "common.hip.h"
#include "clamd_internal.h"
#include "wino_common.hip.h"

namespace clamd {

#ifdef CLAMD_DIAG
// diagnostic build only (python build.py --diag; tools/w24_diag.py): cycles per phase of a tile, summed over workgroups (wave 0)
__device__ unsigned long long g_w24_diag[8];
#define W24_T() __builtin_amdgcn_s_memtime()
#define W24_ADD(i_, v_) do { if (threadIdx.x == 0) atomicAdd(&g_w24_diag[i_], (unsigned long long)(v_)); } while (0)
#else
#define W24_T() 0ull
#define W24_ADD(i_, v_) do { } while (0)
#endif

constexpr int W24_WG = 66;                                    // padded rows per (xi, group)
constexpr int W24_WT_SLOTS = 24 * 2 * W24_WG;
constexpr int W24_EXP = 36;                                   // row pitch (floats) of the epilogue exchange block

template <int TXN, bool RAGGED, bool CLS = false>      // CLS: bias from a border-class table (WinoParams::bias_classes)
__global__ void __launch_bounds__(256, 1) wino24_kernel(const WinoParams p) {
    constexpr int TYN = 32 / TXN;                                         // tiles down x tiles across
    constexpr int PW = 4 * TXN, PH = 2 * TYN;                             // output pixels of the workgroup tile
    constexpr int HW_ = PW + 2, HH_ = PH + 2, PIX = HW_ * HH_;            // input halo
    constexpr int PIXMAX = (HH_ - 1) * HW_ + (HW_ - 1) + ((HH_ - 1) >> 1) + 1;   // slots incl. the row skew
    constexpr int PIXP = PIXMAX + ((10 - PIXMAX % 8) % 8);                // == 2 (mod 8): conflict-free staging stores
    constexpr int IN_SLOTS = 2 * PIXP, STAGE = IN_SLOTS + W24_WT_SLOTS;
    constexpr int NJI = (2 * PIX + 255) / 256;                            // input staging loads per thread
    constexpr int NJW = 12;                                               // filter staging loads per thread (24 xi x 64 x 2 / 256)
    constexpr int NST = 2;
    constexpr int EXB = 4 * 4 * 32 * W24_EXP;                             // floats of one nt exchange block [wave][q][tile][EXP]
    constexpr int LDS = (NST * STAGE * 16 > 2 * EXB * 4 ? NST * STAGE * 16 : 2 * EXB * 4) / 16;
    static_assert(LDS * 16 <= 160 * 1024, "LDS budget");
    __shared__ uint4 smem[LDS];

    // Statistics rows.  One workgroup per tile (gridDim.x == nblk): row = pixel tile, written once.  Persistent grid: row =
    // this workgroup; thread (kind, channel) keeps its running sum in a REGISTER across the tiles of one output slab and adds
    // it to its word of the row (zeroed here; plain load + store by the one thread that owns the word) when the slab changes
    // and at the end -- a read-modify-write per tile would drain the wave's memory counter behind the tile's output stores
    // (5k cycles per tile, measured).  The tile -> workgroup map is static, so the sums are bit-reproducible, and bn_finalize
    // adds <= 256 rows instead of one per tile (4096 at 64 channels x 256^2: 35 us per finalize launch).
    const bool per_wg_rows = p.stats != nullptr && gridDim.x < (unsigned)p.nblk;
    if (per_wg_rows && threadIdx.x < 128)        // thread (k, c) zeroes exactly the words it later accumulates into
        for (int n = threadIdx.x & 63; n < p.Np; n += 64) p.stats[((size_t)blockIdx.x * 2 + (threadIdx.x >> 6)) * p.Np + n] = 0.f;
    float racc = 0.f;
    int rslab = -1;
    auto flush_row = [&]() {          // threads < 128 only; wave-uniform call sites
        if (rslab >= 0 && rslab * 64 + (int)(threadIdx.x & 63) < p.Np) {
            float* dst = p.stats + ((size_t)blockIdx.x * 2 + (threadIdx.x >> 6)) * p.Np + rslab * 64 + (threadIdx.x & 63);
            // this thread's own earlier store to the word (zero fill or an earlier flush) must have reached L2: drain, read from L2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float old = __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst, old + racc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // Persistent grid: a thread's partial sums of its 4 + 4 channels stay in registers across the tiles of one output slab;
    // the cross-lane / cross-wave fold (48 shuffles, two barriers, 3.6k cycles: 6 % of a 64-channel tile) runs once per slab
    // and workgroup instead of once per tile.
    float st1[2][4], st2[2][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { st1[i >> 2][i & 3] = 0.f; st2[i >> 2][i & 3] = 0.f; }
    int cur_tn = -1, cur_tm = 0;                 // slab whose sums the registers hold (-1: none), last tile of it
    // every thread of the workgroup; LDS must be free (called between tiles and after the last one)
    auto fold_stats = [&]() {
        // threads with equal (tid & 7) own the same channels: fold the 8 tiles of a wave (lane bits 3-5), then the 4 waves
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
                if (cur_tn * 64 + c < p.Np) p.stats[((size_t)cur_tm * 2 + k) * p.Np + cur_tn * 64 + c] = t;   // row = pixel tile
            } else {
                rslab = cur_tn; racc = t;
                flush_row();
            }
        }
        __syncthreads();                                                       // the block is free again
        cur_tn = -1;
    };

    // Chunks 0 and 1 of the tile about to start.  The first tile of a workgroup loads them in its prologue; every later tile
    // finds them in these registers: they are requested right behind the previous tile's K loop (the staging registers
    // of the loop are dead there; the epilogue needs ~60 besides the accumulators), so the one exposed HBM latency per tile -- a
    // third of the 13k-cycle prologue, tools/w24_diag.py -- is spent under the exchange, read-back, output transform and stores.  (Carrying three chunks across the WHOLE epilogue spilled: 180 registers beside the accumulators.)
    uint4 pi0[NJI], pw0[NJW], pi1[NJI], pw1[NJW];
    bool pre = false;
    for (int v = blockIdx.x; v < p.nblk; v += gridDim.x) {
        // re-derived per tile from an opaque copy of the thread id (see wino.hip: hoisted constants would spill)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const unsigned long long dt0 = W24_T(); (void)dt0;
        const int lane = tid & 63, w = tid >> 6;
        const int r = lane & 31, h = lane >> 5;
        const int tiles_x = (p.W + PW - 1) / PW, tiles_y = (p.H + PH - 1) / PH;
        const int ntn = (p.Np + 63) >> 6;
        const int bid = xcd_remap(v, p.nblk);
        const int per_band = (p.nblk / ntn) * p.band;
        const int bnd = bid / per_band, rem = bid - bnd * per_band;
        const int tn = bnd * p.band + rem % p.band, tm = rem / p.band;
        const int x0 = (tm % tiles_x) * PW, y0 = ((tm / tiles_x) % tiles_y) * PH, b = tm / (tiles_x * tiles_y);
        const int n0 = tn * 64;
        const int nk = p.Kp >> 3;
        if (cur_tn >= 0 && (tn != cur_tn || !per_wg_rows)) fold_stats();      // workgroup-uniform

        // ---- staging descriptors --------------------------------------------------------------------------------------
        const unsigned img = (unsigned)p.H * (unsigned)p.W * (unsigned)p.x_ldc * 4u;
#ifdef W24_ABLATE_SAMETILE   // measurement only: every workgroup stages the same input tile (all loads hit the caches)
        const __amdgpu_buffer_rsrc_t xrs = make_rsrc((const char*)p.x, img);
#else
        const __amdgpu_buffer_rsrc_t xrs = make_rsrc((const char*)p.x + (size_t)b * img, img);
#endif
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
#ifdef W24_ABLATE_SAMETILE
            const int yy = hy - 1, xx = hx - 1;
#else
            const int yy = y0 + hy - 1, xx = x0 + hx - 1;
#endif
            in_vo[j] = (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) ? (unsigned)(((yy * p.W + xx) * p.x_ldc + 4 * g) * 4) : BUF_OOB;
            in_slot[j] = g * PIXP + hy * HW_ + hx + (hy >> 1);
        }
        // filter slab of one K-chunk: 24 xi x 64 rows x 2 groups = 3072 pieces, piece = tid + 256*j: xi = (tid >> 7) + 2j
        const int wg_ = tid & 1, wn_ = (tid >> 1) & 63, wxi0 = tid >> 7;
        const unsigned w_vo0 = n0 + wn_ < p.Np ? (unsigned)(((wxi0 * p.Np + n0 + wn_) * 8 + 4 * wg_) * 4) : BUF_OOB;
        const unsigned w_vstep = (unsigned)(2 * p.Np * 8 * 4);    // two xi further
        const unsigned w_chunk = (unsigned)(24 * p.Np * 8 * 4);   // bytes of one K-chunk
        const int w_slot0 = IN_SLOTS + (wxi0 * 2 + wg_) * W24_WG + wn_;

        uint4 rin[NJI], rw[NJW];
        auto gload_to = [&](int k, bool live, uint4 (&ri)[NJI], uint4 (&rww)[NJW]) {
            const unsigned so = (unsigned)(k * 8 * 4);
            const __amdgpu_buffer_rsrc_t xr = live ? xrs : xrs_dead, wr = live ? wrs : wrs_dead;
#pragma unroll
            for (int j = 0; j < NJI; ++j) ri[j] = buf_ld16(xr, in_vo[j], so);
#pragma unroll
            for (int j = 0; j < NJW; ++j) rww[j] = buf_ld16(wr, w_vo0, (unsigned)k * w_chunk + j * w_vstep);
        };
        auto lds_store_from = [&](int st, const uint4 (&ri)[NJI], const uint4 (&rww)[NJW]) {
            uint4* sm = smem + st * STAGE;
#pragma unroll
            for (int j = 0; j < NJI; ++j) sm[in_slot[j]] = ri[j];
#pragma unroll
            for (int j = 0; j < NJW; ++j) sm[w_slot0 + j * 4 * W24_WG] = rww[j];
        };
        auto gload = [&](int k, bool live) { gload_to(k, live, rin, rw); };
        auto lds_store = [&](int st) { lds_store_from(st, rin, rw); };

        // ---- fragment addressing: wave w = vertical Winograd row i: t[c] = d[a1][c] + s2 * d[a2][c] (row 2 negated, sign in
        // the packed filters) -------------------------------------------------------------------------------------------
        const int a1 = w == 0 ? 0 : 1, a2 = w == 3 ? 3 : 2;
        const float s2 = w == 1 ? 1.f : -1.f;
        const int ty = r / TXN, tx = r % TXN;
        // halo rows 2ty + a: slot = row * HW_ + col + (row >> 1), (2ty + a) >> 1 = ty + (a >> 1)
        const int pb = h * PIXP + (2 * ty) * HW_ + 4 * tx + ty;
        const int p1 = pb + a1 * HW_ + (a1 >> 1), p2 = pb + a2 * HW_ + (a2 >> 1);
        const int wb = IN_SLOTS + (6 * w * 2 + h) * W24_WG + r;              // + j*2*WG + 32*nt

        f32x16 acc[6][2];                                                     // [j][channel half]
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

        uint4 A[6], Bf[6][2];
        auto frags = [&](int st) {
            const uint4* sm = smem + st * STAGE;
            float4 t[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const uint4 u1 = sm[p1 + c], u2 = sm[p2 + c];
                t[c].x = fmaf(s2, __uint_as_float(u2.x), __uint_as_float(u1.x));
                t[c].y = fmaf(s2, __uint_as_float(u2.y), __uint_as_float(u1.y));
                t[c].z = fmaf(s2, __uint_as_float(u2.z), __uint_as_float(u1.z));
                t[c].w = fmaf(s2, __uint_as_float(u2.w), __uint_as_float(u1.w));
            }
            // B6^T: v0 = 4 t0 - 5 t2 + t4 | v1,2 = (t4 - 4 t2) +- (t3 - 4 t1) | v3,4 = (t4 - t2) +- 2 (t3 - t1) | v5 = 4 t1 - 5 t3 + t5
#define W24_COL(m_)                                                                                               \
    do {                                                                                                          \
        const float pq_ = fmaf(-4.f, t[2].m_, t[4].m_), qq_ = fmaf(-4.f, t[1].m_, t[3].m_);                       \
        const float rr_ = t[4].m_ - t[2].m_, ss_ = t[3].m_ - t[1].m_;                                             \
        o0.m_ = fmaf(4.f, t[0].m_, fmaf(-5.f, t[2].m_, t[4].m_));                                                 \
        o1.m_ = pq_ + qq_; o2.m_ = pq_ - qq_;                                                                     \
        o3.m_ = fmaf(2.f, ss_, rr_); o4.m_ = fmaf(-2.f, ss_, rr_);                                                \
        o5.m_ = fmaf(4.f, t[1].m_, fmaf(-5.f, t[3].m_, t[5].m_));                                                 \
    } while (0)
            float4 o0, o1, o2, o3, o4, o5;
            W24_COL(x); W24_COL(y); W24_COL(z); W24_COL(w);
#undef W24_COL
#define W24_PK(v_) make_uint4(__float_as_uint((v_).x), __float_as_uint((v_).y), __float_as_uint((v_).z), __float_as_uint((v_).w))
            A[0] = W24_PK(o0); A[1] = W24_PK(o1); A[2] = W24_PK(o2); A[3] = W24_PK(o3); A[4] = W24_PK(o4); A[5] = W24_PK(o5);
#undef W24_PK
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) Bf[j][nt] = sm[wb + j * 2 * W24_WG + 32 * nt];
        };

        // prologue: the first three chunks are requested back to back (one memory latency, not three); chunks 0 and 1 are
        // usually here already (see above)
        if (!pre) {                                                           // workgroup-uniform
            gload_to(0, true, pi0, pw0);
            gload_to(1, 1 < nk, pi1, pw1);
        }
        gload(2, 2 < nk);
        lds_store_from(0, pi0, pw0);
        lds_store_from(1, pi1, pw1);
        const unsigned long long dt1 = W24_T(); (void)dt1;
        __syncthreads();
        frags(0);
        __syncthreads();                                                      // stage 0 is free again
        const unsigned long long dt2 = W24_T(); (void)dt2;
        for (int k = 0; k < nk; ++k) {
            uint4 Ac[6], Bc[6][2];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                Ac[j] = A[j];
                Bc[j][0] = Bf[j][0]; Bc[j][1] = Bf[j][1];
            }
            frags((k + 1) % NST);                                             // visible since the last barrier
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) mma16<float>(Ac[j], Bc[j][nt], acc[j][nt]);
            lds_store(k % NST);                                               // chunk k+2 over chunk k's stage
            gload(k + 3, k + 3 < nk);
#ifndef W24_SCHED   // tools/w24_sched_ab.sh: F(2x4) over F(2x2) on the 14 layer shapes 1.158x (0), 1.115x (1), 1.099x (2), 1.030x (3), 1.058x (4)
#define W24_SCHED 0
#endif
#if W24_SCHED == 0
            sched_mfma_slots<48, 24, 25, 25 + NJI + NJW, 30, 30 + NJI + NJW, 2>();
#elif W24_SCHED == 1   // two reads per slot in the first 12, VALU from slot 4
            sched_mfma_slots<48, 12, 25, 25 + NJI + NJW, 30, 30 + NJI + NJW, 2, 4, 2>();
#elif W24_SCHED == 2   // two reads per slot, three VALU per slot from slot 8
            sched_mfma_slots<48, 12, 25, 25 + NJI + NJW, 30, 30 + NJI + NJW, 3, 8, 2>();
#elif W24_SCHED == 3   // one read per slot, VALU from slot 4
            sched_mfma_slots<48, 24, 25, 25 + NJI + NJW, 30, 30 + NJI + NJW, 2, 4, 1>();
#elif W24_SCHED == 4   // one read per slot, three VALU per slot from slot 12
            sched_mfma_slots<48, 24, 28, 28 + NJI + NJW, 30, 30 + NJI + NJW, 3, 12, 1>();
#endif
            __syncthreads();
        }

        // ---- epilogue: Y = A4^T M A6, A6^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]] in-lane (j -> q),
        // A4^T = [[1,1,1,0],[0,1,-1,-1]] across the four waves through LDS -------------------------------------------------
        const unsigned long long dt3 = W24_T(); (void)dt3;
        {   // chunks 0 and 1 of this workgroup's next tile (descriptors as at the top of the loop)
            // issued unconditionally -- behind the last tile with empty descriptors (zeros, no memory traffic) -- so that the
            // registers are redefined on every path: a conditional prefetch would make their OLD contents live through the
            // whole tile (the compiler cannot know that "no next tile" ends the loop) and spill
            const int vn = v + (int)gridDim.x;
            pre = vn < p.nblk;
            {
                const int bidn = xcd_remap(pre ? vn : v, p.nblk);
                const int bndn = bidn / per_band, remn = bidn - bndn * per_band;
                const int tnn = bndn * p.band + remn % p.band, tmn = remn / p.band;
                const int x0n = (tmn % tiles_x) * PW, y0n = ((tmn / tiles_x) % tiles_y) * PH, bn = tmn / (tiles_x * tiles_y);
#ifdef W24_ABLATE_SAMETILE
                const __amdgpu_buffer_rsrc_t xrn = pre ? make_rsrc((const char*)p.x, img) : xrs_dead;
#else
                const __amdgpu_buffer_rsrc_t xrn = pre ? make_rsrc((const char*)p.x + (size_t)bn * img, img) : xrs_dead;
#endif
                const __amdgpu_buffer_rsrc_t wrn = pre ? wrs : wrs_dead;
#pragma unroll
                for (int j = 0; j < NJI; ++j) {
                    int piece = tid + 256 * j;
                    if (piece >= 2 * PIX) piece -= 2 * PIX;
                    const int g = piece & 1, pix = piece >> 1;
                    const int hy = pix / HW_, hx = pix - hy * HW_;
#ifdef W24_ABLATE_SAMETILE
                    const int yy = hy - 1, xx = hx - 1;
#else
                    const int yy = y0n + hy - 1, xx = x0n + hx - 1;
#endif
                    const unsigned vo = (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) ? (unsigned)(((yy * p.W + xx) * p.x_ldc + 4 * g) * 4) : BUF_OOB;
                    pi0[j] = buf_ld16(xrn, vo, 0u);
                    pi1[j] = buf_ld16(xrn, vo, 32u);
                }
                const unsigned wvn = tnn * 64 + wn_ < p.Np ? (unsigned)(((wxi0 * p.Np + tnn * 64 + wn_) * 8 + 4 * wg_) * 4) : BUF_OOB;
#pragma unroll
                for (int j = 0; j < NJW; ++j) {
                    pw0[j] = buf_ld16(wrn, wvn, j * w_vstep);
                    pw1[j] = buf_ld16(wrn, wvn, w_chunk + j * w_vstep);
                }
            }
        }
        float* const ex = reinterpret_cast<float*>(smem);
        const int tl = tid >> 3, ng = tid & 7;                                // reader: tile, 4-channel group
        const float relu_lo = p.relu ? 0.f : -__builtin_inff();
        // data-gradient launches have no bias, ReLU or statistics: 16 VALU per stored float4 less (the wave owns its SIMD, so
        // every VALU instruction of the epilogue is time the matrix pipe idles)
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
                exb[((w * 4 + 0) * 32 + row) * W24_EXP + r] = m0 + sa + sc;
                exb[((w * 4 + 1) * 32 + row) * W24_EXP + r] = fmaf(2.f, sd, sb);
                exb[((w * 4 + 2) * 32 + row) * W24_EXP + r] = fmaf(4.f, sc, sa);
                exb[((w * 4 + 3) * 32 + row) * W24_EXP + r] = fmaf(8.f, sd, sb) + m5;
            }
        }
        __syncthreads();
        const unsigned long long dt4 = W24_T(); (void)dt4;
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
                for (int i = 0; i < 4; ++i) R[i] = *reinterpret_cast<const float4*>(exb + ((i * 4 + q) * 32 + tl) * W24_EXP + 4 * ng);
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
                    if (!RAGGED || (yy < p.H && xx < p.W && n < p.Np)) {
#ifndef W24_ABLATE_ST
                        *reinterpret_cast<float4*>(p.y + (((size_t)b * p.H + yy) * p.W + xx) * p.y_ldc + n) = o;
#endif
                        if (!plain) {
                            st1[nt][0] += o.x; st1[nt][1] += o.y; st1[nt][2] += o.z; st1[nt][3] += o.w;
                            st2[nt][0] = fmaf(o.x, o.x, st2[nt][0]); st2[nt][1] = fmaf(o.y, o.y, st2[nt][1]);
                            st2[nt][2] = fmaf(o.z, o.z, st2[nt][2]); st2[nt][3] = fmaf(o.w, o.w, st2[nt][3]);
                        }
                    }
                }
            }
        }
        const unsigned long long dt5 = W24_T(); (void)dt5;
        if (p.stats) { cur_tn = tn; cur_tm = tm; }
        __syncthreads();                                                   // exchange / statistics blocks are free again
        // [0] setup + first loads + first stores  [1] wait first stage + first fragments  [2] K loop  [3] exchange writes + barrier
        // [4] read-back, output transform, stores  [5] statistics  [6] tiles  [7] K chunks
        W24_ADD(0, dt1 - dt0); W24_ADD(1, dt2 - dt1); W24_ADD(2, dt3 - dt2); W24_ADD(3, dt4 - dt3); W24_ADD(4, dt5 - dt4);
        W24_ADD(5, W24_T() - dt5); W24_ADD(6, 1); W24_ADD(7, nk);
    }
    if (cur_tn >= 0) fold_stats();
}

// ---- filter transform: dst[(k/8)*24 + 6i + j][n][k%8] = (G4 g G6^T)[i][j], row 2 negated (jobs as in wino.hip) ---------
__global__ void __launch_bounds__(256) wino24_pack_kernel(const WinoPackJob* __restrict__ jobs, int njobs, int nblocks, const FoldBias fold) {
    if ((int)blockIdx.x >= nblocks) {       // appended blocks: the border-class bias table of a folded BatchNorm (common.hip.h)
        __shared__ float T[9];
        fold_bias_block(fold, (int)blockIdx.x - nblocks, T);
        return;
    }
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
    // rows: G4 = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] (row 2 carries the sign of the kernel's row transform)
    float t[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        t[0][c] = g[0][c];
        t[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
        t[2][c] = -0.5f * (g[0][c] - g[1][c] + g[2][c]);
        t[3][c] = g[2][c];
    }
    float* d = J.dst + ((size_t)kc * 24 * J.Np + n) * 8 + k8;
    const size_t xs = (size_t)J.Np * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // columns: G6 = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
        const float a = t[i][0], b = t[i][1], c = t[i][2];
        d[(6 * i + 0) * xs] = 0.25f * a;
        d[(6 * i + 1) * xs] = (-1.f / 6.f) * (a + b + c);
        d[(6 * i + 2) * xs] = (-1.f / 6.f) * (a - b + c);
        d[(6 * i + 3) * xs] = (1.f / 24.f) * a + (1.f / 12.f) * b + (1.f / 6.f) * c;
        d[(6 * i + 4) * xs] = (1.f / 24.f) * a - (1.f / 12.f) * b + (1.f / 6.f) * c;
        d[(6 * i + 5) * xs] = c;
    }
}

}  // namespace clamd

using namespace clamd;

int clamd_launch_wino24_pack(const void* jobs_dev, int njobs, int total_blocks, const clamd::FoldBias* fold, hipStream_t stream) {
    if (njobs <= 0 || total_blocks <= 0) return clamd_fail("wino24_pack: empty job table");
    const clamd::FoldBias f = fold ? *fold : clamd::FoldBias{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 9};
    hipLaunchKernelGGL(clamd::wino24_pack_kernel, dim3(total_blocks + (fold ? fold->Cout_p : 0)), dim3(256), 0, stream, (const clamd::WinoPackJob*)jobs_dev, njobs,
                       total_blocks, f);
    return clamd_check_launch("wino24_pack");
}

extern "C" {

int clamd_wino24_pack(const void* jobs_dev, int njobs, int total_blocks, void* stream) {
    return clamd_launch_wino24_pack(jobs_dev, njobs, total_blocks, nullptr, (hipStream_t)stream);
}

#ifdef CLAMD_DIAG
int clamd_debug_w24_diag(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(clamd::g_w24_diag), 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(clamd::g_w24_diag), z, 64) != hipSuccess) return -1; }
    return 0;
}
#endif

int clamd_conv3x3_winograd24(const float* x, int x_ldc, const float* w_wino, const float* bias, float* y, int y_ldc,
                             float* stats, int stat_rows, int B, int H, int W, int Cin_p, int Cout_p, int relu,
                             const clamd_tuning* tune, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return clamd_fail("conv3x3_winograd24: empty problem");
    if ((H & 1) || (W & 3)) return clamd_fail("conv3x3_winograd24: H must be even and W a multiple of 4 (2x4 output tiles)");
    if (Cin_p % 32 || Cout_p % 32 || x_ldc % 8 || y_ldc % 8) return clamd_fail("conv3x3_winograd24: channel counts/pitches must be padded");
    if ((long long)H * W * x_ldc * 4 >= (1ll << 31) || (long long)24 * Cout_p * Cin_p * 4 >= (1ll << 31))
        return clamd_fail("conv3x3_winograd24: image or filter exceeds 2^31 bytes");
    if (int e = clamd_check_tuning(tune)) return e;
    if ((relu & ~3) || ((relu & CLAMD_BIAS_BORDER_CLASSES) && !bias)) return clamd_fail("conv3x3_winograd24: bad relu flags (bit 1 needs the [9][Cout_p] bias table)");
    WinoParams p{x, x_ldc, w_wino, bias, y, y_ldc, stats, B, H, W, Cin_p, Cout_p, relu & 1, 1, 0};
    p.bias_classes = (relu & CLAMD_BIAS_BORDER_CLASSES) ? 1 : 0;
    return launch_wino24(p, clamd_tune(tune), stat_rows, (hipStream_t)stream);
}

}  // extern "C"

namespace clamd {

// workgroup tile: 8 x 32 pixels, or 16 x 16 for images narrower than 32
static inline void w24_tile(int W, int& ph, int& pw) { if (W >= 32) { ph = 8; pw = 32; } else { ph = 16; pw = 16; } }

static long long w24_tiles(int B, int H, int W) {
    int ph, pw;
    w24_tile(W, ph, pw);
    return (long long)B * ((H + ph - 1) / ph) * ((W + pw - 1) / pw);
}

// rows of a launch: one per pixel tile (one workgroup per tile), or one per workgroup of the persistent grid
long long clamd_winograd24_stat_rows(int B, int H, int W, int Cout_p, const clamd_tuning& tn) {
    if (tn.wino_half) return clamd_winograd24_half_stat_rows(B, H, W, Cout_p, tn);
    const long long tiles = w24_tiles(B, H, W), nblk = tiles * ((Cout_p + 63) / 64);
    return (tn.wino_persist && nblk > clamd_usable_cus(tn)) ? clamd_usable_cus(tn) : tiles;
}

int launch_wino24(WinoParams p, const clamd_tuning& tn, int stat_rows, hipStream_t stream) {
    if (tn.wino_half) return launch_wino24_half(p, tn, stat_rows, stream);
    int ph, pw;
    w24_tile(p.W, ph, pw);
    const long long ntn = (p.Np + 63) / 64, tiles = w24_tiles(p.B, p.H, p.W);
    if (tiles * ntn > 0x7fffffff) return clamd_fail("conv3x3_winograd24: grid out of range");
    if (p.stats && stat_rows != clamd_winograd24_stat_rows(p.B, p.H, p.W, p.Np, tn))
        return clamd_fail("conv3x3_winograd24: stat_rows does not match clamd_stat_rows(CLAMD_OP_CONV3X3_WINOGRAD24, ...)");
    p.band = wino_band(tiles, ntn, (double)p.B * p.H * p.W * p.Kp, 24.0 * p.Kp * p.Np, tn.wino_band);
    p.nblk = (int)(tiles * ntn);
    const unsigned grid = tn.wino_persist ? (unsigned)std::min<long long>(p.nblk, clamd_usable_cus(tn)) : (unsigned)p.nblk;
    const bool ragged = (p.H % ph) != 0 || (p.W % pw) != 0 || (p.Np % 64) != 0;
#define W24_LAUNCH(TXN_, RG_)                                                                                          \
    do {                                                                                                               \
        if (p.bias_classes) hipLaunchKernelGGL((wino24_kernel<TXN_, RG_, true>), dim3(grid), dim3(256), 0, stream, p);  \
        else hipLaunchKernelGGL((wino24_kernel<TXN_, RG_, false>), dim3(grid), dim3(256), 0, stream, p);               \
    } while (0)
    if (pw == 32) { if (ragged) W24_LAUNCH(8, true); else W24_LAUNCH(8, false); }
    else { if (ragged) W24_LAUNCH(4, true); else W24_LAUNCH(4, false); }
#undef W24_LAUNCH
    return clamd_check_launch("conv3x3_winograd(F(2x4))");
}

}  // namespace clamd
