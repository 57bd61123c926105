// bf16 3x3 weight-gradient kernel with LDS-DMA staging (gfx950 `buffer_load_dwordx4 ... lds`).
//
// Same GEMM, tile (64 r x 64 c x 9 taps per workgroup, 128-pixel tiles), producer/consumer roles, split-K slabs and
// results as wgrad.hip's WS kernel.  What changes is how a pixel tile reaches LDS.  Measured there (tools/pmc_sq.py,
// ablation build -DWG_ABLATE_STORE): the kernel is LDS-throughput-bound -- 2.2 ds_read_b64_tr_b16 per MFMA from the
// consumers plus the producers' ds_write_b128 (13 cycles per KB: the VGPR->LDS transfer, not the LDS array) keep the
// LDS ~90 % busy; with 10 of 11 staging stores removed it ran 21 % faster.  Here the producers issue LDS-DMA loads:
// no VGPR round trip, no ds_write, the LDS array is written at its own rate.
//
// LDS-DMA writes `wave-uniform base + lane * 16`, so the image cannot be padded per pixel (192-byte stride of wgrad.hip);
// it is [pixel][128 B] with the two 64-byte channel halves of a pixel swapped when bit 1 of the pixel index is set
// (applied to the per-lane SOURCE address; LDS stays lane-linear).  A transposed read touches 4 consecutive pixels x
// 64 bytes per 32-lane group: with the swap those land on four different 64-byte quarters of the 256-byte bank row for
// ANY start pixel -- conflict-free like the padded image.  The swap bit of a read is bit 1 of (start pixel + lane pixel);
// start pixel mod 4 is a compile-time constant per (tile-row parity, tap), so the consumers keep 4 lane-constant offsets
// and spend no address arithmetic in the loop.
//
// Three LDS stages (3 x 42 KB): tile i+2 is in flight while tile i is multiplied; the producers wait with a counted
// s_waitcnt vmcnt(N) (their own 11 DMA instructions of the newest tile stay in flight) and meet the consumers at a raw
// s_barrier -- __syncthreads() would drain the DMA (it is a pending LDS write on the VM counter).
#include "common.hip.h"
#include "wgrad_common.hip.h"
#include "clamd_internal.h"

namespace clamd {

typedef __attribute__((address_space(3))) void lds_void;

// 16 bytes per lane from a buffer straight into LDS at (wave-uniform) lds + lane*16; out-of-range lanes write zeros.
// A __device__ function on purpose: called directly inside a lambda of a kernel template, the target builtin makes the
// host pass drop the kernel's stub without a diagnostic (undefined __device_stub__ at load time).
__device__ inline void dma16(__amdgpu_buffer_rsrc_t rs, char* lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds, 16, voff, soff, 0, 0);
}

template <int TW>
__global__ void __launch_bounds__(512, 2) wgrad_dma_kernel(const WgradParams p) {
    using G = WGeo<bf16_t, WG_CONV3, TW>;
    constexpr int TH = G::TH, NT = 9, BW = G::BW, APIX = G::APIX, BPIX = G::BPIX;
    constexpr int PS = 128;                                   // bytes per pixel: 64 bf16 channels, unpadded
    constexpr int A_CH = APIX * 8 / 64;                       // 1-KB chunks (8 pixels) of the A image
    constexpr int B_CH = (BPIX * 8 + 63) / 64;
    constexpr int NJA = A_CH / 4, NJB = (B_CH + 3) / 4;       // DMA instructions per producer wave and tile
    constexpr int NDMA = NJA + NJB;
    constexpr int A_BYTES = A_CH * 1024, B_BYTES = B_CH * 1024, STAGE = A_BYTES + B_BYTES, NST = 3;
    constexpr int EPI_BYTES = 4 * 8 * 32 * NT * 4;
    static_assert(A_CH % 4 == 0 && TH % 2 == 0, "tile geometry");
    static_assert(EPI_BYTES <= NST * STAGE && NST * STAGE + 1024 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE + 1024];    // + one dummy chunk for padding DMAs

    const int tid = threadIdx.x & 255;                        // index inside the role
    const int lane = tid & 63, wave = tid >> 6;
    const bool producer = threadIdx.x >= 256;
    const int wr = wave >> 1, wc = wave & 1;

    const int rt = (p.Rp + 63) >> 6, ct = (p.Cp + 63) >> 6;
    int bid = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int tr = bid % rt; bid /= rt;
    const int tc = bid % ct; bid /= ct;
    const int split = bid;
    const int r0 = tr * 64, c0 = tc * 64;

    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * p.B;
    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(ntiles, t_begin + p.tiles_per_split);
    const int n = max(t_end - t_begin, 0);

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    if (producer) {
        // ---------------------------------------------------------------------- producers: global -> LDS by DMA
        // lane -> piece: chunk c = 4*j + wave holds pixels 8c..8c+7, lane>>3 = pixel inside the chunk, lane&7 = 16-byte
        // slot; the slot holds channel group (slot ^ 4) when bit 1 of the pixel index is set
        unsigned a_vo[NJA], b_vo[NJB], b_hyx[NJB];
#pragma unroll
        for (int j = 0; j < NJA; ++j) {
            const int pix = 8 * (4 * j + wave) + (lane >> 3);
            const int gs = (lane & 7) ^ (((pix >> 1) & 1) << 2);
            a_vo[j] = r0 + gs * 8 < p.Rp ? (unsigned)((((pix / TW) * p.W + pix % TW) * p.a_ldc + r0 + gs * 8) * 2) : BUF_OOB;
        }
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
            const int c = 4 * j + wave, pix = 8 * c + (lane >> 3);
            const int gs = (lane & 7) ^ (((pix >> 1) & 1) << 2);
            const int hy = pix / BW, hx = pix % BW;
            b_hyx[j] = ((unsigned)hy << 16) | (unsigned)hx;
            b_vo[j] = (c < B_CH && pix < BPIX && c0 + gs * 8 < p.Cp) ? (unsigned)(((hy * p.W + hx) * p.b_ldc + c0 + gs * 8) * 2) : BUF_OOB;
        }
        const unsigned a_img = (unsigned)p.H * p.W * p.a_ldc * 2, b_img = (unsigned)p.H * p.W * p.b_ldc * 2;
        const unsigned b_shift = (unsigned)(p.W + 1) * p.b_ldc * 2;    // descriptor base sits one row + one pixel early

        auto dma_tile = [&](int tile, int st) {
            const int x0 = (tile % tiles_x) * TW, y0 = ((tile / tiles_x) % tiles_y) * TH, b = tile / (tiles_x * tiles_y);
            const __amdgpu_buffer_rsrc_t ars = make_rsrc((const char*)p.a + (size_t)b * a_img, a_img);
            const __amdgpu_buffer_rsrc_t brs = make_rsrc((const char*)p.b + (size_t)b * b_img - b_shift, b_img + b_shift);
            const unsigned a_so = (unsigned)((y0 * p.W + x0) * p.a_ldc * 2), b_so = (unsigned)((y0 * p.W + x0) * p.b_ldc * 2);
            char* const sbase = smem + st * STAGE;
#pragma unroll
            for (int j = 0; j < NJA; ++j) {
                const int pix = 8 * (4 * j + wave) + (lane >> 3);
                const bool ok = y0 + pix / TW < p.H && x0 + pix % TW < p.W;
                dma16(ars, sbase + (4 * j + wave) * 1024, ok ? a_vo[j] : BUF_OOB, a_so);
            }
#pragma unroll
            for (int j = 0; j < NJB; ++j) {
                const int c = 4 * j + wave;
                const int yy = y0 + (int)(b_hyx[j] >> 16) - 1, xx = x0 + (int)(b_hyx[j] & 0xffffu) - 1;
                const bool ok = yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                char* const dst = c < B_CH ? sbase + A_BYTES + c * 1024 : smem + NST * STAGE;    // wave-uniform
                dma16(brs, dst, ok ? b_vo[j] : BUF_OOB, b_so);
            }
        };

        if (n > 0) dma_tile(t_begin, 0);
        if (n > 1) {
            dma_tile(t_begin + 1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");      // tile 0 has landed, tile 1 is in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                          // #0: stage 0 is ready
        for (int i = 0; i < n; ++i) {                          // consumers multiply tile i (stage i % 3)
            if (i + 2 < n) {
                dma_tile(t_begin + i + 2, (i + 2) % NST);      // the stage of tile i-1: released at barrier #i
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");  // tile i+1 has landed
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                      // #(i+1)
        }
    } else {
        // ---------------------------------------------------------------------- consumers: transposed reads -> MFMA
        const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const int kq = 8 * (gq >> 1) + q;                      // pixel of this lane inside a 16-pixel k-step (+4 for the 2nd read)
        const int within = (16 * (gq & 1) + 4 * pp) * 2;       // byte offset inside the wave's 64-byte channel half
        const int a_off = kq * PS + ((wr ^ ((kq >> 1) & 1)) << 6) + within;      // A tile rows start at multiples of 16 pixels
        int b_off[4];                                          // by (start pixel of the read) mod 4
#pragma unroll
        for (int s = 0; s < 4; ++s) b_off[s] = kq * PS + ((wc ^ (((s + (kq & 3)) >> 1) & 1)) << 6) + within;

        constexpr int XPR = TW / 16, NG = 6 * XPR;             // k-steps per tile row; groups (k-step, tap row) per 2 tile rows
        uint4 Ah[2], Bh[2][3];
        // group g of a 2-row iteration: rp = g / (3*XPR) (row inside the pair), xsi = (g/3) % XPR, dy = g % 3.
        // B start pixel = (row+dy)*BW + 16*xsi + dx, BW = 2 (mod 4), ty0 even  =>  (start mod 4) = (2*((rp+dy)&1) + dx) & 3.
#define WD_LOADG(g_)                                                                                               \
    do {                                                                                                           \
        constexpr int rp_ = (g_) / (3 * XPR), xsi_ = ((g_) / 3) % XPR, dy_ = (g_) % 3;                             \
        constexpr int par_ = ((g_) / 3) & 1, slot_ = (g_) & 1;                                                     \
        if constexpr (dy_ == 0) {                                                                                  \
            const char* ap_ = sa + ((ty0 + rp_) * TW + 16 * xsi_) * PS + a_off;                                    \
            const uint2 a0_ = ds_tr16(ap_), a1_ = ds_tr16(ap_ + 4 * PS);                                           \
            Ah[par_] = make_uint4(a0_.x, a0_.y, a1_.x, a1_.y);                                                     \
        }                                                                                                          \
        _Pragma("unroll") for (int dx_ = 0; dx_ < 3; ++dx_) {                                                      \
            const int s_ = (2 * ((rp_ + dy_) & 1) + dx_) & 3;                                                      \
            const char* bp_ = sb + ((ty0 + rp_ + dy_) * BW + 16 * xsi_ + dx_) * PS + b_off[s_];                    \
            const uint2 b0_ = ds_tr16(bp_), b1_ = ds_tr16(bp_ + 4 * PS);                                           \
            Bh[slot_][dx_] = make_uint4(b0_.x, b0_.y, b1_.x, b1_.y);                                               \
        }                                                                                                          \
    } while (0)
#define WD_STEP(g_)                                                                                                \
    do {                                                                                                           \
        WD_LOADG((g_) + 1);                                                        /* prefetch the next group */  \
        constexpr int par_s = ((g_) / 3) & 1, sl_ = (g_) & 1, dy_s = (g_) % 3;                                     \
        _Pragma("unroll") for (int dx = 0; dx < 3; ++dx) mma_bf16(Ah[par_s], Bh[sl_][dx], acc[3 * dy_s + dx]);     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + ((((g_) + 1) % 3 == 0) ? 2 : 0), 0);                       \
    } while (0)

        __builtin_amdgcn_s_barrier();                          // #0: stage 0 is ready
        for (int i = 0; i < n; ++i) {
            const char* const sa = smem + (i % NST) * STAGE;
            const char* const sb = sa + A_BYTES;
            {
                const int ty0 = 0;
                WD_LOADG(0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0); // the first group's reads lead
#pragma unroll 1
            for (int ty0 = 0; ty0 < TH; ty0 += 2) {
                WD_STEP(0); WD_STEP(1); WD_STEP(2); WD_STEP(3); WD_STEP(4); WD_STEP(5);
                if constexpr (NG == 12) { WD_STEP(6); WD_STEP(7); WD_STEP(8); WD_STEP(9); WD_STEP(10); WD_STEP(11); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // every read of this stage has returned
            __builtin_amdgcn_s_barrier();                      // #(i+1): hand the stage back
        }
#undef WD_STEP
#undef WD_LOADG
    }

    // ---- store the partial slab [split][r][c][t]: as wgrad.hip (8 rows x 32 cols x 9 taps per pass through LDS) ----
    __syncthreads();
    float* const wbuf = reinterpret_cast<float*>(smem) + wave * (8 * 32 * NT);
    const bool wave_in = r0 + 32 * wr < p.Rp && c0 + 32 * wc < p.Cp;
    float* const slab = p.partial + (((size_t)split * p.Rp + r0 + 32 * wr) * p.Cp + c0 + 32 * wc) * NT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j) __syncthreads();
        if (!producer) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int ee = 0; ee < 4; ++ee)
                    wbuf[(((lane >> 5) * 4 + ee) * 32 + (lane & 31)) * NT + t] = acc[t][4 * j + ee];
        }
        __syncthreads();
        if (!producer && wave_in) {
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int f = 4 * (k * 64 + lane);
                const int row8 = f / (32 * NT), rem = f % (32 * NT);
                *reinterpret_cast<float4*>(slab + ((size_t)(8 * j + row8) * p.Cp) * NT + rem) =
                    *reinterpret_cast<const float4*>(wbuf + f);
            }
        }
    }
}

int launch_wgrad_dma(const WgradParams& p, hipStream_t s, int grid, int tw) {
    if (tw == 32) hipLaunchKernelGGL((wgrad_dma_kernel<32>), dim3(grid), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((wgrad_dma_kernel<16>), dim3(grid), dim3(512), 0, s, p);
    return clamd_check_launch("wgrad_dma");
}

}  // namespace clamd
