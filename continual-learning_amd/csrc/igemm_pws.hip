// Persistent producer/consumer implicit-GEMM 3x3 convolution for gfx950: the short-K layers (64 / 128 input channels).
//
// igemm_ws.hip runs ONE 512-thread workgroup per CU (two 60-KB LDS stages), so nothing covers a workgroup's prologue
// (tile descriptors, first two stages from HBM) and epilogue; with 2-4 K-steps per tile those cost more than the
// K loop, and the 2-workgroups-per-CU baseline kernel (igemm.hip) stayed faster there.  Here a workgroup is launched
// once per CU and walks its share of the pixel tiles of ONE 64-channel output slab:
//   * waves 4-7 (producers) run ONE flattened (tile, K-step) pipeline: while the consumers finish a tile and write
//     it out, the first stage of the next tile is already in LDS and the second in flight;
//   * waves 0-3 (consumers) do MFMAs, then an epilogue without workgroup barriers: every wave transposes its own
//     32x64 accumulator block through a private LDS region outside the two stages;
//   * the per-channel Sigma / Sigma^2 of the fused BatchNorm statistics stay in registers across tiles and are flushed
//     once per workgroup as ONE partial row (plain stores, row = index of the workgroup inside its slab): the tile ->
//     workgroup map is static, so the rows and their fixed-order sum (bn_finalize) are bit-reproducible.
// Tile (256 pixels x 64 channels), LDS image, fragment maps and per-element results are those of igemm_ws.hip (MT = 2).
#include "common.hip.h"
#include "igemm_common.hip.h"
#include "clamd_internal.h"

#ifndef PWS_EPW
#define PWS_EPW 68
#endif
#if defined(PWS_ABLATE_EPI) && PWS_ABLATE_EPI == 3     // timing ablation (wrong results): no statistics in the channels-in-the-lane epilogue
#define PWS_ABLATE_STATS true
#else
#define PWS_ABLATE_STATS false
#endif
namespace clamd {

#ifdef CLAMD_DIAG
// diagnostic build only (python build.py --diag): consumer-side cycle shares, summed over workgroups
__device__ unsigned long long g_pws_diag[8];
#define PWD_T() __builtin_amdgcn_s_memtime()
#define PWD_ADD(i_, v_) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_pws_diag[i_], (unsigned long long)(v_)); } while (0)
#else
#define PWD_T() 0ull
#define PWD_ADD(i_, v_) do { } while (0)
#endif

// RAGGED = false (image height and width are multiples of the tile): no edge masks anywhere.  With one kernel for both
// cases hipcc merged the `full` and the ragged branch of the epilogue into one select-per-element version (16 compares,
// 30 v_cndmask, 18 s_and per 32x64 block: 283 instead of ~215 instructions for EVERY tile).
//
// CLM (bf16): "channels in the lane" epilogues.  The MFMA takes the FILTER fragment as its row operand and the pixel fragment as its
// column operand, so lane (r, h) ends up with one PIXEL (column r) and 16 output channels per 32-channel block; the filter rows are
// read through a lane permutation (bits 2 and 3 of r exchanged) that makes those 16 channels two runs of 8 CONSECUTIVE ones:
//   accumulator e of lane (r, h), block nt  ->  channel 32 nt + 16 (e >> 3) + 8 h + (e & 7)
// i.e. a lane's registers 8q .. 8q+7 are one 16-byte piece of the NHWC output and go to memory with one buffer store -- no LDS
// transposition (32 ds_write_b32 + 8 ds_read_b128 per 32 x 64 block in the CLM = 0 epilogue), the bias is the accumulator's initial
// value, and the per-channel statistics are running sums per accumulator register (64 registers, reduced over the 32 pixel lanes once
// per workgroup).  Per 256-pixel tile and wave ~170 instead of ~430 non-MFMA instructions on a forward launch.
//   CLM 0: the wave-private transposition epilogue (fp32 / bf16x3 storage only)     1: plain (data gradient: no bias, ReLU, statistics)
//   CLM 2: bias (+ border-class table, CLS) + ReLU + statistics                      3: plain + the two running sums of the consumer's
//                                                                                       BatchNorm backward (sum g, sum g y: rows k = 0, 1)
template <typename T, int TW, bool RAGGED, bool CLS = false, int CLM = 0>      // CLS: bias from a border-class table (IgemmParams::bias_classes)
__global__ void __launch_bounds__(512, 2) igemm_pws_kernel(const IgemmParams p, const int gm, const int w_resident) {
    constexpr int MT = 2;
    using G = WsGeo<TW, MT>;
    constexpr int TH = G::TH, NT = G::NT, HW_ = G::HW_, NPIX = G::NPIX, NPIXP = G::NPIXP, NJ = G::NJ;
    constexpr int KC = DT<T>::KC, VEC = DT<T>::VEC;
    constexpr bool SPLIT = __is_same(T, split_t);
    constexpr int STAGE = G::IN_SLOTS + G::WT_SLOTS;
    // Row pitch of the wave-private transposition block.  68 floats: the two half-waves of a 32x32 accumulator tile share
    // 16 banks when they write; 72 (conflict-free writes) was measured no faster in the write phase (it is VALU-issue
    // bound: 34.7k cycles either way) and slower in the read-back (6.0k -> 8.9k cycles), -DPWS_EPW=72 to reproduce.
    constexpr int EPW = PWS_EPW;
    constexpr bool CL = CLM != 0;
    static_assert(CL == (sizeof(T) == 2), "bf16 storage takes the channels-in-the-lane epilogues, fp32 / bf16x3 storage the transposition");
    static_assert(!CLS || CLM == 0 || CLM == 2, "the border-class table belongs to the bias epilogue");
    constexpr int EPI_SLOTS = CL ? 0 : (4 * 32 * EPW + 8 * 64) / 4;       // 4 waves x [32][EPW] fp32 + statistics hand-over
    static_assert((2 * STAGE + EPI_SLOTS) * 16 <= 160 * 1024, "two LDS stages + the epilogue region must fit one CU");
    __shared__ uint4 smem[2 * STAGE + EPI_SLOTS];
    static_assert(!CL || 2 * STAGE * 16 >= 2 * 64 * 4 * 33 * 4, "the final hand-over of the running sums reuses the two stages");
    // CLS: this workgroup's 64 columns of the [9][Np] border-class bias table (a folded BatchNorm, bnfold.hip), filled by the consumers while
    // they wait for the first stage: a border tile then takes its biases from LDS, not through two dependent global round trips per row tile
    __shared__ float cls_tab[CLS ? 9 * 64 : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int ltid = tid & 255;                    // index inside the role (4 waves each)
    const int cw = wave & 3;                       // consumer wave index (M quarter of the tile)
    const int r = lane & 31, h = lane >> 5;

    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int ntm = tiles_x * tiles_y * p.B, ntn = (p.Np + 63) >> 6;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);            // gridDim.x == ntn * gm
    const int tn = bid % ntn, mg = bid / ntn;                    // this workgroup: pixel tiles mg, mg + gm, ... of slab tn
    const int T_ = (ntm - mg + gm - 1) / gm;                     // >= 1
    const int n0 = tn * 64;
    const int nk = p.Kp / KC;                                    // even (checked by the launcher)
    const int S = T_ * nk;                                       // flattened K-steps

    float st1[2] = {0.f, 0.f}, st2[2] = {0.f, 0.f};              // consumers (CLM 0): running sum / sum of squares, columns 32*nt + r

    if (producer) {
        // ------------------------------------------------------------------ producers: global -> registers -> LDS
        constexpr int ESZ = sizeof(T);
        const int g4 = ltid & 3;
        const unsigned img_elems = (unsigned)p.H * (unsigned)p.W * (unsigned)p.x_ldc;
        const __amdgpu_buffer_rsrc_t wrs = make_rsrc(p.w, (unsigned)(9u * p.Np * p.Kp * ESZ));
        const int wco = ltid >> 2;
        const unsigned w_vo = n0 + wco < p.Np ? (unsigned)(((n0 + wco) * KC + g4 * VEC) * ESZ) : BUF_OOB;
        const unsigned w_slab = (unsigned)(p.Np * KC * ESZ);
        unsigned in_vo[NJ];
        __amdgpu_buffer_rsrc_t xrs;
        // halo pixel of staging slot j; the ragged last pass wraps around and re-stages the first pixels (same data,
        // same LDS slot) so that every load has an unconditional store
        auto slot_pix = [&](int j) { const int pix = (ltid >> 2) + 64 * j; return pix >= NPIX ? pix - NPIX : pix; };
        auto set_tile = [&](int tm) {
            const int x0 = (tm % tiles_x) * TW, y0 = ((tm / tiles_x) % tiles_y) * TH, b = tm / (tiles_x * tiles_y);
            xrs = make_rsrc((const char*)p.x + (size_t)b * img_elems * ESZ, img_elems * ESZ);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int pix = slot_pix(j);
                const int hy = pix / HW_, hx = pix - hy * HW_;
                const int yy = y0 + hy - 1, xx = x0 + hx - 1;
                in_vo[j] = (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W)
                               ? (unsigned)(((yy * p.W + xx) * p.x_ldc + g4 * VEC) * ESZ) : BUF_OOB;
            }
        };
        // Input chunks are fetched in PAIRS (two back-to-back 64-byte pieces = one full 128-byte line per pixel), see
        // igemm_ws.hip; nk is even, so a pair never straddles two tiles.
        uint4 rinA[NJ], rinB[NJ], rw[NT];
#ifdef PWS_PROTO_BNLOAD
        // MEASUREMENT PROTOTYPE (never in the shipped build: -DPWS_PROTO_BNLOAD): the VALU cost of applying the producer's
        // BatchNorm affine while staging the consumer's input -- per 16-byte piece 8 fma + unpack/pack (bf16) and the
        // post-affine zero-padding predicate -- with opaque register coefficients (a LOWER bound: the real form also loads 16
        // per-channel coefficients per K-step).  Results are unchanged (scale 1, shift 0).
        float proto_sc = 1.f, proto_sh = 0.f;
        asm volatile("" : "+v"(proto_sc), "+v"(proto_sh));
        auto proto_affine = [&](const uint4& v, unsigned vo) -> uint4 {
            if constexpr (sizeof(T) == 2) {
                const float m = vo != BUF_OOB ? 1.f : 0.f;
                float f[8] = {__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u),
                              __uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u), __uint_as_float(v.w << 16), __uint_as_float(v.w & 0xffff0000u)};
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] = fmaf(f[i], proto_sc, proto_sh) * m;
                return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
            } else {
                const float m = vo != BUF_OOB ? 1.f : 0.f;
                return make_uint4(__float_as_uint(fmaf(__uint_as_float(v.x), proto_sc, proto_sh) * m), __float_as_uint(fmaf(__uint_as_float(v.y), proto_sc, proto_sh) * m),
                                  __float_as_uint(fmaf(__uint_as_float(v.z), proto_sc, proto_sh) * m), __float_as_uint(fmaf(__uint_as_float(v.w), proto_sc, proto_sh) * m));
            }
        };
#define PWS_PROTO_AFFINE(v_, vo_) proto_affine(v_, vo_)
#else
#define PWS_PROTO_AFFINE(v_, vo_) (v_)
#endif
#define PWS_GLOAD_IN2(kp_)                                                                                        \
    do {                                                                                                          \
        const unsigned so_ = (unsigned)(2 * (kp_) * KC * ESZ);                                                    \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                          \
            rinA[j] = buf_ld16(xrs, in_vo[j], so_);                                                               \
            rinB[j] = buf_ld16(xrs, in_vo[j], so_ + KC * ESZ);                                                    \
        }                                                                                                         \
    } while (0)
#define PWS_GLOAD_W(ks_)                                                                                          \
    do {                                                                                                          \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                            \
            rw[t] = buf_ld16(wrs, w_vo, (unsigned)((ks_) * NT + t) * w_slab);                                     \
    } while (0)
#define PWS_STORE_IN(st_, RIN)                                                                                       \
    do {                                                                                                          \
        uint4* sm_ = smem + (st_) * STAGE;                                                                        \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                          \
            const int pix_ = slot_pix(j);                                                                         \
            /* split_t: the 64-byte K-chunk already is [hi 0-7][hi 8-15][lo 0-7][lo 8-15] = LDS slots 0..3 */     \
            sm_[g4 * NPIXP + pix_] = PWS_PROTO_AFFINE(RIN[j], in_vo[j]);                                          \
        }                                                                                                         \
    } while (0)

#define PWS_STORE_W(st_)                                                                                          \
    do {                                                                                                          \
        uint4* sm_ = smem + (st_) * STAGE;                                                                        \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) sm_[G::IN_SLOTS + (t * 4 + g4) * G::WG + wco] = rw[t];     \
    } while (0)
#define PWS_STORE(st_, RIN) do { PWS_STORE_IN(st_, RIN); PWS_STORE_W(st_); } while (0)

        int l_tile = 0, l_pair = 0;                // load cursor: tile index of this workgroup, K-step pair inside it
        const int npair = nk >> 1;
        set_tile(mg);
        PWS_GLOAD_IN2(0);
        PWS_GLOAD_W(0);
        PWS_STORE(0, rinA);
        PWS_GLOAD_W(1);
        __syncthreads();                                   // step 0 is staged
        if (nk == 2 && w_resident) {
            // Two K-steps per tile (64 input channels in bf16): stage parity == K-step, so the two stages already hold the
            // WHOLE filter slab of this output slab after the first tile -- only the input halos are staged from then on
            // (58 -> 22 KB per K-step).
            for (int q = 0; q < (S >> 1); ++q) {
                const bool more = 2 * q + 2 < S;
                PWS_STORE_IN(1, rinB);
                if (q == 0) PWS_STORE_W(1);                // the slab of K-step 1, fetched by the prologue
                if (more) { ++l_tile; set_tile(mg + l_tile * gm); PWS_GLOAD_IN2(0); }
                __syncthreads();
                if (more) PWS_STORE_IN(0, rinA);
                __syncthreads();
            }
        } else {
            int ks2 = 2 == nk ? 0 : 2;                         // K-step index (inside its tile) of flattened step 2q + 2
            for (int q = 0; q < (S >> 1); ++q) {
                const bool more = 2 * q + 2 < S;               // wave-uniform; false only in the last iteration
                // consumers are on step 2q (stage 0): stage step 2q + 1, then fetch steps 2q + 2 / 2q + 3
                PWS_STORE(1, rinB);
                if (more) {
                    PWS_GLOAD_W(ks2);
                    if (++l_pair == npair) { l_pair = 0; ++l_tile; set_tile(mg + l_tile * gm); }
                    PWS_GLOAD_IN2(l_pair);
                }
                __syncthreads();
                // consumers are on step 2q + 1 (stage 1): stage step 2q + 2, fetch the filter slab of step 2q + 3
                if (more) {
                    PWS_STORE(0, rinA);
                    PWS_GLOAD_W(ks2 + 1);
                }
                __syncthreads();
                ks2 = ks2 + 2 == nk ? 0 : ks2 + 2;
            }
        }
#undef PWS_GLOAD_IN2
#undef PWS_GLOAD_W
#undef PWS_STORE
#undef PWS_STORE_IN
#undef PWS_STORE_W
#undef PWS_PROTO_AFFINE
    } else if constexpr (CL) {
        // ------------------------------------------------------------------ consumers, channels in the lane (see the head of the kernel)
        int apix[MT];
        unsigned st_vo[MT], bn_vo[CLM == 3 ? MT : 1];      // tile-relative byte offsets: pixel of MFMA column r, channel piece 8 h
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = 32 * MT * cw + 32 * mt + r;
            apix[mt] = (m / TW) * HW_ + (m % TW);
            st_vo[mt] = (unsigned)(((m / TW) * p.W + m % TW) * p.y_ldc + n0 + 8 * h) * 2u;
            if constexpr (CLM == 3) bn_vo[mt] = (unsigned)(((m / TW) * p.W + m % TW) * p.Np + n0 + 8 * h) * 2u;
        }
        const int rperm = (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1);      // filter row behind MFMA row r: bits 2 and 3 exchanged
        const bool nt1 = n0 + 32 < p.Np;                                      // the slab's second 32-channel block exists (wave-uniform)
        float bcl[CLM == 2 ? 2 : 1][16];                                      // bias (class 4 of a border-class table) per accumulator register
        float cs1[CLM >= 2 ? 2 : 1][16], cs2[CLM >= 2 ? 2 : 1][16];           // running sums per accumulator register, across tiles
        if constexpr (CLM == 2) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int n = n0 + 32 * nt + 16 * (e >> 3) + 8 * h + (e & 7);
                    bcl[nt][e] = (p.bias && n < p.Np) ? p.bias[(CLS ? 4 * p.Np : 0) + n] : 0.f;
                }
        }
        if constexpr (CLM >= 2) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) { cs1[nt][e] = 0.f; cs2[nt][e] = 0.f; }
        }
        if constexpr (CLS) {      // what a border pixel's bias differs by from the interior's (class 4), natural channel order
            for (int i = ltid; i < 9 * 64; i += 256) {
                const int n = n0 + (i & 63);
                cls_tab[i] = n < p.Np ? p.bias[(i >> 6) * p.Np + n] - p.bias[4 * p.Np + n] : 0.f;
            }
        }
        const float relu_lo = p.relu ? 0.f : -__builtin_inff();
        const unsigned y_img = (unsigned)p.H * (unsigned)p.W * (unsigned)p.y_ldc * 2u;
        const unsigned bn_img = (unsigned)p.H * (unsigned)p.W * (unsigned)p.Np * 2u;      // bn_y: dense pitch Np
        uint4 ypc[CLM == 3 ? MT : 1][4];             // saved-activation pieces [nt * 2 + q] of the tile being finished
        unsigned long long d0 = PWD_T(), d1, d2, dk = 0, db = 0, de = 0;
        (void)d1; (void)d2; (void)dk; (void)db; (void)de;
        __syncthreads();                                   // step 0 is staged
        PWD_ADD(0, PWD_T() - d0);
        for (int ti = 0; ti < T_; ++ti) {
            const int tm = mg + ti * gm;
            const int x0 = (tm % tiles_x) * TW, y0 = ((tm / tiles_x) % tiles_y) * TH, b = tm / (tiles_x * tiles_y);
            const bool full = !RAGGED || (y0 + TH <= p.H && x0 + TW <= p.W);           // wave-uniform
            f32x16 acc[MT][2];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = CLM == 2 ? bcl[j][e] : 0.f;      // the bias is where the sum starts

            for (int ks = 0; ks < nk; ++ks) {
                d0 = PWD_T();
                const uint4* sm = smem + (ks & 1) * STAGE;
                if constexpr (CLM == 3) {
                    if (ks == nk - 1) {                            // wave-uniform: fetch the saved activation under the last K-step
                        const __amdgpu_buffer_rsrc_t brs = make_rsrc((const char*)p.bn_y + (size_t)b * bn_img, bn_img);
                        const unsigned bso = (unsigned)((y0 * p.W + x0) * p.Np) * 2u;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            unsigned vo = bn_vo[mt];
                            if (!full) {
                                const int m = 32 * MT * cw + 32 * mt + r;
                                if (!(y0 + m / TW < p.H && x0 + m % TW < p.W)) vo = BUF_OOB;
                            }
#pragma unroll
                            for (int pc = 0; pc < 4; ++pc)
                                ypc[mt][pc] = buf_ld16(brs, (pc < 2 || nt1) ? vo + 32u * pc : BUF_OOB, bso);
                        }
                    }
                }
                constexpr int NSTEP = 2 * NT;
                uint4 f[2][MT + 2];      // [buffer][pixel fragments a[mt] ..., filter fragments b0, b1]
#define PCL_FRAG(s_, d_)                                                                                          \
    do {                                                                                                          \
        const int t_ = (s_) >> 1, g_ = ((s_) & 1) * 2 + h;                                                        \
        const int ib_ = (t_ / 3) * HW_ + (t_ % 3);                                                                \
        _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) d_[mt_] = sm[ib_ + g_ * NPIXP + apix[mt_]];          \
        d_[MT] = sm[G::IN_SLOTS + (t_ * 4 + g_) * G::WG + rperm];                                                 \
        d_[MT + 1] = sm[G::IN_SLOTS + (t_ * 4 + g_) * G::WG + 32 + rperm];                                        \
    } while (0)
                PCL_FRAG(0, f[0]);
                __builtin_amdgcn_sched_group_barrier(0x100, MT + 2, 0);
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) {
                    if (st + 1 < NSTEP) PCL_FRAG(st + 1, f[(st + 1) & 1]);
                    const uint4* c = f[st & 1];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        mma16<T>(c[MT], c[mt], acc[mt][0]);            // rows = filter rows (channels), columns = pixels
                        mma16<T>(c[MT + 1], c[mt], acc[mt][1]);
                    }
                    if (st + 1 < NSTEP) sched_mfma_reads<2 * MT, MT + 2>();
                    else __builtin_amdgcn_sched_group_barrier(0x008, 2 * MT, 0);
                }
#undef PCL_FRAG
                d1 = PWD_T();
                __syncthreads();                           // hand the stage back to the producers
                dk += d1 - d0; db += PWD_T() - d1;
            }
            d2 = PWD_T();

            // ---- epilogue: a lane owns the pixel of its column; registers 8q .. 8q+7 of block nt are 16 bytes of the output
#if defined(PWS_ABLATE_EPI) && PWS_ABLATE_EPI == 1     // timing ablation (tools/pws_epi_ablate.sh, wrong results): no epilogue at all
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(acc[i][j]));
            continue;
#endif
            const __amdgpu_buffer_rsrc_t yrs = make_rsrc((const char*)p.y + (size_t)b * y_img, y_img);
            const unsigned y_so = (unsigned)((y0 * p.W + x0) * p.y_ldc) * 2u;
            bool border = false;
            if constexpr (CLS) border = x0 == 0 || y0 == 0 || x0 + TW >= p.W || y0 + TH >= p.H;      // wave-uniform
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = 32 * MT * cw + 32 * mt + r;
                bool pok = true;
                if (!full) pok = y0 + m / TW < p.H && x0 + m % TW < p.W;
                const unsigned vo = pok ? st_vo[mt] : BUF_OOB;
                if constexpr (CLS) {
                    if (border) {      // lanes on the image border add what their class's bias differs by (interior lanes read the zero row)
                        const int cls = border_class(y0 + m / TW, x0 + m % TW, p.H, p.W);
                        const float* row = cls_tab + (pok ? cls : 4) * 64 + 8 * h;
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const float4 d0_ = *reinterpret_cast<const float4*>(row + 32 * nt + 16 * q);
                                const float4 d1_ = *reinterpret_cast<const float4*>(row + 32 * nt + 16 * q + 4);
                                acc[mt][nt][8 * q + 0] += d0_.x; acc[mt][nt][8 * q + 1] += d0_.y; acc[mt][nt][8 * q + 2] += d0_.z; acc[mt][nt][8 * q + 3] += d0_.w;
                                acc[mt][nt][8 * q + 4] += d1_.x; acc[mt][nt][8 * q + 5] += d1_.y; acc[mt][nt][8 * q + 6] += d1_.z; acc[mt][nt][8 * q + 7] += d1_.w;
                            }
                    }
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if (nt == 1 && !nt1) break;            // wave-uniform
                    float v[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = CLM == 2 ? vmax_f32(acc[mt][nt][e], relu_lo) : acc[mt][nt][e];
                    if constexpr (CLM == 2 && !PWS_ABLATE_STATS) {
                        if (p.stats) {                     // wave-uniform
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const float vs = full ? v[e] : (pok ? v[e] : 0.f);
                                cs1[nt][e] += vs;
                                cs2[nt][e] = fmaf(vs, vs, cs2[nt][e]);
                            }
                        }
                    }
                    if constexpr (CLM == 3) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const uint4 u = ypc[mt][nt * 2 + q];
                            const float yv[8] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u),
                                                 __uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)};
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const float g = full ? v[8 * q + j] : (pok ? v[8 * q + j] : 0.f);
                                cs1[nt][8 * q + j] += g;
                                cs2[nt][8 * q + j] = fmaf(g, yv[j], cs2[nt][8 * q + j]);
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const uint4 o_ = make_uint4(pack2bf(v[8 * q + 0], v[8 * q + 1]), pack2bf(v[8 * q + 2], v[8 * q + 3]),
                                                    pack2bf(v[8 * q + 4], v[8 * q + 5]), pack2bf(v[8 * q + 6], v[8 * q + 7]));
#if defined(PWS_ABLATE_EPI) && PWS_ABLATE_EPI == 2     // timing ablation (wrong results): everything but the global stores
                        asm volatile("" :: "v"(o_.x), "v"(o_.y), "v"(o_.z), "v"(o_.w));
#elif defined(PWS_ABLATE_EPI) && PWS_ABLATE_EPI == 4   // timing ablation (wrong results): the store PATTERN of an exchanged epilogue (8 full 128-byte lines per instruction)
                        {
                            const int m4_ = 32 * MT * cw + 32 * mt + (lane >> 3) + 8 * (2 * nt + q);
                            buf_st16(yrs, (unsigned)(((m4_ / TW) * p.W + m4_ % TW) * p.y_ldc + n0 + 8 * (lane & 7)) * 2u, y_so, o_);
                        }
#else
                        buf_st16(yrs, vo + (unsigned)(64 * nt + 32 * q), y_so, o_);
#endif
                    }
                }
            }
            de += PWD_T() - d2;
        }
        PWD_ADD(1, dk); PWD_ADD(2, db); PWD_ADD(3, de);
        if (wave == 0) PWD_ADD(7, 1);
        // running sums -> LDS (the stages are free: every wave is past the last K-step's barrier): [kind][channel][wave][pixel lane], pitch 33
        if constexpr (CLM >= 2) {
            if (CLM == 3 || p.stats) {
                float* fb = reinterpret_cast<float*>(smem);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int c = 32 * nt + 16 * (e >> 3) + 8 * h + (e & 7);
                        fb[((0 * 64 + c) * 4 + cw) * 33 + r] = cs1[nt][e];
                        fb[((1 * 64 + c) * 4 + cw) * 33 + r] = cs2[nt][e];
                    }
            }
        }
    } else {
        // ------------------------------------------------------------------ consumers: LDS fragments -> MFMA -> epilogue
        int apix[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = 32 * MT * cw + 32 * mt + r;
            apix[mt] = (m / TW) * HW_ + (m % TW);
        }
        float bcol[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + 32 * nt + r;
            bcol[nt] = (p.bias && n < p.Np) ? p.bias[(CLS ? 4 * p.Np : 0) + n] : 0.f;      // class 4: interior pixels
        }
        if constexpr (CLS) {
            for (int i = ltid; i < 9 * 64; i += 256) {
                const int n = n0 + (i & 63);
                cls_tab[i] = n < p.Np ? p.bias[(i >> 6) * p.Np + n] : 0.f;
            }
        }
        float* const wbuf = reinterpret_cast<float*>(smem + 2 * STAGE) + cw * 32 * EPW;   // this wave's transposition block
        const float relu_lo = p.relu ? 0.f : -__builtin_inff();
        const bool plain = !p.relu && !p.bias && !p.stats;         // wave-uniform
        const unsigned y_img = (unsigned)p.H * (unsigned)p.W * (unsigned)p.y_ldc * (unsigned)sizeof(T);
        unsigned st_vo[MT][4];                     // tile-relative byte offsets of the 8-channel pieces this lane stores
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int m = 32 * MT * cw + 32 * mt + ps * 8 + (lane >> 3), n = n0 + (lane & 7) * 8;
                st_vo[mt][ps] = n < p.Np ? (unsigned)(((m / TW) * p.W + m % TW) * p.y_ldc + n) * (unsigned)sizeof(T) : BUF_OOB;
            }

        unsigned long long d0 = PWD_T(), d1, d2, dk = 0, db = 0, de = 0, dea = 0, deb = 0, dec_ = 0;
        (void)d1; (void)d2; (void)dk; (void)db; (void)de; (void)dea; (void)deb; (void)dec_;
        __syncthreads();                                   // step 0 is staged
        PWD_ADD(0, PWD_T() - d0);                          // [0] wait for the first stage
        for (int ti = 0; ti < T_; ++ti) {
            f32x16 acc[MT][2];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

            for (int ks = 0; ks < nk; ++ks) {
                d0 = PWD_T();
                const uint4* sm = smem + (ks & 1) * STAGE;         // nk is even: the stage parity restarts with every tile
                if constexpr (SPLIT) {
                    uint4 f[2][2 * MT + 4];      // [buffer][a_hi[mt], a_lo[mt] ..., bh0, bl0, bh1, bl1]
#define PWS_FRAG_S(t_, d_)                                                                                        \
    do {                                                                                                          \
        const int ib_ = ((t_) / 3) * HW_ + ((t_) % 3);                                                            \
        _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) {                                                    \
            d_[2 * mt_] = sm[ib_ + h * NPIXP + apix[mt_]];                                                        \
            d_[2 * mt_ + 1] = sm[ib_ + (2 + h) * NPIXP + apix[mt_]];                                              \
        }                                                                                                         \
        d_[2 * MT + 0] = sm[G::IN_SLOTS + ((t_) * 4 + h) * G::WG + r];                                            \
        d_[2 * MT + 1] = sm[G::IN_SLOTS + ((t_) * 4 + 2 + h) * G::WG + r];                                        \
        d_[2 * MT + 2] = sm[G::IN_SLOTS + ((t_) * 4 + h) * G::WG + 32 + r];                                       \
        d_[2 * MT + 3] = sm[G::IN_SLOTS + ((t_) * 4 + 2 + h) * G::WG + 32 + r];                                   \
    } while (0)
                    PWS_FRAG_S(0, f[0]);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT + 4, 0);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if (t + 1 < NT) PWS_FRAG_S(t + 1, f[(t + 1) & 1]);
                        const uint4* c = f[t & 1];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            mma_bf16(c[2 * mt + 1], c[2 * MT + 0], acc[mt][0]); mma_bf16(c[2 * mt], c[2 * MT + 1], acc[mt][0]);
                            mma_bf16(c[2 * mt], c[2 * MT + 0], acc[mt][0]);
                            mma_bf16(c[2 * mt + 1], c[2 * MT + 2], acc[mt][1]); mma_bf16(c[2 * mt], c[2 * MT + 3], acc[mt][1]);
                            mma_bf16(c[2 * mt], c[2 * MT + 2], acc[mt][1]);
                        }
                        if (t + 1 < NT) sched_mfma_reads<6 * MT, 2 * MT + 4>();
                        else __builtin_amdgcn_sched_group_barrier(0x008, 6 * MT, 0);
                    }
#undef PWS_FRAG_S
                } else {
                    constexpr int NSTEP = 2 * NT;
                    constexpr int NMF = sizeof(T) == 2 ? 4 : 16;
                    uint4 f[2][MT + 2];      // [buffer][a[mt] ..., b0, b1]
#define PWS_FRAG(s_, d_)                                                                                          \
    do {                                                                                                          \
        const int t_ = (s_) >> 1, g_ = ((s_) & 1) * 2 + h;                                                        \
        const int ib_ = (t_ / 3) * HW_ + (t_ % 3);                                                                \
        _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) d_[mt_] = sm[ib_ + g_ * NPIXP + apix[mt_]];          \
        d_[MT] = sm[G::IN_SLOTS + (t_ * 4 + g_) * G::WG + r];                                                     \
        d_[MT + 1] = sm[G::IN_SLOTS + (t_ * 4 + g_) * G::WG + 32 + r];                                            \
    } while (0)
                    PWS_FRAG(0, f[0]);
                    __builtin_amdgcn_sched_group_barrier(0x100, MT + 2, 0);
#pragma unroll
                    for (int st = 0; st < NSTEP; ++st) {
                        if (st + 1 < NSTEP) PWS_FRAG(st + 1, f[(st + 1) & 1]);
                        const uint4* c = f[st & 1];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            mma16<T>(c[mt], c[MT], acc[mt][0]);
                            mma16<T>(c[mt], c[MT + 1], acc[mt][1]);
                        }
                        if (st + 1 < NSTEP) sched_mfma_reads<NMF * MT / 2, MT + 2>();
                        else __builtin_amdgcn_sched_group_barrier(0x008, NMF * MT / 2, 0);
                    }
#undef PWS_FRAG
                }
                d1 = PWD_T();
                __syncthreads();                           // hand the stage back to the producers
                dk += d1 - d0; db += PWD_T() - d1;
            }
            d2 = PWD_T();

            // ---- epilogue of this tile: bias, ReLU, statistics, wave-private transposition, 16-byte stores.
            // No workgroup barrier: the producers are already staging the next tile and meet the consumers again at
            // the barrier of its first K-step.  Kept lean on purpose -- with 2-4 K-steps per tile this code, not the
            // MFMA loop, was 44 % of the consumer cycles of the 64-channel bf16 layers (tools/ws_diag.py ... pws):
            // ReLU is a max with 0 / -inf (no select), the stores are range-checked buffer stores whose per-lane offsets
            // are tile-relative constants (tile origin = scalar offset), edge masks exist only on ragged tiles.
#if defined(PWS_ABLATE_EPI) && PWS_ABLATE_EPI == 1     // timing ablation (tools/pws_epi_ablate.sh, wrong results): no epilogue at all
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) asm volatile("" :: "v"(acc[mt][nt][e]));
            continue;
#endif
            const int tm = mg + ti * gm;
            const int x0 = (tm % tiles_x) * TW, y0 = ((tm / tiles_x) % tiles_y) * TH, b = tm / (tiles_x * tiles_y);
            const __amdgpu_buffer_rsrc_t yrs = make_rsrc((const char*)p.y + (size_t)b * y_img, y_img);
            const unsigned y_so = (unsigned)((y0 * p.W + x0) * p.y_ldc) * (unsigned)sizeof(T);
            const bool full = !RAGGED || (y0 + TH <= p.H && x0 + TW <= p.W);           // wave-uniform
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned long long e0 = PWD_T(); (void)e0;
                if (plain) {
                    // data-gradient launch (no bias, no ReLU, no statistics): the accumulators go straight to the transposition.
                    // VALU instructions are not free next to the matrix pipe -- another wave's VALU stream overlaps only ~25 % of an
                    // MFMA stream on the same SIMD (tools/ubench/mfma_valu_overlap.hip) -- so these 4 x 64 skipped per tile count
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) wbuf[acc_row(e, h) * EPW + 32 * nt + r] = acc[mt][nt][e];
                } else if (CLS) {
                    // Border-class bias (a folded BatchNorm): an MFMA row tile is 32 / TW image rows of the tile, so the row class is
                    // wave-uniform and the column class differs from "interior" in at most the first / last pixel of a row -- three table
                    // entries (LDS) per row and channel.  ONE code path for every tile of the launch (a separate path for the 30-56 % border
                    // tiles was cold in the instruction cache every time it ran: +44 % per launch); where the tile is whole along x, WHICH
                    // accumulator registers hold those two pixels is known at compile time (acc_row), the others pay nothing.
                    constexpr int RPM = 32 / TW;
                    const bool whole_x = x0 + TW <= p.W;
                    const bool first_col = x0 == 0 && h == 0, last_col = x0 + TW == p.W && h == 1;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        float bl[RPM], bm[RPM], br[RPM];
#pragma unroll
                        for (int rr = 0; rr < RPM; ++rr) {
                            const int yy = y0 + (MT * cw + mt) * RPM + rr;
                            const int rc = yy == 0 ? 0 : (yy == p.H - 1 ? 6 : 3);
                            bl[rr] = cls_tab[(rc + 0) * 64 + 32 * nt + r];
                            bm[rr] = cls_tab[(rc + 1) * 64 + 32 * nt + r];
                            br[rr] = cls_tab[(rc + 2) * 64 + 32 * nt + r];
                        }
                        if (whole_x) {      // wave-uniform
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int rr = TW == 32 ? 0 : (e >> 3);                 // acc_row(e, h) / TW
                                const int ec = TW == 32 ? (e >> 2) : ((e >> 2) & 1);    // acc_row = (e & 3) + 8 * (e >> 2) + 4 * h
                                float bv = bm[rr];
                                if ((e & 3) == 0 && ec == 0) bv = first_col ? bl[rr] : bv;         // column 0 of the tile: h == 0 lanes
                                if ((e & 3) == 3 && ec == TW / 8 - 1) bv = last_col ? br[rr] : bv;  // column TW - 1: h == 1 lanes
                                const float v = fmaxf(acc[mt][nt][e] + bv, relu_lo);
                                const float vs = (full || y0 + (MT * cw + mt) * RPM + rr < p.H) ? v : 0.f;
                                st1[nt] += vs;
                                st2[nt] = fmaf(vs, vs, st2[nt]);
                                wbuf[acc_row(e, h) * EPW + 32 * nt + r] = v;
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int rr = TW == 32 ? 0 : (e >> 3);
                                const int xx = x0 + acc_row(e, h) - rr * TW, yy = y0 + (MT * cw + mt) * RPM + rr;
                                const float bv = xx == 0 ? bl[rr] : (xx == p.W - 1 ? br[rr] : bm[rr]);
                                const float v = fmaxf(acc[mt][nt][e] + bv, relu_lo);
                                const float vs = (yy < p.H && xx < p.W) ? v : 0.f;
                                st1[nt] += vs;
                                st2[nt] = fmaf(vs, vs, st2[nt]);
                                wbuf[acc_row(e, h) * EPW + 32 * nt + r] = v;
                            }
                        }
                    }
                } else if (full) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float v = fmaxf(acc[mt][nt][e] + bcol[nt], relu_lo);
                            st1[nt] += v;
                            st2[nt] = fmaf(v, v, st2[nt]);
                            wbuf[acc_row(e, h) * EPW + 32 * nt + r] = v;
                        }
                } else {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float v = fmaxf(acc[mt][nt][e] + bcol[nt], relu_lo);
                            const int m = 32 * MT * cw + 32 * mt + acc_row(e, h);
                            const float vs = (y0 + m / TW < p.H && x0 + m % TW < p.W) ? v : 0.f;
                            st1[nt] += vs;
                            st2[nt] = fmaf(vs, vs, st2[nt]);
                            wbuf[acc_row(e, h) * EPW + 32 * nt + r] = v;
                        }
                }
                const unsigned long long e1 = PWD_T(); (void)e1;
                float4 lo[4], hi[4];                       // all four row pieces first: one LDS round trip, not four
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int row = ps * 8 + (lane >> 3), cgp = lane & 7;
                    lo[ps] = *reinterpret_cast<const float4*>(wbuf + row * EPW + cgp * 8);
                    hi[ps] = *reinterpret_cast<const float4*>(wbuf + row * EPW + cgp * 8 + 4);
                }
#ifdef CLAMD_DIAG
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // attribute the LDS round trip to phase [5]
#endif
                const unsigned long long e2 = PWD_T(); (void)e2;
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const float v[8] = {lo[ps].x, lo[ps].y, lo[ps].z, lo[ps].w, hi[ps].x, hi[ps].y, hi[ps].z, hi[ps].w};
                    unsigned vo = st_vo[mt][ps];
                    if (!full) {
                        const int m = 32 * MT * cw + 32 * mt + ps * 8 + (lane >> 3);
                        if (!(y0 + m / TW < p.H && x0 + m % TW < p.W)) vo = BUF_OOB;
                    }
#if defined(PWS_ABLATE_EPI) && PWS_ABLATE_EPI == 2     // timing ablation (wrong results): everything but the global stores
                    asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(vo));
#else
                    buf_st8<T>(yrs, vo, y_so, v);
#endif
                }
                dea += e1 - e0; deb += e2 - e1; dec_ += PWD_T() - e2;
            }
            de += PWD_T() - d2;
        }
        PWD_ADD(1, dk); PWD_ADD(2, db); PWD_ADD(3, de);    // [1] MFMA loops [2] at the K-step barriers [3] epilogues
        PWD_ADD(4, dea); PWD_ADD(5, deb); PWD_ADD(6, dec_); // epilogue: [4] bias/ReLU/stats + LDS writes [5] LDS round trip [6] pack + stores
        if (wave == 0) PWD_ADD(7, 1);                      // [7] workgroups
    }

    // ---------------------------------------------------------------------- statistics: one flush per workgroup
    if constexpr (CLM >= 2) {
        // channels-in-the-lane epilogues: 128 (kind, channel) sums of 4 waves x 32 pixel lanes each, added in a fixed order -- thread
        // (kind, channel, wave) adds its 32 lanes' values, two xor shuffles add the four waves -- and written as this workgroup's row
        if (CLM == 3 || p.stats) {
            __syncthreads();
            const float* fb = reinterpret_cast<const float*>(smem);
            const int kc = tid >> 2;
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) t += fb[tid * 33 + i];
            t += __shfl_xor(t, 1);
            t += __shfl_xor(t, 2);
            const int k = kc >> 6, c = kc & 63;
            if ((tid & 3) == 0 && n0 + c < p.Np) {
                if constexpr (CLM == 2) p.stats[((size_t)mg * 2 + k) * p.Np + n0 + c] = t;
                else p.bn_sums[((size_t)mg * 5 + k) * p.Np + n0 + c] = t;
            }
            if constexpr (CLM == 3) {      // rows 2-4 of the five-sum layout: not taken here (the conv-bias gradient comes from clamd_bn_bwd_apply_sums).
                // Written as NaN, not zeros: a caller that still asks clamd_bn_bwd_finalize for dbias gets NaN, not a silent 0
                if (tid < 192 && n0 + (tid & 63) < p.Np) p.bn_sums[((size_t)mg * 5 + 2 + (tid >> 6)) * p.Np + n0 + (tid & 63)] = __builtin_nanf("");
            }
        }
        return;
    }
    if constexpr (CLM == 1) return;
    if (p.stats) {
        float* sbuf = reinterpret_cast<float*>(smem + 2 * STAGE) + 4 * 32 * EPW;
        if (!producer) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                st1[nt] += __shfl_xor(st1[nt], 32);
                st2[nt] += __shfl_xor(st2[nt], 32);
            }
            if (h == 0) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    sbuf[(cw * 2 + 0) * 64 + 32 * nt + r] = st1[nt];
                    sbuf[(cw * 2 + 1) * 64 + 32 * nt + r] = st2[nt];
                }
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int k = tid >> 6, c = tid & 63;
            const float t = sbuf[(0 * 2 + k) * 64 + c] + sbuf[(1 * 2 + k) * 64 + c] + sbuf[(2 * 2 + k) * 64 + c] +
                            sbuf[(3 * 2 + k) * 64 + c];
            if (n0 + c < p.Np) p.stats[((size_t)mg * 2 + k) * p.Np + n0 + c] = t;
        }
    }
}

}  // namespace clamd
#ifdef CLAMD_DIAG
extern "C" int clamd_debug_pws_diag(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(clamd::g_pws_diag), 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(clamd::g_pws_diag), z, 64) != hipSuccess) return -1; }
    return 0;
}
#endif
namespace clamd {


// Workgroups per 64-channel output slab (= partial statistics rows of a launch), or -1 when this kernel declines the shape.
int pws_rows(const IgemmParams& p, int dtype, const clamd_tuning& tn) {
    const int esz = dtype == CLAMD_BF16 ? 2 : 4, kc = dtype == CLAMD_BF16 ? 32 : 16;
    if (p.bn_y && (esz != 2 || p.bias || p.relu || p.stats || p.bias_classes)) return -1;   // the two-sum epilogue: bf16, plain data-gradient launches
    if (p.Kp % (2 * kc)) return -1;                          // K-steps are staged in pairs
    const int TW = p.W >= 32 ? 32 : 16, TH = 256 / TW;
    const long long ntm = (long long)((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH) * p.B;
    const int ntn = (p.Np + 63) / 64;
    long long gm = clamd_usable_cus(tn) / ntn;
    if (gm < 1) gm = 1;
    if (gm > ntm) gm = ntm;
    return (int)gm;
}

template <typename T>
static int launch_pws_t(const IgemmParams& p, hipStream_t s, int dtype, const clamd_tuning& tn) {
    const long long gm = pws_rows(p, dtype, tn);
    if (gm < 0) return -1;
    const bool wide = p.W >= 32;
    const int TW = wide ? 32 : 16, TH = 256 / TW;
    const long long ntm = (long long)((p.W + TW - 1) / TW) * ((p.H + TH - 1) / TH) * p.B;
    const int ntn = (p.Np + 63) / 64;
    const long long nblk = gm * ntn;
    if (ntm > 0x7fffffff) return clamd_fail("igemm_pws: grid out of range");
    const bool ragged = (p.H % TH) != 0 || (p.W % TW) != 0;
#define PWS_LAUNCH_K(TW_, RG_, CLS_, CLM_) hipLaunchKernelGGL((igemm_pws_kernel<T, TW_, RG_, CLS_, CLM_>), dim3((unsigned)nblk), dim3(512), 0, s, p, (int)gm, tn.pws_wres)
#define PWS_LAUNCH(TW_, RG_)                                                                                                      \
    do {                                                                                                                           \
        if constexpr (sizeof(T) == 2) {      /* bf16: the channels-in-the-lane epilogues */                                         \
            const bool plain_ = !p.bias && !p.relu && !p.stats && !p.bias_classes;                                                 \
            if (p.bn_y) PWS_LAUNCH_K(TW_, RG_, false, 3);                /* plain, checked by pws_rows */                           \
            else if (plain_) PWS_LAUNCH_K(TW_, RG_, false, 1);                                                                     \
            else if (p.bias_classes) PWS_LAUNCH_K(TW_, RG_, true, 2);                                                              \
            else PWS_LAUNCH_K(TW_, RG_, false, 2);                                                                                 \
        } else {                                                                                                                   \
            if (p.bias_classes) PWS_LAUNCH_K(TW_, RG_, true, 0);                                                                   \
            else PWS_LAUNCH_K(TW_, RG_, false, 0);                                                                                 \
        }                                                                                                                          \
    } while (0)
    if (wide) { if (ragged) PWS_LAUNCH(32, true); else PWS_LAUNCH(32, false); }
    else { if (ragged) PWS_LAUNCH(16, true); else PWS_LAUNCH(16, false); }
#undef PWS_LAUNCH
#undef PWS_LAUNCH_K
    return clamd_check_launch("igemm_pws");
}

int launch_igemm_pws(const IgemmParams& p, int dtype, hipStream_t s, const clamd_tuning& tn) {
    if (dtype == CLAMD_BF16) return launch_pws_t<bf16_t>(p, s, dtype, tn);
    if (dtype == CLAMD_F32) return launch_pws_t<float>(p, s, dtype, tn);
    if (dtype == CLAMD_SPLIT) return launch_pws_t<split_t>(p, s, dtype, tn);
    return clamd_fail("igemm_pws: bad dtype");
}

}  // namespace clamd
