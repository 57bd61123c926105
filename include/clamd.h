/* clamd.h -- C ABI of libclamd.so: the MI355X (gfx950) kernels behind the UNet segmentation train step of
 * LorenzoFramba/Continual-Learning.
 *
 * The reference defines NO FFI for this path: its hot loop (trainer.py:168-176) calls PyTorch modules
 * (models/unet.py:8-92) whose kernels live in ATen/cuDNN.  This header is therefore the interface a maintainer
 * would bind INSTEAD of those torch operators; each entry point names the reference call site it replaces.
 * INTEGRATION.md shows the ctypes binding and the torch.autograd.Function glue.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, ints, doubles; `stream` is a hipStream_t passed as void*.
 *   - every call only ENQUEUES work on `stream`; no allocation, no host synchronisation, graph-capturable.
 *     Buffers are owned by the caller (PyTorch's allocator) and only borrowed for the enqueue.
 *   - return 0 on success, negative on error; clamd_last_error() returns the (thread-local) message.
 *   - activations are NHWC in the compute dtype with an explicit channel pitch `ldc` (elements) so that a
 *     tensor may be a channel slice of a concat buffer (replaces torch.cat, models/unet.py:83-87).
 *     Physical channel counts (`*_p`) are padded to a power of two >= 32; padded channels hold zeros.
 *   - dtype: CLAMD_F32 computes with v_mfma_f32_32x32x2_f32 (exact fp32), CLAMD_BF16 stores activations and
 *     packed weights as bf16 and accumulates in fp32 (v_mfma_f32_32x32x16_bf16).  CLAMD_SPLIT ("bf16x3") stores
 *     every activation and packed weight as the bf16 pair hi = rne(x), lo = rne(x - hi) (4 bytes per element) and
 *     multiplies hi*hi + hi*lo + lo*hi with fp32 accumulation (~2^-17 relative product error).  Its tensors keep the
 *     fp32 geometry (element counts, `ldc`, byte sizes), but each 16-channel group of a pixel is laid out as
 *     [16 x bf16 hi][16 x bf16 lo]; tensor and channel-slice bases must be 64-byte aligned and `ldc` a multiple of
 *     16 (checked on entry).  clamd_nchw_to_nhwc / clamd_nhwc_to_nchw convert at the boundary.
 */
#ifndef CLAMD_H
#define CLAMD_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { CLAMD_F32 = 0, CLAMD_BF16 = 1, CLAMD_SPLIT = 2 };
enum { CLAMD_WGRAD_CONV3 = 0, CLAMD_WGRAD_PW = 1, CLAMD_WGRAD_UP2 = 2 };

const char* clamd_last_error(void);
int clamd_version(void);
/* layout constants the host side needs to build device tables */
int clamd_sizeof_pack_job(void);
int clamd_sizeof_adam_tensor(void);
int clamd_adam_chunk_elems(void);
int clamd_pack_tile(void);            /* clamd_pack: blocks per job = ceil(Np/tile) * ceil(Kp/tile) */
int clamd_bn_bwd_nsums(void);

/* ---- per-call kernel-structure selection --------------------------------------------------------------------
 * The library keeps NO mutable process state (nn.DataParallel, trainer.py:122, drives one Python thread
 * per GPU through the same code).  Every entry point that chooses between kernel structures takes a
 * `const clamd_tuning*`; NULL = the defaults clamd_tuning_init() writes.  Activations are bit-identical under every
 * setting; statistics / weight gradients differ only in (fixed) summation order.
 *   igemm_pws      0|1|2   persistent conv kernel: never | <= 256 input channels | always
 *   igemm_ws       0..4    producer/consumer kernel: never | 256-px | heuristic | 512-px | 128-px tiles
 *   igemm_variant  0|1|2   baseline kernel prefetch variants
 *   pws_wres       0|1     persistent kernel: filter slab kept in LDS across tiles when a tile has two K-steps
 *   wgrad_ws 0|1, wgrad_dma 0|1|2 (LDS-DMA staging: never | heuristic | always), wgrad_xcd 0|1,
 *   wgrad_blocks 1..1024 (split-K target in workgroups on a whole chip, two per CU = 512 by default), wgrad_tw16 0|1
 *   wino_band      0 (per-launch choice) | 1..32 output-channel slabs per band of the Winograd block order
 *   wino_persist   0|1     one workgroup per tile | persistent tile loop
 *   wino_mt        0|1|2   tile height: per-launch choice | 8 | 16 pixels
 *   bn_reduce_blocks / chsum_blocks   0 (per-launch choice) | n: grid cap of the per-channel reductions
 *   cu_reserve     CUs the persistent grids leave free (for RCCL channel workgroups under data parallelism)
 *   wgrad_streamk  0|1|2   clamd_wgrad_winograd24_pre / 44_pre: split-K plan in whole rounds of the chip | per launch (stream-K from 96 items on) |
 *                          always the k-steps of all (plane, block) items dealt evenly to one workgroup per CU (static map, an item's slots
 *                          added in slot order: deterministic)
 *   wino_half      0|1     clamd_conv3x3_winograd24: 32 tiles x 64 channels, one workgroup per CU | 32 x 32, two per CU (wino24n.hip) */
typedef struct clamd_tuning {
    int igemm_pws, igemm_ws, igemm_variant, pws_wres;
    int wgrad_ws, wgrad_dma, wgrad_xcd, wgrad_blocks, wgrad_tw16;
    int wino_band, wino_persist, wino_mt;
    int bn_reduce_blocks, chsum_blocks;
    int cu_reserve;
    int wino_half;
    int wgrad_streamk;
    int reserved[7];
} clamd_tuning;
int clamd_sizeof_tuning(void);
void clamd_tuning_init(clamd_tuning* t);

/* ---- deterministic per-channel reductions ---------------------------------------------------------------------
 * BatchNorm statistics (sum, sum of squares) and the five BatchNorm-backward sums are never accumulated with float
 * atomics: every producing launch writes `rows` partial rows [row][nk][Cp] (nk = 2 or 5) with plain stores, one row per
 * workgroup (or per tile), and clamd_bn_finalize / clamd_bn_bwd_finalize add rows 0..rows-1 in a fixed order in fp64.
 * Two runs on the same inputs are bit-identical.  The caller sizes the buffer with clamd_stat_rows() for the SAME
 * arguments it launches with and passes that row count to the launch (checked) and to the finalize call. */
enum { CLAMD_OP_CONV3X3 = 0, CLAMD_OP_CONV3X3_WINOGRAD = 1, CLAMD_OP_CONV1X1 = 2, CLAMD_OP_CONVT2X2_DGRAD = 3,
       CLAMD_OP_BN_BWD_REDUCE = 4, CLAMD_OP_CONV3X3_WINOGRAD24 = 5, CLAMD_OP_CONV3X3_WINOGRAD44 = 6 };
/* rows a launch of `op` writes: (B,H,W) = pixel grid of the launch, Cin_p/Cout_p as passed to it (BN_BWD_REDUCE: Cout_p = Cp,
 * Cin_p != 0 means the pooled variant), fused_bn != 0 when bn_y/bn_sums are passed.  Negative on error. */
int clamd_stat_rows(int op, int B, int H, int W, int Cin_p, int Cout_p, int dtype, int fused_bn, const clamd_tuning* tune);

/* ---- implicit-GEMM convolutions (igemm.hip) ---------------------------------------------------------------
 * nn.Conv2d(k3,s1,p1)+bias followed by nn.ReLU (models/unet.py:13-14,16-17,28-29,31-32,50-51,53-54,66-67,69-70):
 *   y = relu?(conv3x3(x, w) + bias), and (if stats != NULL) per-channel sum / sum-of-squares of y written as
 *   partial rows stats[row][2][Cout_p] for the following nn.BatchNorm2d (unet.py:15,...).  The same entry point run
 *   on the flipped/transposed packing computes the data gradient of that conv (loss.backward(), trainer.py:175).
 *   w_packed: [9][Cout_p][Cin_p] compute dtype, Cin innermost (see clamd_pack).  m_fastest: block order hint.
 *   bn_y/bn_sums (optional, data-gradient launches): when y of THIS launch is the gradient w.r.t. a BatchNorm output,
 *   also accumulate the five per-channel sums of clamd_bn_bwd_reduce (bn_y = that unit's saved activation
 *   [B,H,W,Cout_p], bn_sums = [row][5][Cout_p]) in the epilogue, so the separate reduce pass is not needed.
 *   stat_rows = clamd_stat_rows(CLAMD_OP_CONV3X3, ...) (ignored when neither stats nor bn_sums is given).
 *   relu: bit 0 = apply ReLU; bit 1 (CLAMD_BIAS_BORDER_CLASSES, forward launches behind a folded BatchNorm, see
 *   clamd_bn_fold_bias below) = `bias` is a [9][Cout_p] table indexed by the border class of the output pixel.  Bit 1 is taken by
 *   clamd_conv3x3 where clamd_conv3x3_border_bias_ok() says so (the persistent kernel), by clamd_conv3x3_winograd24 and by
 *   clamd_conv3x3_winograd24_direct_filters; every other entry point rejects it. */
enum { CLAMD_RELU = 1, CLAMD_BIAS_BORDER_CLASSES = 2 };
int clamd_conv3x3(const void* x, int x_ldc, const void* w_packed, const float* bias, void* y, int y_ldc,
                  float* stats, const void* bn_y, float* bn_sums, int stat_rows, int B, int H, int W, int Cin_p, int Cout_p,
                  int relu, int m_fastest, int dtype, const clamd_tuning* tune, void* stream);
int clamd_conv3x3_border_bias_ok(int B, int H, int W, int Cin_p, int Cout_p, int dtype, const clamd_tuning* tune);
/* How many of the five sums a PLAIN (no bias, ReLU, statistics) clamd_conv3x3 launch with bn_y / bn_sums takes: 5, or 2 where the
 * persistent bf16 kernel runs it (channels-in-the-lane epilogue: sum g and sum g y as running sums per accumulator register; rows
 * k = 2..4 are written as NaN -- clamd_bn_bwd_finalize with dbias != NULL on such rows returns NaN bias gradients instead of silent zeros --
 * and the convolution's bias gradient comes from clamd_bn_bwd_apply_sums, below).  0 on bad arguments. */
int clamd_conv3x3_bn_sums(int B, int H, int W, int Cin_p, int Cout_p, int dtype, const clamd_tuning* tune);
/* ---- nn.BatchNorm2d folded into the nn.Conv2d(k3,p1) behind it (models/unet.py:15-16,30-31; bnfold.hip) ---------------------
 * x = scale * r + shift with r the producer's saved conv+ReLU output and scale / shift from clamd_bn_finalize:
 *   conv3x3(x, W) = conv3x3(r, W * scale[ci]) + the sum of T[co][tap] = sum_ci W[co][ci][tap] * shift[ci] over the taps that read
 *   inside the image (nn.Conv2d pads x, not r, with zeros) -- nine border classes: 3 * (0 top row, 1 interior, 2 bottom row) +
 *   (0 left column, 1 interior, 2 right column); H, W >= 2.
 * The caller packs the filters with `kscale` = scale (PackJob / WinoPackJob, ops.PackTable / ops.WinoPackTable), fills the table
 * with clamd_bn_fold_bias (table[class][co] = bias[co] + sum of T over the class's valid taps; padded channels 0) and launches
 * the convolution on r with relu | CLAMD_BIAS_BORDER_CLASSES: the clamd_bn_apply pass over the producer's output is not needed.
 * Weight gradient (trainer.py:175): run the weight-gradient entry point on r instead of x, then clamd_bn_fold_wgrad in place:
 *   dW[co][ci][tap] = scale[ci] * dW[co][ci][tap] + shift[ci] * S[tap][co],  S = sum of gz over the pixels whose tap reads inside
 *   the image = sum_gz (the convolution's bias gradient, as clamd_bn_bwd_finalize writes it) - border row - border column + corner,
 *   the border sums taken here from gz in a fixed order.  dw = [Cout][Cin][3][3] fp32 (the parameter's own layout); workspace >=
 *   clamd_bn_fold_wgrad_workspace_bytes(B, Cout_p).  The data gradient (w.r.t. x) is unchanged. */
int clamd_bn_fold_bias(const float* w, const float* shift, const float* bias, float* table, int Cout, int Cin, int Cout_p, void* stream);
/* the filter pack of that convolution and its bias table in ONE launch (both sit between the producer's clamd_bn_finalize and the
 * convolution, on the critical path of the forward pass): form 0 = clamd_pack(jobs_dev, njobs, total_blocks, dtype), 16 =
 * clamd_wino_pack, 24 = clamd_wino24_pack, with the blocks of clamd_bn_fold_bias appended to the grid (taps = 9).
 * The 1x1 head behind the last BatchNorm (models/unet.py:70-72) folds the same way without border classes: taps = 1 (form 0), w
 * [Cout][Cin], table = one row [Cout_p] = bias + W . shift; its weight gradient on the un-normalised tensor is fixed up by
 * clamd_bn_fold_wgrad_pointwise: dW[co][ci] = scale[ci] * dW[co][ci] + shift[ci] * sum_g[co] (sum_g = the head's bias gradient). */
int clamd_bn_fold_pack(int form, const void* jobs_dev, int njobs, int total_blocks, int dtype, const float* w, int taps, const float* shift,
                       const float* bias, float* table, int Cout, int Cin, int Cout_p, void* stream);
int clamd_bn_fold_wgrad_pointwise(const float* sum_g, const float* scale, const float* shift, float* dw, int Cout, int Cin, void* stream);
size_t clamd_bn_fold_wgrad_workspace_bytes(int B, int Cout_p);
int clamd_bn_fold_wgrad(const void* gz, int gz_ldc, const float* sum_gz, const float* scale, const float* shift, float* dw,
                        void* workspace, size_t ws_bytes, int B, int H, int W, int Cout_p, int Cout, int Cin, int dtype, void* stream);
/* The same convolution (fp32 only; H, W even) by Winograd F(2x2,3x3): 2.25x fewer multiply-adds, fp32 transforms
 * (relative error vs fp64 3.5e-7 against 2.3e-7 for the direct sum).  w_wino = [Cin_p/8][16][Cout_p][8] written by
 * clamd_wino_pack (jobs: device table of WinoPackJob, see ops.WinoPackTable; the data gradient uses the tap-flipped,
 * transposed filter).  Epilogue: bias, ReLU, BN statistics as clamd_conv3x3 (wino.hip). */
int clamd_sizeof_wino_pack_job(void);
int clamd_wino_pack(const void* jobs_dev, int njobs, int total_blocks, void* stream);
int clamd_conv3x3_winograd(const float* x, int x_ldc, const float* w_wino, const float* bias, float* y, int y_ldc,
                           float* stats, int stat_rows, int B, int H, int W, int Cin_p, int Cout_p, int relu,
                           const clamd_tuning* tune, void* stream);
/* The same convolution by the hybrid Winograd F(2x4,3x3) (fp32 only; H even, W a multiple of 4): F(2,3) down the rows, F(4,3)
 * along the columns -- 3 multiply-adds per output instead of 4 (F(2x2)) or 9 (direct); fp32 error vs fp64 1e-6.
 * w_wino = [Cin_p/8][24][Cout_p][8] written by clamd_wino24_pack (same job table as clamd_wino_pack).  Arguments and epilogue
 * as clamd_conv3x3_winograd; stat_rows = clamd_stat_rows(CLAMD_OP_CONV3X3_WINOGRAD24, ...) (wino24.hip). */
int clamd_wino24_pack(const void* jobs_dev, int njobs, int total_blocks, void* stream);
int clamd_conv3x3_winograd24(const float* x, int x_ldc, const float* w_wino, const float* bias, float* y, int y_ldc,
                             float* stats, int stat_rows, int B, int H, int W, int Cin_p, int Cout_p, int relu,
                             const clamd_tuning* tune, void* stream);
/* Weight gradient of the same convolution by the hybrid F(2x4,3x3) (fp32; H even, W a multiple of 4): 24 instead of 32
 * (F(2x2)) or 72 (direct) multiply-adds per 8 output pixels; arguments as clamd_wgrad_winograd (wino24_wgrad.hip). */
size_t clamd_wgrad_winograd24_workspace_bytes(int Rp, int Cp);
int clamd_wgrad_winograd24(const float* gz, int gz_ldc, const float* x, int x_ldc, float* workspace, size_t ws_bytes, float* out,
                           int B, int H, int W, int Rp, int Cp, int R, int C, int r_seg0, int r_seg0p, int c_seg0, int c_seg0p,
                           const clamd_tuning* tune, void* stream);
/* ---- F(2x4,3x3) with PRE-TRANSFORMED operands (wino24g.hip): the wide (>= 256-channel) 3x3 convolutions of
 * models/unet.py:28-33,50-55 (enc3.4, enc4, dec1, dec2, dec3) and their gradients (trainer.py:175).
 * clamd_conv3x3_winograd24 forms B^T d B inside the K loop of every output-slab workgroup; here it is formed ONCE per
 * tensor (clamd_winograd24_transform_input: x [B,H,W,ldc] -> v, clamd_winograd24_input_elems() floats, 3x the
 * activation) and clamd_conv3x3_winograd24_pre runs a transform-free K loop (48 MFMAs + 18 buffer loads straight into
 * the operand registers, no LDS, no VALU) on the SAME packed filters, tile grid, epilogue and statistics rows:
 * bit-identical to clamd_conv3x3_winograd24.  Needs Cin_p % 32 == 0, Cin_p >= 64, Cout_p % 64 == 0.
 * scale / shift (optional, [Cp] each): the transform reads x * scale + shift instead of x -- the nn.BatchNorm2d in front of the
 * convolution (models/unet.py:15-16,30-31: clamd_bn_finalize's scale / shift on the producer's conv+ReLU output) folded into
 * the load, with the zero padding applied AFTER the affine as nn.Conv2d does; the clamd_bn_apply pass of that unit is then not
 * needed when nothing else reads its output. */
size_t clamd_winograd24_input_elems(int B, int H, int W, int Cp);
int clamd_winograd24_transform_input(const float* x, int x_ldc, const float* scale, const float* shift, float* v, int B, int H, int W,
                                     int Cp, void* stream);
int clamd_conv3x3_winograd24_pre(const float* v, const float* w_wino, const float* bias, float* y, int y_ldc,
                                 float* stats, int stat_rows, int B, int H, int W, int Cin_p,
                                 int Cout_p, int relu, const clamd_tuning* tune, void* stream);
/* The narrow layers (64 / 128 channels, levels 0-1 of models/unet.py:49-72, where 3x the activation bytes through HBM would
 * cost more than the in-kernel transform): clamd_conv3x3_winograd24 with the FILTER fragments loaded straight into the MFMA
 * operand registers instead of being staged through LDS (wino24h_kernel, wino24g.hip).  Same arguments, same packed filters,
 * bit-identical results and statistics rows; needs Cout_p % 64 == 0 and Cin_p % 32 == 0. */
int clamd_conv3x3_winograd24_direct_filters(const float* x, int x_ldc, const float* w_wino, const float* bias, float* y, int y_ldc,
                                            float* stats, int stat_rows, int B, int H, int W,
                                            int Cin_p, int Cout_p, int relu, const clamd_tuning* tune, void* stream);
/* Weight gradient of the same convolution as a batched GEMM over the 24 Winograd planes (K = tiles) on operands transformed
 * once: v = the forward image of the convolution INPUT (clamd_winograd24_transform_input, kept from the forward pass: it is
 * read in place as the x-side operand), yt = caller-provided scratch of clamd_wgrad_winograd24_pre_operand_elems(B,H,W,Rp)
 * floats for A4 dY A6^T of gz ([24][tiles][Rp], written here).  Every wave owns a 128 x 128 block of one plane (16 MFMAs per
 * two 16-byte loads, no LDS); split-K slabs in `workspace` (>= clamd_wgrad_winograd24_pre_workspace_bytes), fixed-order
 * reduce with G4^T . G6: deterministic.  Rp and Cp multiples of 128 (round 5; multiples of 256 before); other arguments as
 * clamd_wgrad_winograd24. */
size_t clamd_wgrad_winograd24_pre_operand_elems(int B, int H, int W, int Rp);
/* the gradient-side transform alone (a caller with several streams can run it beside another launch's GEMM); clamd_wgrad_winograd24_pre
 * with gz == NULL then takes yt as already transformed */
int clamd_wgrad_winograd24_pre_transform(const float* gz, int gz_ldc, float* yt, int B, int H, int W, int Rp, void* stream);
size_t clamd_wgrad_winograd24_pre_workspace_bytes(int B, int H, int W, int Rp, int Cp);
int clamd_wgrad_winograd24_pre(const float* gz, int gz_ldc, const float* v, float* yt, float* workspace, size_t ws_bytes, float* out,
                               int B, int H, int W, int Rp, int Cp, int R, int C, int r_seg0, int r_seg0p, int c_seg0, int c_seg0p,
                               const clamd_tuning* tune, void* stream);
/* ---- F(4x4,3x3) with PRE-TRANSFORMED operands (wino44g.hip): the wide 3x3 convolutions at 32x32 and 64x64 of models/unet.py:28-33,
 * 50-55 (enc3, enc4, dec2, dec3) and their gradients (trainer.py:175).  36 multiply-adds per 4 x 4 outputs = 2.25 per output against 3
 * (F(2x4)) and 9 (direct); fp32 error vs fp64 2-3.5e-6 relative.  H and W multiples of 4.  The same call structure as the F(2x4) family
 * above, argument for argument:
 *   clamd_wino44_pack                   filters G6 g G6^T as [Cin_p/8][36][Cout_p][8] (same job table as clamd_wino_pack)
 *   clamd_winograd44_transform_input    x [B,H,W,ldc] (optionally x * scale + shift, zero padding after the affine) -> v,
 *                                       clamd_winograd44_input_elems() floats = 2.25x the activation
 *   clamd_conv3x3_winograd44_pre        transform-free K loop (per wave 24 MFMAs + 12 buffer loads into the operand registers per
 *                                       8-channel chunk; 12 waves per workgroup: six Winograd rows x two 32-channel halves), bias, ReLU,
 *                                       statistics rows (stat_rows = clamd_stat_rows(CLAMD_OP_CONV3X3_WINOGRAD44, ...)); Cout_p % 64 == 0
 *   clamd_wgrad_winograd44_pre[_transform|_operand_elems|_workspace_bytes]   weight gradient as the batched plane GEMM of
 *                                       clamd_wgrad_winograd24_pre over 36 planes: v = the kept forward image, yt = A6 dY A6^T scratch,
 *                                       fixed-order reduce with G6^T . G6 (deterministic); Rp, Cp multiples of 128 (256 x 256 workgroup blocks where both are
 *                                       multiples of 256, 128 x 128 wave blocks dealt k-step by k-step otherwise) */
int clamd_wino44_pack(const void* jobs_dev, int njobs, int total_blocks, void* stream);
size_t clamd_winograd44_input_elems(int B, int H, int W, int Cp);
int clamd_winograd44_transform_input(const float* x, int x_ldc, const float* scale, const float* shift, float* v, int B, int H, int W,
                                     int Cp, void* stream);
int clamd_conv3x3_winograd44_pre(const float* v, const float* w_wino, const float* bias, float* y, int y_ldc,
                                 float* stats, int stat_rows, int B, int H, int W, int Cin_p,
                                 int Cout_p, int relu, const clamd_tuning* tune, void* stream);
size_t clamd_wgrad_winograd44_pre_operand_elems(int B, int H, int W, int Rp);
int clamd_wgrad_winograd44_pre_transform(const float* gz, int gz_ldc, float* yt, int B, int H, int W, int Rp, void* stream);
size_t clamd_wgrad_winograd44_pre_workspace_bytes(int B, int H, int W, int Rp, int Cp);
int clamd_wgrad_winograd44_pre(const float* gz, int gz_ldc, const float* v, float* yt, float* workspace, size_t ws_bytes, float* out,
                               int B, int H, int W, int Rp, int Cp, int R, int C, int r_seg0, int r_seg0p, int c_seg0, int c_seg0p,
                               const clamd_tuning* tune, void* stream);
/* Weight gradient of the same convolution by Winograd (fp32, H and W even): out [R][C][3][3] = G^T (sum over tiles of
 * (A dY A^T) x (B^T d B)) G; arguments as clamd_wgrad(CLAMD_WGRAD_CONV3, ...) (gz = d loss / d conv output, x = conv input). */
size_t clamd_wgrad_winograd_workspace_bytes(int Rp, int Cp);
int clamd_wgrad_winograd(const float* gz, int gz_ldc, const float* x, int x_ldc, float* workspace, size_t ws_bytes, float* out,
                         int B, int H, int W, int Rp, int Cp, int R, int C, int r_seg0, int r_seg0p, int c_seg0, int c_seg0p,
                         const clamd_tuning* tune, void* stream);
/* 1x1 convolution, NHWC output, same epilogue options as clamd_conv3x3 (bias, ReLU, BN statistics).  Used for the
 * data gradient of the head (unet.py:72) and, on an im2col'ed input (clamd_nchw_im2col3), for the first conv
 * enc1.0 (unet.py:50, Cin = 3).  w_packed [1][Cout_p][Cin_p]. */
int clamd_conv1x1(const void* x, int x_ldc, const void* w_packed, const float* bias, void* y, int y_ldc,
                  float* stats, const void* bn_y, float* bn_sums, int stat_rows, int B, int H, int W, int Cin_p, int Cout_p,
                  int relu, int dtype, void* stream);
/* the head nn.Conv2d(conv_dim, num_classes, k1) (unet.py:72): logits written as fp32 NCHW [B,num_classes,H,W]. */
int clamd_conv1x1_logits(const void* x, int x_ldc, const void* w_packed, const float* bias, float* logits_nchw,
                         int B, int H, int W, int Cin_p, int Cout_p, int num_classes, int dtype, void* stream);
/* The head with the arg-max over classes fused into its epilogue (eval forward, trainer.py:279 `torch.max(outputs, 1)`;
 * SURVEY.md §8f row 4): pred int64 [B,H,W], first maximum wins; logits_nchw may be NULL (the logits are then never written). */
int clamd_conv1x1_argmax(const void* x, int x_ldc, const void* w_packed, const float* bias, long long* pred, float* logits_nchw,
                         int B, int H, int W, int Cin_p, int Cout_p, int num_classes, int dtype, void* stream);
/* nn.ConvTranspose2d(k2,s2)+bias (unet.py:34): x [B,h,w,Cin_p] -> y [B,2h,2w,(ldc)] channels [0,Cout_p) of the
 * slice y points at.  w_packed [4][Cout_p][Cin_p] (tap q = 2*dy+dx). */
int clamd_convT2x2_fwd(const void* x, int x_ldc, const void* w_packed, const float* bias, void* y, int y_ldc, int B,
                       int h, int w, int Cin_p, int Cout_p, int dtype, void* stream);
/* data gradient of the above: gy [B,2h,2w,...] -> gx [B,h,w,Cin_p].  w_packed [Cin_p][4][Cout_p]. */
int clamd_convT2x2_dgrad(const void* gy, int gy_ldc, const void* w_packed, void* gx, int gx_ldc, const void* bn_y,
                         float* bn_sums, int stat_rows, int B, int h, int w, int Cin_p, int Cout_p, int dtype, void* stream);

/* ---- weight gradients (wgrad.hip) --------------------------------------------------------------------------
 * out[r][c][t] = sum_pixels a[p, r] * b[nbr_t(p), c]  written in the parameter's own fp32 layout:
 *   CONV3: a = d(conv output), b = conv input  -> d weight [Cout][Cin][3][3]
 *   PW   : a = d logits,       b = head input  -> d weight [K][Cin][1][1]
 *   UP2  : a = convT input,    b = d(convT output) -> d weight [Cin][Cout][2][2]
 * R,C logical sizes.  Physical channel p maps to logical p (p < seg0p and p < seg0), to padding (seg0 <= p < seg0p),
 * or to seg0 + (p - seg0p): concat inputs keep each half padded separately; a single segment is (seg0, seg0p) =
 * (logical size, physical size).  workspace: fp32 split-K slabs, size >= clamd_wgrad_workspace_bytes(). */
size_t clamd_wgrad_workspace_bytes(int mode, int B, int H, int W, int Rp, int Cp, int dtype);
int clamd_wgrad(int mode, const void* a, int a_ldc, const void* b, int b_ldc, float* workspace, size_t ws_bytes,
                float* out, int B, int H, int W, int Rp, int Cp, int R, int C, int r_seg0, int r_seg0p, int c_seg0,
                int c_seg0p, int dtype, const clamd_tuning* tune, void* stream);

/* ---- BatchNorm / ReLU / MaxPool / concat plumbing (elementwise.hip) ------------------------------------------
 * nn.BatchNorm2d train mode (unet.py:15): partial rows stats[stat_rows][2][Cp] -> scale/shift (+ running stats,
 * momentum 0.1, unbiased var); rows are added in a fixed order in fp64, mean/variance formed in fp64.
 * stats == NULL: eval mode, normalise with the running statistics.  num_batches_tracked (optional, int64 scalar): nn.BatchNorm2d's
 * counter, incremented in train mode (stats != NULL). */
int clamd_bn_finalize(const float* stats, int stat_rows, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float* scale, float* shift, float* save_mean, float* save_istd,
                      int Cp, int C, double count, double momentum, double eps, long long* num_batches_tracked, void* stream);
/* out = y*scale+shift into `out` (possibly a concat slice: replaces torch.cat, unet.py:83-87); pooled (optional)
 * = nn.MaxPool2d(2,2) of out (unet.py:12,80). */
int clamd_bn_apply(const void* y, int y_ldc, const float* scale, const float* shift, void* out, int out_ldc,
                   void* pooled, int p_ldc, int B, int H, int W, int Cp, int dtype, void* stream);
/* backward of ReLU->BatchNorm (+ max-pool routing of `gp`, the gradient w.r.t. the pooled tensor):
 * reduce -> partial rows sums[sum_rows][5][Cp] (sum_rows = clamd_stat_rows(CLAMD_OP_BN_BWD_REDUCE, ...)),
 * finalize -> k0,k1,k2 + d gamma, d beta, d conv-bias, apply -> g_z. */
int clamd_bn_bwd_reduce(const void* ga, int ga_ldc, const void* gp, int gp_ldc, const void* y, int y_ldc,
                        const float* scale, const float* shift, float* sums, int sum_rows, int B, int H, int W, int Cp,
                        int dtype, const clamd_tuning* tune, void* stream);
int clamd_bn_bwd_finalize(const float* sums, int sum_rows, const float* gamma, const float* save_mean, const float* save_istd,
                          float* k012, float* dgamma, float* dbeta, float* dbias, int Cp, int C, double count,
                          void* stream);
int clamd_bn_bwd_apply(const void* ga, int ga_ldc, const void* gp, int gp_ldc, const void* y, int y_ldc,
                       const float* scale, const float* shift, const float* k012, void* gz, int gz_ldc, int B,
                       int H, int W, int Cp, int dtype, void* stream);
/* Two-sum form (the persistent bf16 convolution kernel, see clamd_conv3x3_bn_sums): the producing data-gradient launch wrote only
 * sum g and sum g y (rows k = 0, 1; k = 2..4 NaN), which is all k0, k1, k2, d gamma and d beta need; pass dbias = NULL to
 * clamd_bn_bwd_finalize and take the convolution's bias gradient (models/unet.py:13,16: d conv-bias = sum of g_z) where g_z is formed:
 * clamd_bn_bwd_apply_sums = clamd_bn_bwd_apply without pooling + partial rows gz_rows[nrows][Cp] of sum g_z (nrows =
 * clamd_bn_bwd_apply_sums_rows(B, H, W, Cp), one row per workgroup, plain stores), then clamd_rows_sum adds rows 0..nrows-1 in a
 * fixed order in fp64 and OVERWRITES out[0..C). */
int clamd_bn_bwd_apply_sums_rows(int B, int H, int W, int Cp);
int clamd_bn_bwd_apply_sums(const void* ga, int ga_ldc, const void* y, int y_ldc, const float* k012, void* gz, int gz_ldc,
                            float* gz_rows, int nrows, int B, int H, int W, int Cp, int dtype, void* stream);
int clamd_rows_sum(const float* rows, int nrows, float* out, int Cp, int C, void* stream);
/* nn.MaxPool2d(2,2) alone (models/unet.py:12: the first layer of a DownBlock run as a stand-alone block, blocks.py; inside the UNet step
 * the pool is part of clamd_bn_apply / clamd_bn_bwd_*): x [B,H,W,ldc] -> pooled [B,H/2,W/2,ldc]; backward: gx [B,H,W,ldc] = gp at the first
 * maximum of each window (the tie rule of clamd_bn_apply and of torch's CPU kernel), 0 elsewhere.
 * sign (optional, [Cp]): channels with sign[c] < 0 take the window MINIMUM instead -- the pool of a tensor scale * x + shift that is never
 * written (a BatchNorm folded into the consumers of the pooled tensor, see clamd_bn_fold_bias) taken on x: max(s x + t) = s min(x) + t. */
int clamd_maxpool2x2(const void* x, int x_ldc, const float* sign, void* pooled, int p_ldc, int B, int H, int W, int Cp, int dtype, void* stream);
int clamd_maxpool2x2_bwd(const void* x, int x_ldc, const float* sign, const void* gp, int gp_ldc, void* gx, int gx_ldc, int B, int H, int W, int Cp,
                         int dtype, void* stream);
/* out[c] = sum_pixels g[p,c]  (bias gradients of convT / head): per-block partial rows in `workspace`
 * (>= clamd_channel_sum_workspace_bytes(Cp)), then a fixed-order fp64 sum -- no float atomics, out is overwritten. */
size_t clamd_channel_sum_workspace_bytes(int Cp);
int clamd_channel_sum(const void* g, int ldc, float* out, long long npix, int Cp, int C, int dtype, float* workspace,
                      size_t ws_bytes, const clamd_tuning* tune, void* stream);
/* boundary layout conversion: visible tensors are fp32 NCHW; replaces images.to(device) layout
 * handling (trainer.py:168) and feeds grad_output into the backward. */
int clamd_nchw_to_nhwc(const float* src, void* dst, int ldc, int B, int C, int H, int W, int Cp, double mul,
                       int dtype, void* stream);
int clamd_nhwc_to_nchw(const void* src, int ldc, float* dst, int B, int C, int H, int W, int dtype, void* stream);
/* NCHW fp32 image -> NHWC with the 3x3 neighbourhood folded into channels: dst[p, c*9 + ky*3 + kx] (9*C <= Cp). */
int clamd_nchw_im2col3(const float* src, void* dst, int ldc, int B, int C, int H, int W, int Cp, int dtype, void* stream);

/* ---- parameters, loss, optimiser, metrics (misc.hip) ---------------------------------------------------------
 * clamd_pack: one fused launch re-packing every fp32 master parameter into the layouts above (job table built by
 * the host). */
int clamd_pack(const void* jobs_dev, int njobs, int total_blocks, int dtype, void* stream);
/* nn.CrossEntropyLoss() forward+backward (trainer.py:113,174-175) on fp32 NCHW logits / int64 labels, plus the
 * build-defined distillation term when old_logits != NULL.  loss3 = {total, ce, kd}. */
size_t clamd_ce_workspace_bytes(void);
/* byte offset inside the workspace of an unsigned int: pixels whose label is neither ignore_index nor a class (torch's
 * CrossEntropyLoss asserts on those; here they are left out of the mean and counted for the caller to check). */
size_t clamd_ce_bad_label_count_offset(void);
int clamd_ce_fwd_bwd(const float* logits, const long long* labels, const float* old_logits, int K_old_total, int c_old,
                     double temperature, double lam, float* dlogits, float* loss3, void* workspace, size_t ws_bytes,
                     int B, int K, int H, int W, long long ignore_index, double grad_scale, void* stream);
/* The same loss without the distillation term, in two launches that need no memset and no atomics: clamd_ce_count writes the count of
 * the pixels that take part in the mean as one partial pair per workgroup into the workspace (plain stores), the loss / gradient pass
 * (same workspace, ordered behind it by the caller) adds them.  Needs H * W % 4 == 0.  dl_nhwc (optional): d logits ALSO as
 * an NHWC tensor [B,H,W,dl_ldc] of compute dtype dl_dtype, channels K .. dl_ldc-1 zero (dl_ldc >= 32) -- the operand of the 1x1
 * head's data gradient (models/unet.py:72, trainer.py:175), which then needs no clamd_nchw_to_nhwc pass over d logits. */
int clamd_ce_count(const long long* labels, int B, int K, int H, int W, long long ignore_index, void* workspace, size_t ws_bytes, void* stream);
int clamd_ce_fwd_bwd_counted(const float* logits, const long long* labels, float* dlogits, void* dl_nhwc, int dl_ldc, int dl_dtype,
                             float* loss3, void* workspace, size_t ws_bytes, int B, int K, int H, int W, long long ignore_index,
                             double grad_scale, void* stream);
/* torch.optim.Adam.step over all parameters in one launch (trainer.py:108-110,176); hyper/step/derived live on the
 * device so a captured graph can be replayed with a new learning rate.  l2_accum_dev (optional, with the L2-to-old-weights
 * term): 1 + nchunks floats, [0] = sum ||theta - theta_old||^2 of this step, [1..] = per-workgroup partials added in a fixed
 * order (no float atomics: the value is bit-reproducible). */
int clamd_adam_step(const void* tensors_dev, const void* chunks_dev, int nchunks, const float* hyper_dev, int* step_dev,
                    float* derived_dev, float* l2_accum_dev, void* stream);
/* argmax over classes + confusion matrix (trainer.py:183-188, metrics.py:32-38). */
int clamd_argmax_confusion(const float* logits, const long long* labels, long long* pred, unsigned long long* conf,
                           int B, int K, int Kc, int H, int W, void* stream);
/* Data path: Pad(10)+CenterCrop+ToTensor+Normalize(0.5,0.5) of an interleaved uint8 RGB image
 * (main.py:18-23) -> image_out fp32 [3,h,w]; Pad+CenterCrop+voc.to_mask of the RGB mask (datasets/voc.py:56-72,
 * 140-142) -> label_out int64 [h,w] (void -> 0).  (oy, ox): source coordinate of output pixel (0,0).  Pixels whose
 * colour is not in the palette are counted in *bad_count (the reference raises ValueError) and set to 0. */
int clamd_voc_prepare(const unsigned char* img_rgb, const unsigned char* mask_rgb, float* image_out, long long* label_out,
                      int Hs, int Ws, int oy, int ox, int h, int w, unsigned int* bad_count, void* stream);
/* voc.to_rgb (datasets/voc.py:74-89): labels int64 [N,H,W] -> palette colours [N,3,H,W] (0..255 as float). */
int clamd_label_to_rgb(const long long* labels, float* rgb, long long n_img, long long hw, void* stream);
int clamd_fill_f32(float* p, long long n, double v, void* stream);
/* `ncus` workgroups (1..256) that each keep one whole CU (96 KB of LDS) for `usec` microseconds of wall clock and touch no memory:
 * what an RCCL channel workgroup does to the one-workgroup-per-CU kernels of this library during a collective (trainer.py:120-122
 * becomes one process per GPU, ddp.py).  The data-parallel host code uses it with ncus = 1 to MEASURE how many hardware queues the
 * runtime multiplexes its streams onto (ddp.hw_queues); tools/cu_steal.py rehearses held CUs with it. */
int clamd_hold_cus(int ncus, int usec, void* stream);
/* Gradient exchange in bf16 (replaces the reduce of nn.DataParallel, trainer.py:120-122, for BASELINE.json configs[2]/[4]
 * "bf16 DDP"): a bucket of the flat fp32 gradient buffer rounded to bf16 (rne) for the RCCL all-reduce and widened back.
 * A bucket may start at any element: the bf16 buffer must sit at the same element phase, (address / element size) % 8. */
int clamd_f32_to_bf16(const float* src, void* dst_bf16, long long n, void* stream);
int clamd_bf16_to_f32(const void* src_bf16, float* dst, long long n, void* stream);
/* p[i] *= *scale_dev for a DEVICE scalar, nothing at all when it is exactly 1 (the upstream gradient loss.backward()
 * hands to the loss function, trainer.py:175): no host sync, no pass over d logits in the common case. */
int clamd_scale_by_device_scalar(float* p, long long n, const float* scale_dev, void* stream);
/* ... and for an NHWC tensor of a compute dtype (n logical elements, a multiple of 8; the second copy of d logits above); also_f32
 * (optional, n_f32 elements): an fp32 tensor scaled by the same launch. */
int clamd_scale_by_device_scalar_nhwc(void* p, long long n, int dtype, const float* scale_dev, float* also_f32, long long n_f32, void* stream);

#ifdef __cplusplus
}
#endif
#endif
