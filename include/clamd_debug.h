/* clamd_debug.h -- test and measurement scaffolding exported by libclamd.so.  NOT part of the product ABI: include/clamd.h
 * does not include this file, nothing in continual-learning_amd/ calls these on the train-step path, and a maintainer binding
 * the reference never needs them.  Used by tools/ and by bench.py's calibration line.
 */
#ifndef CLAMD_DEBUG_H
#define CLAMD_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* Older name of clamd_hold_cus (include/clamd.h), kept for the tools of earlier rounds. */
int clamd_debug_hold_cus(int ncus, int usec, void* stream);

/* Calibration for bench.py: a bare MFMA loop (v_mfma_f32_32x32x16_bf16 for CLAMD_BF16 / CLAMD_SPLIT, v_mfma_f32_32x32x2_f32 for
 * CLAMD_F32) on pseudo-random register operands, 256 workgroups x 4 waves x iters x 32 MFMAs.  Returns the FLOP of the launch (-1 on
 * error); the caller times it with events.  bf16 MFMA loops on real data are power-limited well below the nominal peak. */
long long clamd_debug_mfma_rate(int dtype, int iters, float* sink_65536, void* stream);

/* Diagnostic builds only (python continual-learning_amd/build.py --diag, -DCLAMD_DIAG): in-kernel cycle stamps summed over
 * workgroups, read and optionally reset (tools/w24_diag.py, ws_diag.py, wg_diag.py, ww_diag.py).  Absent from the shipped
 * library. */
#ifdef CLAMD_DIAG
int clamd_debug_w24_diag(unsigned long long* out8, int reset);
int clamd_debug_w44_diag(unsigned long long* out8, int reset);
int clamd_debug_ww_diag(unsigned long long* out4, int reset);
int clamd_debug_ws_diag(unsigned long long* out8, int reset);
int clamd_debug_pws_diag(unsigned long long* out8, int reset);
int clamd_debug_wg_diag(unsigned long long* out8, int reset);
#endif

#ifdef __cplusplus
}
#endif
#endif
