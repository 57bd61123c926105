/* clamd_debug.h -- test and measurement scaffolding exported by libclamd.so.  NOT part of the product ABI: include/clamd.h
 * does not include this file, nothing in continual-learning_amd/ calls these on the train-step path, and a maintainer binding
 * the reference never needs them.  Used by tools/ only.
 */
#ifndef CLAMD_DEBUG_H
#define CLAMD_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* Rehearsal aid for data parallelism on a one-GPU box: `ncus` workgroups that each hold a whole CU for `usec` microseconds
 * and do nothing else -- what an RCCL channel workgroup does to the one-workgroup-per-CU MFMA kernels during a collective
 * (tools/cu_steal.py measures the step with and without clamd_tuning::cu_reserve). */
int clamd_debug_hold_cus(int ncus, int usec, void* stream);

/* Diagnostic builds only (python continual-learning_amd/build.py --diag, -DCLAMD_DIAG): in-kernel cycle stamps summed over
 * workgroups, read and optionally reset (tools/w24_diag.py, ws_diag.py, wg_diag.py, ww_diag.py).  Absent from the shipped
 * library. */
#ifdef CLAMD_DIAG
int clamd_debug_w24_diag(unsigned long long* out8, int reset);
int clamd_debug_ww_diag(unsigned long long* out4, int reset);
int clamd_debug_ws_diag(unsigned long long* out8, int reset);
int clamd_debug_pws_diag(unsigned long long* out8, int reset);
int clamd_debug_wg_diag(unsigned long long* out8, int reset);
#endif

#ifdef __cplusplus
}
#endif
#endif
