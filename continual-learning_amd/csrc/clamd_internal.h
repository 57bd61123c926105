// Internal glue shared by the translation units of libclamd.so (error reporting, tuning defaults, device facts).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/clamd.h"

// Records a message retrievable through clamd_last_error() and returns a negative status.
int clamd_fail(const char* msg);
// hipGetLastError() after a launch -> 0 or a negative status with the HIP error string recorded.
int clamd_check_launch(const char* what);

// The library has no mutable state: kernel-structure choices arrive per call (include/clamd.h `clamd_tuning`).
// NULL -> these defaults (measured choices).
inline const clamd_tuning& clamd_default_tuning() {
    static const clamd_tuning d = {/*igemm_pws*/ 1, /*igemm_ws*/ 2, /*igemm_variant*/ 0, /*pws_wres*/ 1,
                                   /*wgrad_ws*/ 1, /*wgrad_dma*/ 1, /*wgrad_xcd*/ 1, /*wgrad_blocks*/ 512, /*wgrad_tw16*/ 0,
                                   /*wino_band*/ 0, /*wino_persist*/ 1, /*wino_mt*/ 0,
                                   /*bn_reduce_blocks*/ 0, /*chsum_blocks*/ 0, /*cu_reserve*/ 0, /*wino_half*/ 0, /*wgrad_streamk*/ 1, {0, 0, 0, 0, 0, 0, 0}};
    return d;
}
inline const clamd_tuning& clamd_tune(const clamd_tuning* t) { return t ? *t : clamd_default_tuning(); }
// 0 or a negative status with the reason recorded
int clamd_check_tuning(const clamd_tuning* t);

// bf16x3 (CLAMD_SPLIT) activations live in hi/lo planes per 16-channel group (common.hip.h, Vec8<split_t>): the kernels
// find a group from the address bits, so a tensor (or a channel slice of one) must start on 64 bytes and have a pitch
// that is a whole number of groups.  0, or a negative status with the reason recorded; other dtypes always pass.
inline int clamd_check_split(int dtype, const void* p, int ldc) {
    if (dtype != CLAMD_SPLIT || !p) return 0;
    if (((unsigned long long)p & 63ull) || ldc % 16)
        return clamd_fail("bf16x3 tensors need a 64-byte aligned base and a pitch that is a multiple of 16 channels");
    return 0;
}

// CU count of the current device (a device fact, not state: every MI355X answers 256).
inline int clamd_query_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
}
inline int clamd_num_cus() {
    static const int n = clamd_query_cus();      // initialised once (thread-safe), never written again
    return n;
}
// CUs the persistent grids may occupy
inline int clamd_usable_cus(const clamd_tuning& t) {
    const int n = clamd_num_cus() - (t.cu_reserve > 0 ? t.cu_reserve : 0);
    return n < 8 ? 8 : n;
}

// partial statistics rows of a launch (see clamd_stat_rows): pixel tiles of the Winograd forward kernel (wino.hip) and
// workgroups of the BatchNorm-backward reduction (elementwise.hip)
long long clamd_winograd_stat_rows(int B, int H, int W, int Cout_p, const clamd_tuning& tn);
long long clamd_bn_bwd_reduce_rows(int B, int H, int W, int Cp, bool pooled, const clamd_tuning& tn);

// the three filter-pack launches with an optional border-class bias table appended (grid = total_blocks + Cout_p; bnfold.hip)
namespace clamd { struct FoldBias; }
int clamd_launch_pack(const void* jobs_dev, int njobs, int total_blocks, int dtype, const clamd::FoldBias* fold, hipStream_t stream);
int clamd_launch_wino_pack(const void* jobs_dev, int njobs, int total_blocks, const clamd::FoldBias* fold, hipStream_t stream);
int clamd_launch_wino24_pack(const void* jobs_dev, int njobs, int total_blocks, const clamd::FoldBias* fold, hipStream_t stream);
