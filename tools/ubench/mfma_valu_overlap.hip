// Micro-benchmark (gfx950): do VALU / LDS instructions of one wave overlap with the MFMA stream of ANOTHER wave on the same
// SIMD?  256 workgroups x 512 threads (one workgroup per CU, 2 waves per SIMD): waves 0-3 run back-to-back independent MFMAs,
// waves 4-7 run independent VALU fma chains (or LDS reads).  Times: MFMA alone, VALU alone, both.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ovl tools/ubench/mfma_valu_overlap.hip && /tmp/ovl
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE_MFMA, int MODE_OTHER>   // MODE_MFMA: 0 none, 1 bf16 32x32x16, 2 f32 32x32x2;  MODE_OTHER: 0 none, 1 VALU fma, 2 LDS read b128
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    __shared__ float4 lds[4096];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    if (wave < 4) {
        if (MODE_MFMA == 0) return;
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)1.0f; }
        float fa = threadIdx.x, fb = 1.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MODE_MFMA == 1) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
                    else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[j], 0, 0, 0);
                }
        }
        float s = 0;
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        if (MODE_OTHER == 0) return;
        if (MODE_OTHER == 1) {
            float x[8];
            for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j;
            const float m = 1.0001f, c = 0.5f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 32; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], m, c);
            }
            float s = 0;
            for (int j = 0; j < 8; ++j) s += x[j];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        } else {
            float4 s = make_float4(0, 0, 0, 0);
            int idx = threadIdx.x & 255;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    const float4 v = lds[(idx + u * 64) & 4095];
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
            }
            out[blockIdx.x * 512 + threadIdx.x] = s.x + s.y + s.z + s.w;
        }
    }
}

template <int A, int B> float run(float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int it = 2000;
    printf("per workgroup: 4 waves x %d x 32 MFMAs; other 4 waves x %d x 256 VALU fma / x 32 ds_read_b128\n", it, it);
    printf("bf16 MFMA alone        %8.1f us\n", run<1, 0>(out, it));
    printf("f32  MFMA alone        %8.1f us\n", run<2, 0>(out, it));
    printf("VALU alone             %8.1f us\n", run<0, 1>(out, it));
    printf("LDS reads alone        %8.1f us\n", run<0, 2>(out, it));
    printf("bf16 MFMA + VALU       %8.1f us\n", run<1, 1>(out, it));
    printf("f32  MFMA + VALU       %8.1f us\n", run<2, 1>(out, it));
    printf("bf16 MFMA + LDS reads  %8.1f us\n", run<1, 2>(out, it));
    printf("f32  MFMA + LDS reads  %8.1f us\n", run<2, 2>(out, it));
    return 0;
}
