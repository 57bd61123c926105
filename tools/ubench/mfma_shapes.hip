// Micro-benchmark (gfx950): sustained rate of bare MFMA loops, v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 and
// v_mfma_f32_32x32x2_f32 against v_mfma_f32_16x16x4_f32, on pseudo-random operands (the sustained clock depends on the data), one or two
// waves per SIMD, every CU busy.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shapes tools/ubench/mfma_shapes.hip && /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int SHAPE>   // 0: 32x32x16, 4 independent accumulators;  1: 16x16x32, 16 independent accumulators (the same 64x64 output block per wave)
__global__ void __launch_bounds__(512) k(float* out, int iters, int zero) {
    bf16x8 a[4], b[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 8; ++e) {
            const unsigned ha = hash(threadIdx.x * 64 + j * 8 + e + blockIdx.x * 7919), hb = hash(ha + 12345);
            a[j][e] = (__bf16)(zero ? 0.f : ((int)(ha & 0xffff) - 32768) * (1.f / 32768.f));
            b[j][e] = (__bf16)(zero ? 0.f : ((int)(hb & 0xffff) - 32768) * (1.f / 32768.f));
        }
    float s = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)          // 8 x 4 MFMAs x 32768 FLOP
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + u) & 3], b[(j + 2 + u) & 3], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    } else {
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)          // 4 x 16 MFMAs x 16384 FLOP = the same FLOP per iteration as above
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + u) & 3], b[(j + u + 1) & 3], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int SHAPE>   // 0: 32x32x2 f32, 4 accumulators;  1: 16x16x4 f32, 16 accumulators
__global__ void __launch_bounds__(512) kf(float* out, int iters, int zero) {
    float a[4], b[4];
    for (int j = 0; j < 4; ++j) {
        const unsigned ha = hash(threadIdx.x * 64 + j + blockIdx.x * 7919), hb = hash(ha + 12345);
        a[j] = zero ? 0.f : ((int)(ha & 0xffffff) - 8388608) * (1.f / 8388608.f);
        b[j] = zero ? 0.f : ((int)(hb & 0xffffff) - 8388608) * (1.f / 8388608.f);
    }
    float s = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)          // 8 x 4 MFMAs x 4096 FLOP
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + u) & 3], b[(j + 2 + u) & 3], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    } else {
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)          // 4 x 16 MFMAs x 2048 FLOP
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + u) & 3], b[(j + u + 1) & 3], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int SHAPE> void runf(float* out, int threads, int iters, int zero, const char* name) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((kf<SHAPE>), dim3(256), dim3(threads), 0, 0, out, iters, zero);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((kf<SHAPE>), dim3(256), dim3(threads), 0, 0, out, iters, zero);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double flop = 256.0 * (threads / 64) * (double)iters * 32 * 4096.0;
    printf("%-28s %d waves/SIMD %s: %8.3f ms  %7.1f TFLOP/s  (implied matrix-pipe clock %.2f GHz)\n", name, threads / 256, zero ? "zeros " : "random",
           best, flop / best / 1e9, flop / best / 1e9 / 157.3 * 2.4);
}

template <int SHAPE> void run(float* out, int threads, int iters, int zero, const char* name) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE>), dim3(256), dim3(threads), 0, 0, out, iters, zero);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE>), dim3(256), dim3(threads), 0, 0, out, iters, zero);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double flop = 256.0 * (threads / 64) * (double)iters * 32 * 32768.0;
    // cycles per MFMA-equivalent of 32768 FLOP if the pipe never idles: 32 -> implied clock
    printf("%-28s %d waves/SIMD %s: %8.3f ms  %7.1f TFLOP/s  (implied matrix-pipe clock %.2f GHz)\n", name, threads / 256, zero ? "zeros " : "random",
           best, flop / best / 1e9, flop / best / 1e9 / 2500.0 * 2.4);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    const int it = 20000;
    for (int zero = 0; zero < 2; ++zero)
        for (int threads = 256; threads <= 512; threads += 256) {
            run<0>(out, threads, it, zero, "v_mfma_f32_32x32x16_bf16");
            run<1>(out, threads, it, zero, "v_mfma_f32_16x16x32_bf16");
        }
    for (int zero = 0; zero < 2; ++zero) {
        runf<0>(out, 256, it / 2, zero, "v_mfma_f32_32x32x2_f32");
        runf<1>(out, 256, it / 2, zero, "v_mfma_f32_16x16x4_f32");
    }
    return 0;
}
