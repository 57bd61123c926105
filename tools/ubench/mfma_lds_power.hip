// Micro-benchmark (gfx950): what does a second, busy wave per SIMD cost a power-limited bf16 MFMA loop -- clock or cycles?  256 workgroups x 512
// threads: waves 0-3 (one per SIMD) run back-to-back v_mfma_f32_32x32x16_bf16 on pseudo-random operands; waves 4-7 sleep, run VALU chains, read or
// write LDS.  Reports the MFMA loop's own wall time (s_memrealtime), its shader cycles (s_memtime) and the clock = cycles / time.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mlp tools/ubench/mfma_lds_power.hip && /tmp/mlp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int R2, int MODE = 0>   // R2: operations per loop trip of each second wave (0 = the wave exits); MODE 0: ds_read_b128, 1: s_sleep only, 2: VALU xor chains only, 3: ds_write_b128, 4: wait at a barrier the MFMA waves reach after their loop
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, int iters, const uint4* big = nullptr) {
    __shared__ uint4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) { const unsigned h = hash(i + blockIdx.x * 4096); lds[i] = make_uint4(h, hash(h), hash(h + 1), hash(h + 2)); }
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        bf16x8 a[4], b[4];
        for (int j = 0; j < 4; ++j)
            for (int e = 0; e < 8; ++e) {
                const unsigned ha = hash(threadIdx.x * 64 + j * 8 + e + blockIdx.x * 7919), hb = hash(ha + 12345);
                a[j][e] = (__bf16)(((int)(ha & 0xffff) - 32768) * (1.f / 32768.f));
                b[j][e] = (__bf16)(((int)(hb & 0xffff) - 32768) * (1.f / 32768.f));
            }
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + u) & 3], b[(j + 2 + u) & 3], acc[i][j], 0, 0, 0);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        float s = 0.f;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (MODE == 4) __syncthreads();
        if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }      // shader cycles, 100-MHz ticks of the MFMA loop itself
    } else if (MODE == 4) {
        __syncthreads();      // the second waves wait at the barrier for the whole MFMA loop
    } else if (R2 > 0) {
        uint4 s = make_uint4(threadIdx.x, 1, 2, 3);
        int idx = threadIdx.x & 255;
        for (int it = 0; it < iters * 24; ++it) {      // longer than the MFMA waves run: their loop is timed with the second waves active throughout
#pragma unroll
            for (int r = 0; r < R2; ++r) {
                if (MODE == 0) { const uint4 v = lds[(idx + r * 64 + it * 7) & 4095]; s.x ^= v.x; s.y ^= v.y; s.z ^= v.z; s.w ^= v.w; }
                else if (MODE == 2) { s.x = s.x * 1664525u + 1013904223u; s.y ^= s.x; s.z += s.y; s.w ^= s.z; }
                else if (MODE == 3) { lds[(idx + r * 64 + it * 7) & 4095] = s; s.x += 1; }
                else if (MODE == 5) { const uint4 v = big[((size_t)(s.x & 0xfffffu) * 64 + (threadIdx.x & 63)) & ((1u << 26) - 1)]; s.x = s.x * 1664525u + v.x + 1u; }      // a dependent HBM load: the wave waits on vmcnt
            }
            if (MODE != 5) __builtin_amdgcn_s_sleep(MODE == 1 ? 8 : 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = __uint_as_float(s.x ^ s.y ^ s.z ^ s.w);
    }
}

template <int R2, int MODE = 0> void run(float* out, unsigned long long* cyc, int iters, const uint4* big = nullptr) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<R2, MODE>), dim3(256), dim3(512), 0, 0, out, cyc, MODE == 5 ? iters / 8 : iters, big);
    (void)hipDeviceSynchronize();
    float best = 1e30f; unsigned long long c[2] = {0, 0};
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<R2, MODE>), dim3(256), dim3(512), 0, 0, out, cyc, MODE == 5 ? iters / 8 : iters, big);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost); }
    }
    if (MODE == 5) iters /= 8;
    const double flop = 256.0 * 4 * (double)iters * 32 * 32768.0;
    const double us = c[1] / 100.0;      // the MFMA loop's own wall time
    static const char* what[] = {"ds_read_b128", "nothing (s_sleep 8)", "VALU mul/xor/add", "ds_write_b128", "waiting at s_barrier", "dependent 16-byte HBM load (waits on vmcnt)"};
    printf("second wave per SIMD: %d x %s per trip: MFMA loop %8.1f us  %7.1f TFLOP/s  %.1f cycles per MFMA  clock %.2f GHz  (kernel %.3f ms)\n", R2, what[MODE],
           us, flop / us / 1e6, (double)c[0] / ((double)iters * 32), (double)c[0] / us / 1e3, best);
}

int main() {
    float* out; unsigned long long* cyc; (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 16);
    const int it = 20000;
    run<0>(out, cyc, it); run<1, 1>(out, cyc, it); run<4, 2>(out, cyc, it); run<16, 2>(out, cyc, it); run<1>(out, cyc, it); run<4>(out, cyc, it);
    run<1, 3>(out, cyc, it); run<4, 3>(out, cyc, it); run<1, 4>(out, cyc, it);
    uint4* big; (void)hipMalloc(&big, (size_t)1 << 30); (void)hipMemset(big, 1, (size_t)1 << 30);
    run<1, 5>(out, cyc, it, big); run<0>(out, cyc, it);
    return 0;
}
