// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the UNet train-step path.
// Wave = 64 lanes; MFMA tiles are 32x32 (v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// PASS_PRIO(): raised wave priority for the HBM-bound passes of the critical chain (elementwise.hip has the measurement).
#ifndef CLAMD_NO_PASS_PRIO
#define PASS_PRIO() __builtin_amdgcn_s_setprio(3)
#else
#define PASS_PRIO() do { } while (0)
#endif
// SIDE_PRIO(): the same for the small passes of the weight-gradient streams (operand transforms, partial-row reductions, bias-gradient sums):
// beside the F(4x4) convolution of the critical chain they ran at a sixth of their speed (tools/corun_lab.py), and the weight-gradient GEMM
// behind them waits for every one of them.  fp32 step 19.05 -> 18.95 ms in two interleaved pairs (profiles/r05_step_ab.txt).
#define SIDE_PRIO() PASS_PRIO()

namespace clamd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct bf16_t { uint16_t v; };   // storage type tag for bf16 tensors
// "split" compute type (bf16x3): every value x is STORED as the bf16 pair hi = rne(x), lo = rne(x - hi) -- 4 bytes per
// element, laid out in hi/lo planes per 16-channel group (see Vec8<split_t>) so that the MFMA kernels stage activations
// with plain 16-byte copies, as they do the packed weights -- and multiplied as hi*hi + hi*lo + lo*hi in fp32 with three
// bf16 MFMAs (relative product error ~2^-17 instead of bf16's 2^-9, at 3/16 of the fp32-MFMA cost).  The producer of a
// tensor splits it ONCE in its epilogue; round 1 stored fp32 and re-split in every consumer's staging loop (3-10x per
// element), which cost the convolutions 10-16% (tools ablation).
struct split_t { float v; };

// ---- dtype traits -------------------------------------------------------------------------------
template <typename T> struct DT;
template <> struct DT<float> {
    static constexpr int BYTES = 4;
    static constexpr int VEC = 4;        // elements per 16-byte group
    static constexpr int KC = 16;        // channels per 64-byte K-chunk
    __device__ static inline float ld(const float* p) { return *p; }
};
template <> struct DT<bf16_t> {
    static constexpr int BYTES = 2;
    static constexpr int VEC = 8;
    static constexpr int KC = 32;
};
template <> struct DT<split_t> {
    static constexpr int BYTES = 4;
    static constexpr int VEC = 4;
    static constexpr int KC = 16;
};

__device__ inline float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even; plain cast form keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
__device__ inline uint16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ inline uint32_t pack2bf(float lo, float hi) {      // one v_cvt_pk_bf16_f32
    bf16x2_t r;
    r.x = (__bf16)lo;
    r.y = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, r);
}

// 8 consecutive channels <-> fp32 registers
template <typename T> struct Vec8;
template <> struct Vec8<float> {
    __device__ static inline void load(const float* p, float (&v)[8]) {
        float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    __device__ static inline void store(float* p, const float (&v)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct Vec8<bf16_t> {
    __device__ static inline void load(const bf16_t* p, float (&v)[8]) {
        uint4 u = *reinterpret_cast<const uint4*>(p);
        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
        v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
    }
    __device__ static inline void store(bf16_t* p, const float (&v)[8]) {
        uint4 u;
        u.x = pack2bf(v[0], v[1]); u.y = pack2bf(v[2], v[3]);
        u.z = pack2bf(v[4], v[5]); u.w = pack2bf(v[6], v[7]);
        *reinterpret_cast<uint4*>(p) = u;
    }
};

// split_t tensors: every 16-channel group of a pixel is 64 bytes, [16 x bf16 hi][16 x bf16 lo] (the element index
// (pixel * ldc + c) keeps its fp32 meaning, so ldc, offsets and sizes are those of an fp32 tensor).  Needs a 64-byte
// aligned base and ldc % 16 == 0 (channel counts are padded to 32).  The 8 channels c..c+7 (c % 8 == 0) of element
// address a live at hi = (a & ~63) + ((a >> 1) & 16), lo = hi + 32.
__device__ inline void split8(const float (&v)[8], uint4& hi, uint4& lo) {
    hi = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    lo.x = pack2bf(v[0] - __uint_as_float(hi.x << 16), v[1] - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = pack2bf(v[2] - __uint_as_float(hi.y << 16), v[3] - __uint_as_float(hi.y & 0xffff0000u));
    lo.z = pack2bf(v[4] - __uint_as_float(hi.z << 16), v[5] - __uint_as_float(hi.z & 0xffff0000u));
    lo.w = pack2bf(v[6] - __uint_as_float(hi.w << 16), v[7] - __uint_as_float(hi.w & 0xffff0000u));
}
template <> struct Vec8<split_t> {
    __device__ static inline void load(const split_t* p, float (&v)[8]) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        const char* b = reinterpret_cast<const char*>((a & ~(uintptr_t)63) + ((a >> 1) & 16));
        const uint4 h = *reinterpret_cast<const uint4*>(b), l = *reinterpret_cast<const uint4*>(b + 32);
        v[0] = __uint_as_float(h.x << 16) + __uint_as_float(l.x << 16); v[1] = __uint_as_float(h.x & 0xffff0000u) + __uint_as_float(l.x & 0xffff0000u);
        v[2] = __uint_as_float(h.y << 16) + __uint_as_float(l.y << 16); v[3] = __uint_as_float(h.y & 0xffff0000u) + __uint_as_float(l.y & 0xffff0000u);
        v[4] = __uint_as_float(h.z << 16) + __uint_as_float(l.z << 16); v[5] = __uint_as_float(h.z & 0xffff0000u) + __uint_as_float(l.z & 0xffff0000u);
        v[6] = __uint_as_float(h.w << 16) + __uint_as_float(l.w << 16); v[7] = __uint_as_float(h.w & 0xffff0000u) + __uint_as_float(l.w & 0xffff0000u);
    }
    __device__ static inline void store(split_t* p, const float (&v)[8]) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        char* b = reinterpret_cast<char*>((a & ~(uintptr_t)63) + ((a >> 1) & 16));
        uint4 h, l;
        split8(v, h, l);
        *reinterpret_cast<uint4*>(b) = h;
        *reinterpret_cast<uint4*>(b + 32) = l;
    }
};

__device__ inline void mma_bf16(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

template <typename T> __device__ inline float ld1(const T* p);
template <> __device__ inline float ld1<float>(const float* p) { return *p; }
template <> __device__ inline float ld1<bf16_t>(const bf16_t* p) { return bf2f(p->v); }
template <typename T> __device__ inline void st1(T* p, float v);
template <> __device__ inline void st1<float>(float* p, float v) { *p = v; }
template <> __device__ inline void st1<bf16_t>(bf16_t* p, float v) { p->v = f2bf(v); }
// single element of a split_t tensor: channel k = (a / 4) % 16 of its 64-byte group -> hi at 2k, lo at 32 + 2k
template <> __device__ inline float ld1<split_t>(const split_t* p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint16_t* b = reinterpret_cast<const uint16_t*>((a & ~(uintptr_t)63) + ((a & 63) >> 1));
    return bf2f(b[0]) + bf2f(b[16]);
}
template <> __device__ inline void st1<split_t>(split_t* p, float v) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    uint16_t* b = reinterpret_cast<uint16_t*>((a & ~(uintptr_t)63) + ((a & 63) >> 1));
    const uint16_t h = f2bf(v);
    b[0] = h;
    b[16] = f2bf(v - bf2f(h));
}

// ---- MFMA step on one 16-byte A group and one 16-byte B group ------------------------------------
// bf16: 8 k-values per lane -> one 32x32x16 MFMA.
// f32 : 4 k-values per lane -> four 32x32x2 MFMAs; MFMA j contracts k-pair {j (lanes 0-31), 4+j (lanes 32-63)}
//       of the 8 channels the two half-waves loaded, identically for A and B, so any 8-channel block is
//       summed exactly once (bit-exact f32 fma chain per the ISA).
template <typename T> __device__ inline void mma16(const uint4& a, const uint4& b, f32x16& acc);
template <> __device__ inline void mma16<bf16_t>(const uint4& a, const uint4& b, f32x16& acc) { mma_bf16(a, b, acc); }
template <> __device__ inline void mma16<float>(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// Predicated 16-byte global load (returns zeros when !ok).  Written as a by-value helper on purpose: a ternary
// between `*ptr` and a zero VARIABLE is an lvalue conditional and hipcc lowers it to a select between the global
// address and the private (scratch) address of the zero -- flat loads and scratch traffic in the hot loop.
__device__ inline uint4 ldg16(const void* p, bool ok) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (ok) v = *reinterpret_cast<const uint4*>(p);
    return v;
}

// 16-byte buffer load with hardware range checking: one VMEM instruction, per-lane 32-bit byte offset computed ONCE,
// wave-uniform (SGPR) chunk offset, zero returned for lanes whose offset is >= num_records.  Halo / padding lanes use
// BUF_OOB.  Measured with in-kernel stamps (tools/ws_diag.py): the predicated 64-bit global loads used before cost
// ~100 issue cycles each (exec-mask branches + 64-bit address VALU) and made the staging code, not the MFMAs or
// the memory system, the critical path of the convolution kernels.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr unsigned BUF_OOB = 0x80000000u;
__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ inline uint4 buf_ld16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// The data registers of a 16-byte store must not be rewritten by the very next vector instructions: the store reads them dword by dword
// after it has issued.  hipcc inserts the wait state only for stores WITHOUT an SGPR offset (GCNHazardRecognizer: "no hazard if soffset is
// a register"); on gfx950 a packed fma that rewrote dwords 2-3 one instruction behind such a store corrupted them in lanes 12-15 / 28-31 of
// each half-wave (the 1-D Winograd experiment of round 4, in the git history: wrong .w components, found by its kernel parity test).  The asm keeps the four registers alive -- tied
// in and out -- until two wait states behind the store.
__device__ inline void buf_st16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, const uint4& v) {
    u32x4 d; d.x = v.x; d.y = v.y; d.z = v.z; d.w = v.w;
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, voff, soff, 0);
    asm volatile("s_nop 1" : "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));
}
// 8 consecutive channels (fp32 registers) -> one (bf16) or two (fp32 storage) range-checked 16-byte buffer stores;
// lanes whose offset is BUF_OOB store nothing
template <typename T> __device__ inline void buf_st8(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        buf_st16(rs, voff, soff, make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])));
    } else if constexpr (__is_same(T, split_t)) {
        // hi/lo planes (Vec8<split_t>); soff is a whole number of pixels, so the group remap applies to voff alone,
        // and BUF_OOB stays out of range
        const unsigned vs = (voff & ~63u) + ((voff >> 1) & 16u);
        uint4 h, l;
        split8(v, h, l);
        buf_st16(rs, vs, soff, h);
        buf_st16(rs, vs, soff + 32, l);
    } else {
        buf_st16(rs, voff, soff, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
        buf_st16(rs, voff, soff + 16, make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])));
    }
}

// max without the canonicalising v_max hipcc puts in front of fmaxf on an MFMA result (a NaN operand yields the other one, as fmaxf)
__device__ inline float vmax_f32(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Row of accumulator register `reg` (0..15) for lane half h in a 32x32 MFMA tile; column = lane & 31.
__device__ inline int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ---- XCD-aware block remap (8 XCDs, blocks dealt round-robin): give each XCD a contiguous id range so
// neighbouring tiles share that XCD's L2.  Bijective for any grid size.  Speed only, never correctness.
__device__ inline int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// Border class of an output pixel of a zero-padded 3x3 convolution: 3 * (0 top row, 1 interior, 2 bottom row) + (0 left column, 1 interior,
// 2 right column).  Which of the nine taps read padding depends on nothing else (H, W >= 2).
__device__ inline int border_class(int yy, int xx, int H, int W) {
    return (yy == 0 ? 0 : (yy == H - 1 ? 6 : 3)) + (xx == 0 ? 0 : (xx == W - 1 ? 2 : 1));
}

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Border-class bias table of a BatchNorm folded into the convolution behind it (bnfold.hip): table[class][co] = bias[co] + the sum of
// T[co][tap] = sum_ci w[co][ci][tap] * shift[ci] over the taps of that class which read inside the image.  Appended to the filter-pack launch
// of the same convolution (blocks beyond the pack's own: one per physical output channel, 256 threads).
// taps = 9: w [Cout][Cin][3][3], nine classes; taps = 1: a pointwise convolution w [Cout][Cin] (the 1x1 head): no padding, one row.
struct FoldBias { const float* w; const float* shift; const float* bias; float* table; int Cout, Cin, Cout_p, taps; };
__device__ inline void fold_bias_block(const FoldBias& f, int co, float* T /* __shared__ [9] */) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int tap = wv; tap < f.taps; tap += 4) {    // a wave per tap: lanes stride 64 over the input channels, then the xor butterfly
        float a = 0.f;
        if (co < f.Cout)
            for (int ci = lane; ci < f.Cin; ci += 64) a = fmaf(f.w[((size_t)co * f.Cin + ci) * f.taps + tap], f.shift[ci], a);
        a = wave_sum(a);
        if (lane == 0) T[tap] = a;
    }
    __syncthreads();
    if (f.taps == 1) {
        if (threadIdx.x == 0) f.table[co] = co < f.Cout ? (f.bias ? f.bias[co] : 0.f) + T[0] : 0.f;
        return;
    }
    if (threadIdx.x < 9) {
        const int rc = threadIdx.x / 3, cc = threadIdx.x % 3;       // row class: 0 top (tap row 0 reads padding), 2 bottom (tap row 2 does)
        float s = (co < f.Cout && f.bias) ? f.bias[co] : 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const bool out = (rc == 0 && ky == 0) || (rc == 2 && ky == 2) || (cc == 0 && kx == 0) || (cc == 2 && kx == 2);
                if (!out) s += T[ky * 3 + kx];
            }
        f.table[(size_t)threadIdx.x * f.Cout_p + co] = co < f.Cout ? s : 0.f;
    }
}

// Issue-order hint for a software-pipelined MFMA loop: NM MFMAs of the current step interleaved with the NR LDS reads
// that prefetch the next one, reads spread evenly BEHIND the MFMAs (first an MFMA, then its share of reads, ...), so the
// reads issue while an MFMA occupies the matrix pipe instead of in a burst between two MFMA bursts.
template <int NM, int NR, int I = 0>
__device__ inline void sched_mfma_reads() {
    if constexpr (I < NM) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int n = ((I + 1) * NR) / NM - (I * NR) / NM;
        if constexpr (n > 0) __builtin_amdgcn_sched_group_barrier(0x100, n, 0);
        sched_mfma_reads<NM, NR, I + 1>();
    }
}

// Issue-order hint for a fully software-pipelined MFMA loop of a single wave per SIMD: slot i = one MFMA followed by
// NV VALU ops and, depending on the slot range, one LDS read [0, R1), one LDS write [W0, W1) or one buffer load [L0, L1).
// (sched_group_barrier masks: 0x008 MFMA, 0x002 VALU, 0x100 DS read, 0x200 DS write, 0x020 VMEM read.)
// VALU groups start at slot V0: the VALU ops that consume an LDS read must not share its slot, or the wave waits out the
// LDS latency with a single MFMA queued (measured: 5.6k instead of 4.1k cycles per 64-MFMA chunk).
// RPS LDS reads per slot in slots [0, R1): with one read per slot and its consumers in the same slot the compiler has to
// wait for the read it has just issued (lgkmcnt(0) behind every read: the whole LDS latency under ONE 64-cycle MFMA);
// RPS = 2 and V0 > 0 put the reads a few slots ahead of the VALU ops that use them.
template <int NM, int R1, int W0, int W1, int L0, int L1, int NV, int V0 = 0, int RPS = 1, int I = 0>
__device__ inline void sched_mfma_slots() {
    if constexpr (I < NM) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (I < R1) __builtin_amdgcn_sched_group_barrier(0x100, RPS, 0);
        if constexpr (I >= W0 && I < W1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if constexpr (I >= L0 && I < L1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if constexpr (NV > 0 && I >= V0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
        sched_mfma_slots<NM, R1, W0, W1, L0, L1, NV, V0, RPS, I + 1>();
    }
}

}  // namespace clamd
