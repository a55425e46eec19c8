// Shared device/host helpers for the SSN gfx950 kernels.
// Everything here is wave64 / CDNA4-only by construction (no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SSN_WAVE 64

// ---- error plumbing (C ABI returns 0 / negative code; message via ssn_last_error) ----
enum {
    SSN_OK = 0,
    SSN_ERR_ARG = -1,      // bad argument / unsupported shape
    SSN_ERR_LAUNCH = -2,   // hipLaunch failed
    SSN_ERR_WORKSPACE = -3 // workspace too small
};
void ssn_set_error(const char* fmt, ...);
#define SSN_CHECK_ARG(cond, ...)             \
    do {                                     \
        if (!(cond)) {                       \
            ssn_set_error(__VA_ARGS__);      \
            return SSN_ERR_ARG;              \
        }                                    \
    } while (0)
#define SSN_CHECK_LAUNCH(name)                                                      \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            ssn_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));    \
            return SSN_ERR_LAUNCH;                                                  \
        }                                                                           \
    } while (0)

// ---- division by a runtime-constant divisor (host computes the magic once per launch) ----
// Valid for 0 <= n < 2^31 and 1 <= d < 2^31 (round-up method, 64-bit intermediate).
struct FastDiv {
    uint32_t mul;
    uint32_t shift;
    uint32_t d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    if (d == 1) {
        f.mul = 0;
        f.shift = 0;
        return f;
    }
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;  // ceil(log2 d)
    uint64_t m = ((1ull << (32 + l)) + d - 1) / d;  // ceil(2^(32+l)/d), fits in 33 bits
    f.mul = (uint32_t)(m - (1ull << 32));            // low 32 bits; implicit +2^32
    f.shift = l;
    return f;
}
__device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv& f) {
    if (f.d == 1) return n;
    // q = (n + mulhi(n, mul)) >> shift, done in 64 bits to keep the carry
    uint64_t t = (uint64_t)__umulhi(n, f.mul) + (uint64_t)n;
    return (uint32_t)(t >> f.shift);
}
__device__ __forceinline__ void fd_divmod(uint32_t n, const FastDiv& f, uint32_t& q, uint32_t& r) {
    q = fd_div(n, f);
    r = n - q * f.d;
}

// ---- split-K planning (host): how many workgroups share one output tile of a weight-gradient GEMM ----
// All workgroups of a launch do the same amount of work, so the launch takes `rounds` = ceil(blocks / slots)
// block-times, slots = 256 CUs x co-resident workgroups; a block-time is the chunks of one split plus a fixed
// prologue/epilogue cost (~`fixed_chunks` chunks).  Pick the split count that minimises rounds x block-time plus
// `reduce_chunks_per_split` (the time, in chunk units, the reduce pass spends on each additional partial slab).
static inline void plan_split_k(long tiles, long chunks, int occupancy, int min_chunks, int fixed_chunks,
                                double reduce_chunks_per_split, int* splits, int* chunks_per_split) {
    const long slots = 256L * (occupancy > 0 ? occupancy : 1);
    long max_splits = chunks / (min_chunks > 0 ? min_chunks : 1);
    if (max_splits < 1) max_splits = 1;
    double best = 1e300;
    long best_s = 1;
    for (long m = 1; m <= 16; ++m) {
        long s = (m * slots) / tiles;
        if (s < 1) s = 1;
        if (s > max_splits) s = max_splits;
        const long cps = (chunks + s - 1) / s;
        const long s_eff = (chunks + cps - 1) / cps;
        const long rounds = (tiles * s_eff + slots - 1) / slots;
        const double cost = (double)rounds * (double)(cps + fixed_chunks) + reduce_chunks_per_split * (double)s_eff;
        if (cost < best) {
            best = cost;
            best_s = s_eff;
        }
        if (s == max_splits) break;
    }
    const long cps = (chunks + best_s - 1) / best_s;
    *chunks_per_split = (int)cps;
    *splits = (int)((chunks + cps - 1) / cps);
}

// ---- wave64 reductions ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Promote a value that is known to be wave-uniform to an SGPR.  (Wrapped in a typed function: in hipcc's host
// pass the raw builtin has no type, which silently poisons every expression it touches.)
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// LDS-DMA helpers (buffer_load ... lds): the destination is a wave-uniform LDS address + lane * size.
#ifndef SSN_LDS_PTR
#define SSN_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#endif
// A pointer to device memory that (a) holds the same bytes for the whole kernel and (b) is read at wave-uniform addresses, as a
// CONSTANT-address-space pointer: the compiler then fetches through the scalar unit (s_load into SGPRs, like kernel arguments) instead
// of copying the structure through VGPRs into scratch.  For device-resident problem tables that a previous launch wrote.
#ifndef SSN_CONST_PTR
#if defined(__HIP_DEVICE_COMPILE__)
#define SSN_CONST_PTR(T, p) ((const __attribute__((address_space(4))) T*)(p))
#else
#define SSN_CONST_PTR(T, p) ((const T*)(p))      /* (hipcc's host pass only parses the kernels) */
#endif
#endif
#ifndef SSN_WAIT_VMCNT
#define SSN_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#endif
#ifndef SSN_WAIT_LGKM0
// as a builtin (not inline asm), so that the compiler's own wait-count bookkeeping sees it: vmcnt 63, expcnt 7, lgkmcnt 0
#define SSN_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
#endif

// ---- fp32 operands on the f16 matrix cores: 2-way split, 3 products (the "split" convolution kernels) ----
// x * s = hi + lo + e with hi = f16(x s), lo = f16(x s - hi), both ROUND-TO-NEAREST-EVEN (v_cvt_pk_f16_f32 on gfx950),
// |e| <= 2^-22 |x s|: an f16 carries 11 significant bits, so two terms hold 22 of the 24 bits of an fp32 value, and a
// product a * b is accumulated in fp32 from THREE partial products  a_lo b_hi + a_hi b_lo + a_hi b_hi  (the dropped
// a_lo b_lo is <= 2^-22 |a b|): 3 v_mfma_f32_32x32x16_f16 per k16 step instead of 16 exact-f32 MFMAs, measured error
// against float64 within 2x of an fp32 FMA chain (tests/test_kernels.py::test_conv_split_error_growth_with_k).
//
// f16 has 5 exponent bits, so every operand TENSOR is scaled by a power of two s (exact) that puts its largest
// magnitude just below 2^15: nothing overflows (65504), and lo -- 2^-11 of its element -- stays a normal f16 for every
// element down to 2^-17 of the tensor's maximum (smaller ones keep an absolute accuracy of 2^-40 of the maximum).  The
// maximum comes from the kernel that PRODUCED the tensor (amax_emit below: every kernel that writes a tracked tensor
// max-es what it stores into the tensor's slot; the slot is therefore an upper bound of the final contents), the
// consumer derives s from it, and the epilogue multiplies the accumulators by 1 / (s_a s_b), again exact.
typedef _Float16 ssn_f16x2 __attribute__((ext_vector_type(2)));
typedef float ssn_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// power-of-two scale for a tensor whose largest magnitude is `amax` (>= 0): amax * s lies in [2^14, 2^15).
// amax == 0 / denormal / Inf / NaN -> 1 (an all-zero tensor needs no scale; non-finite data propagates as in fp32).
__host__ __device__ inline float f16_scale_of(float amax) {
    const uint32_t b = __builtin_bit_cast(uint32_t, amax);
    const int e = (int)((b >> 23) & 0xFFu);            // amax in [2^(e-127), 2^(e-126))
    if (e == 0 || e == 255) return 1.f;
    int se = 268 - e;                                  // biased exponent of 2^(141 - e)
    se = se < 67 ? 67 : (se > 187 ? 187 : se);         // s in [2^-60, 2^60]: s_a * s_b and its reciprocal stay finite
    return __builtin_bit_cast(float, (uint32_t)se << 23);
}
// f16 pair of two (already scaled) fp32 values, `even` in the low half (k even -> low half of an MFMA operand dword)
__device__ __forceinline__ uint32_t f16_pair_rne(float even, float odd) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(ssn_f32x2{even, odd}, ssn_f16x2));
}
__device__ __forceinline__ float f16_pair_lo(uint32_t pk) { return (float)__builtin_bit_cast(ssn_f16x2, pk)[0]; }
__device__ __forceinline__ float f16_pair_hi(uint32_t pk) { return (float)__builtin_bit_cast(ssn_f16x2, pk)[1]; }
// the two plane dwords of one k-pair of UNSCALED values
__device__ __forceinline__ void f16_split2_pair(float v0, float v1, float s, uint32_t& hi, uint32_t& lo) {
    const float x0 = v0 * s, x1 = v1 * s;
    hi = f16_pair_rne(x0, x1);
    lo = f16_pair_rne(x0 - f16_pair_lo(hi), x1 - f16_pair_hi(hi));
}

// ---- per-tensor maximum magnitude ("amax slot": one float, zeroed by the host before the tensor's first writer) ----
// Call with each lane's running max of |values it stored| (wave-uniform control flow).  Non-negative floats order like
// their bit patterns, so the slot is maintained with an unsigned atomic max; a wave only issues the atomic when it
// would raise the slot (a stale read costs an unneeded atomic, never a wrong result).
__device__ __forceinline__ void amax_emit(float* slot, float lane_max) {
    if (!slot) return;
    const float m = wave_max(lane_max);
    if ((threadIdx.x & 63) == 0) {
        const unsigned int b = __builtin_bit_cast(unsigned int, m);
        unsigned int* u = reinterpret_cast<unsigned int*>(slot);
        if (b > *reinterpret_cast<volatile unsigned int*>(u)) atomicMax(u, b);
    }
}

// The same with ONE slot read (+ atomic) per WORKGROUP: every thread of the block calls it, once, at the end of the kernel.  Device-scope
// atomics on one address -- and the read in front of them, a memory round trip at the very end of a wave's life -- are what recording a
// maximum costs (profiles/r6_grid_cap.txt: per-wave -> per-workgroup in the convolution epilogue was -0.14 ms per step).
__device__ __forceinline__ void amax_emit_block(float* slot, float lane_max) {
    if (!slot) return;      // (block-uniform)
    __shared__ float wave_maxima[16];
    const float wv = wave_max(lane_max);
    if ((threadIdx.x & 63) == 0) wave_maxima[threadIdx.x >> 6] = wv;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int nw = (int)(blockDim.x + 63) >> 6;
        float m = wave_maxima[0];
        for (int w = 1; w < nw; ++w) m = fmaxf(m, wave_maxima[w]);
        amax_emit(slot, m);
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// XCD-aware remap of a 1-D block id: consecutive logical ids land on the same XCD
// (hardware places block b on XCD b % 8 -- speed only, never correctness).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblk) {
    const uint32_t q = nblk >> 3, r = nblk & 7;
    const uint32_t xcd = bid & 7, slot = bid >> 3;
    const uint32_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}
