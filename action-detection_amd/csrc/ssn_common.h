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
#ifndef SSN_WAIT_VMCNT
#define SSN_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#endif
#ifndef SSN_WAIT_LGKM0
// as a builtin (not inline asm), so that the compiler's own wait-count bookkeeping sees it: vmcnt 63, expcnt 7, lgkmcnt 0
#define SSN_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
#endif

// ---- exact 3-way bf16 split of fp32 values (the "x6" kernels) ----
// x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2), every conversion ROUND-TO-NEAREST-EVEN
// (v_cvt_pk_bf16_f32 on gfx950: two values per instruction, the same instruction count as a truncating split).  The sum
// is exact -- the remainder after a rounding to 8 significant bits has at most 16, then at most 8 of them -- and,
// unlike truncation, the terms have no preferred sign: the partial products an x6 kernel drops (x2*w3, x3*w2, x3*w3)
// then neither add up systematically nor exceed 2^-26 |x w| each (tests/test_kernels.py::test_conv_x6_error_growth_with_k).
typedef __bf16 ssn_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ssn_f32x2 __attribute__((ext_vector_type(2)));
// bf16 pair of two fp32 values, `even` in the low half (k even -> low half of an MFMA operand dword)
__device__ __forceinline__ uint32_t bf16_pair_rne(float even, float odd) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(ssn_f32x2{even, odd}, ssn_bf16x2));
}
__device__ __forceinline__ float bf16_pair_lo(uint32_t pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float bf16_pair_hi(uint32_t pk) { return __builtin_bit_cast(float, pk & 0xFFFF0000u); }
// the three plane dwords of one k-pair
__device__ __forceinline__ void bf16_split3_pair(float v0, float v1, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = bf16_pair_rne(v0, v1);
    const float r0 = v0 - bf16_pair_lo(p0), r1 = v1 - bf16_pair_hi(p0);
    p1 = bf16_pair_rne(r0, r1);
    p2 = bf16_pair_rne(r0 - bf16_pair_lo(p1), r1 - bf16_pair_hi(p1));
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
