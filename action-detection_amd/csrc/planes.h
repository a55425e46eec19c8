// "Planes" tensors: the MFMA-native activation / gradient format of the split-precision path (gfx950).
//
// An fp32 activation x is kept as TWO f16 terms of x * s (s = the tensor's power-of-two scale):
//        x s = hi + lo + e,   hi = f16(x s),  lo = f16(x s - hi)   (both round-to-nearest-even),  |e| <= 2^-22 |x s|
// -- exactly the operand pair the split convolution kernels multiply (ssn_common.h) -- but produced ONCE, by the
// kernel that computes x, instead of by every consumer of x (each m-tile of each tap of each consuming convolution
// re-split it: 7-13 VALU per MFMA in the round-2 kernels).  4 bytes per element, like the fp32 value it replaces.
//
// Layout ("NC8HW8", channel-blocked so that 8 consecutive channels of a pixel are one 16-byte MFMA k-half):
//        plane p (0 = hi, 1 = lo) | image n | channel group g = c / 8 | pixel q = h W + w | channel c % 8     (f16)
// i.e. byte offset inside a plane = ((n G + g) HW + q) * 16 + (c % 8) * 2, G = C / 8 (C is a multiple of 8; layers
// with fewer channels are zero-padded).  The two planes are separate allocations (one buffer descriptor each, so a
// tensor may hold 2 GiB PER PLANE), a channel slice is a group offset, an Inception block output is written slice by
// slice as before.  A 64-lane LDS-DMA instruction moves 64 pixels x 8 channels of one plane (1 KiB, fully
// coalesced), a forward / dgrad B fragment is ONE ds_read_b128 per plane, an accumulator tile stores 8 bytes (4
// channels) per lane and plane, and the weight gradient -- whose reduction index is the pixel -- reads its operands
// with the LDS transpose read (ds_read_b64_tr_b16).
//
// Scales are "delayed": a tensor's scale for step t is derived (scales_update_kernel) from the largest magnitude its
// producers recorded in step t-1 (amax slot, as before), with PL_HEADROOM_BITS bits of head-room; producers clamp to the
// f16 range, record the true magnitude, and the update kernel raises a sticky flag when a tensor outgrew its
// head-room, so the host can re-calibrate (run the step again with fresh scales).  The first step is calibrated by
// running it until the scales stop moving (planes_exec.py).
#pragma once
#include "ssn_common.h"

namespace pl {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int PL_HEADROOM_BITS = 2;       // a tensor's largest magnitude lands in [2^12, 2^13) when it repeats last step's
constexpr float PL_F16_MAX = 65504.f;
// range guard (range_check_kernel): a tensor whose largest scaled magnitude fell below this -- 2^8 under the [2^12, 2^13) target, i.e.
// it shrank more than 256x since the pass its scale was derived from -- is reported like an overflow (elements below 2^-12 of the
// tensor's maximum have lost low-plane bits); anything above is absorbed by the format (22 bits down to 2^-16 of the target)
constexpr float PL_UNDERFLOW_FLOOR = 16.f;
constexpr uint32_t PL_OOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pl_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// scale for next step from this step's amax: amax * s in [2^(14-H), 2^(15-H)); 0 / non-finite -> keep `old`
__host__ __device__ inline float pl_scale_from_amax(float amax, float old) {
    const uint32_t b = __builtin_bit_cast(uint32_t, amax);
    const int e = (int)((b >> 23) & 0xFFu);
    if (e == 0 || e == 255) return old;
    int se = 268 - PL_HEADROOM_BITS - e;
    se = se < 67 ? 67 : (se > 187 ? 187 : se);
    return __builtin_bit_cast(float, (uint32_t)se << 23);
}

__device__ __forceinline__ float pl_clamp(float v) { return __builtin_fminf(__builtin_fmaxf(v, -PL_F16_MAX), PL_F16_MAX); }
// v_med3_f32: relu / no-relu floor and the f16 ceiling in one instruction (floor = 0 or -65504)
__device__ __forceinline__ float pl_clamp_floor(float v, float floor) { return __builtin_amdgcn_fmed3f(v, floor, PL_F16_MAX); }

// low-plane word of a channel pair: f16(v0 - hi.lo), f16(v1 - hi.hi).  [r6] v_fma_mixlo / mixhi_f16 take the f16 half of `hi` as an
// operand and round v * 1 - hi straight to f16: two instructions per pair instead of four (2 x cvt_f32_f16, pk_add, cvt_pk) -- a
// seventh of the convolution epilogue's VALU work.  Bit-identical: v - hi is exact in fp32 (hi is v rounded to 11 bits), so both
// forms round the same number once.  (The host emulator of the CPU test tier defines its own.)
#ifndef PL_LO_PAIR
#define PL_LO_PAIR(dst, v0, v1, hi)                                                                       \
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"                            \
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"                                 \
        : "=&v"(dst)                                                                                      \
        : "v"(v0), "v"(v1), "v"(hi))
#endif
__device__ __forceinline__ uint32_t pl_lo_pair(float v0, float v1, uint32_t hi) {
    uint32_t lo;
    PL_LO_PAIR(lo, v0, v1, hi);
    return lo;
}

// split four (already scaled and clamped) values into the two plane words of 4 consecutive channels
__device__ __forceinline__ void pl_split4(const float (&v)[4], u32x2& hi, u32x2& lo) {
    hi[0] = f16_pair_rne(v[0], v[1]);
    hi[1] = f16_pair_rne(v[2], v[3]);
    lo[0] = pl_lo_pair(v[0], v[1], hi[0]);
    lo[1] = pl_lo_pair(v[2], v[3], hi[1]);
}
// the scaled values (hi + lo) of 4 consecutive channels
__device__ __forceinline__ void pl_join4(const u32x2& hi, const u32x2& lo, float (&v)[4]) {
    v[0] = f16_pair_lo(hi[0]) + f16_pair_lo(lo[0]);
    v[1] = f16_pair_hi(hi[0]) + f16_pair_hi(lo[0]);
    v[2] = f16_pair_lo(hi[1]) + f16_pair_lo(lo[1]);
    v[3] = f16_pair_hi(hi[1]) + f16_pair_hi(lo[1]);
}
__device__ __forceinline__ void pl_join8(const u32x4& hi, const u32x4& lo, float (&v)[8]) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        v[2 * d] = f16_pair_lo(hi[d]) + f16_pair_lo(lo[d]);
        v[2 * d + 1] = f16_pair_hi(hi[d]) + f16_pair_hi(lo[d]);
    }
}
__device__ __forceinline__ void pl_split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        hi[d] = f16_pair_rne(v[2 * d], v[2 * d + 1]);
        lo[d] = pl_lo_pair(v[2 * d], v[2 * d + 1], hi[d]);
    }
}

// raw buffer stores of 8 / 16 bytes
__device__ __forceinline__ void pl_store_b64(u32x2 v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}
// 16-byte buffer store.  HAZARD (found on the MI355X in round 6, profiles/r6_b128_store_hazard.txt): a VMEM store with more than 64 bits
// of data reads its data VGPRs over more than one cycle, and a VALU instruction issued right behind it may overwrite the LAST data
// dword before it has been read (the GCN ">64-bit store data" hazard, one wait state).  The compiler's hazard recognizer only pads
// that case when the store's soffset is NOT a register -- these stores use an SGPR soffset, it padded nothing, and the compiled
// epilogue of the TM = 3 / TN = 1 tiles had `buffer_store_dwordx4 v[42:45] ...; v_or_b32 v45, ...` back to back: a few elements in
// 10^8 lost the last dword of their low plane.  The asm below keeps the data registers alive until two wait states behind the store
// (an instruction scheduled in between cannot be given those registers), which closes the window whatever the scheduler does.
#ifndef SSN_STORE_DATA_GUARD      // (the host emulator of the CPU test tier defines it away)
#define SSN_STORE_DATA_GUARD(v) asm volatile("s_nop 1" : : "v"(v) : "memory")
#endif
__device__ __forceinline__ void pl_store_b128(u32x4 v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
    SSN_STORE_DATA_GUARD(v);
}

}  // namespace pl

// gfx950 LDS transpose read: every lane supplies the address of 4 consecutive f16 (8-byte aligned); inside each group of 16
// lanes, lanes 4j..4j+3 supply row j (16 elements) and lane c receives column c of rows 0..3.  (The host emulator of the
// CPU test tier supplies its own definition.)
#ifndef SSN_DS_READ_TR16_B64_AT
#if defined(__HIP_DEVICE_COMPILE__)
// ds_read_b64_tr_b16 of the LDS bytes at ptr + imm (imm: compile-time constant < 64 KiB).  INLINE ASM, not
// __builtin_amdgcn_ds_read_tr16_b64_*: the compiler cannot tell what an LDS access through the builtin aliases, so it puts
// `s_waitcnt vmcnt(0)` -- wait for EVERY LDS-DMA fetch in flight, the ones just issued for later k-steps included -- in front of
// the first such read after each fetch instruction, which serialises fetch and compute (measured: the weight-gradient loops ran
// 25 - 45 % of the matrix pipe).  The asm form is invisible to that bookkeeping, so the CALLER orders it against the DMA
// (SSN_WAIT_VMCNT + barrier) and waits for the result (SSN_WAIT_LGKM0 + sched_barrier) before its first use.
#define SSN_DS_READ_TR16_B64_AT(dst, ptr, imm)                                                                      \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2"                                                              \
                 : "=v"(dst)                                                                                        \
                 : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(ptr)), "n"(imm))
#else
#define SSN_DS_READ_TR16_B64_AT(dst, ptr, imm) ((dst) = pl::u32x2{0u, 0u})
#endif
#endif
