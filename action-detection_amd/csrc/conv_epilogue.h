// Shared epilogue of the implicit-GEMM convolution kernels (conv_igemm.hip, conv_x6.hip).
//
// A wave holds TM x TN accumulator tiles of the 32x32 MFMA layout (lane (li, lh), register r:
// row = (r & 3) + 8 (r >> 2) + 4 lh, column = li).  The store applies, per output channel (row m):
//   forward : y = relu?(acc * inv * scale[m] + shift[m])
//   dgrad   : y = maskfn((accumulate ? y_old : 0) + acc * inv),  maskfn = fused backward of the producer's ReLU + frozen BN
// (inv = 1 / (operand scales) of the split kernels, 1 for the exact-f32 kernel), and -- when the output tensor has an
// amax slot -- max-es the magnitudes it stores into it (ssn_common.h: amax_emit).
// Two things keep this phase short (it used to cost as much as a third of the main loop):
//   * the per-channel vectors are staged in LDS once per workgroup -- a global load placed after a store to a
//     pointer the compiler cannot disambiguate is never hoisted, i.e. one dependent memory round trip PER ELEMENT;
//   * all global accesses are raw buffer operations with branch-free addressing: rows / pixels outside the tensor
//     carry an out-of-range offset (loads return 0, stores are dropped), so there is no divergent control flow and
//     the compiler counts outstanding operations exactly (gfx9 counts stores in vmcnt: after a divergent join it
//     drains the queue with vmcnt(0), which serialises the stores on the memory round trip).
#pragma once
#include "ssn_common.h"

namespace {

constexpr uint32_t EPI_OOB = 0x80000000u;

struct EpiArgs {
    float* y;
    const float* mask_y;   // nullptr: no mask
    uint32_t y_bytes, mask_bytes;
    uint32_t howo4;        // bytes between consecutive channels of y (and of mask_y)
    int M;
    int relu, accumulate;
    float* amax;           // nullptr: the output tensor is not tracked
    float* amax2;          // second slot raised by the same value (a launch whose rows land in two regions of a tensor), or nullptr
    // Output rows m >= row_split are stored row_gap channels further up the tensor (both multiples of 32, so a 32-row MFMA
    // tile never straddles the split): the fused launch on an Inception block input writes the block's 1x1 branch to the head
    // of the block-output tensor and the reduce / projection rows behind the block's own channels.  No split: INT_MAX, 0.
    int row_split, row_gap;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t epi_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// ch[0,BM) = inv * scale (inv if none), ch[BM,2BM) = shift (0), ch[2BM,3BM) = mask scale (NaN: pass through),
// ch[3BM,4BM) = the ReLU floor (0, or -inf for rows that take no ReLU).  Rows m >= raw_from are "raw": no affine and no
// ReLU whatever scale / relu say (the bias-free pool projection that rides in the same launch as the reduce pair).  All
// threads of the workgroup call this between two barriers (LDS is free once the main loop has ended).
template <int BM, int NT>
__device__ __forceinline__ void epi_stage_channels(float* ch, const float* scale, const float* shift,
                                                   const float* mask_scale, int m0, int M, int tid, float inv = 1.f,
                                                   int relu = 0, int raw_from = 0x7fffffff, int row_split = 0x7fffffff,
                                                   int row_gap = 0) {
    for (int r = tid; r < BM; r += NT) {
        const int m = m0 + r;
        const bool ok = m < M;
        const bool aff = ok && scale && m < raw_from;
        // `scale` is indexed by the output CHANNEL (it is the per-tensor vector the backward masks read too), i.e. displaced
        // like the stores of rows >= row_split; `shift` by the launch's own row
        ch[r] = aff ? scale[m + (m >= row_split ? row_gap : 0)] * inv : inv;
        ch[BM + r] = aff ? shift[m] : 0.f;
        ch[2 * BM + r] = (ok && mask_scale) ? mask_scale[m] : __builtin_nanf("");
        ch[3 * BM + r] = (relu && m < raw_from) ? 0.f : -__builtin_inff();
    }
}

// yoff[j] / moff[j]: byte offset of (image, pixel) of accumulator column j at channel m0 + 4 lh, or EPI_OOB when the
// pixel does not exist.  row0 = first tile row of this wave inside the workgroup tile ((wm * TM) * 32).
template <int TM, int TN, int BM>
__device__ __forceinline__ void conv_epilogue(const f32x16 (&acc)[TM][TN], const float* ch, const EpiArgs& e,
                                              const uint32_t (&yoff)[TN], const uint32_t (&moff)[TN], int row0, int lh,
                                              int m0) {
    const __amdgpu_buffer_rsrc_t yrsrc = epi_rsrc(e.y, e.y_bytes);
    const int mlim = e.M - m0 - 4 * lh;   // rows srow (without the lane-half term) below this are inside the tensor
    auto srow = [&](int i, int r) { return row0 + i * 32 + (r & 3) + 8 * (r >> 2); };
    // wave-uniform channel displacement of MFMA tile i (see EpiArgs::row_split)
    auto gap = [&](int i) { return (m0 + row0 + i * 32 >= e.row_split) ? e.row_gap : 0; };
    // amax of what is stored: rows past the tensor compute exact zeros (zero weight rows, shift staged as 0), so only
    // the pixel columns that do not exist have to be kept out
    float vmax = 0.f;
    if (!e.accumulate && !e.mask_y) {
        // pure stores: nothing in this path waits on memory
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float cmax = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int sr = srow(i, r);
                    float v = acc[i][j][r] * ch[sr + 4 * lh] + ch[BM + sr + 4 * lh];
                    if (e.relu) v = fmaxf(v, ch[3 * BM + sr + 4 * lh]);
                    cmax = fmaxf(cmax, fabsf(v));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yrsrc,
                                                          sr < mlim ? yoff[j] : EPI_OOB, (uint32_t)(sr + gap(i)) * e.howo4, 0);
                }
            vmax = fmaxf(vmax, yoff[j] != EPI_OOB ? cmax : 0.f);
        }
    } else {
        // read-modify-write: the operands of half-tile g+1 are requested before half-tile g is stored
        const __amdgpu_buffer_rsrc_t orsrc = epi_rsrc(e.y, e.accumulate ? e.y_bytes : 0u);
        const __amdgpu_buffer_rsrc_t mrsrc = epi_rsrc(e.mask_y ? e.mask_y : e.y, e.mask_y ? e.mask_bytes : 0u);
        constexpr int G = TM * TN * 2;   // half accumulator tiles (8 values per lane)
        float old[2][8], mk[2][8];
        auto fetch = [&](int g, int b) {
            const int j = (g >> 1) / TM, i = (g >> 1) % TM;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int sr = srow(i, (g & 1) * 8 + q);
                old[b][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                          orsrc, sr < mlim ? yoff[j] : EPI_OOB, (uint32_t)sr * e.howo4, 0));
                mk[b][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                         mrsrc, sr < mlim ? moff[j] : EPI_OOB, (uint32_t)sr * e.howo4, 0));
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g + 1 < G) fetch(g + 1, (g + 1) & 1);
            const int j = (g >> 1) / TM, i = (g >> 1) % TM, b = g & 1;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = (g & 1) * 8 + q;
                const int sr = srow(i, r);
                float v = acc[i][j][r] * ch[sr + 4 * lh] + ch[BM + sr + 4 * lh];
                if (e.relu) v = fmaxf(v, ch[3 * BM + sr + 4 * lh]);
                v += old[b][q];
                const float sc = ch[2 * BM + sr + 4 * lh];
                v = (sc != sc) ? v : (mk[b][q] > 0.f ? v * sc : 0.f);   // NaN marks a channel that is not a ReLU output
                vmax = fmaxf(vmax, (sr < mlim && yoff[j] != EPI_OOB) ? fabsf(v) : 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yrsrc,
                                                      sr < mlim ? yoff[j] : EPI_OOB, (uint32_t)sr * e.howo4, 0);
            }
        }
    }
    amax_emit(e.amax, vmax);
    amax_emit(e.amax2, vmax);
}

}  // namespace
