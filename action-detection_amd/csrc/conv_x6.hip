// Implicit-GEMM convolution with fp32-class accuracy on the f16 matrix cores (the "split" path; file and symbol names
// keep the "x6" of the first version, which split into three bf16 terms and multiplied six partial products), gfx950.
//
// Same role and same operands as conv_igemm.hip (forward conv + frozen-BN + ReLU, and dgrad, of the
// BN-Inception layers behind /root/reference/ssn_models.py:266,298), but the multiply runs on
// v_mfma_f32_32x32x16_f16 instead of the 16x slower exact-f32 MFMA:
//
//   every fp32 operand x, scaled by a per-tensor power of two s, is split into two f16 terms (round to nearest),
//        x s = hi + lo + e,   hi = f16(x s),  lo = f16(x s - hi),  |e| <= 2^-22 |x s|,
//   and a*b is accumulated in fp32 from three partial products  a_lo b_hi + a_hi b_hi + a_hi b_lo; the dropped
//   a_lo b_lo is <= 2^-22 |ab|.  The accumulators are multiplied by 1 / (s_a s_b) in the epilogue (exact).  Measured
//   error against float64: within 2x of an fp32 FMA chain (test_conv_split_error_growth_with_k), at 3/16 of the
//   exact-f32 kernel's matrix-pipe time -- and half of the first version's (three bf16 terms, six products).
//   The scales come from per-tensor maxima: weights are measured when they are packed (the value travels behind the
//   packed rows), activations / gradients by the kernels that write them (ssn_common.h: amax_emit; x_amax / y_amax).
//
// Structure (what keeps the loop matrix-bound now that a slab is only ~0.2-0.4k matrix cycles per wave):
//  * K runs in slabs of 16 rows = one f16 MFMA.  1x1: 16 channels.  3x3: 16 channels x ONE tap, taps innermost:
//    all rows of a slab share the tap, so a lane needs one gather offset per slab (base + scalar tap delta, a
//    9-bit validity mask) and the channel part of the address is a scalar soffset.
//  * Operands reach LDS by LDS-DMA (`buffer_load ... lds`), two slabs ahead in a 3-stage ring: no VGPR staging,
//    no VALU, no ds_write; the global round trip (~1.5-2k cycles under load) is hidden behind two slabs of MFMAs.
//      - weights: scaled, split and packed once per step (ssn_conv_x6_pack_weights_multi) into exactly the LDS image
//        (row = 4 chunks of 16 B: 2 planes x 2 k-halves, chunk-swizzled), copied 1 KiB per instruction;
//      - activations: RAW fp32, 16 k-rows x BN pixels (the DMA deposits lane-consecutive dwords = pixel-major
//        rows); a wave scales and splits them into f16 planes while building its B fragments (8 ds_read_b32 + ~24
//        VALU per 32-pixel fragment and slab).
//  * NG = 2 ("ping-pong") workgroups have 8 waves = two groups that share the A tile and own half of the pixel
//    columns each.  Two s_barriers per slab hold the groups half a slab apart: while one group issues its MFMAs the
//    other reads LDS and splits, so each SIMD's matrix pipe always has one wave feeding it (free-running co-resident
//    workgroups drift into lock-step instead: both wait, then both compete -- measured 60 % pipe utilisation).
//  * Epilogue: per-channel vectors via LDS, branch-free buffer addressing (out-of-range rows/pixels fall off the
//    buffer), read-modify-write operands fetched one accumulator tile ahead.
#include "conv_x6_kernel.h"
#include <unordered_map>
#include <vector>

namespace {

using namespace x6;

// ---- weight scale + split + pack: out[slab][m][4 chunks x 4 dwords], then [amax, 0, 0, 0] ----
// slab = g (1x1) or g * 9 + tap (3x3); row k of a slab = channel 16 g + k.  Logical 16-byte chunk c = 2 * plane +
// khalf holds f16 k = 8 * khalf + 0..7 of that plane; it is stored at chunk position c ^ ((m >> 2) & 3), which makes
// the straight LDS copy conflict-free for the ds_read_b128 fragment reads.
//   mode 0 (forward operand): A[m][c][tap] = w[m][c][tap];  mode 1 (dgrad operand): A[m][c][tap] = w[c][m][tap];
//   mode 2 (dgrad operand of a rectangular-tap layer, run as a forward correlation): A[m][c][tap] = w[c][m][KK - 1 - tap]
// Up to four sources (fused launches on an Inception block input: 1x1 branch, reduce pair, pool projection): output
// channel co < split comes from w0, split <= co < split2 from w1, split2 <= co < split3 from w2, the rest from w3.
// Three launches: clear the amax tails, max |w| of every entry into its tail (one workgroup per XP_ACHUNK source
// elements, unsigned atomic max), then the packing proper, which derives the entry's power-of-two scale from that tail.
constexpr int XP_MAX = 40;
constexpr int XP_CHUNK = 4096;   // (slab, m, kpair) triples per block
constexpr int XP_ACHUNK = 8192;  // source elements per block of the amax launch
struct X6PackTable {
    const float* w0[XP_MAX];
    const float* w1[XP_MAX];
    const float* w2[XP_MAX];
    const float* w3[XP_MAX];
    uint32_t* out[XP_MAX];
    long rows_dw[XP_MAX];       // dwords of the packed rows; the tail starts there
    int cout[XP_MAX], cin[XP_MAX], kk[XP_MAX], mode[XP_MAX], split[XP_MAX], split2[XP_MAX], split3[XP_MAX];   // kk = kh * kw
    int srckk[XP_MAX];          // taps per channel of the SOURCE weight (== kk unless a tap subset is packed)
    unsigned tapmap[XP_MAX];    // srckk != kk: nibble t = source tap of packed tap t
    int blk0[XP_MAX + 1];
    int ablk0[XP_MAX + 1];      // block ranges of the amax launch
    int count;
};
// One entry of a packing launch, as the kernels see it (the by-value table above, or a row of a device-resident plan)
struct X6PackEntry {
    const float* w[4];
    uint32_t* out;
    long rows_dw;
    int cout, cin, kk, mode, split, split2, split3, srckk;
    unsigned tapmap;
    int blk0, ablk0;    // first block of the entry in the packing / amax launch
    int pad_;
};
__device__ __forceinline__ X6PackEntry pack_entry_of(const X6PackTable& t, int ti) {
    X6PackEntry e;
    e.w[0] = t.w0[ti];
    e.w[1] = t.w1[ti];
    e.w[2] = t.w2[ti];
    e.w[3] = t.w3[ti];
    e.out = t.out[ti];
    e.rows_dw = t.rows_dw[ti];
    e.cout = t.cout[ti];
    e.cin = t.cin[ti];
    e.kk = t.kk[ti];
    e.mode = t.mode[ti];
    e.split = t.split[ti];
    e.split2 = t.split2[ti];
    e.split3 = t.split3[ti];
    e.srckk = t.srckk[ti];
    e.tapmap = t.tapmap[ti];
    e.blk0 = t.blk0[ti];
    e.ablk0 = t.ablk0[ti];
    e.pad_ = 0;
    return e;
}
// block `lb` (entry-local) of the amax launch: max |w| over XP_ACHUNK source elements into the entry's tail
__device__ __forceinline__ void pack_entry_amax(const X6PackEntry& e, int lb) {
    const long per = (long)e.cin * e.srckk;
    const long n0 = (long)e.split * per;                        // elements of w0
    const long n1 = (long)(e.split2 - e.split) * per;           // ... of w1 (fused pair)
    const long n2 = (long)(e.split3 - e.split2) * per;          // ... of w2
    const long n3 = (long)(e.cout - e.split3) * per;            // ... of w3
    const long base = (long)lb * XP_ACHUNK;
    long end = base + XP_ACHUNK;
    if (end > n0 + n1 + n2 + n3) end = n0 + n1 + n2 + n3;
    float m = 0.f;
    for (long i = base + threadIdx.x; i < end; i += 256)
        m = fmaxf(m, fabsf(i < n0 ? e.w[0][i] : (i < n0 + n1 ? e.w[1][i - n0] : (i < n0 + n1 + n2 ? e.w[2][i - n0 - n1]
                                                                               : e.w[3][i - n0 - n1 - n2]))));
    amax_emit(reinterpret_cast<float*>(e.out + e.rows_dw), m);
}
// block `lb` (entry-local) of the packing launch: XP_CHUNK (slab, m, kpair) triples
__device__ __forceinline__ void pack_entry_rows(const X6PackEntry& e, int lb) {
    const int mode = e.mode, Cout = e.cout, Cin = e.cin, KK = e.kk;
    const int M = mode ? Cin : Cout, C = mode ? Cout : Cin;
    const int ngroups = (C + 15) / 16;
    const long total = (long)ngroups * KK * M * 8;     // kpairs
    const long base = (long)lb * XP_CHUNK;
    long end = base + XP_CHUNK;
    if (end > total) end = total;
    const float sa = f16_scale_of(__builtin_bit_cast(float, e.out[e.rows_dw]));
    for (long idx = base + threadIdx.x; idx < end; idx += 256) {
        const int kp = (int)(idx & 7);
        const long sm = idx >> 3;
        const int m = (int)(sm % M);
        const int slab = (int)(sm / M);
        const int g = slab / KK, tap = slab - g * KK;
        float v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = 16 * g + 2 * kp + q;
            float x = 0.f;
            if (c < C) {
                const int co = mode ? c : m, ci = mode ? m : c;
                const float* src = co < e.split ? e.w[0] : (co < e.split2 ? e.w[1] : (co < e.split3 ? e.w[2] : e.w[3]));
                const int cor = co < e.split ? co : (co < e.split2 ? co - e.split : (co < e.split3 ? co - e.split2 : co - e.split3));
                const int stap = e.srckk == KK ? (mode == 2 ? KK - 1 - tap : tap) : (int)((e.tapmap >> (4 * tap)) & 15u);
                x = src[((long)cor * Cin + ci) * e.srckk + stap];
            }
            v[q] = x;
        }
        uint32_t hi, lo;
        f16_split2_pair(v[0], v[1], sa, hi, lo);
        uint32_t* row = e.out + ((long)slab * M + m) * APITCH;
        const int swz = (m >> 2) & 3;
        const int khalf = kp >> 2, w = kp & 3;
        row[((0 + khalf) ^ swz) * 4 + w] = hi;
        row[((2 + khalf) ^ swz) * 4 + w] = lo;
    }
}

__global__ __launch_bounds__(64) void pack_x6_clear_tail_kernel(X6PackTable t) {
    for (int i = threadIdx.x; i < t.count * ATAIL; i += 64) t.out[i / ATAIL][t.rows_dw[i / ATAIL] + i % ATAIL] = 0u;
}
__global__ __launch_bounds__(256) void pack_x6_amax_kernel(X6PackTable t) {
    int ti = 0;
    while (ti + 1 < t.count && (int)blockIdx.x >= t.ablk0[ti + 1]) ++ti;
    pack_entry_amax(pack_entry_of(t, ti), (int)blockIdx.x - t.ablk0[ti]);
}
__global__ __launch_bounds__(256) void pack_x6_kernel(X6PackTable t) {
    int ti = 0;
    while (ti + 1 < t.count && (int)blockIdx.x >= t.blk0[ti + 1]) ++ti;
    pack_entry_rows(pack_entry_of(t, ti), (int)blockIdx.x - t.blk0[ti]);
}

// ---- a device-resident PLAN: every weight operand of a step (forward, dgrad, stride-2 dgrad, rectangular, stem) in three
// launches.  The entries are the ones the ssn_conv_x6_pack_* calls between ssn_conv_x6_pack_batch_begin / _end would have
// launched table by table (<= XP_MAX entries per kernel argument); _end writes them into the caller's device buffer -- only when
// they differ from what that buffer already holds: parameters and operand buffers keep their addresses from step to step --
// and launches clear / amax / pack ONCE over all of them.
__global__ __launch_bounds__(64) void pack_x6_plan_write_kernel(X6PackTable t, X6PackEntry* plan, int first, int blk_base,
                                                                int ablk_base) {
    for (int i = threadIdx.x; i < t.count; i += 64) {
        X6PackEntry e = pack_entry_of(t, i);
        e.blk0 += blk_base;
        e.ablk0 += ablk_base;
        plan[first + i] = e;
    }
}
// entry of block b: the last one whose first block is <= b (entries without blocks share their successor's first block)
__device__ __forceinline__ int plan_entry_of_block(const X6PackEntry* plan, int count, int b, bool amax) {
    int lo = 0, hi = count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((amax ? plan[mid].ablk0 : plan[mid].blk0) <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ __launch_bounds__(256) void pack_x6_plan_clear_kernel(const X6PackEntry* plan, int count) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < count * ATAIL; i += gridDim.x * 256)
        plan[i / ATAIL].out[plan[i / ATAIL].rows_dw + i % ATAIL] = 0u;
}
__global__ __launch_bounds__(256) void pack_x6_plan_amax_kernel(const X6PackEntry* plan, int count) {
    const X6PackEntry e = plan[plan_entry_of_block(plan, count, (int)blockIdx.x, true)];
    pack_entry_amax(e, (int)blockIdx.x - e.ablk0);
}
__global__ __launch_bounds__(256) void pack_x6_plan_kernel(const X6PackEntry* plan, int count) {
    const X6PackEntry e = plan[plan_entry_of_block(plan, count, (int)blockIdx.x, false)];
    pack_entry_rows(e, (int)blockIdx.x - e.blk0);
}

bool g_pack_batching = false;
std::vector<X6PackTable> g_pack_batch;                 // the tables collected since ssn_conv_x6_pack_batch_begin
std::unordered_map<const void*, uint64_t> g_plan_hash; // device plan buffer -> hash of the entries it holds

// the three launches of a filled table (blk0[count] = blocks of the packing launch) -- or, inside a batch, its collection
int launch_pack(X6PackTable& t, hipStream_t stream) {
    if (t.count == 0 || t.blk0[t.count] == 0) return SSN_OK;
    int ablocks = 0;
    for (int i = 0; i < t.count; ++i) {
        t.ablk0[i] = ablocks;
        ablocks += (int)(((long)t.cout[i] * t.cin[i] * t.srckk[i] + XP_ACHUNK - 1) / XP_ACHUNK);
    }
    t.ablk0[t.count] = ablocks;
    if (g_pack_batching) {
        // (unused columns zeroed: the table is hashed)
        for (int i = t.count; i < XP_MAX; ++i) {
            t.w0[i] = t.w1[i] = t.w2[i] = t.w3[i] = nullptr;
            t.out[i] = nullptr;
            t.rows_dw[i] = 0;
            t.cout[i] = t.cin[i] = t.kk[i] = t.mode[i] = t.split[i] = t.split2[i] = t.split3[i] = t.srckk[i] = 0;
            t.tapmap[i] = 0;
        }
        for (int i = t.count + 1; i <= XP_MAX; ++i) t.blk0[i] = t.ablk0[i] = 0;
        g_pack_batch.push_back(t);
        return SSN_OK;
    }
    hipLaunchKernelGGL(pack_x6_clear_tail_kernel, dim3(1), dim3(64), 0, stream, t);
    hipLaunchKernelGGL(pack_x6_amax_kernel, dim3((unsigned)ablocks), dim3(256), 0, stream, t);
    hipLaunchKernelGGL(pack_x6_kernel, dim3((unsigned)t.blk0[t.count]), dim3(256), 0, stream, t);
    return SSN_OK;
}

int g_x6_dbg = 0;   // tooling: bit 4 (16) disables the 16-byte activation loads
unsigned long long* g_x6_trace = nullptr;

template <int KH, int KW, int S, int MODE, int NG, int WM, int WN, int TM, int TN>
int launch_cfg(X6Args& a, hipStream_t stream) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = NG * WN * TN * 32;
    a.hw_real = a.Ho * a.Wo;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.div_mt = make_fastdiv((uint32_t)a.n_mtiles);
    // 16-byte activation loads: stride 1, same-size, a tile width the 1 KiB pieces divide, and the caller's guarantee that the
    // bytes in front of x are readable; images that are not a multiple of 4 pixels are enumerated padded (a lane's 4 pixels
    // must not straddle two images)
    constexpr bool wide_ok = (S == 1) && BN <= 256 && BN / 16 >= 4 * NG;
    if constexpr (wide_ok) {
        if (a.x_guard >= 256 && a.Ho == a.H && a.Wo == a.W && x6_reach_bytes(a.pad_h, a.pad_w, KH, KW, a.W) <= 256 &&
            !(g_x6_dbg & 16)) {
            x6_pad_enumeration(a);
            a.n_ptiles = (a.P + BN - 1) / BN;
            const unsigned nblk = (unsigned)a.n_ptiles * (unsigned)a.n_mtiles;
            hipLaunchKernelGGL((conv_x6_kernel<KH, KW, S, MODE, true, NG, WM, WN, TM, TN>), dim3(nblk), dim3(256 * NG), 0,
                               stream, a);
            SSN_CHECK_LAUNCH("conv_x6 (wide)");
            return SSN_OK;
        }
    }
    a.n_ptiles = (a.P + BN - 1) / BN;
    const unsigned nblk = (unsigned)a.n_ptiles * (unsigned)a.n_mtiles;
    hipLaunchKernelGGL((conv_x6_kernel<KH, KW, S, MODE, false, NG, WM, WN, TM, TN>), dim3(nblk), dim3(256 * NG), 0, stream,
                       a);
    SSN_CHECK_LAUNCH("conv_x6");
    return SSN_OK;
}

// Tile ids (rows x pixels).  0-7: 4-wave workgroups:
//   0 128x128, 1 64x128, 2 96x128, 3 64x64, 4 32x128, 5 128x128 (1x4 waves), 6 64x128 (1x4 waves), 7 128x64
// 8-15: 8-wave ping-pong workgroups (two groups side by side along the pixel axis):
//   8 128x256, 9 64x256, 10 96x256, 11 64x128, 12 160x256, 13 128x128, 14 64x256 (1x4 waves), 15 128x256 (1x4 waves)
// 16-19: 4-wave workgroups with 128x64 .. 192x64 register tiles per wave, one workgroup per CU:   16 128x256, 17 96x256, 18 160x256, 19 192x256
// 20-21: tall tiles that cover the 160-192 output channels of the 14x14 stage in ONE m-tile at two workgroups per CU:
//   20 192x128 (1x4 waves), 21 160x128 (1x4 waves)
template <int KH, int KW, int S, int MODE>
int launch_tile(X6Args& a, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_cfg<KH, KW, S, MODE, 1, 2, 2, 2, 2>(a, stream);
        case 1: return launch_cfg<KH, KW, S, MODE, 1, 2, 2, 1, 2>(a, stream);
        case 2: return launch_cfg<KH, KW, S, MODE, 1, 1, 4, 3, 1>(a, stream);
        case 3: return launch_cfg<KH, KW, S, MODE, 1, 2, 2, 1, 1>(a, stream);
        case 4: return launch_cfg<KH, KW, S, MODE, 1, 1, 4, 1, 1>(a, stream);
        case 5: return launch_cfg<KH, KW, S, MODE, 1, 1, 4, 4, 1>(a, stream);
        case 6: return launch_cfg<KH, KW, S, MODE, 1, 1, 4, 2, 1>(a, stream);
        case 7: return launch_cfg<KH, KW, S, MODE, 1, 2, 2, 2, 1>(a, stream);
        case 8: return launch_cfg<KH, KW, S, MODE, 2, 2, 2, 2, 2>(a, stream);
        case 9: return launch_cfg<KH, KW, S, MODE, 2, 2, 2, 1, 2>(a, stream);
        case 10: return launch_cfg<KH, KW, S, MODE, 2, 1, 4, 3, 1>(a, stream);
        case 11: return launch_cfg<KH, KW, S, MODE, 2, 2, 2, 1, 1>(a, stream);
        case 12: return launch_cfg<KH, KW, S, MODE, 2, 1, 4, 5, 1>(a, stream);
        case 13: return launch_cfg<KH, KW, S, MODE, 2, 2, 2, 2, 1>(a, stream);
        case 14: return launch_cfg<KH, KW, S, MODE, 2, 1, 4, 2, 1>(a, stream);
        case 15: return launch_cfg<KH, KW, S, MODE, 2, 1, 4, 4, 1>(a, stream);
        case 16: return launch_cfg<KH, KW, S, MODE, 1, 1, 4, 4, 2>(a, stream);
        case 17: return launch_cfg<KH, KW, S, MODE, 1, 1, 4, 3, 2>(a, stream);
        case 18: return launch_cfg<KH, KW, S, MODE, 1, 1, 4, 5, 2>(a, stream);
        case 19: return launch_cfg<KH, KW, S, MODE, 1, 1, 4, 6, 2>(a, stream);
        case 20: return launch_cfg<KH, KW, S, MODE, 1, 1, 4, 6, 1>(a, stream);
        case 21: return launch_cfg<KH, KW, S, MODE, 1, 1, 4, 5, 1>(a, stream);
    }
    ssn_set_error("conv_x6: unknown tile config %d", cfg);
    return SSN_ERR_ARG;
}


int default_tile(int M, long P) {
    if (M % 128 == 0 || M > 256) return 0;
    if (M % 96 == 0) return 2;
    // no exact fit: the tile height with the fewest padded rows (80 rows: 96, not 2 x 64), the taller one on a tie
    const int p128 = (M + 127) / 128 * 128, p96 = (M + 95) / 96 * 96, p64 = (M + 63) / 64 * 64;
    if (p96 < p128 && p96 < p64) return 2;
    if (p128 <= p64) return 0;
    return (P >= 100000) ? 1 : 3;
}

long x6_packed_dwords(int Cout, int Cin, int ksize, int transposed) {
    return x6_packed_dwords_kk(Cout, Cin, ksize * ksize, transposed);
}

int fill_args(X6Args& a, const float* x, const uint32_t* ap, float* y, int N, int C, int H, int W, long xs, int M,
              int Ho, int Wo, long ys, int ksize, int pad, const float* x_amax, float* y_amax, int row_split, int row_gap,
              int k_split, int k_gap, const char* what) {
    if ((row_gap && (row_split % 32 || row_gap % 32 || row_split <= 0 || row_split >= M)) ||
        (k_gap && (k_split % 16 || k_gap % 16 || k_split <= 0 || k_split >= C || C % 16))) {
        ssn_set_error("%s: a row split must be a multiple of 32 inside (0, M), a channel split a multiple of 16 inside (0, C)", what);
        return SSN_ERR_ARG;
    }
    a.row_split = row_gap ? row_split : 0x7fffffff;
    a.row_gap = row_gap;
    a.k_split = k_split;
    a.k_gap = k_gap;
    if (!x_amax) {
        ssn_set_error("%s: the source tensor's amax slot is required (operand scaling of the f16 split)", what);
        return SSN_ERR_ARG;
    }
    a.x = x;
    a.ap = ap;
    a.y = y;
    a.x_amax = x_amax;
    a.y_amax = y_amax;
    a.x_amax2 = nullptr;
    a.y_amax2 = nullptr;
    a.N = N;
    a.C = C;
    a.H = H;
    a.W = W;
    a.x_img_stride = xs;
    a.M = M;
    a.Ho = Ho;
    a.Wo = Wo;
    a.y_img_stride = ys;
    a.P = N * Ho * Wo;
    a.pad_h = pad;
    a.pad_w = pad;
    a.ngroups = (C + 15) / 16;
    a.div_hw = make_fastdiv((uint32_t)(Ho * Wo));
    a.div_w = make_fastdiv((uint32_t)Wo);
    const long xb = ((long)(N - 1) * xs + (long)(C + k_gap) * H * W) * 4;
    const long ab = (long)a.ngroups * ksize * ksize * M * APITCH * 4;   // packed rows (the amax tail follows them)
    if (!(xb < (1l << 31) && ab < (1l << 31) && (long)N * Ho * Wo < (1l << 31))) {
        ssn_set_error("%s: operand larger than 2 GiB (buffer addressing)", what);
        return SSN_ERR_ARG;
    }
    a.dbg = g_x6_dbg;
    a.trace = g_x6_trace;
    a.sub_a = a.sub_b = a.sub_W = a.sub_HW = 0;
    a.x_bytes = (uint32_t)xb;
    a.a_bytes = (uint32_t)ab;
    const long yb = ((long)(N - 1) * ys + (long)(M + row_gap) * Ho * Wo) * 4;
    if (!(yb < (1l << 31))) {
        ssn_set_error("%s: output larger than 2 GiB (buffer addressing)", what);
        return SSN_ERR_ARG;
    }
    a.y_bytes = (uint32_t)yb;
    a.mask_bytes = 0;
    return SSN_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
extern "C" void ssn_conv_x6_debug_flags(int flags) {
    g_x6_dbg = flags;
}
extern "C" void ssn_conv_x6_debug_trace(unsigned long long* buf) { g_x6_trace = buf; }

extern "C" long ssn_conv_x6_packed_floats(int Cout, int Cin, int ksize, int transposed) {
    return x6_packed_dwords(Cout, Cin, ksize, transposed);
}

// ---- batched packing: every ssn_conv_x6_pack_* call between _begin and _end only records its entries; _end issues them all in three
// launches through a device-resident plan (plan: >= ssn_conv_x6_pack_batch_entries() * ssn_conv_x6_pack_entry_bytes() bytes of
// device memory that the caller keeps from step to step -- it is rewritten, by ceil(entries / 40) small launches, only when the
// recorded entries differ from the ones last written there, or with force_write: a buffer the caller has just allocated).  Replaces the ~24 launches per training step that packing layer group by layer
// group costs (cuDNN reads the caller's weights as they are: /root/reference/ssn_models.py:266, ssn_train.py:236 have no such step).
extern "C" int ssn_conv_x6_pack_batch_begin(void) {
    SSN_CHECK_ARG(!g_pack_batching, "conv x6 pack batch: already open");
    g_pack_batch.clear();
    g_pack_batching = true;
    return SSN_OK;
}
extern "C" int ssn_conv_x6_pack_batch_entries(void) {
    int n = 0;
    for (const X6PackTable& t : g_pack_batch) n += t.count;
    return n;
}
extern "C" long ssn_conv_x6_pack_entry_bytes(void) { return (long)sizeof(X6PackEntry); }
extern "C" void ssn_conv_x6_pack_batch_abort(void) {
    g_pack_batching = false;
    g_pack_batch.clear();
}
extern "C" int ssn_conv_x6_pack_batch_end(void* plan, long plan_bytes, int force_write, hipStream_t stream) {
    SSN_CHECK_ARG(g_pack_batching, "conv x6 pack batch: not open");
    g_pack_batching = false;
    const int count = ssn_conv_x6_pack_batch_entries();
    if (count == 0) return SSN_OK;
    if (!plan || plan_bytes < (long)count * (long)sizeof(X6PackEntry)) {
        g_pack_batch.clear();
        ssn_set_error("conv x6 pack batch: plan buffer %ld < %ld bytes", plan_bytes, (long)count * (long)sizeof(X6PackEntry));
        return SSN_ERR_WORKSPACE;
    }
    uint64_t h = 1469598103934665603ull;      // FNV-1a over the recorded tables
    for (const X6PackTable& t : g_pack_batch) {
        // (field by field: the struct's padding bytes are indeterminate)
        auto mix = [&](const void* p, size_t n) {
            const unsigned char* q = static_cast<const unsigned char*>(p);
            for (size_t i = 0; i < n; ++i) h = (h ^ q[i]) * 1099511628211ull;
        };
        mix(t.w0, sizeof(t.w0)); mix(t.w1, sizeof(t.w1)); mix(t.w2, sizeof(t.w2)); mix(t.w3, sizeof(t.w3));
        mix(t.out, sizeof(t.out)); mix(t.rows_dw, sizeof(t.rows_dw)); mix(t.cout, sizeof(t.cout)); mix(t.cin, sizeof(t.cin));
        mix(t.kk, sizeof(t.kk)); mix(t.mode, sizeof(t.mode)); mix(t.split, sizeof(t.split)); mix(t.split2, sizeof(t.split2));
        mix(t.split3, sizeof(t.split3)); mix(t.srckk, sizeof(t.srckk)); mix(t.tapmap, sizeof(t.tapmap));
        mix(t.blk0, sizeof(t.blk0)); mix(t.ablk0, sizeof(t.ablk0)); mix(&t.count, sizeof(t.count));
    }
    int blocks = 0, ablocks = 0;
    for (const X6PackTable& t : g_pack_batch) {
        blocks += t.blk0[t.count];
        ablocks += t.ablk0[t.count];
    }
    X6PackEntry* dev = static_cast<X6PackEntry*>(plan);
    auto it = g_plan_hash.find(plan);
    if (force_write || it == g_plan_hash.end() || it->second != h) {
        int first = 0, bb = 0, ab = 0;
        for (const X6PackTable& t : g_pack_batch) {
            hipLaunchKernelGGL(pack_x6_plan_write_kernel, dim3(1), dim3(64), 0, stream, t, dev, first, bb, ab);
            first += t.count;
            bb += t.blk0[t.count];
            ab += t.ablk0[t.count];
        }
        g_plan_hash[plan] = h;
    }
    g_pack_batch.clear();
    hipLaunchKernelGGL(pack_x6_plan_clear_kernel, dim3((unsigned)((count * ATAIL + 255) / 256)), dim3(256), 0, stream,
                       (const X6PackEntry*)dev, count);
    hipLaunchKernelGGL(pack_x6_plan_amax_kernel, dim3((unsigned)ablocks), dim3(256), 0, stream, (const X6PackEntry*)dev, count);
    hipLaunchKernelGGL(pack_x6_plan_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const X6PackEntry*)dev, count);
    SSN_CHECK_LAUNCH("conv_x6_pack_batch_end");
    return SSN_OK;
}

// Scale + split + pack `count` weights (HOST arrays, one entry per layer; see ssn_conv_pack_weights_multi for w1/split).
extern "C" int ssn_conv_x6_pack_weights_multi(int count, const float* const* w0, const float* const* w1,
                                              const float* const* w2, const float* const* w3, float* const* out,
                                              const int* cout, const int* cin, const int* ksize, const int* mode,
                                              const int* split, const int* split2, const int* split3, hipStream_t stream) {
    SSN_CHECK_ARG(count >= 0 && (count == 0 || (w0 && w1 && w2 && w3 && out && cout && cin && ksize && mode && split &&
                                                split2 && split3)),
                  "conv x6 pack: bad arguments");
    for (int base = 0; base < count; base += XP_MAX) {
        X6PackTable t;
        t.count = count - base < XP_MAX ? count - base : XP_MAX;
        int blocks = 0;
        for (int i = 0; i < t.count; ++i) {
            const int j = base + i;
            SSN_CHECK_ARG(ksize[j] == 1 || ksize[j] == 3, "conv x6 pack: ksize %d unsupported", ksize[j]);
            SSN_CHECK_ARG(mode[j] == 0 || mode[j] == 1, "conv x6 pack: mode %d", mode[j]);
            SSN_CHECK_ARG(w0[j] && out[j] && (w1[j] || split[j] >= cout[j]) && (w2[j] || split2[j] >= cout[j]) &&
                              (w3[j] || split3[j] >= cout[j]) && split[j] <= split2[j] && split2[j] <= split3[j],
                          "conv x6 pack: null pointer / bad splits");
            t.w0[i] = w0[j];
            t.w1[i] = w1[j];
            t.w2[i] = w2[j];
            t.w3[i] = w3[j];
            t.split2[i] = split2[j];
            t.split3[i] = split3[j];
            t.out[i] = (uint32_t*)out[j];
            t.cout[i] = cout[j];
            t.cin[i] = cin[j];
            t.kk[i] = ksize[j] * ksize[j];
            t.srckk[i] = t.kk[i];
            t.tapmap[i] = 0;
            t.mode[i] = mode[j];
            t.split[i] = split[j];
            t.blk0[i] = blocks;
            t.rows_dw[i] = x6_row_dwords_kk(cout[j], cin[j], t.kk[i], mode[j]);
            const long triples = t.rows_dw[i] / APITCH * 8;
            blocks += (int)((triples + XP_CHUNK - 1) / XP_CHUNK);
        }
        t.blk0[t.count] = blocks;
        launch_pack(t, stream);
    }
    SSN_CHECK_LAUNCH("conv_x6_pack_weights_multi");
    return SSN_OK;
}

// `count` rectangular-tap weights [cout][cin][kh][kw] in ceil(count / 40) x 3 launches (HOST arrays, one entry per layer):
// mode 0 = forward operand (ssn_conv_x6_packed_floats_rect floats), mode 2 = dgrad operand (transposed, taps reversed;
// ssn_conv_x6_packed_floats_dgrad_rect floats).  The Inception-v3 plan has ~40 such layers per pass.
extern "C" int ssn_conv_x6_pack_rect_multi(int count, const float* const* w, float* const* out, const int* cout,
                                           const int* cin, const int* kh, const int* kw, const int* mode, hipStream_t stream) {
    SSN_CHECK_ARG(count >= 0 && (count == 0 || (w && out && cout && cin && kh && kw && mode)), "conv x6 pack rect multi: bad arguments");
    for (int base = 0; base < count; base += XP_MAX) {
        X6PackTable t;
        t.count = count - base < XP_MAX ? count - base : XP_MAX;
        int blocks = 0;
        for (int i = 0; i < t.count; ++i) {
            const int j = base + i;
            SSN_CHECK_ARG(w[j] && out[j] && cout[j] > 0 && cin[j] > 0 && kh[j] > 0 && kw[j] > 0 && kh[j] * kw[j] <= 32 &&
                              (mode[j] == 0 || mode[j] == 2),
                          "conv x6 pack rect multi: bad entry %d", j);
            t.w0[i] = w[j];
            t.w1[i] = t.w2[i] = t.w3[i] = nullptr;
            t.split[i] = t.split2[i] = t.split3[i] = cout[j];
            t.out[i] = (uint32_t*)out[j];
            t.cout[i] = cout[j];
            t.cin[i] = cin[j];
            t.kk[i] = kh[j] * kw[j];
            t.srckk[i] = t.kk[i];
            t.tapmap[i] = 0;
            t.mode[i] = mode[j];
            t.blk0[i] = blocks;
            t.rows_dw[i] = x6_row_dwords_kk(cout[j], cin[j], t.kk[i], mode[j] ? 1 : 0);
            blocks += (int)((t.rows_dw[i] / APITCH * 8 + XP_CHUNK - 1) / XP_CHUNK);
        }
        t.blk0[t.count] = blocks;
        launch_pack(t, stream);
    }
    SSN_CHECK_LAUNCH("conv_x6_pack_rect_multi");
    return SSN_OK;
}

// Split + pack ONE forward weight [cout][cin][kh][kw] (rectangular taps: conv_x6_rect.hip); out holds
// ssn_conv_x6_packed_floats_rect() floats.
extern "C" int ssn_conv_x6_pack_weights_rect(const float* w, float* out, int cout, int cin, int kh, int kw,
                                             hipStream_t stream) {
    SSN_CHECK_ARG(w && out && cout > 0 && cin > 0 && kh > 0 && kw > 0 && kh * kw <= 32, "conv x6 pack rect: bad arguments");
    X6PackTable t;
    t.count = 1;
    t.w0[0] = w;
    t.w1[0] = nullptr;
    t.w2[0] = t.w3[0] = nullptr;
    t.split2[0] = t.split3[0] = cout;
    t.out[0] = (uint32_t*)out;
    t.cout[0] = cout;
    t.cin[0] = cin;
    t.kk[0] = kh * kw;
    t.srckk[0] = kh * kw;
    t.tapmap[0] = 0;
    t.mode[0] = 0;
    t.split[0] = cout;
    t.blk0[0] = 0;
    t.rows_dw[0] = x6_row_dwords_kk(cout, cin, kh * kw, 0);
    const long triples = t.rows_dw[0] / APITCH * 8;
    t.blk0[1] = (int)((triples + XP_CHUNK - 1) / XP_CHUNK);
    launch_pack(t, stream);
    SSN_CHECK_LAUNCH("conv_x6_pack_weights_rect");
    return SSN_OK;
}

// dgrad operand of a stride-1 layer with kh x kw taps [cout][cin][kh][kw] for ssn_conv_x6_dgrad_rect: transposed and
// tap-reversed, so that the data gradient is the FORWARD correlation of dy with it (padding kh-1-pad_h, kw-1-pad_w).
// out holds ssn_conv_x6_packed_floats_dgrad_rect() floats.
extern "C" long ssn_conv_x6_packed_floats_dgrad_rect(int Cout, int Cin, int kh, int kw) {
    return x6_packed_dwords_kk(Cout, Cin, kh * kw, 1);
}
extern "C" int ssn_conv_x6_pack_dgrad_rect(const float* w, float* out, int cout, int cin, int kh, int kw,
                                           hipStream_t stream) {
    SSN_CHECK_ARG(w && out && cout > 0 && cin > 0 && kh > 0 && kw > 0 && kh * kw <= 32, "conv x6 pack dgrad rect: bad arguments");
    X6PackTable t;
    t.count = 1;
    t.w0[0] = w;
    t.w1[0] = nullptr;
    t.w2[0] = t.w3[0] = nullptr;
    t.split[0] = t.split2[0] = t.split3[0] = cout;
    t.out[0] = (uint32_t*)out;
    t.cout[0] = cout;
    t.cin[0] = cin;
    t.kk[0] = kh * kw;
    t.srckk[0] = kh * kw;
    t.tapmap[0] = 0;
    t.mode[0] = 2;
    t.blk0[0] = 0;
    t.rows_dw[0] = x6_row_dwords_kk(cout, cin, kh * kw, 1);
    const long triples = t.rows_dw[0] / APITCH * 8;
    t.blk0[1] = (int)((triples + XP_CHUNK - 1) / XP_CHUNK);
    launch_pack(t, stream);
    SSN_CHECK_LAUNCH("conv_x6_pack_dgrad_rect");
    return SSN_OK;
}

// dgrad operand of a 3x3 / stride-2 / pad-1 convolution, one section per parity class (a, b) = (hi & 1, wi & 1) of
// the input pixel, class = 2a + b, sections back to back (ssn_conv_x6_dgrad_s2_packed_floats): class (a, b) only
// receives the taps r = a + 1 (mod 2), s = b + 1 (mod 2), i.e. a (1 + a) x (1 + b) stride-1 problem over dy with
//   A[m = ci][k = co][tap (dr, ds)] = w[co][ci][a ? 2 - 2 dr : 1][b ? 2 - 2 ds : 1].
extern "C" int ssn_conv_x6_pack_dgrad_s2(const float* w, float* out, int cout, int cin, hipStream_t stream) {
    SSN_CHECK_ARG(w && out && cout > 0 && cin > 0, "conv x6 pack dgrad s2: bad arguments");
    X6PackTable t;
    t.count = 4;
    int blocks = 0;
    long off = 0;
    for (int cls = 0; cls < 4; ++cls) {
        const int a = cls >> 1, b = cls & 1, kh = 1 + a, kw = 1 + b;
        unsigned map = 0;
        for (int dr = 0; dr < kh; ++dr)
            for (int ds = 0; ds < kw; ++ds) {
                const int r = a ? 2 - 2 * dr : 1, sx = b ? 2 - 2 * ds : 1;
                map |= (unsigned)(r * 3 + sx) << (4 * (dr * kw + ds));
            }
        t.w0[cls] = w;
        t.w1[cls] = nullptr;
        t.w2[cls] = t.w3[cls] = nullptr;
        t.split2[cls] = t.split3[cls] = cout;
        t.out[cls] = (uint32_t*)out + off;
        t.cout[cls] = cout;
        t.cin[cls] = cin;
        t.kk[cls] = kh * kw;
        t.srckk[cls] = 9;
        t.tapmap[cls] = map;
        t.mode[cls] = 1;
        t.split[cls] = cout;
        t.blk0[cls] = blocks;
        t.rows_dw[cls] = x6_row_dwords_kk(cout, cin, kh * kw, 1);
        blocks += (int)((t.rows_dw[cls] / APITCH * 8 + XP_CHUNK - 1) / XP_CHUNK);
        off += t.rows_dw[cls] + ATAIL;
    }
    t.blk0[4] = blocks;
    launch_pack(t, stream);
    SSN_CHECK_LAUNCH("conv_x6_pack_dgrad_s2");
    return SSN_OK;
}

extern "C" int ssn_conv_x6_fwd(const float* x, const float* w_packed, const float* scale, const float* shift,
                               float* y, int N, int Cin, int H, int W, long x_img_stride, int Cout, int Ho, int Wo,
                               long y_img_stride, int ksize, int stride, int pad, int relu, int x_guard_bytes,
                               int tile_cfg, const float* x_amax, float* y_amax, int raw_from, int row_split,
                               int row_gap, float* y_amax2, hipStream_t stream) {
    SSN_CHECK_ARG(x && w_packed && y, "conv x6 fwd: null pointer");
    SSN_CHECK_ARG(ksize == 1 || ksize == 3, "conv x6 fwd: ksize %d unsupported", ksize);
    SSN_CHECK_ARG(stride == 1 || stride == 2, "conv x6 fwd: stride %d unsupported", stride);
    X6Args a;
    int rc = fill_args(a, x, (const uint32_t*)w_packed, y, N, Cin, H, W, x_img_stride, Cout, Ho, Wo, y_img_stride,
                       ksize, pad, x_amax, y_amax, row_split, row_gap, 0, 0, "conv x6 fwd");
    if (rc != SSN_OK) return rc;
    a.x_guard = x_guard_bytes;
    a.scale = scale;
    a.shift = shift;
    a.relu = relu;
    a.raw_from = raw_from > 0 ? raw_from : 0x7fffffff;    // <= 0: every output row takes the affine / ReLU
    a.y_amax2 = y_amax2;
    a.accumulate = 0;
    a.mask_y = nullptr;
    a.mask_scale = nullptr;
    a.mask_img_stride = 0;
    const int cfg = tile_cfg >= 0 ? tile_cfg : default_tile(Cout, a.P);
    if (ksize == 1 && stride == 1) return launch_tile<1, 1, 1, MODE_FWD>(a, cfg, stream);
    if (ksize == 3 && stride == 1) return launch_tile<3, 3, 1, MODE_FWD>(a, cfg, stream);
    if (ksize == 3 && stride == 2) return launch_tile<3, 3, 2, MODE_FWD>(a, cfg, stream);
    ssn_set_error("conv x6 fwd: (k=%d, s=%d) has no kernel", ksize, stride);
    return SSN_ERR_ARG;
}

// stride-1 data gradient (stride-2 layers use ssn_conv_dgrad's parity path)
extern "C" int ssn_conv_x6_dgrad(const float* dy, const float* wt_packed, float* dx, int N, int Cout, int Ho, int Wo,
                                 long dy_img_stride, int Cin, int H, int W, long dx_img_stride, int ksize, int pad,
                                 int accumulate, const float* mask_y, long mask_img_stride, const float* mask_scale,
                                 int dy_guard_bytes, int tile_cfg, const float* dy_amax, float* dx_amax, int k_split,
                                 int k_gap, const float* dy_amax2, hipStream_t stream) {
    SSN_CHECK_ARG(dy && wt_packed && dx, "conv x6 dgrad: null pointer");
    SSN_CHECK_ARG(ksize == 1 || ksize == 3, "conv x6 dgrad: ksize %d unsupported", ksize);
    X6Args a;
    int rc = fill_args(a, dy, (const uint32_t*)wt_packed, dx, N, Cout, Ho, Wo, dy_img_stride, Cin, H, W,
                       dx_img_stride, ksize, pad, dy_amax, dx_amax, 0, 0, k_split, k_gap, "conv x6 dgrad");
    if (rc != SSN_OK) return rc;
    a.x_guard = dy_guard_bytes;
    a.x_amax2 = dy_amax2;
    a.scale = nullptr;
    a.shift = nullptr;
    a.relu = 0;
    a.raw_from = 0x7fffffff;
    a.accumulate = accumulate;
    a.mask_y = mask_scale ? mask_y : nullptr;
    a.mask_scale = mask_y ? mask_scale : nullptr;
    a.mask_img_stride = mask_img_stride;
    if (a.mask_y) {
        const long mb = ((long)(N - 1) * mask_img_stride + (long)Cin * H * W) * 4;
        SSN_CHECK_ARG(mb < (1l << 31), "conv x6 dgrad: mask tensor larger than 2 GiB (buffer addressing)");
        a.mask_bytes = (uint32_t)mb;
    }
    const int cfg = tile_cfg >= 0 ? tile_cfg : default_tile(Cin, a.P);
    if (ksize == 1) return launch_tile<1, 1, 1, MODE_DGRAD>(a, cfg, stream);
    return launch_tile<3, 3, 1, MODE_DGRAD>(a, cfg, stream);
}
