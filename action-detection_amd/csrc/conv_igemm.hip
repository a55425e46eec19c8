// Implicit-GEMM convolution on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// Replaces the cuDNN conv fwd / dgrad calls the reference reaches through
// model_zoo.BNInception (/root/reference/ssn_models.py:266,298) for the frozen-BN
// Conv+BN+ReLU blocks of the backbone.
//
//   D[m][p] = sum_k A[m][k] * G[k][p]
//     m : output channel            (A = weights, k = (c, r, s))
//     p : flattened (n, ho, wo)     (NCHW: p is contiguous inside an image plane, so both the
//                                    gather loads and the epilogue stores are coalesced)
//     G : the im2col gather of the NCHW input, never materialised.
//
// MODE_FWD      G[k][p] = x[n][c][ho*S - pad + r][wo*S - pad + s]
// MODE_DGRAD    pixels enumerate the conv INPUT (n, hi, wi); the source is dY:
//               G[k][p] = dy[n][co][(hi + pad - r)/S][(wi + pad - s)/S]   (0 unless divisible)
//
// K loop design (what makes the loop MFMA-bound instead of VALU-bound):
//  * K is consumed in slabs of CPS whole channels x KS*KS taps (1x1: 16 ch, 3x3: 2 ch = 18 rows,
//    7x7: 1 ch = 49 rows).  Because a slab always starts on a channel boundary, the (r, s) tap of
//    every gather row -- and therefore its border validity and its byte offset relative to the
//    slab's first channel -- is the same in every slab.  Each thread computes its gather offsets
//    ONCE (invalid taps / pixel tails get an out-of-range offset) and the loop body is just
//    `buffer_load_dword v, voff[i], rsrc, soffset` with the slab base in an SGPR: no per-element
//    address arithmetic, no bounds checks, no branches (the buffer unit returns 0 out of range).
//  * Weights are pre-packed (ssn_conv_pack_weights) into the same slab structure, split by the
//    MFMA lane-half that consumes them: Ap[slab][m][h*HP + t] = A[m][slab, k = 2t + h].  A tile is
//    then a contiguous block: float4 loads at loop-invariant offsets, and each lane's A operands
//    for four consecutive MFMAs are one conflict-free ds_read_b128.
//  * Two LDS buffers, one barrier per slab; the next slab's loads are issued before the MFMAs of
//    the current one.
//
// Tiling: 256 threads = 4 waves as WM x WN; each wave owns TM x TN MFMA tiles of 32x32.
#include "conv_epilogue.h"
#include "ssn_common.h"

namespace {

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

// slab geometry per kernel size (host and device agree through these)
template <int KS>
struct Slab {
    static constexpr int KK = KS * KS;
    static constexpr int CPS = KS == 1 ? 16 : (KS == 3 ? 2 : 1);   // channels per slab
    static constexpr int ROWS = CPS * KK;                          // real k rows per slab
    static constexpr int T = (ROWS + 1) / 2;                       // MFMA steps per slab (k pairs)
    static constexpr int HP = (T + 3) / 4 * 4;                     // per-half row length, float4 padded
    static constexpr int AP = 2 * HP;                              // packed weight row length (floats)
};

struct ConvArgs {
    const float* x;      // gather source (channel-slice base)
    const float* ap;     // packed weights [nslab][M][AP]
    float* y;            // output (channel-slice base)
    const float* scale;  // [M] or nullptr
    const float* shift;  // [M] or nullptr
    int N, C, H, W;      // gather-source dims (C = channels of source, H x W its plane)
    long x_img_stride;   // floats between consecutive images of the source
    int M;               // output channels
    int Ho, Wo;          // enumerated pixel grid
    long y_img_stride;
    int P;               // N*Ho*Wo
    int pad;
    int relu, accumulate;
    const float* mask_y;      // dgrad only: activation of the tensor whose gradient is being finalised
    const float* mask_scale;  // its per-channel folded-BN scale (NaN: channel is not a ReLU output, gradient passes through)
    long mask_img_stride;
    int n_ptiles, n_mtiles, nslab;
    int dbg;             // ablation switches for tools/ablate_conv.py (0 in production)
    int par;             // stride-2 dgrad: parity-major pixel order + tap-major slab rows (see below)
    FastDiv div_nq, div_q, div_wh;  // N*(H/2)*(W/2), (H/2)*(W/2), W/2 of the enumerated grid (parity-major decode)
    uint32_t x_bytes, a_bytes;  // extents of the gather source / packed weights (buffer descriptors)
    uint32_t y_bytes, mask_bytes;   // extents of the output / mask tensors (epilogue buffer descriptors)
    FastDiv div_hw, div_w, div_mt;
    float* y_amax;       // amax slot of the output tensor (nullptr: not tracked); see ssn_common.h: amax_emit
};

// Raw buffer descriptor over [base, base + bytes): loads whose byte offset is >= bytes return 0, which is
// how padding taps, pixel tails and channel tails are zero-filled without a branch in the gather.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// Out-of-range marker: extents are < 2^31 (checked on the host) and the per-slab scalar offset is added
// by the hardware in 32 bits, so 0x80000000 + soffset stays >= num_records for every slab.
constexpr uint32_t OOB = 0x80000000u;

// ---- stride-2 dgrad without the 4x zero-feed ------------------------------------------------------------
// For a 3x3 / stride-2 / pad-1 convolution an input pixel (hi, wi) only receives taps with r = hi+1 (mod 2),
// s = wi+1 (mod 2): 1, 2, 2 or 4 of the 9 taps depending on its parity class (hi&1, wi&1).  In PAR mode the
// pixels are enumerated class by class (n, class, hi/2, wi/2) so that a tile is (almost) single-class, and the
// slab rows are tap-major with the taps grouped by class -- slot order 4 | 3 5 | 1 7 | 0 2 6 8 -- so the live
// MFMA steps of class 00 / 01 / 10 / 11 are the contiguous ranges [0,1) / [1,3) / [3,5) / [5,9): a tile runs
// only the steps of the classes it touches instead of all nine.
__device__ __forceinline__ int par_tap(int slot) { return (int)((0x862071534ull >> (4 * slot)) & 15); }
__device__ __forceinline__ int par_lo(int cls) { return (0x5310 >> (4 * cls)) & 15; }
__device__ __forceinline__ int par_hi(int cls) { return (0x9531 >> (4 * cls)) & 15; }

// flattened pixel index -> image, row, column of the enumerated grid.  Parity mode enumerates the whole batch
// class by class: p = ((cls * N + n) * H/2 + u) * W/2 + v  with (hi, wi) = (2u + cls/2, 2v + cls%2), so only the
// three tiles that contain a class boundary mix classes.
__device__ __forceinline__ void decode_pixel(const ConvArgs& p, bool par, uint32_t pp, uint32_t& n, uint32_t& h,
                                             uint32_t& w) {
    if (par) {
        uint32_t cls, rem, r2, u, v;
        fd_divmod(pp, p.div_nq, cls, rem);
        fd_divmod(rem, p.div_q, n, r2);
        fd_divmod(r2, p.div_wh, u, v);
        h = 2 * u + (cls >> 1);
        w = 2 * v + (cls & 1);
    } else {
        uint32_t rem;
        fd_divmod(pp, p.div_hw, n, rem);
        fd_divmod(rem, p.div_w, h, w);
    }
}

template <int KS, int S, int MODE, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs p) {
    using SL = Slab<KS>;
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int KSTEP = 256 / BN;                          // gather rows handled concurrently
    constexpr int NB = (2 * SL::T + KSTEP - 1) / KSTEP;      // gather elements per thread per slab
    constexpr int BROWS = NB * KSTEP;                        // LDS rows of the gathered slab (>= 2T)
    constexpr int AP = SL::AP, HP = SL::HP, T = SL::T;
    constexpr int APITCH = AP + 4;                           // floats; keeps ds_read_b128 conflict-free
    constexpr int A_F4 = BM * AP / 4;                        // float4s in one weight tile
    constexpr int NA = (A_F4 + 255) / 256;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BN >= 64 && BN <= 256 && (256 % BN) == 0, "a wave must gather whole rows: BN in {64,128,256}");
    constexpr bool PAR = (MODE == MODE_DGRAD) && (S == 2) && (KS == 3);   // parity mode is possible at all

    __shared__ __attribute__((aligned(16))) float lds[2 * (BM * APITCH + BROWS * BN)];
    float* As0 = lds;
    float* Bs0 = lds + 2 * BM * APITCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    // ---- tile coordinates (XCD-aware: the M tiles of one pixel tile share an L2) ----
    const uint32_t nblk = (uint32_t)p.n_ptiles * (uint32_t)p.n_mtiles;
    const uint32_t logical = xcd_remap(blockIdx.x, nblk);
    uint32_t ptile, mtile;
    fd_divmod(logical, p.div_mt, ptile, mtile);
    const int m0 = (int)mtile * BM;
    const int p0 = (int)ptile * BN;

    // ---- live MFMA steps of this tile (all of them unless parity mode narrows the range) ----
    int t_lo = 0, t_hi = SL::T;
    if (PAR && p.par) {
        const uint32_t first = (uint32_t)p0;
        uint32_t last = (uint32_t)(p0 + BN - 1);
        if (last > (uint32_t)(p.P - 1)) last = (uint32_t)(p.P - 1);
        // classes are visited in order, so the union of the step ranges of the touched classes is contiguous
        t_lo = par_lo((int)fd_div(first, p.div_nq));
        t_hi = par_hi((int)fd_div(last, p.div_nq));
    }

    // ---- loop-invariant gather offsets of this thread's pixel column ----
    const int gcol = tid % BN;
    const int gk0 = wave_uniform(tid / BN);
    uint32_t voff[NB];
    {
        const int gp = p0 + gcol;
        const bool gvalid = gp < p.P;
        uint32_t n, ho, wo;
        decode_pixel(p, PAR && p.par, (uint32_t)(gvalid ? gp : 0), n, ho, wo);
        const uint32_t gbase = (uint32_t)((long)n * p.x_img_stride * 4);
        const int HW = p.H * p.W;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = gk0 + KSTEP * i;   // wave-uniform
            int cl = row / SL::KK;
            int tap = row - cl * SL::KK;
            if (PAR && p.par && row < SL::ROWS) {   // tap-major rows, taps grouped by parity class
                cl = row % SL::CPS;
                tap = par_tap(row / SL::CPS);
            }
            const int r = tap / KS, s = tap - r * KS;
            int hi, wi;
            bool ok = gvalid && (row < SL::ROWS);
            if (MODE == MODE_FWD) {
                hi = (int)ho * S - p.pad + r;
                wi = (int)wo * S - p.pad + s;
            } else {
                hi = (int)ho + p.pad - r;
                wi = (int)wo + p.pad - s;
                if (S == 2) {
                    ok = ok && (((hi | wi) & 1) == 0);
                    hi >>= 1;
                    wi >>= 1;
                }
            }
            ok = ok && ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W);
            voff[i] = ok ? gbase + (uint32_t)(cl * HW + hi * p.W + wi) * 4u : OOB;
        }
    }
    // ---- loop-invariant weight-tile offsets ----
    uint32_t aoff[NA];
    int alds[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        const int f = tid + 256 * q;
        const int row = f / (AP / 4), c4 = f - row * (AP / 4);
        const bool ok = (f < A_F4) && (m0 + row < p.M);
        aoff[q] = ok ? (uint32_t)((m0 + row) * AP + c4 * 4) * 4u : OOB;
        alds[q] = (f < A_F4) ? row * APITCH + c4 * 4 : -1;
    }
    const __amdgpu_buffer_rsrc_t xrsrc = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(p.ap, p.a_bytes);
    const uint32_t x_step = (uint32_t)(SL::CPS * p.H * p.W) * 4u;   // bytes per slab in the source
    const uint32_t a_step = (uint32_t)(p.M * AP) * 4u;             // bytes per slab in the packed weights

    float breg[NB];
    f32x4 areg[NA];
    // c_left < CPS only for the last slab of a source whose channel count is not a multiple of CPS (never
    // in BN-Inception; arbitrary shapes stay correct): rows of channels past the end are forced out of range.
    auto load_slab = [&](uint32_t xso, uint32_t aso, int c_left) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = gk0 + KSTEP * i;
            const int cl = (PAR && p.par) ? row % SL::CPS : row / SL::KK;
            const uint32_t vo = (cl < c_left) ? voff[i] : OOB;
            breg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, vo, xso, 0));
        }
#pragma unroll
        for (int q = 0; q < NA; ++q)
            areg[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff[q], aso, 0));
    };
    auto load_full = [&](uint32_t xso, uint32_t aso) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
            breg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, voff[i], xso, 0));
#pragma unroll
        for (int q = 0; q < NA; ++q)
            areg[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff[q], aso, 0));
    };
    const int c_tail = p.C - (p.nslab - 1) * SL::CPS;   // channels in the last slab (== CPS when exact)
    auto store_slab = [&](int buf) {
        float* As = As0 + buf * BM * APITCH;
        float* Bs = Bs0 + buf * BROWS * BN + gk0 * BN + gcol;
#pragma unroll
        for (int i = 0; i < NB; ++i) Bs[KSTEP * i * BN] = breg[i];
#pragma unroll
        for (int q = 0; q < NA; ++q)
            if (alds[q] >= 0) *reinterpret_cast<f32x4*>(&As[alds[q]]) = areg[q];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint32_t xso = 0, aso = 0;
    if (p.nslab == 1)
        load_slab(xso, aso, c_tail);
    else
        load_full(xso, aso);
    store_slab(0);
    __syncthreads();

    for (int t = 0; t < p.nslab; ++t) {
        const int buf = t & 1;
        xso += x_step;
        aso += a_step;
        if (!(p.dbg & 1)) {
            if (t + 2 < p.nslab)
                load_full(xso, aso);
            else if (t + 1 < p.nslab)
                load_slab(xso, aso, c_tail);
        }

        const float* As = As0 + buf * BM * APITCH + (wm * TM * 32 + li) * APITCH + lh * HP;
        const float* Bs = Bs0 + buf * BROWS * BN + lh * BN + wn * TN * 32 + li;
        // Fragment groups of 4 MFMA steps, software-pipelined: the LDS reads of group q+1 are issued before the
        // MFMAs of group q, so a wave does not sit on LDS latency between every pair of matrix instructions.
        constexpr int NQ = HP / 4;
        f32x4 af[2][TM];
        float bf[2][4][TN];
        auto load_group = [&](int q, int slot) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[slot][i] = *reinterpret_cast<const f32x4*>(As + i * 32 * APITCH + 4 * q);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (4 * q + s < T) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[slot][s][j] = Bs[(2 * (4 * q + s)) * BN + j * 32];
                }
        };
        load_group(0, 0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int cur = q & 1;
            if (q + 1 < NQ) load_group(q + 1, cur ^ 1);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (4 * q + s < T && (!PAR || (4 * q + s >= t_lo && 4 * q + s < t_hi))) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][s], bf[cur][s][j], acc[i][j],
                                                                            0, 0, 0);
                }
        }

        if (t + 1 < p.nslab && !(p.dbg & 2)) store_slab(buf ^ 1);
        if (!(p.dbg & 4)) __syncthreads();
    }

    // ---- epilogue: folded BN affine + ReLU (fwd) or accumulate + fused ReLU/BN backward (dgrad); conv_epilogue.h ----
    __syncthreads();
    float* ch = lds;
    epi_stage_channels<BM, 256>(ch, p.scale, p.shift, p.mask_scale, m0, p.M, tid, 1.f, p.relu);
    __syncthreads();
    EpiArgs e;
    e.y = p.y;
    e.mask_y = p.mask_y;
    e.y_bytes = p.y_bytes;
    e.mask_bytes = p.mask_bytes;
    e.howo4 = (uint32_t)(p.Ho * p.Wo) * 4u;
    e.M = p.M;
    e.relu = p.relu;
    e.accumulate = p.accumulate;
    e.amax = p.y_amax;
    e.amax2 = nullptr;
    e.row_split = 0x7fffffff;
    e.row_gap = 0;
    uint32_t yoff[TN], moff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pp = p0 + (wn * TN + j) * 32 + li;
        const bool pv = pp < p.P;
        uint32_t n, hw;
        if (PAR && p.par) {
            uint32_t eh, ew;
            decode_pixel(p, true, (uint32_t)(pv ? pp : 0), n, eh, ew);
            hw = eh * (uint32_t)p.Wo + ew;
        } else {
            fd_divmod((uint32_t)(pv ? pp : 0), p.div_hw, n, hw);
        }
        const uint32_t row0 = (uint32_t)(m0 + 4 * lh) * e.howo4 + hw * 4u;
        yoff[j] = pv ? (uint32_t)((long)n * p.y_img_stride * 4) + row0 : EPI_OOB;
        moff[j] = pv ? (uint32_t)((long)n * p.mask_img_stride * 4) + row0 : EPI_OOB;
    }
    conv_epilogue<TM, TN, BM>(acc, ch, e, yoff, moff, wm * TM * 32, lh, m0);
}

// Ap[slab][m][h*HP + t] = A[m][c = slab*CPS + cl][tap], (cl, tap) = decode(k = 2t + h); zero padded.
//   transposed == 0 (forward operand):  A[m][c][tap] = w[m][c][tap]          (w is [M][C][KK])
//   transposed == 1 (dgrad operand):    A[m][c][tap] = w[c][m][tap]          (w is [C][M][KK])
//   transposed == 2 (3x3 only): as 1, with the parity-mode row order of the stride-2 dgrad (see par_tap)
template <int KS>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* w, float* ap, int M, int C, int nslab,
                                                           int transposed) {
    using SL = Slab<KS>;
    const long total = (long)nslab * M * SL::AP;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int e = (int)(idx % SL::AP);
        const long sm = idx / SL::AP;
        const int m = (int)(sm % M);
        const int slab = (int)(sm / M);
        const int h = e / SL::HP, t = e - h * SL::HP;
        const int k = 2 * t + h;
        float v = 0.f;
        if (t < SL::T && k < SL::ROWS) {
            int cl = k / SL::KK, tap = k - cl * SL::KK;
            if (transposed == 2) {   // stride-2 dgrad operand: tap-major rows, taps grouped by parity class
                cl = k % SL::CPS;
                tap = par_tap(k / SL::CPS);
            }
            const int c = slab * SL::CPS + cl;
            if (c < C) v = transposed ? w[((long)c * M + m) * SL::KK + tap] : w[((long)m * C + c) * SL::KK + tap];
        }
        ap[idx] = v;
    }
}

// ---- table-driven packing of many layers in one launch (the per-layer pack launches are ~4 us each, ~120
// per training step).  An entry may name two sources: the fused pair of 1x1 reduce convolutions is packed as
// one operand whose output-channel index co < split comes from w0 and the rest from w1.
constexpr int PK_MAX = 40;
constexpr int PK_CHUNK = 8192;
struct PackTable {
    const float* w0[PK_MAX];
    const float* w1[PK_MAX];
    float* out[PK_MAX];
    int cout[PK_MAX], cin[PK_MAX], ks[PK_MAX], mode[PK_MAX], split[PK_MAX];
    int blk0[PK_MAX + 1];
    int count;
};
template <int KS>
__device__ __forceinline__ void pack_entry(const PackTable& t, int ti, long base) {
    using SL = Slab<KS>;
    const int mode = t.mode[ti];
    const int Cout = t.cout[ti], Cin = t.cin[ti];
    const int M = mode ? Cin : Cout, C = mode ? Cout : Cin;
    const int nslab = (C + SL::CPS - 1) / SL::CPS;
    const long total = (long)nslab * M * SL::AP;
    long end = base + PK_CHUNK;
    if (end > total) end = total;
    for (long idx = base + threadIdx.x; idx < end; idx += 256) {
        const int e = (int)(idx % SL::AP);
        const long sm = idx / SL::AP;
        const int m = (int)(sm % M);
        const int slab = (int)(sm / M);
        const int h = e / SL::HP, tt = e - h * SL::HP;
        const int k = 2 * tt + h;
        float v = 0.f;
        if (tt < SL::T && k < SL::ROWS) {
            int cl = k / SL::KK, tap = k - cl * SL::KK;
            if (mode == 2) {
                cl = k % SL::CPS;
                tap = par_tap(k / SL::CPS);
            }
            const int c = slab * SL::CPS + cl;
            if (c < C) {
                const int co = mode ? c : m, ci = mode ? m : c;
                const float* src = co < t.split[ti] ? t.w0[ti] : t.w1[ti];
                const int cor = co < t.split[ti] ? co : co - t.split[ti];
                v = src[((long)cor * Cin + ci) * SL::KK + tap];
            }
        }
        t.out[ti][idx] = v;
    }
}
__global__ __launch_bounds__(256) void pack_multi_kernel(PackTable t) {
    int ti = 0;
    while (ti + 1 < t.count && (int)blockIdx.x >= t.blk0[ti + 1]) ++ti;
    const long base = (long)((int)blockIdx.x - t.blk0[ti]) * PK_CHUNK;
    if (t.ks[ti] == 1)
        pack_entry<1>(t, ti, base);
    else if (t.ks[ti] == 3)
        pack_entry<3>(t, ti, base);
    else
        pack_entry<7>(t, ti, base);
}

template <int KS, int S, int MODE, int WM, int WN, int TM, int TN>
int launch_cfg(ConvArgs& a, hipStream_t stream) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    a.n_ptiles = (a.P + BN - 1) / BN;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.div_mt = make_fastdiv((uint32_t)a.n_mtiles);
    const unsigned nblk = (unsigned)a.n_ptiles * (unsigned)a.n_mtiles;
    hipLaunchKernelGGL((conv_igemm_kernel<KS, S, MODE, WM, WN, TM, TN>), dim3(nblk), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("conv_igemm");
    return SSN_OK;
}

// Tile configurations: id -> (WM, WN, TM, TN) -> BM x BN
//   0: 2,2,2,2 -> 128x128     1: 2,2,1,2 -> 64x128     2: 1,4,3,1 -> 96x128
//   3: 2,2,1,1 -> 64x64       4: 1,4,1,1 -> 32x128     5: 1,4,5,1 -> 160x128
//   6: 1,4,2,1 -> 64x128 (4 waves along pixels)        7: 2,2,2,1 -> 128x64
template <int KS, int S, int MODE>
int launch_tile(ConvArgs& a, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_cfg<KS, S, MODE, 2, 2, 2, 2>(a, stream);
        case 1: return launch_cfg<KS, S, MODE, 2, 2, 1, 2>(a, stream);
        case 2: return launch_cfg<KS, S, MODE, 1, 4, 3, 1>(a, stream);
        case 3: return launch_cfg<KS, S, MODE, 2, 2, 1, 1>(a, stream);
        case 4: return launch_cfg<KS, S, MODE, 1, 4, 1, 1>(a, stream);
        case 5: return launch_cfg<KS, S, MODE, 1, 4, 5, 1>(a, stream);
        case 6: return launch_cfg<KS, S, MODE, 1, 4, 2, 1>(a, stream);
        case 7: return launch_cfg<KS, S, MODE, 2, 2, 2, 1>(a, stream);
    }
    ssn_set_error("conv_igemm: unknown tile config %d", cfg);
    return SSN_ERR_ARG;
}

int g_conv_debug = 0;
const int kTileBM[8] = {128, 64, 96, 64, 32, 160, 64, 128};
const int kTileBN[8] = {128, 128, 128, 64, 128, 128, 128, 64};
const double kTileEff[8] = {1.0, 1.05, 1.0, 1.12, 1.2, 1.0, 1.05, 1.05};

// Heuristic used when no autotuned entry exists: padded MACs x grid quantisation x per-tile overhead.
int pick_tile(int M, long P) {
    double best = 1e300;
    int best_cfg = 0;
    for (int c = 0; c < 8; ++c) {
        const long mt = (M + kTileBM[c] - 1) / kTileBM[c];
        const long pt = (P + kTileBN[c] - 1) / kTileBN[c];
        const double padded = (double)mt * kTileBM[c] * (double)pt * kTileBN[c];
        const double blocks = (double)mt * pt;
        const double rounds = blocks / 768.0;  // ~3 workgroups per CU resident
        const double quant = (rounds < 1.0) ? 1.0 / (rounds > 0.05 ? rounds : 0.05) : ((long)rounds + 1) / rounds;
        const double cost = padded * quant * kTileEff[c];
        if (cost < best) {
            best = cost;
            best_cfg = c;
        }
    }
    return best_cfg;
}

int slab_count(int C, int ksize) {
    const int cps = ksize == 1 ? Slab<1>::CPS : (ksize == 3 ? Slab<3>::CPS : Slab<7>::CPS);
    return (C + cps - 1) / cps;
}
int packed_row(int ksize) { return ksize == 1 ? Slab<1>::AP : (ksize == 3 ? Slab<3>::AP : Slab<7>::AP); }

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI (declared in include/ssn_hip.h)
// ------------------------------------------------------------------------------------------

// floats needed for the packed form of a [Cout][Cin][k][k] weight.  transposed=0: forward operand
// (M = Cout, channels = Cin); transposed=1: dgrad operand (M = Cin, channels = Cout).
extern "C" long ssn_conv_packed_floats(int Cout, int Cin, int ksize, int transposed) {
    const int M = transposed ? Cin : Cout, C = transposed ? Cout : Cin;
    return (long)slab_count(C, ksize) * M * packed_row(ksize);
}

extern "C" int ssn_conv_pack_weights(const float* w, float* packed, int Cout, int Cin, int ksize, int transposed,
                                     hipStream_t stream) {
    SSN_CHECK_ARG(w && packed, "conv pack: null pointer");
    SSN_CHECK_ARG(ksize == 1 || ksize == 3 || ksize == 7, "conv pack: ksize %d unsupported", ksize);
    SSN_CHECK_ARG(transposed == 0 || transposed == 1 || (transposed == 2 && ksize == 3), "conv pack: bad layout %d",
                  transposed);
    const int M = transposed ? Cin : Cout, C = transposed ? Cout : Cin;
    const int nslab = slab_count(C, ksize);
    const long total = (long)nslab * M * packed_row(ksize);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    // the source is always the torch-layout weight [Cout][Cin][KK]; transposed selects which index is "m"
    if (ksize == 1)
        hipLaunchKernelGGL(pack_weights_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, w, packed, M, C, nslab,
                           transposed);
    else if (ksize == 3)
        hipLaunchKernelGGL(pack_weights_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, stream, w, packed, M, C, nslab,
                           transposed);
    else
        hipLaunchKernelGGL(pack_weights_kernel<7>, dim3((unsigned)blocks), dim3(256), 0, stream, w, packed, M, C, nslab,
                           transposed);
    SSN_CHECK_LAUNCH("conv_pack_weights");
    return SSN_OK;
}

// Pack `count` layers in ceil(count / 40) launches.  All arguments are HOST arrays with one entry per layer:
// w0 / w1 device pointers of the torch-layout weights ([cout][cin][k][k]; w1 = second source of a fused pair
// holding output channels >= split, or NULL with split = cout), out = device destination of
// ssn_conv_packed_floats() floats, mode as `transposed` of ssn_conv_pack_weights.
extern "C" int ssn_conv_pack_weights_multi(int count, const float* const* w0, const float* const* w1,
                                           float* const* out, const int* cout, const int* cin, const int* ksize,
                                           const int* mode, const int* split, hipStream_t stream) {
    SSN_CHECK_ARG(count >= 0 && (count == 0 || (w0 && w1 && out && cout && cin && ksize && mode && split)),
                  "conv pack multi: bad arguments");
    for (int base = 0; base < count; base += PK_MAX) {
        PackTable t;
        t.count = count - base < PK_MAX ? count - base : PK_MAX;
        int blocks = 0;
        for (int i = 0; i < t.count; ++i) {
            const int j = base + i;
            SSN_CHECK_ARG(ksize[j] == 1 || ksize[j] == 3 || ksize[j] == 7, "conv pack multi: ksize %d", ksize[j]);
            SSN_CHECK_ARG(mode[j] >= 0 && mode[j] <= 2 && (mode[j] != 2 || ksize[j] == 3), "conv pack multi: mode");
            SSN_CHECK_ARG(w0[j] && out[j] && (w1[j] || split[j] >= cout[j]), "conv pack multi: null pointer");
            t.w0[i] = w0[j];
            t.w1[i] = w1[j];
            t.out[i] = out[j];
            t.cout[i] = cout[j];
            t.cin[i] = cin[j];
            t.ks[i] = ksize[j];
            t.mode[i] = mode[j];
            t.split[i] = split[j];
            t.blk0[i] = blocks;
            const long total = ssn_conv_packed_floats(cout[j], cin[j], ksize[j], mode[j] ? 1 : 0);
            blocks += (int)((total + PK_CHUNK - 1) / PK_CHUNK);
        }
        t.blk0[t.count] = blocks;
        if (blocks)
            hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, t);
    }
    SSN_CHECK_LAUNCH("conv_pack_weights_multi");
    return SSN_OK;
}

extern "C" int ssn_conv_bn_relu_fwd(const float* x, const float* w_packed, const float* scale, const float* shift,
                                    float* y, int N, int Cin, int H, int W, long x_img_stride, int Cout, int Ho,
                                    int Wo, long y_img_stride, int ksize, int stride, int pad, int relu,
                                    int tile_cfg, float* y_amax, hipStream_t stream) {
    SSN_CHECK_ARG(x && w_packed && y, "conv fwd: null pointer");
    SSN_CHECK_ARG(ksize == 1 || ksize == 3 || ksize == 7, "conv fwd: ksize %d unsupported", ksize);
    SSN_CHECK_ARG(stride == 1 || stride == 2, "conv fwd: stride %d unsupported", stride);
    SSN_CHECK_ARG((long)N * Ho * Wo < (1l << 31), "conv fwd: too many pixels");
    ConvArgs a;
    a.x = x;
    a.ap = w_packed;
    a.y = y;
    a.scale = scale;
    a.shift = shift;
    a.N = N;
    a.C = Cin;
    a.H = H;
    a.W = W;
    a.x_img_stride = x_img_stride;
    a.M = Cout;
    a.Ho = Ho;
    a.Wo = Wo;
    a.y_img_stride = y_img_stride;
    a.P = N * Ho * Wo;
    a.pad = pad;
    a.relu = relu;
    a.accumulate = 0;
    a.y_amax = y_amax;
    a.mask_y = nullptr;
    a.mask_scale = nullptr;
    a.mask_img_stride = 0;
    a.nslab = slab_count(Cin, ksize);
    a.dbg = g_conv_debug;
    a.par = 0;
    a.div_nq = make_fastdiv(1);
    a.div_q = make_fastdiv(1);
    a.div_wh = make_fastdiv(1);
    a.div_hw = make_fastdiv((uint32_t)(Ho * Wo));
    a.div_w = make_fastdiv((uint32_t)Wo);
    const long xb = ((long)(N - 1) * x_img_stride + (long)Cin * H * W) * 4;
    const long ab = ssn_conv_packed_floats(Cout, Cin, ksize, 0) * 4;
    const long yb = ((long)(N - 1) * y_img_stride + (long)Cout * Ho * Wo) * 4;
    SSN_CHECK_ARG(xb < (1l << 31) && ab < (1l << 31) && yb < (1l << 31),
                  "conv fwd: operand larger than 2 GiB (buffer addressing)");
    a.x_bytes = (uint32_t)xb;
    a.a_bytes = (uint32_t)ab;
    a.y_bytes = (uint32_t)yb;
    a.mask_bytes = 0;
    const int cfg = tile_cfg >= 0 ? tile_cfg : pick_tile(Cout, a.P);
    if (ksize == 1 && stride == 1) return launch_tile<1, 1, MODE_FWD>(a, cfg, stream);
    if (ksize == 3 && stride == 1) return launch_tile<3, 1, MODE_FWD>(a, cfg, stream);
    if (ksize == 3 && stride == 2) return launch_tile<3, 2, MODE_FWD>(a, cfg, stream);
    if (ksize == 7 && stride == 2) return launch_tile<7, 2, MODE_FWD>(a, cfg, stream);
    ssn_set_error("conv fwd: (k=%d, s=%d) has no kernel", ksize, stride);
    return SSN_ERR_ARG;
}

// Which packed-weight layout ssn_conv_dgrad wants for this conv: 2 (parity mode) for 3x3 / stride-2 / pad-1 with an
// even input size, else 1.  Pass the result to ssn_conv_pack_weights(transposed=...) and to ssn_conv_dgrad.
extern "C" int ssn_conv_dgrad_layout(int ksize, int stride, int pad, int H, int W) {
    return (ksize == 3 && stride == 2 && pad == 1 && (H % 2) == 0 && (W % 2) == 0) ? 2 : 1;
}

// dx[n][ci][hi][wi] (+)= sum_{co,r,s} w[co][ci][r][s] * dy[n][co][(hi+pad-r)/S][(wi+pad-s)/S]
// wt_packed = ssn_conv_pack_weights(w, ..., transposed = 1)
// mask_y / mask_scale (optional): when this launch is the LAST writer of dx, apply the backward of the ReLU +
// frozen BN that produced the tensor dx is the gradient of: dx <- dx * (mask_y > 0) * mask_scale[ci]
// (a NaN mask_scale[ci] marks a channel that is not a ReLU output: dx passes through; any finite scale, negative
// ones included -- BN gammas of trained checkpoints can be negative -- is applied with its sign).
extern "C" int ssn_conv_dgrad(const float* dy, const float* wt_packed, float* dx, int N, int Cout, int Ho, int Wo,
                              long dy_img_stride, int Cin, int H, int W, long dx_img_stride, int ksize,
                              int stride, int pad, int accumulate, const float* mask_y, long mask_img_stride,
                              const float* mask_scale, int wt_layout, int tile_cfg, float* dx_amax,
                              hipStream_t stream) {
    SSN_CHECK_ARG(dy && wt_packed && dx, "conv dgrad: null pointer");
    SSN_CHECK_ARG(ksize == 1 || ksize == 3, "conv dgrad: ksize %d unsupported", ksize);
    SSN_CHECK_ARG(stride == 1 || stride == 2, "conv dgrad: stride %d unsupported", stride);
    ConvArgs a;
    a.x = dy;
    a.ap = wt_packed;
    a.y = dx;
    a.scale = nullptr;
    a.shift = nullptr;
    a.N = N;
    a.C = Cout;
    a.H = Ho;
    a.W = Wo;
    a.x_img_stride = dy_img_stride;
    a.M = Cin;
    a.Ho = H;
    a.Wo = W;
    a.y_img_stride = dx_img_stride;
    a.P = N * H * W;
    a.pad = pad;
    a.relu = 0;
    a.accumulate = accumulate;
    a.y_amax = dx_amax;
    a.mask_y = mask_scale ? mask_y : nullptr;
    a.mask_scale = mask_y ? mask_scale : nullptr;
    a.mask_img_stride = mask_img_stride;
    a.nslab = slab_count(Cout, ksize);
    a.dbg = g_conv_debug;
    a.par = (wt_layout == 2);
    SSN_CHECK_ARG(wt_layout == 1 || wt_layout == 2, "conv dgrad: wt_layout must be 1 or 2");
    SSN_CHECK_ARG(!a.par || ssn_conv_dgrad_layout(ksize, stride, pad, H, W) == 2,
                  "conv dgrad: parity layout needs a 3x3 / stride-2 / pad-1 conv with even input size");
    a.div_nq = make_fastdiv(a.par ? (uint32_t)(N * (H / 2) * (W / 2)) : 1u);
    a.div_q = make_fastdiv(a.par ? (uint32_t)((H / 2) * (W / 2)) : 1u);
    a.div_wh = make_fastdiv(a.par ? (uint32_t)(W / 2) : 1u);
    a.div_hw = make_fastdiv((uint32_t)(H * W));
    a.div_w = make_fastdiv((uint32_t)W);
    const long xb = ((long)(N - 1) * dy_img_stride + (long)Cout * Ho * Wo) * 4;
    const long ab = ssn_conv_packed_floats(Cout, Cin, ksize, 1) * 4;
    const long yb = ((long)(N - 1) * dx_img_stride + (long)Cin * H * W) * 4;
    const long mb = a.mask_y ? ((long)(N - 1) * mask_img_stride + (long)Cin * H * W) * 4 : 0;
    SSN_CHECK_ARG(xb < (1l << 31) && ab < (1l << 31) && yb < (1l << 31) && mb < (1l << 31),
                  "conv dgrad: operand larger than 2 GiB (buffer addressing)");
    a.x_bytes = (uint32_t)xb;
    a.a_bytes = (uint32_t)ab;
    a.y_bytes = (uint32_t)yb;
    a.mask_bytes = (uint32_t)mb;
    const int cfg = tile_cfg >= 0 ? tile_cfg : pick_tile(Cin, a.P);
    if (ksize == 1 && stride == 1) return launch_tile<1, 1, MODE_DGRAD>(a, cfg, stream);
    if (ksize == 3 && stride == 1) return launch_tile<3, 1, MODE_DGRAD>(a, cfg, stream);
    if (ksize == 3 && stride == 2) return launch_tile<3, 2, MODE_DGRAD>(a, cfg, stream);
    ssn_set_error("conv dgrad: (k=%d, s=%d) has no kernel", ksize, stride);
    return SSN_ERR_ARG;
}

extern "C" int ssn_conv_pick_tile(int M, long P) { return pick_tile(M, P); }

// Tooling only (tools/ablate_conv.py): bit 0 skip global loads, 1 skip LDS stores, 2 skip barriers, 3 skip MFMAs.
// Results are garbage with any bit set; production code never calls this.
extern "C" int ssn_conv_debug_flags(int flags) {
    const int old = g_conv_debug;
    g_conv_debug = flags;
    return old;
}
