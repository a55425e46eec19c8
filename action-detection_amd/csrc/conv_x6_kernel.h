// The split-operand implicit-GEMM convolution kernel template (fp32 operands as two f16 terms, three f16 MFMAs per k16
// step; the family keeps its historical "x6" name from the first version, which used three bf16 terms and six
// products), shared by conv_x6.hip (the square 1x1 / 3x3 layers of BN-Inception, forward and dgrad) and
// conv_x6_rect.hip (rectangular taps: 5x5, 1x7, 7x1, 1x3, 3x1 forward, and the stride-2 dgrad classes).
// Design notes: see the head of conv_x6.hip.
#pragma once
#include "conv_epilogue.h"
#include "ssn_common.h"
#include <type_traits>

namespace x6 {

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int APITCH = 16;    // dwords per packed weight row: 4 chunks of 16 B (2 planes x 2 k-halves), chunk-swizzled
constexpr int ATAIL = 4;      // dwords behind the packed rows of a weight: [amax of the weight, 0, 0, 0]
constexpr uint32_t OOB = 0x80000000u;

struct X6Args {
    const float* x;       // gather source (channel-slice base), fp32 NCHW
    const uint32_t* ap;   // packed split weights [nslab][M][APITCH], then ATAIL dwords (the weight's amax)
    const float* x_amax;  // amax slot of the gather source's tensor (required)
    float* y_amax;        // amax slot of the output tensor (nullptr: not tracked)
    // A tensor that holds a block's output AND (behind it) the block's reduce rows has one slot per region, so that every slot
    // is final before anybody reads it (the two regions are written and read by concurrent launches): a launch whose rows /
    // channels span both regions raises both (y_amax2) resp. takes the larger of both (x_amax2).  nullptr: one region.
    const float* x_amax2;
    float* y_amax2;
    float* y;
    const float* scale;
    const float* shift;
    int N, C, H, W;       // gather-source dims
    long x_img_stride;
    int M;
    int Ho, Wo;           // enumerated pixel grid
    long y_img_stride;
    int P;                // enumerated pixel slots (N * Ho * Wo, or N * the padded plane size: see hw_real)
    // 16-byte activation loads need 4 consecutive pixel slots to lie in ONE image: planes whose size is not a multiple of 4 are
    // enumerated with their size rounded up (div_hw divides by the padded size); slots >= hw_real of an image do not exist --
    // they gather whatever follows the plane, contribute to nothing that is stored and are kept out of the amax.
    int hw_real;
    int pad_h, pad_w;
    int relu, accumulate;
    int raw_from;         // output rows >= raw_from take no affine and no ReLU (conv_epilogue.h); M or more: none
    int row_split, row_gap;   // forward: output rows >= row_split are stored row_gap channels further up (conv_epilogue.h)
    int k_split, k_gap;       // gather source: channels >= k_split sit k_gap channels further up its tensor (multiples of 16)
    const float* mask_y;
    const float* mask_scale;
    long mask_img_stride;
    int n_ptiles, n_mtiles, ngroups;   // ngroups = ceil(C / 16)
    uint32_t x_bytes, a_bytes, y_bytes, mask_bytes;
    int x_guard;   // readable bytes in front of x (>= 256 enables the 16-byte activation loads)
    // sub-sampled output (stride-2 dgrad as four stride-1 problems, one per parity class of the input pixel): the
    // enumerated pixel (u, v) is stored at (2u + sub_a, 2v + sub_b) of planes sub_W wide with sub_HW elements.
    // sub_HW == 0: dense output.
    int sub_a, sub_b, sub_W, sub_HW;
    unsigned long long* trace;   // tooling only: per-block phase timestamps (tools/trace_x6.py), normally null
    int dbg;      // tooling only (tools/ablate_x6.py)
    FastDiv div_hw, div_w, div_mt;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// (operand split and scaling: f16_pair_rne / f16_pair_lo / f16_pair_hi / f16_scale_of of ssn_common.h)

// The 16-byte LDS-DMA form only exists for gfx950; hipcc's HOST pass (no target features) rejects it and then
// silently drops the kernel's launch stub, so it is compiled for the device pass only.
#if defined(__HIP_DEVICE_COMPILE__)
#define X6_DMA_B128(rsrc_, dst_, voff_, soff_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_, SSN_LDS_PTR(dst_), 16, voff_, soff_, 0, 0)
#else
#define X6_DMA_B128(rsrc_, dst_, voff_, soff_) ((void)(dst_), (void)(voff_), (void)(soff_))
#endif

// tooling build only (tools/build_trace_lib.sh): per-phase cycle accumulators of wave 0 and wave 4
#ifdef X6_PHASE_TRACE
#define X6_PH_DECL unsigned long long ph_[6] = {0, 0, 0, 0, 0, 0}, phc_ = __builtin_readcyclecounter()
#define X6_PH(k)                                                  \
    do {                                                          \
        const unsigned long long n_ = __builtin_readcyclecounter(); \
        ph_[k] += n_ - phc_;                                      \
        phc_ = n_;                                                \
    } while (0)
#else
#define X6_PH_DECL
#define X6_PH(k)
#endif

template <int KH, int KW, int S, int MODE, bool WIDE, int NG, int WM, int WN, int TM, int TN>
// big register tiles (TM * TN >= 6, e.g. 128 x 64 per wave) run one wave per SIMD with up to 512 VGPRs: a third of the
// LDS traffic per MFMA of the 1 x 4-wave tiles, the software pipeline hides the LDS round trip without a partner wave
__global__ __launch_bounds__(256 * NG, (NG == 1 && TM * TN >= 6 && TN >= 2) ? 1 : 2) void conv_x6_kernel(X6Args p) {
    constexpr int NW = 4 * NG;              // waves per workgroup
    constexpr int NT = 64 * NW;
    constexpr int BM = WM * TM * 32;
    constexpr int BNG = WN * TN * 32;       // pixel columns of one wave group
    constexpr int BN = NG * BNG;
    constexpr int KK = KH * KW;
    constexpr int A_PIECES = BM / 16;       // 1 KiB pieces of the BM x 64 B weight tile
    constexpr int NA = (A_PIECES + NW - 1) / NW;
    constexpr int A_STAGE = NA * NW * 256;  // dwords; padded so that every wave copies exactly NA pieces
    constexpr int B_STAGE = 16 * BN;        // dwords
    constexpr int STAGE = A_STAGE + B_STAGE;
    constexpr int NSTAGE = 3;
    constexpr int SEGS = BN / 64;           // 64-pixel segments per k-row
    // activation DMA instructions per wave and slab: 4 B per lane (256 B pieces = one k-row of 64 pixels), or,
    // WIDE, 16 B per lane (1 KiB pieces = 256 consecutive (k-row, pixel) slots, 4 pixels per lane)
    constexpr int NB = WIDE ? BN / 16 / NW : 16 * SEGS / NW;
    constexpr int KSTEP = NW / SEGS;
    constexpr int RPP = 256 / BN > 0 ? 256 / BN : 1;   // WIDE: k-rows per piece
    constexpr uint32_t GUARD = WIDE ? 256u : 0u;       // WIDE: readable bytes in front of x (contract)
    static_assert(!WIDE || (S == 1 && BN <= 256 && BN / 16 >= NW), "WIDE: stride 1, 64*NW/4 <= BN <= 256");
    constexpr int NLOAD = NA + NB;
    static_assert(WM * WN == 4, "4 waves per group");
    static_assert(NW % SEGS == 0 && (16 * SEGS) % NW == 0, "unsupported tile width");

    // WIDE 3x3: rows of zeros behind the ring; a lane whose tap falls on padding reads its fragment from there
    constexpr int ZROWS = (WIDE && KK > 1) ? 7 * BN + 64 : 0;
    __shared__ __attribute__((aligned(1024))) uint32_t lds[NSTAGE * STAGE + ZROWS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int grp = wave >> 2, gw = wave & 3;
    const int wm = gw / WN, wn = gw % WN;
    const int li = lane & 31, lh = lane >> 5;

    const uint32_t nblk = (uint32_t)p.n_ptiles * (uint32_t)p.n_mtiles;
    const uint32_t logical = xcd_remap(blockIdx.x, nblk);
    uint32_t ptile, mtile;
    fd_divmod(logical, p.div_mt, ptile, mtile);
    const int m0 = (int)mtile * BM;
    const int p0 = (int)ptile * BN;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
    if (p.trace) tr0 = __builtin_readcyclecounter();

    // ---- B gather state ----
    //  narrow: this lane fetches pixel (seg * 64 + lane) of k-rows krow0 + i * KSTEP; taps that fall outside the
    //          image are turned into out-of-range offsets (-> zeros) per slab.
    //  WIDE:   this lane fetches the 4 consecutive pixels wpx.. of k-row (piece * RPP + wkl).  Address arithmetic is
    //          linear in the pixel index for stride 1, so no per-pixel border handling is possible at the load:
    //          whatever lies next to the image in memory is fetched (x must be preceded by GUARD readable bytes),
    //          and the padding positions are zeroed when the fragments are read (fmask).
    const int seg = wave % SEGS;
    const int krow0 = wave / SEGS;
    const int wkl = WIDE ? (4 * lane) / BN : 0;
    const int wpx = WIDE ? (4 * lane) % BN : seg * 64 + lane;
    uint32_t gbase;                        // byte offset of the tap-(0,0) input element (may wrap below zero)
    uint32_t gmask = 0;                    // bit t: tap t reads inside the image (WIDE: bit 0 = pixel exists)
    const uint32_t hw_bytes = (uint32_t)(p.H * p.W) * 4u;
    auto tap_mask = [&](int gp, uint32_t& n, int& h0, int& w0) {
        bool gvalid = gp < p.P;
        uint32_t hw, ho, wo, m = 0;
        fd_divmod((uint32_t)(gvalid ? gp : 0), p.div_hw, n, hw);
        gvalid = gvalid && hw < (uint32_t)p.hw_real;
        fd_divmod(hw, p.div_w, ho, wo);
        h0 = (MODE == MODE_FWD) ? (int)ho * S - p.pad_h : (int)ho + p.pad_h;
        w0 = (MODE == MODE_FWD) ? (int)wo * S - p.pad_w : (int)wo + p.pad_w;
#pragma unroll
        for (int t = 0; t < KK; ++t) {
            const int r = t / KW, s = t - r * KW;
            const int hi = (MODE == MODE_FWD) ? h0 + r : h0 - r;
            const int wi = (MODE == MODE_FWD) ? w0 + s : w0 - s;
            if (gvalid && ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W)) m |= 1u << t;
        }
        return m;
    };
    {
        uint32_t n;
        int h0, w0;
        const uint32_t m = tap_mask(p0 + wpx, n, h0, w0);
        gmask = WIDE ? (uint32_t)(p0 + wpx < p.P) : m;
        gbase = (uint32_t)((long)n * p.x_img_stride * 4) + (uint32_t)((h0 * p.W + w0) * 4) + (uint32_t)wkl * hw_bytes +
                GUARD;
    }
    // WIDE: validity of the taps at the pixels of this lane's B fragments
    uint32_t fmask[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        uint32_t n;
        int h0, w0;
        fmask[j] = (WIDE && KK > 1) ? tap_mask(p0 + grp * BNG + (wn * TN + j) * 32 + li, n, h0, w0) : 0u;
    }

    // ---- A copy: wave w moves 1 KiB pieces (q * NW + w) of the tile ----
    uint32_t aoff[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        const int f = (q * NW + wave) * 64 + lane;   // 16-byte chunk of the tile
        aoff[q] = (f < BM * 4 && m0 + f / 4 < p.M) ? (uint32_t)(m0 * APITCH) * 4u + (uint32_t)f * 16u : OOB;
    }
    const __amdgpu_buffer_rsrc_t xrsrc = make_rsrc(reinterpret_cast<const char*>(p.x) - GUARD, p.x_bytes + GUARD);
    const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(p.ap, p.a_bytes);
    const uint32_t a_step = (uint32_t)(p.M * APITCH) * 4u;
    const int nslab = p.ngroups * KK;
    const int c_last = p.C - (p.ngroups - 1) * 16;   // channels in the last group (16 when exact)

    // DMA of slab (ig, itap) into a ring stage, as NLOAD separate pieces (NB activation rows, then NA weight pieces)
    // so that they can be dealt out between the MFMAs: the texture path takes 64 B/clk per CU, and a wave that
    // issues its whole share in one burst (together with the 7 other waves) stalls ~800 cycles at the issue.
    // Slabs past the end (the ring always runs two ahead) are issued too, with every offset out of range, so the
    // K loop has no "is there still something to fetch" branches: such a piece deposits zeros in a ring slot that is
    // never read again.
    // Producer state = the slab that is fetched next, kept as running scalars (no multiplies in the loop): groups
    // still to fetch, tap, tap displacement in bytes, channel-group offset of the source and slab offset of the weights.
    int pf_left = (p.dbg & 1) ? 0 : p.ngroups;
    int pf_tap = 0, pf_col = 0;
    uint32_t pf_d = 0;                                                       // (r * W + s) * 4 of the tap
    uint32_t pf_x = (uint32_t)(WIDE ? wave * RPP : krow0) * hw_bytes;        // + 16 channels per group
    uint32_t pf_a = 0;                                                       // + a_step per slab
    const uint32_t row_wrap = (uint32_t)(p.W - KW) * 4u, group_step = 16u * hw_bytes;
    // a gap in the channel axis of the gather source (the dgrad of a fused block-input launch reads the gradient of the 1x1
    // branch at the head of the block-output gradient and the reduce / projection gradients behind the block's channels)
    const int gap_left = p.k_gap ? p.ngroups - p.k_split / 16 : -(1 << 30);   // value of pf_left when the gap is crossed
    const uint32_t gap_bytes = (uint32_t)p.k_gap * hw_bytes;
    uint32_t d_vo, d_so, d_aso, d_dead;
    uint32_t *d_b, *d_a;
    bool d_tail;
    auto issue_begin = [&](uint32_t st_off) {   // st_off = dword offset of the destination ring slot
        const uint32_t live = pf_left > 0 ? 1u : 0u;
        d_dead = live ? 0u : OOB;
        uint32_t cand = gbase, ok = gmask & 1u;
        if (KK > 1) {
            cand = (MODE == MODE_FWD) ? gbase + pf_d : gbase - pf_d;
            if (!WIDE) ok = (gmask >> pf_tap) & 1u;
        }
        d_vo = (ok & live) ? cand : OOB;
        d_tail = (pf_left == 1) && (c_last != 16);
        d_so = pf_x;
        d_b = lds + st_off + A_STAGE + (WIDE ? wave * 256 : krow0 * BN + seg * 64);
        d_aso = pf_a;
        d_a = lds + st_off + wave * 256;
        // advance to the next slab
        pf_a += a_step;
        if (KK == 1) {
            pf_x += group_step;
            if (--pf_left == gap_left) pf_x += gap_bytes;
        } else {
            pf_d += 4u;
            if (++pf_col == KW) {
                pf_col = 0;
                pf_d += row_wrap;
            }
            if (++pf_tap == KK) {
                pf_tap = 0;
                pf_d = 0;
                pf_x += group_step;
                if (--pf_left == gap_left) pf_x += gap_bytes;
            }
        }
    };
    auto issue_piece = [&](int k) {   // k is a compile-time constant at every call site
        if (k < NB && WIDE) {
            uint32_t v = d_vo;
            if (d_tail && ((k * NW + wave) * RPP + wkl >= c_last)) v = OOB;
            X6_DMA_B128(xrsrc, d_b + k * NW * 256, v, d_so + (uint32_t)(k * NW * RPP) * hw_bytes);
        } else if (k < NB) {
            uint32_t v = d_vo;
            if (d_tail && (krow0 + k * KSTEP >= c_last)) v = OOB;   // channels past the end (never in BN-Inception)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, SSN_LDS_PTR(d_b + k * KSTEP * BN), 4, v,
                                                     d_so + (uint32_t)(k * KSTEP) * hw_bytes, 0, 0);
        } else {
            X6_DMA_B128(arsrc, d_a + (k - NB) * NW * 256, aoff[k - NB] | d_dead, d_aso);
        }
    };
    auto issue = [&](uint32_t st_off) {
        issue_begin(st_off);
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) issue_piece(k);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (ZROWS) {
        for (int i = tid; i < ZROWS; i += NT) lds[NSTAGE * STAGE + i] = 0u;
        __syncthreads();
    }
    issue(0);
    issue(STAGE);   // past the last slab the pieces turn into out-of-range (all-zero) copies: no branches in the loop
    if (p.trace) tr1 = __builtin_readcyclecounter();

    // operand scales (powers of two, exact): the weight's amax sits behind its packed rows, the activations' in the
    // slot their producers maintained
    const float sa = f16_scale_of(__builtin_bit_cast(float, p.ap[p.a_bytes >> 2]));
    const float sb = f16_scale_of(p.x_amax2 ? fmaxf(*p.x_amax, *p.x_amax2) : *p.x_amax);
    const float inv = 1.f / (sa * sb);

    // fragment addressing: A row (wm*TM+i)*32 + li, 16-byte chunk (2*plane + lh) ^ ((row >> 2) & 3): the 16 lanes of a
    // ds_read_b128 group (li in {0-3,12-15,20-27} or {4-11,16-19,28-31}) then cover all 16 bank quads once
    const int swz = (li >> 2) & 3;
    int achunk[2];
#pragma unroll
    for (int pn = 0; pn < 2; ++pn) achunk[pn] = ((2 * pn + lh) ^ swz) * 4;
    const int arow = (wm * TM * 32 + li) * APITCH;
    const int bcol = A_STAGE + (8 * lh) * BN + grp * BNG + wn * TN * 32 + li;

    // Operand registers of one slab: the raw fp32 activations of this lane's fragments and the weight fragments.  Two
    // sets: a free-running (NG == 1) wave reads slab t+1 from LDS while it multiplies slab t.
    struct Frags {
        float raw[TN][8];
        f16x8 af[2][TM];
    };
    Frags fr0, fr1;
    uint32_t pl[2][TN][4];   // the two f16 planes of the slab being multiplied, as k-pairs
    // The low plane of pair e (k = 2e, 2e+1) of fragment j: residual against the top plane + conversion, 4 VALU (two
    // v_cvt_f32_f16, one packed subtract, one v_cvt_pk_f16_f32) -- what fits in the shadow of half an MFMA.
    // (f.raw already holds the SCALED values and pl[0][j][e] the top plane: top_plane() runs before any low step.)
    auto low_step = [&](const Frags& f, int j, int e) {
        pl[1][j][e] = f16_pair_rne(f.raw[j][2 * e] - f16_pair_lo(pl[0][j][e]), f.raw[j][2 * e + 1] - f16_pair_hi(pl[0][j][e]));
    };
    // LDS -> registers: raw activations (a tap on padding reads the rows of zeros instead) and weight fragments, as
    // NREAD separate steps (a k-pair of one activation fragment = one ds_read2st64_b32, or one 16-byte weight read) so
    // that a pipelined wave can deal them out between its MFMAs: eight waves that all burst 13+ reads right after the
    // barrier queue up behind the LDS pipeline for ~200 cycles before anybody's first MFMA issues.
    int ctap = 0;   // tap of the slab being read (WIDE 3x3 only)
    constexpr int NREAD = 4 * TN + 2 * TM;
    const uint32_t* rd_src[TN];
    const uint32_t* rd_a;
    auto read_begin = [&](uint32_t st_off) {
        const uint32_t* Ls = lds + st_off;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            rd_src[j] = Ls + bcol + j * 32;
            if (WIDE && KK > 1) {
                const bool inside = (fmask[j] >> ctap) & 1u;
                rd_src[j] = inside ? rd_src[j] : lds + NSTAGE * STAGE + li;
            }
        }
        rd_a = Ls + arow;
        if (WIDE && KK > 1) ctap = (ctap + 1 == KK) ? 0 : ctap + 1;
    };
    auto read_step = [&](Frags& f, int k) {   // k is a compile-time constant at every call site
        if (k < 4 * TN) {
            const int j = k / 4, e = k % 4;
            f.raw[j][2 * e] = __builtin_bit_cast(float, rd_src[j][(2 * e) * BN]);
            f.raw[j][2 * e + 1] = __builtin_bit_cast(float, rd_src[j][(2 * e + 1) * BN]);
        } else {
            const int q = k - 4 * TN, pn = q / TM, i = q % TM;
            f.af[pn][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(rd_a + i * 32 * APITCH + achunk[pn]));
        }
    };
    auto read_frags = [&](uint32_t st_off, Frags& f) {
        read_begin(st_off);
#pragma unroll
        for (int k = 0; k < NREAD; ++k) read_step(f, k);
    };
    // scale the raw values in place and take the top plane (one packed multiply + one v_cvt_pk_f16_f32 per k-pair);
    // `all_planes`: also the low one (the ping-pong groups split while the other group multiplies; a free-running wave
    // does that in the shadow of its own MFMAs instead, see mfma)
    auto top_plane = [&](Frags& f, bool all_planes) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f.raw[j][2 * e] *= sb;
                f.raw[j][2 * e + 1] *= sb;
                pl[0][j][e] = f16_pair_rne(f.raw[j][2 * e], f.raw[j][2 * e + 1]);
                if (all_planes) low_step(f, j, e);
            }
    };
    auto front = [&](uint32_t st_off, bool all_planes) {
        read_frags(st_off, fr0);
        top_plane(fr0, all_planes);
    };
    // Three partial products per accumulator tile: w_lo x_hi, w_hi x_hi, w_hi x_lo.  The two that only need the TOP plane
    // of the activations go first and (`interleave`) the 4 * TN low steps that produce the low plane are dealt out
    // behind those MFMAs, fenced in place: the matrix pipe starts as soon as the LDS reads are back and the ~16 VALU
    // per fragment run in its shadow.  (The running fp32 accumulator already holds the earlier slabs, so the order of
    // the products inside a slab is immaterial for rounding.)  With `dma` the pieces of the slab two ahead are dealt
    // out between the MFMAs as well.
    auto mfma = [&](auto dma_tag, auto il_tag, const Frags& f, uint32_t dma_stage, Frags& nxt) {
        constexpr bool dma = decltype(dma_tag)::value;
        constexpr bool interleave = decltype(il_tag)::value;
        constexpr int PA[3] = {1, 0, 0};
        constexpr int PB[3] = {0, 0, 1};
        constexpr int NM = 3 * TM * TN;
        constexpr int NTOP = 2 * TM * TN;   // MFMAs that need plane 0 only
        constexpr int NSTEP = 4 * TN;       // low steps
        constexpr int EVERY = NM / NLOAD > 0 ? NM / NLOAD : 1;
        if (dma) issue_begin(dma_stage);
        if (interleave) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f16x8 b = __builtin_bit_cast(
                        f16x8, u32x4{pl[PB[c]][j][0], pl[PB[c]][j][1], pl[PB[c]][j][2], pl[PB[c]][j][3]});
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.af[PA[c]][i], b, acc[i][j], 0, 0, 0);
                    const int idx = (c * TM + i) * TN + j;
                    if (interleave && idx < NTOP) {
#pragma unroll
                        for (int st = idx * NSTEP / NTOP; st < (idx + 1) * NSTEP / NTOP; ++st) low_step(f, st / 4, st % 4);
                    }
                    if (interleave) {   // the LDS reads of the next slab (into the other register set)
#pragma unroll
                        for (int k = idx * NREAD / NM; k < (idx + 1) * NREAD / NM; ++k) read_step(nxt, k);
                    }
                    if (dma && (idx + 1) % EVERY == 0 && (idx + 1) / EVERY <= NLOAD) issue_piece((idx + 1) / EVERY - 1);
                    if (interleave) __builtin_amdgcn_sched_barrier(0);
                }
        if (dma) {
#pragma unroll
            for (int k = NM / EVERY; k < NLOAD; ++k) issue_piece(k);
        }
    };

    // tooling (tools/prio_x6.py): dbg bit 5 = static priority for the waves in odd hardware wave slots, bit 6 = raised
    // priority around every MFMA block
    if ((p.dbg & 32) && wave_uniform((int)(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11)) & 1u)))
        __builtin_amdgcn_s_setprio(2);
    X6_PH_DECL;
    if (NG == 1) {
        // Free-running waves, software-pipelined through two register sets: in iteration t the wave multiplies slab t out
        // of registers while its LDS reads of slab t+1 and the DMA of slab t+3 are in flight, so neither the LDS round
        // trip nor the split sits in front of the matrix pipe.  Ring slots: t+1 (being read), t+2 (landing), and the
        // slot of slab t, free since every wave finished reading it before the barrier, takes slab t+3.
        issue(2 * STAGE);
        SSN_WAIT_VMCNT(2 * NLOAD);
        __builtin_amdgcn_s_barrier();
        read_frags(0, fr0);
        uint32_t s_cur = 0, s_n1 = STAGE, s_n2 = 2 * STAGE;   // ring slots of slabs t, t+1, t+2
        auto half = [&](Frags& cur, Frags& nxt) {
            SSN_WAIT_VMCNT(NLOAD);   // this wave's pieces of slab t+1 (slab t+2's may still be in flight)
            SSN_WAIT_LGKM0();        // ... and its reads of slab t are back
            X6_PH(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            X6_PH(1);
            read_begin(s_n1);
            top_plane(cur, false);
            X6_PH(3);
            mfma(std::true_type{}, std::true_type{}, cur, s_cur, nxt);
            X6_PH(5);
            const uint32_t o = s_cur;
            s_cur = s_n1;
            s_n1 = s_n2;
            s_n2 = o;
        };
        // two slabs per trip (the register sets swap roles); an odd slab count runs one all-zero slab at the end
        for (int t = 0; t < nslab; t += 2) {
            half(fr0, fr1);
            half(fr1, fr0);
        }
        SSN_WAIT_LGKM0();
    } else {
        uint32_t stage = 0;   // dword offset of the ring slot holding slab t
        for (int t = 0; t < nslab; ++t) {
            // slab t has landed once at most the NLOAD pieces of slab t+1 (real or out-of-range) are still in flight
            SSN_WAIT_VMCNT(NLOAD);
            X6_PH(0);
            __builtin_amdgcn_s_barrier();   // (a) every wave's share of slab t is visible, (b) slab t-1 is consumed
            __builtin_amdgcn_sched_barrier(0);
            X6_PH(1);
            const uint32_t dst = stage == 0 ? 2 * STAGE : stage - STAGE;   // ring slot of slab t-1, refilled with slab t+2
            if (grp == 0) {
                front(stage, true);
                X6_PH(3);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                X6_PH(4);
                mfma(std::true_type{}, std::false_type{}, fr0, dst, fr1);
                X6_PH(5);
            } else {
                if (t > 0)
                    mfma(std::true_type{}, std::false_type{}, fr0, dst, fr1);   // slab t-1, split during the previous half-phase
                else
                    issue(dst);
                X6_PH(3);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                X6_PH(4);
                front(stage, true);
                X6_PH(5);
            }
            stage = stage == (NSTAGE - 1) * STAGE ? 0 : stage + STAGE;
        }
        if (grp == 1) mfma(std::false_type{}, std::false_type{}, fr0, 0u, fr1);
    }
    SSN_WAIT_VMCNT(0);   // the out-of-range tail pieces still write (zeros) into the ring the epilogue is about to reuse
    if (p.trace) tr2 = __builtin_readcyclecounter();

    // ---- epilogue: BN affine + ReLU (forward), or accumulate + fused ReLU/BN backward (dgrad) ----
    __syncthreads();
    float* ch = reinterpret_cast<float*>(lds);
    epi_stage_channels<BM, NT>(ch, p.scale, p.shift, p.mask_scale, m0, p.M, tid, inv, p.relu, p.raw_from, p.row_split,
                               p.row_gap);
    __syncthreads();
    EpiArgs e;
    e.y = p.y;
    e.mask_y = p.mask_y;
    e.y_bytes = p.y_bytes;
    e.mask_bytes = p.mask_bytes;
    e.howo4 = (uint32_t)(p.sub_HW ? p.sub_HW : p.Ho * p.Wo) * 4u;
    e.M = p.M;
    e.relu = p.relu;
    e.accumulate = p.accumulate;
    e.amax = p.y_amax;
    e.amax2 = p.y_amax2;
    e.row_split = p.row_split;
    e.row_gap = p.row_gap;
    uint32_t yoff[TN], moff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pp = p0 + grp * BNG + (wn * TN + j) * 32 + li;
        uint32_t n, hw;
        fd_divmod((uint32_t)(pp < p.P ? pp : 0), p.div_hw, n, hw);
        const bool exists = pp < p.P && hw < (uint32_t)p.hw_real;
        if (p.sub_HW) {
            uint32_t u, v;
            fd_divmod(hw, p.div_w, u, v);
            hw = (2u * u + (uint32_t)p.sub_a) * (uint32_t)p.sub_W + 2u * v + (uint32_t)p.sub_b;
        }
        const uint32_t row0 = (uint32_t)(m0 + 4 * lh) * e.howo4 + hw * 4u;
        yoff[j] = exists ? (uint32_t)((long)n * p.y_img_stride * 4) + row0 : EPI_OOB;
        moff[j] = exists ? (uint32_t)((long)n * p.mask_img_stride * 4) + row0 : EPI_OOB;
    }
    conv_epilogue<TM, TN, BM>(acc, ch, e, yoff, moff, wm * TM * 32, lh, m0);
#ifdef X6_PHASE_TRACE
    if (p.trace && lane == 0 && (wave & 3) == 0) {
        unsigned long long* t = p.trace + (size_t)blockIdx.x * 32 + 8 + 8 * (wave >> 2);
        for (int k = 0; k < 6; ++k) t[k] = ph_[k];
    }
#endif
    if (p.trace && tid == 0) {
        unsigned long long* t = p.trace + (size_t)blockIdx.x * 32;
        t[0] = tr0;
        t[1] = tr1;
        t[2] = tr2;
        t[3] = __builtin_readcyclecounter();
        t[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID (wave/simd/cu/sh/se bits)
        t[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
    }
}
#undef X6_DMA_B128

// bytes a 16-byte (WIDE) activation load may reach in front of / behind the tile's own pixels: the largest tap
// displacement of a same-size stride-1 convolution
// host side, before a launch with 16-byte activation loads: planes that are not a multiple of 4 pixels are enumerated padded
static inline void x6_pad_enumeration(X6Args& a) {
    const int hw = a.Ho * a.Wo, hwp = (hw + 3) / 4 * 4;
    a.hw_real = hw;
    if (hwp != hw) {
        a.P = a.N * hwp;
        a.div_hw = make_fastdiv((uint32_t)hwp);
    }
}
static inline int x6_reach_bytes(int pad_h, int pad_w, int kh, int kw, int W) {
    const int rh = pad_h > kh - 1 - pad_h ? pad_h : kh - 1 - pad_h;
    const int rw = pad_w > kw - 1 - pad_w ? pad_w : kw - 1 - pad_w;
    return (rh * W + rw) * 4;
}

// dwords of the packed rows of a split weight: [ceil(C / 16) * kk slabs][M rows][APITCH] ...
static inline long x6_row_dwords_kk(int Cout, int Cin, int kk, int transposed) {
    const int M = transposed ? Cin : Cout, C = transposed ? Cout : Cin;
    return (long)((C + 15) / 16) * kk * M * APITCH;
}
// ... and of the whole packed operand (rows + amax tail)
static inline long x6_packed_dwords_kk(int Cout, int Cin, int kk, int transposed) {
    return x6_row_dwords_kk(Cout, Cin, kk, transposed) + ATAIL;
}

}  // namespace x6
