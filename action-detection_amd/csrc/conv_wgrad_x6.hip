// Weight (and bias) gradient of the stride-1 1x1 / 3x3 backbone convolutions on the f16 matrix cores with
// fp32-class accuracy (the split scheme of conv_x6.hip: every fp32 operand scaled by a per-tensor power of two and
// split into two f16 terms, three partial products per k16 step accumulated in fp32; "x6" is the family's historical
// name), gfx950.
//
// Same contract as conv_wgrad.hip (the cuDNN wgrad behind loss.backward(), /root/reference/ssn_train.py:236):
//   dW[co][kk] = sum_p G[co][p] * X[kk][p]        kk = (ci, r, s),  p = (n, ho, wo)
//   db[co]     = sum_p G[co][p]                   (summed in fp32 from the unsplit values)
// split-K over the pixel range, partial slabs reduced in a fixed order by the shared reduce kernel (deterministic).
//
// What is different from the f32 kernel, and why:
//  * The reduction index (pixels) is the contiguous axis of BOTH operands in NCHW, so every global access is a
//    16-byte load of 4 consecutive pixels.  The texture path accepts one wave instruction per ~16 cycles whatever
//    its width; the dword-per-pixel loads of the f32 kernel keep it as busy as the (64-cycle) f32 MFMAs, and would
//    starve a 32-cycle bf16 MFMA.  Address arithmetic is linear in the pixel index for stride 1, so a 4-pixel group
//    is one load even when it runs over the end of an image row; the taps that fall on padding are zeroed in
//    registers afterwards (two compares per pixel and row).  Loads of the shifted taps reach up to (W+1) floats in
//    front of x: the caller guarantees 256 readable bytes there (x_guard_bytes, see include/ssn_hip.h).
//  * Both operands are activations, so both are scaled and split on the fly: once per element, by the thread that
//    stages it (12 VALU per 4 pixels), written to LDS as two f16 planes ([operand][plane][row][16 k]: 32-byte rows, the
//    two 16-byte halves of a row swapped in every other group of 8 rows).  The scales come from the amax slots of the
//    two tensors (g_amax, x_amax; ssn_common.h), their reciprocal is applied when the partial slab is stored.  With that image every MFMA operand is one
//    conflict-free ds_read_b128 AND the 8-byte plane stores of the staging threads (4 rows x 4 pixel groups per
//    16-lane store group = 32 consecutive dwords) are conflict-free too -- the 112-byte row pitch of the first version
//    put a third of the store cycles into bank conflicts (profiles/r1_pmc_summary_x6.json).
//  * Loads run two chunks (2 x 16 pixels) ahead of the MFMAs in two alternating register sets: one chunk of MFMAs
//    (~0.8k cycles per wave) does not cover a global round trip.
#include "ssn_common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct WgX6Args {
    const float* g;  // [N][..Cout..][H][W] channel-slice base (stride 1, same-size convolution)
    const float* x;  // [N][..Cin..][H][W] channel-slice base
    float* part;     // [splits][M][ldp]
    const float* g_amax;   // amax slots of the tensors g and x belong to (required)
    const float* x_amax;
    const float* g_amax2;         // second region of g's tensor (rows behind the gap), or nullptr: the larger of both counts
    int g_row_split, g_row_gap;   // rows m >= g_row_split of G sit g_row_gap channels further up its tensor (fused block-input launch)
    uint32_t guard;               // readable bytes in front of x the shifted taps may reach into (multiple of 256)
    int N, Cin, H, W;
    long x_img_stride;
    int M;
    long g_img_stride;
    int K;    // Cin*KS*KS
    int ldp;  // K + 1 (last column = bias gradient)
    int P;    // N * HWp pixel slots
    int HWp;  // pixel slots per image: H*W rounded up to a multiple of 4 (PAD: the slots past H*W hold zeros)
    int pad;                       // square taps (KS > 0)
    int kh, kw, pad_h, pad_w;      // runtime taps (KS == 0): rectangular layers
    int splits, chunks_per_split;  // chunk = 16 pixels
    int n_mtiles, n_ktiles;
    uint32_t g_bytes, x_bytes;
    FastDiv div_hw, div_w, div_tiles, div_kt;
};

constexpr int CP = 16;          // pixels per chunk = one f16 MFMA k-step
constexpr int ROW_DW = 8;       // LDS row of one plane: 16 f16 = 32 B
constexpr uint32_t OOB = 0x80000000u;
constexpr uint32_t GUARD = 256u;   // the default (3x3 taps on rows of <= 63 pixels); wider reaches use a larger multiple

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// (operand scaling and split: f16_scale_of / f16_split2_pair of ssn_common.h)

// KS = 0: kh x kw taps with per-axis padding taken from the arguments (the 5x5 / 1x7 / 7x1 / 1x3 / 3x1 layers of
// Inception-v3): the tap geometry only enters the per-row constants in front of the loop.
// tiles of 8+ MFMA tiles per wave run one wave per SIMD (up to 512 VGPRs): the elements to split per MFMA drop with
// the tile size, which moves the kernel from VALU-bound towards matrix-bound
template <int KS, bool PAD, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256, TM * TN >= 8 ? 1 : 2) void wgrad_x6_kernel(WgX6Args p) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int NAR = (BM + 63) / 64;   // G rows per thread (thread = one row x 4 pixels of a chunk)
    constexpr int NBR = (BN + 63) / 64;   // X rows per thread
    const int KW_ = KS ? KS : p.kw;
    const int KK = KS ? KS * KS : p.kh * p.kw;
    const int PH = KS ? p.pad : p.pad_h, PW = KS ? p.pad : p.pad_w;
    constexpr int STAGE = 2 * (BM + BN) * ROW_DW;   // [A plane 0..1][B plane 0..1], rows x 8 dwords each
    static_assert(WM * WN == 4, "4 waves per workgroup");

    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const uint32_t tiles = (uint32_t)p.n_mtiles * (uint32_t)p.n_ktiles;
    const uint32_t nblk = tiles * (uint32_t)p.splits;
    const uint32_t logical = xcd_remap(blockIdx.x, nblk);
    uint32_t z, tile, mt, kt;
    fd_divmod(logical, p.div_tiles, z, tile);
    fd_divmod(tile, p.div_kt, mt, kt);
    const int m0 = (int)mt * BM;
    const int kk0 = (int)kt * BN;

    const float sg = f16_scale_of(p.g_amax2 ? fmaxf(*p.g_amax, *p.g_amax2) : *p.g_amax);   // power-of-two operand scales
    const float sx = f16_scale_of(*p.x_amax);
    const float inv = 1.f / (sg * sx);

    const int pxg = tid & 3;       // which 4-pixel group of the chunk
    const int row0 = tid >> 2;     // 0..63; this thread stages rows row0 + 64 i
    const int HW = p.H * p.W;

    // ---- loop-invariant row constants ----
    uint32_t a_const[NAR];   // byte offset of row m inside an image of G, or OOB
#pragma unroll
    for (int i = 0; i < NAR; ++i) {
        const int r = row0 + 64 * i, m = m0 + r;
        a_const[i] = (r < BM && m < p.M) ? (uint32_t)((m + (m >= p.g_row_split ? p.g_row_gap : 0)) * HW) * 4u : OOB;
    }
    uint32_t b_const[NBR];   // byte offset of (c, r, s) relative to the pixel (guard included), or OOB
    int b_dh[NBR], b_dw[NBR];   // tap displacement (r - pad, s - pad)
#pragma unroll
    for (int i = 0; i < NBR; ++i) {
        const int r = row0 + 64 * i, kk = kk0 + r;
        int c = kk, tap = 0;
        if (KS != 1) {
            c = kk / KK;
            tap = kk - c * KK;
        }
        const int tr = tap / KW_, ts = tap - tr * KW_;
        b_dh[i] = tr - PH;
        b_dw[i] = ts - PW;
        b_const[i] = (r < BN && kk < p.K) ? (uint32_t)((c * HW + b_dh[i] * p.W + b_dw[i]) * 4 + (int)p.guard) : OOB;
    }
    const __amdgpu_buffer_rsrc_t grsrc = wg_rsrc(p.g, p.g_bytes);
    const __amdgpu_buffer_rsrc_t xrsrc = wg_rsrc(reinterpret_cast<const char*>(p.x) - p.guard, p.x_bytes + p.guard);

    struct Staged {
        u32x4 a[NAR], b[NBR];
        uint32_t ho, wo;   // first pixel of the group
        uint32_t hw;
    };
    const int total_chunks = (p.P + CP - 1) / CP;
    const int c_begin = (int)z * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > total_chunks) c_end = total_chunks;
    const int nch = c_end - c_begin;
    // chunks past the end of this workgroup's split load as zeros (every offset out of range): the pipeline below
    // runs without "is there another chunk" branches and an odd chunk count ends with one all-zero chunk
    auto load_chunk = [&](int chunk, Staged& s) {
        const int pp = chunk * CP + pxg * 4;
        const bool valid = pp < p.P && chunk < c_end;
        uint32_t n, hw;
        fd_divmod((uint32_t)(valid ? pp : 0), p.div_hw, n, hw);
        fd_divmod(hw, p.div_w, s.ho, s.wo);
        s.hw = hw;
        const uint32_t abase = (uint32_t)((long)n * p.g_img_stride * 4) + hw * 4u;
        const uint32_t bbase = (uint32_t)((long)n * p.x_img_stride * 4) + hw * 4u;
#pragma unroll
        for (int i = 0; i < NAR; ++i)
            s.a[i] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, (valid && a_const[i] != OOB) ? abase + a_const[i] : OOB,
                                                           0, 0);
#pragma unroll
        for (int i = 0; i < NBR; ++i)
            s.b[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (valid && b_const[i] != OOB) ? bbase + b_const[i] : OOB,
                                                           0, 0);
    };

    float rowsum[NAR];
#pragma unroll
    for (int i = 0; i < NAR; ++i) rowsum[i] = 0.f;
    const bool do_bias = (kt == 0);
    // a wave stages 16 consecutive rows, so "this staging row exists" is wave-uniform (BM, BN are multiples of 32)
    bool a_live[NAR], b_live[NBR];
#pragma unroll
    for (int i = 0; i < NAR; ++i) a_live[i] = wave_uniform((row0 & ~15) + 64 * i) < BM;
#pragma unroll
    for (int i = 0; i < NBR; ++i) b_live[i] = wave_uniform((row0 & ~15) + 64 * i) < BN;

    // Split of a staged chunk into bf16 planes + LDS store, cut into 2 * (NAR + NBR) steps of ~12-16 VALU (a k-pair of
    // one staged row; the second step of a row also writes its three 8-byte plane pieces), so that the steps can be
    // dealt out behind the MFMAs of the chunk that is being multiplied: the kernel is VALU-bound (~10 VALU per MFMA),
    // and with the split running AFTER the MFMAs the two co-resident waves of a SIMD fall into lock-step -- both
    // multiply, then both split -- which leaves the matrix pipe idle two thirds of the time.
    constexpr int NROW = NAR + NBR, NSTEP = 2 * NROW;
    int hh[4], ww[4];          // coordinates of the 4 pixels of the staged group (a group may run over a row end)
    uint32_t plw[2][2];
    auto stage_begin = [&](const Staged& s) {
        if (KS != 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int w = (int)s.wo + e, h = (int)s.ho;
                if (w >= p.W) {
                    w -= p.W;
                    h += 1;
                }
                hh[e] = h;
                ww[e] = w;
            }
        }
    };
    auto stage_step = [&](const Staged& s, int buf, int st) {   // st is a compile-time constant at every call site
        const int row = st / 2, half = st % 2;
        const bool is_a = row < NAR;
        const int i = is_a ? row : row - NAR;
        if (is_a ? !a_live[i] : !b_live[i]) return;
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t u = is_a ? s.a[i][2 * half + e] : s.b[i][2 * half + e];
            v[e] = __builtin_bit_cast(float, u);
            if (!is_a && KS != 1) {
                const bool ok = ((unsigned)(hh[2 * half + e] + b_dh[i]) < (unsigned)p.H) &&
                                ((unsigned)(ww[2 * half + e] + b_dw[i]) < (unsigned)p.W);
                v[e] = ok ? v[e] : 0.f;
            }
            // planes that are not a multiple of 4 pixels: the last group of an image runs past the plane
            if (PAD) v[e] = (s.hw + (uint32_t)(2 * half + e) < (uint32_t)HW) ? v[e] : 0.f;
        }
        if (is_a && do_bias) rowsum[i] += v[0] + v[1];
        f16_split2_pair(v[0], v[1], is_a ? sg : sx, plw[0][half], plw[1][half]);
        if (half == 1) {
            // 16-byte half (pxg >> 1), swapped in odd groups of 8 rows (row0 and row0 + 64 i share bit 3)
            uint32_t* dst = lds + buf * STAGE + (is_a ? 0 : 2 * BM * ROW_DW) + (row0 + 64 * i) * ROW_DW +
                            (((pxg >> 1) ^ ((row0 >> 3) & 1)) * 4) + (pxg & 1) * 2;
#pragma unroll
            for (int pn = 0; pn < 2; ++pn)
                *reinterpret_cast<uint2*>(dst + pn * (is_a ? BM : BN) * ROW_DW) = uint2{plw[pn][0], plw[pn][1]};
        }
    };
    auto store_chunk = [&](const Staged& s, int buf) {
        stage_begin(s);
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) stage_step(s, buf, st);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // multiply the chunk in LDS buffer `buf`; behind the MFMAs: split + store the staged chunk `st` into the other
    // buffer, then fetch chunk `next_chunk` into the register set that has just been drained
    auto compute = [&](int buf, Staged& st, int next_chunk) {
        const int half = (lh ^ ((li >> 3) & 1)) * 4;   // this lane's k-half of its row (tile rows are 32 apart: bit 3 = li's)
        const uint32_t* As = lds + buf * STAGE + (wm * TM * 32 + li) * ROW_DW + half;
        const uint32_t* Bs = lds + buf * STAGE + 2 * BM * ROW_DW + (wn * TN * 32 + li) * ROW_DW + half;
        f16x8 af[2][TM], bf[2][TN];
#pragma unroll
        for (int pn = 0; pn < 2; ++pn) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[pn][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(As + i * 32 * ROW_DW + pn * BM * ROW_DW));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[pn][j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(Bs + j * 32 * ROW_DW + pn * BN * ROW_DW));
        }
        stage_begin(st);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int PA[3] = {1, 0, 0};   // g_lo x_hi + g_hi x_lo + g_hi x_hi
        constexpr int PB[3] = {0, 1, 0};
        constexpr int NM = 3 * TM * TN;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[PA[c]][i], bf[PB[c]][j], acc[i][j], 0, 0, 0);
                    const int idx = (c * TM + i) * TN + j;
#pragma unroll
                    for (int k = idx * NSTEP / NM; k < (idx + 1) * NSTEP / NM; ++k) stage_step(st, buf ^ 1, k);
                    __builtin_amdgcn_sched_barrier(0);
                }
        load_chunk(next_chunk, st);
    };

    // chunk t lives in register set t & 1 from two iterations before it is multiplied until one iteration before
    Staged s0, s1;
    load_chunk(c_begin, s0);
    load_chunk(c_begin + 1, s1);
    store_chunk(s0, 0);
    load_chunk(c_begin + 2, s0);
    __syncthreads();
    for (int t = 0; t < nch; t += 2) {
        // even chunk t: LDS buffer 0; registers: s1 = chunk t+1 (-> buffer 1), then chunk t+3
        compute(0, s1, c_begin + t + 3);
        __syncthreads();
        // odd chunk t+1: LDS buffer 1; registers: s0 = chunk t+2 (-> buffer 0), then chunk t+4
        compute(1, s0, c_begin + t + 4);
        __syncthreads();
    }

    // ---- partial slab store: part[z][m][kk] (kk contiguous across lanes) ----
    float* out = p.part + (long)z * p.M * p.ldp;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int kk = kk0 + (wn * TN + j) * 32 + li;
        if (kk >= p.K) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.M) out[(long)m * p.ldp + kk] = acc[i][j][r] * inv;
            }
        }
    }
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < NAR; ++i) {
            float tot = rowsum[i];
            tot += __shfl_xor(tot, 1, 64);   // the 4 pixel groups of a row sit in 4 adjacent lanes
            tot += __shfl_xor(tot, 2, 64);
            const int r = row0 + 64 * i, m = m0 + r;
            if (pxg == 0 && r < BM && m < p.M) out[(long)m * p.ldp + p.K] = tot;
        }
    }
}

template <int KS, int WM, int WN, int TM, int TN>
int launch_wgx6(WgX6Args& a, hipStream_t stream) {
    const bool pad = a.HWp != a.H * a.W;
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.n_ktiles = (a.K + BN - 1) / BN;
    const unsigned tiles = (unsigned)a.n_mtiles * (unsigned)a.n_ktiles;
    a.div_tiles = make_fastdiv(tiles);
    a.div_kt = make_fastdiv((uint32_t)a.n_ktiles);
    if (pad)
        hipLaunchKernelGGL((wgrad_x6_kernel<KS, true, WM, WN, TM, TN>), dim3(tiles * (unsigned)a.splits), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((wgrad_x6_kernel<KS, false, WM, WN, TM, TN>), dim3(tiles * (unsigned)a.splits), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("conv_wgrad_x6");
    return SSN_OK;
}

// tile configs as in conv_wgrad.hip:
//   0: 64(co) x 64(kk)   1: 32 x 128   2: 128 x 128   3: 64 x 128   4: 96 x 128   5: 64 x 128 (waves along kk)   6: 128 x 64
// and larger ones (fewer elements to split per MFMA: the kernel is VALU-bound):   7: 96 x 256   8: 192 x 128
// one workgroup per CU, one wave per SIMD:   9: 128 x 256   10: 160 x 256   11: 192 x 256
constexpr int NCFG = 12;
const int kBM[NCFG] = {64, 32, 128, 64, 96, 64, 128, 96, 192, 128, 160, 192};
const int kBN[NCFG] = {64, 128, 128, 128, 128, 128, 64, 256, 128, 256, 256, 256};

template <int KS>
int launch_wgx6_tile(WgX6Args& a, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_wgx6<KS, 2, 2, 1, 1>(a, stream);
        case 1: return launch_wgx6<KS, 1, 4, 1, 1>(a, stream);
        case 2: return launch_wgx6<KS, 2, 2, 2, 2>(a, stream);
        case 3: return launch_wgx6<KS, 2, 2, 1, 2>(a, stream);
        case 4: return launch_wgx6<KS, 1, 4, 3, 1>(a, stream);
        case 5: return launch_wgx6<KS, 1, 4, 2, 1>(a, stream);
        case 6: return launch_wgx6<KS, 2, 2, 2, 1>(a, stream);
        case 7: return launch_wgx6<KS, 1, 4, 3, 2>(a, stream);
        case 8: return launch_wgx6<KS, 2, 2, 3, 2>(a, stream);
        case 9: return launch_wgx6<KS, 1, 4, 4, 2>(a, stream);
        case 10: return launch_wgx6<KS, 1, 4, 5, 2>(a, stream);
        case 11: return launch_wgx6<KS, 2, 2, 3, 4>(a, stream);
    }
    ssn_set_error("conv_wgrad_x6: unknown tile config %d", cfg);
    return SSN_ERR_ARG;
}

// 4x4 taps (the space-to-depth form of the 7x7 / stride-2 stem convolution, 64 x 192 weights): a few small tiles only
int launch_wgx6_tile4(WgX6Args& a, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_wgx6<4, 2, 2, 1, 1>(a, stream);
        case 3: return launch_wgx6<4, 2, 2, 1, 2>(a, stream);
        case 6: return launch_wgx6<4, 2, 2, 2, 1>(a, stream);
        default: return launch_wgx6<4, 1, 4, 2, 1>(a, stream);   // 5: 64 x 128, waves along kk
    }
}

// runtime taps (rectangular layers): the mid-size tiles
int launch_wgx6_tile_rt(WgX6Args& a, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_wgx6<0, 2, 2, 1, 1>(a, stream);
        case 2: return launch_wgx6<0, 2, 2, 2, 2>(a, stream);
        case 3: return launch_wgx6<0, 2, 2, 1, 2>(a, stream);
        case 6: return launch_wgx6<0, 2, 2, 2, 1>(a, stream);
        default: return launch_wgx6<0, 1, 4, 2, 1>(a, stream);   // 5: 64 x 128, waves along kk
    }
}
int rt_tile(int cfg, int M, int K);

int pick_tile(int M, int K) {
    double best = 1e300;
    int bc = 0;
    for (int c = 0; c < 7; ++c) {   // (the two large tiles are only taken when the autotuner picks them)
        const double padded = (double)((M + kBM[c] - 1) / kBM[c]) * kBM[c] * (double)((K + kBN[c] - 1) / kBN[c]) * kBN[c];
        const double reuse = (kBM[c] * kBN[c] >= 128 * 64) ? 1.0 : 1.12;
        if (padded * reuse < best) {
            best = padded * reuse;
            bc = c;
        }
    }
    return bc;
}

// co-resident workgroups per CU of each tile config (LDS 2 x (BM + BN) x 64 B, VGPRs as compiled)
const int kOcc[NCFG] = {5, 4, 2, 3, 2, 3, 3, 2, 2, 1, 1, 1};

void plan(int M, int K, long P, int cfg, int* splits, int* chunks_per_split) {
    const long tiles = (long)((M + kBM[cfg] - 1) / kBM[cfg]) * ((K + kBN[cfg] - 1) / kBN[cfg]);
    // reduce pass: ~8 bytes per output element and split at ~2 TB/s, in units of a ~0.6 us chunk
    const long chunks = (P + CP - 1) / CP;
    plan_split_k(tiles, chunks, kOcc[cfg], 32, 24, 0.02 + (double)M * K * 6.7e-6, splits, chunks_per_split);
    if (*chunks_per_split & 1) {   // the kernel's pipeline runs two chunks per trip
        ++*chunks_per_split;
        *splits = (int)((chunks + *chunks_per_split - 1) / *chunks_per_split);
    }
}

int rt_tile(int cfg, int M, int K) {
    if (cfg < 0) cfg = pick_tile(M, K);
    return (cfg == 0 || cfg == 2 || cfg == 3 || cfg == 6) ? cfg : 5;
}

}  // namespace

// reduce kernel shared with conv_wgrad.hip
extern "C" int ssn_wgrad_reduce(const float* part, float* dw, float* db, int M, int K, int splits, hipStream_t stream);

extern "C" long ssn_conv_wgrad_x6_workspace_bytes(int N, int Cin, int Cout, int H, int W, int ksize, int tile_cfg) {
    const int K = Cin * ksize * ksize;
    int cfg = (tile_cfg >= 0 && tile_cfg < NCFG) ? tile_cfg : pick_tile(Cout, K);
    if (ksize == 4 && cfg != 0 && cfg != 3 && cfg != 6) cfg = 5;      // the tiles instantiated for 4x4 taps
    int splits, cps;
    plan(Cout, K, (long)N * ((H * W + 3) / 4 * 4), cfg, &splits, &cps);
    return (long)splits * Cout * (K + 1) * (long)sizeof(float);
}

// Stride-1, same-size (2*pad == ksize-1) convolutions; x_guard_bytes >= 256 (see header).  Planes whose H*W is not a
// multiple of 4 (the 7x7 stage) are enumerated in groups of 4 pixel slots per image, the slots past the plane zeroed.
extern "C" long ssn_conv_wgrad_x6_rect_workspace_bytes(int N, int Cin, int Cout, int H, int W, int kh, int kw, int tile_cfg) {
    const int K = Cin * kh * kw;
    const int cfg = rt_tile(tile_cfg < NCFG ? tile_cfg : -1, Cout, K);
    int splits, cps;
    plan(Cout, K, (long)N * ((H * W + 3) / 4 * 4), cfg, &splits, &cps);
    return (long)splits * Cout * (K + 1) * (long)sizeof(float);
}

static int wgrad_x6_impl(const float* g, const float* x, float* dw, float* db, int N, int Cin, int H, int W,
                         long x_img_stride, int Cout, long g_img_stride, int ksize, int pad, int kh, int kw, int pad_h,
                         int pad_w, int x_guard_bytes, void* workspace, long ws_bytes, int tile_cfg, const float* g_amax,
                         const float* x_amax, int g_row_split, int g_row_gap, const float* g_amax2, hipStream_t stream);

// Weight gradient of a stride-1, same-size layer with kh x kw taps (5x5, 1x7, 7x1, 1x3, 3x1: the layers
// ssn_conv_x6_fwd_rect runs forward); dw [Cout][Cin][kh][kw].  x needs (pad_h * W + pad_w) * 4 readable bytes in front of
// it, rounded up to a multiple of 256.  Other arguments as ssn_conv_wgrad_x6.
extern "C" int ssn_conv_wgrad_x6_rect(const float* g, const float* x, float* dw, float* db, int N, int Cin, int H, int W,
                                      long x_img_stride, int Cout, long g_img_stride, int kh, int kw, int pad_h, int pad_w,
                                      int x_guard_bytes, void* workspace, long ws_bytes, int tile_cfg, const float* g_amax,
                                      const float* x_amax, hipStream_t stream) {
    // (pad 0, 0: an unpadded convolution whose output gradient the caller embedded into planes of the input's size)
    SSN_CHECK_ARG(kh >= 1 && kw >= 1 && kh * kw <= 49 && ((2 * pad_h == kh - 1 && 2 * pad_w == kw - 1) || (pad_h == 0 && pad_w == 0)),
                  "conv wgrad x6 rect: only same-grid stride-1 convolutions (%dx%d taps, pad %d,%d)", kh, kw, pad_h, pad_w);
    return wgrad_x6_impl(g, x, dw, db, N, Cin, H, W, x_img_stride, Cout, g_img_stride, 0, 0, kh, kw, pad_h, pad_w,
                         x_guard_bytes, workspace, ws_bytes, tile_cfg, g_amax, x_amax, 0, 0, nullptr, stream);
}

extern "C" int ssn_conv_wgrad_x6(const float* g, const float* x, float* dw, float* db, int N, int Cin, int H, int W,
                                 long x_img_stride, int Cout, long g_img_stride, int ksize, int pad, int x_guard_bytes,
                                 void* workspace, long ws_bytes, int tile_cfg, const float* g_amax, const float* x_amax,
                                 int g_row_split, int g_row_gap, const float* g_amax2, hipStream_t stream) {
    SSN_CHECK_ARG(ksize == 1 || ksize == 3 || ksize == 4, "conv wgrad x6: ksize %d unsupported", ksize);
    // same-size stride-1 convolutions; ksize 4 = the space-to-depth stem: taps -pad .. ksize-1-pad with pad = 2 (one tap less
    // behind the pixel than in front of it)
    SSN_CHECK_ARG(2 * pad == ksize - 1 || (ksize == 4 && pad == 2),
                  "conv wgrad x6: only same-size stride-1 convolutions (pad %d, ksize %d)", pad, ksize);
    return wgrad_x6_impl(g, x, dw, db, N, Cin, H, W, x_img_stride, Cout, g_img_stride, ksize, pad, ksize, ksize, pad, pad,
                         x_guard_bytes, workspace, ws_bytes, tile_cfg, g_amax, x_amax, g_row_split, g_row_gap, g_amax2, stream);
}

static int wgrad_x6_impl(const float* g, const float* x, float* dw, float* db, int N, int Cin, int H, int W,
                         long x_img_stride, int Cout, long g_img_stride, int ksize, int pad, int kh, int kw, int pad_h,
                         int pad_w, int x_guard_bytes, void* workspace, long ws_bytes, int tile_cfg, const float* g_amax,
                         const float* x_amax, int g_row_split, int g_row_gap, const float* g_amax2, hipStream_t stream) {
    SSN_CHECK_ARG(g && x && dw && workspace, "conv wgrad x6: null pointer");
    SSN_CHECK_ARG(g_amax && x_amax, "conv wgrad x6: the amax slots of both operand tensors are required");
    const uint32_t guard = (uint32_t)(((pad_h * W + pad_w) * 4 + (int)GUARD - 1) / (int)GUARD * (int)GUARD);
    SSN_CHECK_ARG(x_guard_bytes >= (int)(guard ? guard : GUARD),
                  "conv wgrad x6: needs %u readable bytes in front of x (got %d): the taps in front of a pixel reach that far", guard,
                  x_guard_bytes);
    WgX6Args a;
    a.g = g;
    a.x = x;
    a.part = (float*)workspace;
    a.g_amax = g_amax;
    a.x_amax = x_amax;
    a.g_amax2 = g_amax2;
    a.guard = guard ? guard : GUARD;
    SSN_CHECK_ARG(g_row_gap >= 0 && (g_row_gap == 0 || (g_row_split > 0 && g_row_split < Cout)), "conv wgrad x6: bad row split");
    a.g_row_split = g_row_gap ? g_row_split : 0x7fffffff;
    a.g_row_gap = g_row_gap;
    a.N = N;
    a.Cin = Cin;
    a.H = H;
    a.W = W;
    a.x_img_stride = x_img_stride;
    a.M = Cout;
    a.g_img_stride = g_img_stride;
    a.K = Cin * kh * kw;
    a.ldp = a.K + 1;
    a.HWp = (H * W + 3) / 4 * 4;
    a.P = N * a.HWp;
    a.pad = pad;
    a.kh = kh;
    a.kw = kw;
    a.pad_h = pad_h;
    a.pad_w = pad_w;
    a.div_hw = make_fastdiv((uint32_t)a.HWp);
    a.div_w = make_fastdiv((uint32_t)W);
    const long gb = ((long)(N - 1) * g_img_stride + (long)(Cout + g_row_gap) * H * W) * 4;
    const long xb = ((long)(N - 1) * x_img_stride + (long)Cin * H * W) * 4;
    SSN_CHECK_ARG(gb < (1l << 31) && xb < (1l << 31) - 512, "conv wgrad x6: operand larger than 2 GiB (buffer addressing)");
    a.g_bytes = (uint32_t)gb;
    a.x_bytes = (uint32_t)xb;
    SSN_CHECK_ARG(tile_cfg < NCFG, "conv wgrad x6: unknown tile config %d", tile_cfg);
    int cfg = tile_cfg >= 0 ? tile_cfg : pick_tile(Cout, a.K);
    if (ksize == 4 && cfg != 0 && cfg != 3 && cfg != 6) cfg = 5;      // the tiles instantiated for 4x4 taps
    if (ksize == 0) cfg = rt_tile(cfg, Cout, a.K);
    plan(Cout, a.K, a.P, cfg, &a.splits, &a.chunks_per_split);
    const long need = (long)a.splits * Cout * a.ldp * (long)sizeof(float);
    if (ws_bytes < need) {
        ssn_set_error("conv wgrad x6: workspace %ld < %ld bytes", ws_bytes, need);
        return SSN_ERR_WORKSPACE;
    }
    const int rc = ksize == 0 ? launch_wgx6_tile_rt(a, cfg, stream)
                 : ksize == 1 ? launch_wgx6_tile<1>(a, cfg, stream)
                 : (ksize == 3 ? launch_wgx6_tile<3>(a, cfg, stream) : launch_wgx6_tile4(a, cfg, stream));
    if (rc != SSN_OK) return rc;
    return ssn_wgrad_reduce(a.part, dw, db, Cout, a.K, a.splits, stream);
}
