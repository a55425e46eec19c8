// Implicit-GEMM convolution with fp32-class accuracy on the bf16 matrix cores ("x6" path), gfx950.
//
// Same role and same operands as conv_igemm.hip (forward conv + frozen-BN + ReLU, and dgrad, of the
// BN-Inception layers behind /root/reference/ssn_models.py:266,298), but the multiply runs on
// v_mfma_f32_32x32x16_bf16 instead of the 16x slower exact-f32 MFMA:
//
//   every fp32 operand x is split EXACTLY into three bf16 terms by truncation,
//        x = x1 + x2 + x3,   x1 = top 16 bits of x,  x2 = top 16 bits of (x - x1),  x3 = x - x1 - x2
//   (8 + 8 + 8 = 24 significand bits, so nothing is lost), and a*b is accumulated in fp32 from the six
//   partial products with i + j <= 4:  a3b1 + a1b3 + a2b2 + a2b1 + a1b2 + a1b1  (smallest first).
//   The three dropped products are <= 2^-24 |ab|; measured error against float64 equals that of the f32
//   MFMA (tools/proto/run_bf16x6.py: 2.4e-7 vs 3.0e-7 of sum|ab|), at 6/16 of its matrix-pipe time.
//
// Structure (what keeps the loop matrix-bound now that an MFMA is 32 cycles):
//  * K runs in slabs of 16 rows = one bf16 MFMA.  1x1: 16 channels.  3x3: 16 channels x ONE tap, taps
//    innermost -- all rows of a slab share the tap, so a thread needs one border-checked gather offset per
//    tap (9 registers, computed once) and the channel part of the address is a scalar:
//    `buffer_load_dword v, tapoff[tap], rsrc, soffset`.  No per-element address arithmetic in the loop.
//  * Weights are split and packed once per step (ssn_conv_pack_weights_x6*) into exactly the LDS image
//    (row = [3 planes][16 bf16], 112-byte pitch): the A tile is a contiguous block copied with b128 loads.
//  * Activations are split on the fly while staging the gathered slab to LDS: ~5.5 VALU per element
//    (and/sub for the residuals, one v_perm per bf16 pair), written as one ds_write_b128 per plane.
//  * LDS rows are k-contiguous for both operands, so every MFMA operand is one conflict-free ds_read_b128.
#include "ssn_common.h"

namespace {

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int PITCH_DW = 28;   // dwords per operand row in LDS and in the packed weights (96 B data + 16 B pad)
constexpr uint32_t OOB = 0x80000000u;

struct X6Args {
    const float* x;       // gather source (channel-slice base), fp32 NCHW
    const uint32_t* ap;   // packed split weights [nslab][M][PITCH_DW]
    float* y;
    const float* scale;
    const float* shift;
    int N, C, H, W;       // gather-source dims
    long x_img_stride;
    int M;
    int Ho, Wo;           // enumerated pixel grid
    long y_img_stride;
    int P;
    int pad;
    int relu, accumulate;
    const float* mask_y;
    const float* mask_scale;
    long mask_img_stride;
    int n_ptiles, n_mtiles, ngroups;   // ngroups = ceil(C / 16)
    uint32_t x_bytes, a_bytes;
    unsigned long long* trace;   // tooling only: per-block phase timestamps (tools/trace_x6.py), normally null
    int desync;   // first-round start stagger between co-resident workgroups, in units of 1024 cycles (0 = off)
    int dbg;   // tooling only (tools/ablate_x6.py): 1 no global loads, 2 no LDS stores/split, 4 no barrier
    FastDiv div_hw, div_w, div_mt;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// bf16 pair (k even -> low half, k odd -> high half) of the top 16 bits of two fp32 values
__device__ __forceinline__ uint32_t pack_hi16(uint32_t even, uint32_t odd) {
    return __builtin_amdgcn_perm(odd, even, 0x07060302u);
}
__device__ __forceinline__ float residual(float x) {
    return x - __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x) & 0xFFFF0000u);
}

template <int KS, int S, int MODE, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void conv_x6_kernel(X6Args p) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int TPP = 256 / BN;            // threads per pixel column
    constexpr int RPT = 16 / TPP;            // slab rows (channels) per thread: 4, 8 or 16
    constexpr int KK = KS * KS;
    constexpr int A_CH = BM * PITCH_DW / 4;  // 16-byte chunks of one weight tile
    constexpr int NA = (A_CH + 255) / 256;
    constexpr int STAGE = (BM + BN) * PITCH_DW;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BN == 64 || BN == 128 || BN == 256, "BN in {64,128,256}");

    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const uint32_t nblk = (uint32_t)p.n_ptiles * (uint32_t)p.n_mtiles;
    const uint32_t logical = xcd_remap(blockIdx.x, nblk);
    uint32_t ptile, mtile;
    fd_divmod(logical, p.div_mt, ptile, mtile);
    const int m0 = (int)mtile * BM;
    const int p0 = (int)ptile * BN;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
    if (p.trace) tr0 = __builtin_readcyclecounter();
    // Workgroups of equal work started together stay in lock-step for the whole launch, so every CU would run its
    // store-bound epilogues at the same moment with the matrix pipe idle.  Delay the second (third) workgroup that
    // lands on each CU once, in the first round; the phase shift then persists.
    if (p.desync > 0 && blockIdx.x >= 256u && blockIdx.x < 768u) {
        const int n = (int)(blockIdx.x >> 8) * p.desync;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
    }

    // ---- per-thread gather state: one border-checked byte offset per tap ----
    const int gcol = tid % BN;
    const int gq = wave_uniform(tid / BN);    // which RPT-row group of the slab this thread stages
    uint32_t tapoff[KK];
    {
        const int gp = p0 + gcol;
        const bool gvalid = gp < p.P;
        uint32_t n, hw, ho, wo;
        fd_divmod((uint32_t)(gvalid ? gp : 0), p.div_hw, n, hw);
        fd_divmod(hw, p.div_w, ho, wo);
        const uint32_t gbase = (uint32_t)((long)n * p.x_img_stride * 4);
#pragma unroll
        for (int t = 0; t < KK; ++t) {
            const int r = t / KS, s = t - r * KS;
            int hi, wi;
            if (MODE == MODE_FWD) {
                hi = (int)ho * S - p.pad + r;
                wi = (int)wo * S - p.pad + s;
            } else {
                hi = (int)ho + p.pad - r;
                wi = (int)wo + p.pad - s;
            }
            const bool ok = gvalid && ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W);
            tapoff[t] = ok ? gbase + (uint32_t)(hi * p.W + wi) * 4u : OOB;
        }
    }
    const uint32_t hw_bytes = (uint32_t)(p.H * p.W) * 4u;
    const uint32_t row0_off = (uint32_t)(gq * RPT) * hw_bytes;   // first channel row of this thread inside a group

    // ---- weight tile: one contiguous block of the packed operand per slab ----
    uint32_t aoff[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        const int f = tid + 256 * q;
        const int row = (f * 4) / PITCH_DW;
        const bool ok = (f < A_CH) && (m0 + row < p.M);
        aoff[q] = ok ? (uint32_t)(m0 * PITCH_DW) * 4u + (uint32_t)f * 16u : OOB;
    }
    const __amdgpu_buffer_rsrc_t xrsrc = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(p.ap, p.a_bytes);
    const uint32_t a_step = (uint32_t)(p.M * PITCH_DW) * 4u;
    const int c_last = p.C - (p.ngroups - 1) * 16;   // channels in the last group (16 when exact)

    float breg[RPT];
    u32x4 areg[NA];
    // slab (g, tap): taps innermost; `tap` is a compile-time constant at every call site (unrolled tap loop), so
    // tapoff[] stays in registers
    auto load_slab = [&](int g, int tap) {
        const uint32_t vo = tapoff[tap];
        const uint32_t so = (uint32_t)g * 16u * hw_bytes + row0_off;
        const bool tail = (g == p.ngroups - 1) && (c_last != 16);
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            uint32_t v = vo;
            if (tail && (gq * RPT + i >= c_last)) v = OOB;   // channels past the end (never in BN-Inception)
            breg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, v, so + (uint32_t)i * hw_bytes, 0));
        }
        const uint32_t aso = (uint32_t)(g * KK + tap) * a_step;
#pragma unroll
        for (int q = 0; q < NA; ++q) areg[q] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff[q], aso, 0);
    };
    // split the staged fp32 rows into 3 bf16 planes and write them (k-contiguous) next to the weights
    auto store_slab = [&](int buf) {
        uint32_t* As = lds + buf * STAGE;
        uint32_t* Bs = As + BM * PITCH_DW + gcol * PITCH_DW + gq * (RPT / 2);
#pragma unroll
        for (int q = 0; q < NA; ++q)
            if (tid + 256 * q < A_CH) *reinterpret_cast<u32x4*>(As + (tid + 256 * q) * 4) = areg[q];
        uint32_t pl[3][RPT / 2];
#pragma unroll
        for (int i = 0; i < RPT / 2; ++i) {
            const float x0 = breg[2 * i], x1 = breg[2 * i + 1];
            const float r0 = residual(x0), r1 = residual(x1);
            const float s0 = residual(r0), s1 = residual(r1);
            pl[0][i] = pack_hi16(__builtin_bit_cast(uint32_t, x0), __builtin_bit_cast(uint32_t, x1));
            pl[1][i] = pack_hi16(__builtin_bit_cast(uint32_t, r0), __builtin_bit_cast(uint32_t, r1));
            pl[2][i] = pack_hi16(__builtin_bit_cast(uint32_t, s0), __builtin_bit_cast(uint32_t, s1));
        }
#pragma unroll
        for (int pn = 0; pn < 3; ++pn) {
            if (RPT == 4) {
                *reinterpret_cast<uint2*>(Bs + pn * 8) = uint2{pl[pn][0], pl[pn][1]};
            } else {
#pragma unroll
                for (int v = 0; v < RPT / 8; ++v)
                    *reinterpret_cast<u32x4*>(Bs + pn * 8 + 4 * v) =
                        u32x4{pl[pn][4 * v], pl[pn][4 * v + 1], pl[pn][4 * v + 2], pl[pn][4 * v + 3]};
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_slab(0, 0);
    store_slab(0);
    __syncthreads();
    if (p.trace) tr1 = __builtin_readcyclecounter();

    int buf = 0;
    for (int g = 0; g < p.ngroups; ++g) {
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            const bool more = (tap + 1 < KK) || (g + 1 < p.ngroups);
            if (!(p.dbg & 1)) {
                if (tap + 1 < KK)
                    load_slab(g, tap + 1);
                else if (g + 1 < p.ngroups)
                    load_slab(g + 1, 0);
            }

            const uint32_t* As = lds + buf * STAGE + (wm * TM * 32 + li) * PITCH_DW + lh * 4;
            const uint32_t* Bs = lds + buf * STAGE + BM * PITCH_DW + (wn * TN * 32 + li) * PITCH_DW + lh * 4;
            bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
            for (int pn = 0; pn < 3; ++pn) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[pn][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(As + i * 32 * PITCH_DW + pn * 8));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[pn][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Bs + j * 32 * PITCH_DW + pn * 8));
            }
            // six partial products, smallest magnitude first
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
            constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[c]][i], bf[PB[c]][j], acc[i][j], 0, 0, 0);

            if (more && !(p.dbg & 2)) store_slab(buf ^ 1);
            if (!(p.dbg & 4)) __syncthreads();
            buf ^= 1;
        }
    }

    if (p.trace) tr2 = __builtin_readcyclecounter();
    // ---- epilogue (identical to conv_igemm.hip): BN affine + ReLU, or accumulate + fused ReLU/BN backward ----
    const int howo = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pp = p0 + (wn * TN + j) * 32 + li;
        if (pp >= p.P) continue;
        uint32_t n, hw;
        fd_divmod((uint32_t)pp, p.div_hw, n, hw);
        float* yb = p.y + (long)n * p.y_img_stride + hw;
        const float* mb = p.mask_y ? p.mask_y + (long)n * p.mask_img_stride + hw : nullptr;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= p.M) continue;
                float v = acc[i][j][r];
                if (p.scale) v = v * p.scale[m] + p.shift[m];
                if (p.relu) v = fmaxf(v, 0.f);
                float* dst = yb + (long)m * howo;
                if (p.accumulate) v += *dst;
                if (mb) {
                    const float sc = p.mask_scale[m];
                    v = (sc < 0.f) ? v * -sc : (mb[(long)m * howo] > 0.f ? v * sc : 0.f);
                }
                *dst = v;
            }
        }
    }
    if (p.trace && tid == 0) {
        unsigned long long* t = p.trace + (size_t)blockIdx.x * 8;
        t[0] = tr0;
        t[1] = tr1;
        t[2] = tr2;
        t[3] = __builtin_readcyclecounter();
        t[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID (wave/simd/cu/sh/se bits)
        t[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
    }
}

// ---- weight split + pack: out[slab][m][plane*8 + kpair] (PITCH_DW dwords per row, zero padded) ----
// slab = g (1x1) or g * 9 + tap (3x3); row k of a slab = channel 16 g + k.
//   mode 0 (forward operand): A[m][c][tap] = w[m][c][tap];  mode 1 (dgrad operand): A[m][c][tap] = w[c][m][tap]
// Two sources (fused pair): output channel co < split comes from w0, the rest from w1.
constexpr int XP_MAX = 40;
constexpr int XP_CHUNK = 4096;   // (slab, m, kpair) triples per block
struct X6PackTable {
    const float* w0[XP_MAX];
    const float* w1[XP_MAX];
    uint32_t* out[XP_MAX];
    int cout[XP_MAX], cin[XP_MAX], ks[XP_MAX], mode[XP_MAX], split[XP_MAX];
    int blk0[XP_MAX + 1];
    int count;
};
__global__ __launch_bounds__(256) void pack_x6_kernel(X6PackTable t) {
    int ti = 0;
    while (ti + 1 < t.count && (int)blockIdx.x >= t.blk0[ti + 1]) ++ti;
    const int mode = t.mode[ti], Cout = t.cout[ti], Cin = t.cin[ti], ks = t.ks[ti];
    const int KK = ks * ks;
    const int M = mode ? Cin : Cout, C = mode ? Cout : Cin;
    const int ngroups = (C + 15) / 16;
    const long total = (long)ngroups * KK * M * 8;     // kpairs
    const long base = (long)((int)blockIdx.x - t.blk0[ti]) * XP_CHUNK;
    long end = base + XP_CHUNK;
    if (end > total) end = total;
    for (long idx = base + threadIdx.x; idx < end; idx += 256) {
        const int kp = (int)(idx & 7);
        const long sm = idx >> 3;
        const int m = (int)(sm % M);
        const int slab = (int)(sm / M);
        const int g = slab / KK, tap = slab - g * KK;
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = 16 * g + 2 * kp + e;
            float x = 0.f;
            if (c < C) {
                const int co = mode ? c : m, ci = mode ? m : c;
                const float* src = co < t.split[ti] ? t.w0[ti] : t.w1[ti];
                const int cor = co < t.split[ti] ? co : co - t.split[ti];
                x = src[((long)cor * Cin + ci) * KK + tap];
            }
            v[e] = x;
        }
        const float r0 = residual(v[0]), r1 = residual(v[1]);
        const float s0 = residual(r0), s1 = residual(r1);
        uint32_t* row = t.out[ti] + ((long)slab * M + m) * PITCH_DW;
        row[kp] = pack_hi16(__builtin_bit_cast(uint32_t, v[0]), __builtin_bit_cast(uint32_t, v[1]));
        row[8 + kp] = pack_hi16(__builtin_bit_cast(uint32_t, r0), __builtin_bit_cast(uint32_t, r1));
        row[16 + kp] = pack_hi16(__builtin_bit_cast(uint32_t, s0), __builtin_bit_cast(uint32_t, s1));
        if (kp < 4) row[24 + kp] = 0u;   // row padding
    }
}

template <int KS, int S, int MODE, int WM, int WN, int TM, int TN>
int launch_cfg(X6Args& a, hipStream_t stream) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    a.n_ptiles = (a.P + BN - 1) / BN;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.div_mt = make_fastdiv((uint32_t)a.n_mtiles);
    const unsigned nblk = (unsigned)a.n_ptiles * (unsigned)a.n_mtiles;
    hipLaunchKernelGGL((conv_x6_kernel<KS, S, MODE, WM, WN, TM, TN>), dim3(nblk), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("conv_x6");
    return SSN_OK;
}

// tile ids as in conv_igemm.hip: 0 128x128, 1 64x128, 2 96x128, 3 64x64, 4 32x128, 5 160x128, 6 64x128 (1x4), 7 128x64
template <int KS, int S, int MODE>
int launch_tile(X6Args& a, int cfg, hipStream_t stream) {
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
    ssn_set_error("conv_x6: unknown tile config %d", cfg);
    return SSN_ERR_ARG;
}

int g_x6_dbg = 0;
int g_x6_desync = 0;
unsigned long long* g_x6_trace = nullptr;

int default_tile(int M, long P) {
    if (M % 128 == 0 || M > 256) return 0;
    if (M % 96 == 0) return 2;
    return (P >= 100000) ? 1 : 3;
}

long x6_packed_dwords(int Cout, int Cin, int ksize, int transposed) {
    const int M = transposed ? Cin : Cout, C = transposed ? Cout : Cin;
    return (long)((C + 15) / 16) * ksize * ksize * M * PITCH_DW;
}

int fill_args(X6Args& a, const float* x, const uint32_t* ap, float* y, int N, int C, int H, int W, long xs, int M,
              int Ho, int Wo, long ys, int ksize, int pad, const char* what) {
    a.x = x;
    a.ap = ap;
    a.y = y;
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
    a.pad = pad;
    a.ngroups = (C + 15) / 16;
    a.div_hw = make_fastdiv((uint32_t)(Ho * Wo));
    a.div_w = make_fastdiv((uint32_t)Wo);
    const long xb = ((long)(N - 1) * xs + (long)C * H * W) * 4;
    const long ab = (long)a.ngroups * ksize * ksize * M * PITCH_DW * 4;
    if (!(xb < (1l << 31) && ab < (1l << 31) && (long)N * Ho * Wo < (1l << 31))) {
        ssn_set_error("%s: operand larger than 2 GiB (buffer addressing)", what);
        return SSN_ERR_ARG;
    }
    a.dbg = g_x6_dbg;
    a.desync = g_x6_desync;
    a.trace = g_x6_trace;
    a.x_bytes = (uint32_t)xb;
    a.a_bytes = (uint32_t)ab;
    return SSN_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
extern "C" void ssn_conv_x6_debug_flags(int flags) {
    g_x6_dbg = flags & 0xFF;
    g_x6_desync = flags >> 8;
}
extern "C" void ssn_conv_x6_debug_trace(unsigned long long* buf) { g_x6_trace = buf; }

extern "C" long ssn_conv_x6_packed_floats(int Cout, int Cin, int ksize, int transposed) {
    return x6_packed_dwords(Cout, Cin, ksize, transposed);
}

// Split + pack `count` weights (HOST arrays, one entry per layer; see ssn_conv_pack_weights_multi for w1/split).
extern "C" int ssn_conv_x6_pack_weights_multi(int count, const float* const* w0, const float* const* w1,
                                              float* const* out, const int* cout, const int* cin, const int* ksize,
                                              const int* mode, const int* split, hipStream_t stream) {
    SSN_CHECK_ARG(count >= 0 && (count == 0 || (w0 && w1 && out && cout && cin && ksize && mode && split)),
                  "conv x6 pack: bad arguments");
    for (int base = 0; base < count; base += XP_MAX) {
        X6PackTable t;
        t.count = count - base < XP_MAX ? count - base : XP_MAX;
        int blocks = 0;
        for (int i = 0; i < t.count; ++i) {
            const int j = base + i;
            SSN_CHECK_ARG(ksize[j] == 1 || ksize[j] == 3, "conv x6 pack: ksize %d unsupported", ksize[j]);
            SSN_CHECK_ARG(mode[j] == 0 || mode[j] == 1, "conv x6 pack: mode %d", mode[j]);
            SSN_CHECK_ARG(w0[j] && out[j] && (w1[j] || split[j] >= cout[j]), "conv x6 pack: null pointer");
            t.w0[i] = w0[j];
            t.w1[i] = w1[j];
            t.out[i] = (uint32_t*)out[j];
            t.cout[i] = cout[j];
            t.cin[i] = cin[j];
            t.ks[i] = ksize[j];
            t.mode[i] = mode[j];
            t.split[i] = split[j];
            t.blk0[i] = blocks;
            const long triples = x6_packed_dwords(cout[j], cin[j], ksize[j], mode[j]) / PITCH_DW * 8;
            blocks += (int)((triples + XP_CHUNK - 1) / XP_CHUNK);
        }
        t.blk0[t.count] = blocks;
        if (blocks) hipLaunchKernelGGL(pack_x6_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, t);
    }
    SSN_CHECK_LAUNCH("conv_x6_pack_weights_multi");
    return SSN_OK;
}

extern "C" int ssn_conv_x6_fwd(const float* x, const float* w_packed, const float* scale, const float* shift,
                               float* y, int N, int Cin, int H, int W, long x_img_stride, int Cout, int Ho, int Wo,
                               long y_img_stride, int ksize, int stride, int pad, int relu, int tile_cfg,
                               hipStream_t stream) {
    SSN_CHECK_ARG(x && w_packed && y, "conv x6 fwd: null pointer");
    SSN_CHECK_ARG(ksize == 1 || ksize == 3, "conv x6 fwd: ksize %d unsupported", ksize);
    SSN_CHECK_ARG(stride == 1 || stride == 2, "conv x6 fwd: stride %d unsupported", stride);
    X6Args a;
    int rc = fill_args(a, x, (const uint32_t*)w_packed, y, N, Cin, H, W, x_img_stride, Cout, Ho, Wo, y_img_stride,
                       ksize, pad, "conv x6 fwd");
    if (rc != SSN_OK) return rc;
    a.scale = scale;
    a.shift = shift;
    a.relu = relu;
    a.accumulate = 0;
    a.mask_y = nullptr;
    a.mask_scale = nullptr;
    a.mask_img_stride = 0;
    const int cfg = tile_cfg >= 0 ? tile_cfg : default_tile(Cout, a.P);
    if (ksize == 1 && stride == 1) return launch_tile<1, 1, MODE_FWD>(a, cfg, stream);
    if (ksize == 3 && stride == 1) return launch_tile<3, 1, MODE_FWD>(a, cfg, stream);
    if (ksize == 3 && stride == 2) return launch_tile<3, 2, MODE_FWD>(a, cfg, stream);
    ssn_set_error("conv x6 fwd: (k=%d, s=%d) has no kernel", ksize, stride);
    return SSN_ERR_ARG;
}

// stride-1 data gradient (stride-2 layers use ssn_conv_dgrad's parity path)
extern "C" int ssn_conv_x6_dgrad(const float* dy, const float* wt_packed, float* dx, int N, int Cout, int Ho, int Wo,
                                 long dy_img_stride, int Cin, int H, int W, long dx_img_stride, int ksize, int pad,
                                 int accumulate, const float* mask_y, long mask_img_stride, const float* mask_scale,
                                 int tile_cfg, hipStream_t stream) {
    SSN_CHECK_ARG(dy && wt_packed && dx, "conv x6 dgrad: null pointer");
    SSN_CHECK_ARG(ksize == 1 || ksize == 3, "conv x6 dgrad: ksize %d unsupported", ksize);
    X6Args a;
    int rc = fill_args(a, dy, (const uint32_t*)wt_packed, dx, N, Cout, Ho, Wo, dy_img_stride, Cin, H, W,
                       dx_img_stride, ksize, pad, "conv x6 dgrad");
    if (rc != SSN_OK) return rc;
    a.scale = nullptr;
    a.shift = nullptr;
    a.relu = 0;
    a.accumulate = accumulate;
    a.mask_y = mask_scale ? mask_y : nullptr;
    a.mask_scale = mask_y ? mask_scale : nullptr;
    a.mask_img_stride = mask_img_stride;
    const int cfg = tile_cfg >= 0 ? tile_cfg : default_tile(Cin, a.P);
    if (ksize == 1) return launch_tile<1, 1, MODE_DGRAD>(a, cfg, stream);
    return launch_tile<3, 1, MODE_DGRAD>(a, cfg, stream);
}
