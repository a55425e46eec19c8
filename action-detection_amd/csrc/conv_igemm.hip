// Implicit-GEMM convolution on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// Replaces the cuDNN conv fwd / dgrad calls the reference reaches through
// model_zoo.BNInception (/root/reference/ssn_models.py:266,298) for the frozen-BN
// Conv+BN+ReLU blocks of the backbone.
//
//   D[m][p] = sum_k A[m][k] * G[k][p]
//     m : output channel            (A = weights, row-major [M][K], k = (c, r, s))
//     p : flattened (n, ho, wo)     (NCHW: p is contiguous inside an image plane, so both the
//                                    gather loads and the epilogue stores are coalesced)
//     G : the im2col gather of the NCHW input, never materialised: each workgroup gathers a
//         [16 x BN] slab straight into LDS (zero-filled at borders / tails).
//
// MODE_FWD      G[k][p] = x[n][c][ho*S - pad + r][wo*S - pad + s]
// MODE_DGRAD    pixels enumerate the conv INPUT (n, hi, wi); the source is dY:
//               G[k][p] = dy[n][co][(hi + pad - r)/S][(wi + pad - s)/S]   (0 unless divisible)
//               with A = W transposed to [Cin][Cout*KH*KW] (see ssn_weight_transpose).
//
// Tiling: 256 threads = 4 waves as WM x WN; each wave owns TM x TN MFMA tiles of 32x32.
// K is consumed in slabs of 16 through two LDS buffers (one barrier per slab); the next
// slab's global loads are issued before the MFMAs of the current one.  Within a slab the
// k index is permuted (k = 8u + 4h + s for lane-half h) so that A fragments are one
// ds_read_b128 per 4 MFMAs; A and B use the same permutation, so only the fp32 summation
// order changes.
#include "ssn_common.h"

namespace {

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

struct ConvArgs {
    const float* x;      // gather source (channel-slice base)
    const float* a;      // [M][K]
    float* y;            // output (channel-slice base)
    const float* scale;  // [M] or nullptr
    const float* shift;  // [M] or nullptr
    int N, C, H, W;      // gather-source dims (C = channels of source, H x W its plane)
    long x_img_stride;   // floats between consecutive images of the source
    int M;               // output channels
    int Ho, Wo;          // enumerated pixel grid
    long y_img_stride;
    int K, P;            // K = C*KS*KS, P = N*Ho*Wo
    int pad;
    int relu, accumulate;
    int n_ptiles, n_mtiles;
    uint32_t x_bytes, a_bytes;  // extents of the gather source / weight matrix (buffer descriptors)
    FastDiv div_hw, div_w, div_mt;
};

// Raw buffer descriptor over [base, base + bytes): loads whose byte offset is >= bytes return 0, which is
// how padding taps, pixel tails and K tails are zero-filled without a single branch in the gather.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
constexpr uint32_t OOB = 0x80000000u;  // any offset >= num_records (extents are checked < 2^31 on the host)

constexpr int BK = 16;
constexpr int A_PITCH = 20;  // floats; 80 B rows keep ds_read_b128 conflict-free

template <int KS, int S, int MODE, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int KSTEP = 256 / BN;         // k rows gathered concurrently
    constexpr int NB = BK / KSTEP;          // gather elements per thread per slab
    constexpr int A_PASSES = (BM + 63) / 64;
    constexpr int KK = KS * KS;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BN >= 64 && BN <= 256 && (256 % BN) == 0, "a wave must gather one k row: BN in {64,128,256}");

    __shared__ __attribute__((aligned(16))) float lds[2 * (BM * A_PITCH + BK * BN)];
    float* As0 = lds;
    float* Bs0 = lds + 2 * BM * A_PITCH;

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

    // ---- per-thread gather column; the k rows a wave gathers are wave-uniform (SGPR decode) ----
    const int gcol = tid % BN;
    const int gk0 = __builtin_amdgcn_readfirstlane(tid / BN);
    const int gp = p0 + gcol;
    const bool gvalid = gp < p.P;
    int gh0 = 0, gw0 = 0;
    uint32_t gbase = 0;  // byte offset of this pixel's image inside the gather source
    {
        uint32_t n, hw, ho, wo;
        fd_divmod((uint32_t)(gvalid ? gp : 0), p.div_hw, n, hw);
        fd_divmod(hw, p.div_w, ho, wo);
        gbase = (uint32_t)((long)n * p.x_img_stride * 4);
        if (MODE == MODE_FWD) {
            gh0 = (int)ho * S - p.pad;
            gw0 = (int)wo * S - p.pad;
        } else {
            gh0 = (int)ho + p.pad;
            gw0 = (int)wo + p.pad;
        }
    }
    const int HW = p.H * p.W;
    const __amdgpu_buffer_rsrc_t xrsrc = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(p.a, p.a_bytes);

    auto gather = [&](int k) -> float {  // k is wave-uniform: (c, r, s) decode runs on the scalar unit
        int c, r, s;
        if (KS == 1) {
            c = k;
            r = 0;
            s = 0;
        } else {
            c = k / KK;
            const int rem = k - c * KK;
            r = rem / KS;
            s = rem - r * KS;
        }
        int hi, wi;
        bool ok = gvalid && (k < p.K);
        if (MODE == MODE_FWD) {
            hi = gh0 + r;
            wi = gw0 + s;
        } else {
            hi = gh0 - r;
            wi = gw0 - s;
            if (S == 2) {
                ok = ok && (((hi | wi) & 1) == 0);
                hi >>= 1;
                wi >>= 1;
            }
        }
        ok = ok && ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W);
        const uint32_t off = gbase + (uint32_t)(c * HW + hi * p.W + wi) * 4u;
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, ok ? off : OOB, 0, 0));
    };

    // ---- per-thread weight-tile slot ----
    const int arow = tid >> 2;
    const int akq = (tid & 3) * 4;
    const bool a_vec = (p.K & 15) == 0;  // every layer but the 7x7 stem (K = 147 / 490)

    float breg[NB];
    f32x4 areg[A_PASSES];

    auto load_slab = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NB; ++i) breg[i] = gather(k0 + gk0 + KSTEP * i);
#pragma unroll
        for (int q = 0; q < A_PASSES; ++q) {
            const int row = arow + 64 * q;
            const int m = m0 + row;
            const int k = k0 + akq;
            const bool rok = (row < BM) && (m < p.M);
            const uint32_t off = (uint32_t)(m * p.K + k) * 4u;
            f32x4 v;
            if (a_vec) {
                v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, rok ? off : OOB, 0, 0));
            } else {
                v.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(arsrc, (rok && k + 0 < p.K) ? off : OOB, 0, 0));
                v.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(arsrc, (rok && k + 1 < p.K) ? off + 4 : OOB, 0, 0));
                v.z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(arsrc, (rok && k + 2 < p.K) ? off + 8 : OOB, 0, 0));
                v.w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(arsrc, (rok && k + 3 < p.K) ? off + 12 : OOB, 0, 0));
            }
            areg[q] = v;
        }
    };
    auto store_slab = [&](int buf) {
        float* As = As0 + buf * BM * A_PITCH;
        float* Bs = Bs0 + buf * BK * BN;
#pragma unroll
        for (int i = 0; i < NB; ++i) Bs[(gk0 + KSTEP * i) * BN + gcol] = breg[i];
#pragma unroll
        for (int q = 0; q < A_PASSES; ++q) {
            const int row = arow + 64 * q;
            if (row < BM) *reinterpret_cast<f32x4*>(&As[row * A_PITCH + akq]) = areg[q];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nslab = (p.K + BK - 1) / BK;
    load_slab(0);
    store_slab(0);
    __syncthreads();

    for (int t = 0; t < nslab; ++t) {
        const int buf = t & 1;
        if (t + 1 < nslab) load_slab((t + 1) * BK);

        const float* As = As0 + buf * BM * A_PITCH + (wm * TM * 32 + li) * A_PITCH + 4 * lh;
        const float* Bs = Bs0 + buf * BK * BN + (4 * lh) * BN + wn * TN * 32 + li;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 af[TM];
            float bf[4][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * A_PITCH + 8 * u);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[s][j] = Bs[(8 * u + s) * BN + j * 32];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[s][j], acc[i][j], 0, 0, 0);
        }

        if (t + 1 < nslab) store_slab(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: folded BN affine + ReLU (fwd) or accumulate (dgrad), NCHW stores ----
    const int howo = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pp = p0 + (wn * TN + j) * 32 + li;
        if (pp >= p.P) continue;
        uint32_t n, hw;
        fd_divmod((uint32_t)pp, p.div_hw, n, hw);
        float* yb = p.y + (long)n * p.y_img_stride + hw;
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
                *dst = v;
            }
        }
    }
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
//   6: alias of 3            7: 2,2,2,1 -> 128x64
template <int KS, int S, int MODE>
int launch_tile(ConvArgs& a, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_cfg<KS, S, MODE, 2, 2, 2, 2>(a, stream);
        case 1: return launch_cfg<KS, S, MODE, 2, 2, 1, 2>(a, stream);
        case 2: return launch_cfg<KS, S, MODE, 1, 4, 3, 1>(a, stream);
        case 3: return launch_cfg<KS, S, MODE, 2, 2, 1, 1>(a, stream);
        case 4: return launch_cfg<KS, S, MODE, 1, 4, 1, 1>(a, stream);
        case 5: return launch_cfg<KS, S, MODE, 1, 4, 5, 1>(a, stream);
        case 6: return launch_cfg<KS, S, MODE, 2, 2, 1, 1>(a, stream);  // (retired 128x32: alias of 3)
        case 7: return launch_cfg<KS, S, MODE, 2, 2, 2, 1>(a, stream);
    }
    ssn_set_error("conv_igemm: unknown tile config %d", cfg);
    return SSN_ERR_ARG;
}

const int kTileBM[8] = {128, 64, 96, 64, 32, 160, 64, 128};
const int kTileBN[8] = {128, 128, 128, 64, 128, 128, 64, 64};

// Pick the tile that minimises (padded MACs) x (wave-quantisation of the grid over 256 CUs).
int pick_tile(int M, long P) {
    double best = 1e300;
    int best_cfg = 0;
    for (int c = 0; c < 8; ++c) {
        const long mt = (M + kTileBM[c] - 1) / kTileBM[c];
        const long pt = (P + kTileBN[c] - 1) / kTileBN[c];
        const double padded = (double)mt * kTileBM[c] * (double)pt * kTileBN[c];
        const double blocks = (double)mt * pt;
        // two workgroups per CU resident: a "round" is 512 blocks
        const double rounds = blocks / 512.0;
        const double quant = (rounds < 1.0) ? 1.0 / (rounds > 0.05 ? rounds : 0.05) : ((long)rounds + 1) / rounds;
        // small tiles re-gather more: mild penalty for BM < 64
        const double reuse = (kTileBM[c] < 64) ? 1.15 : (kTileBN[c] < 64 ? 1.1 : 1.0);
        const double cost = padded * quant * reuse;
        if (cost < best) {
            best = cost;
            best_cfg = c;
        }
    }
    return best_cfg;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI (declared in include/ssn_hip.h)
// ------------------------------------------------------------------------------------------
extern "C" int ssn_conv_bn_relu_fwd(const float* x, const float* w, const float* scale, const float* shift,
                                    float* y, int N, int Cin, int H, int W, long x_img_stride, int Cout, int Ho,
                                    int Wo, long y_img_stride, int ksize, int stride, int pad, int relu,
                                    int tile_cfg, hipStream_t stream) {
    SSN_CHECK_ARG(x && w && y, "conv fwd: null pointer");
    SSN_CHECK_ARG(ksize == 1 || ksize == 3 || ksize == 7, "conv fwd: ksize %d unsupported", ksize);
    SSN_CHECK_ARG(stride == 1 || stride == 2, "conv fwd: stride %d unsupported", stride);
    SSN_CHECK_ARG((long)N * Ho * Wo < (1l << 31), "conv fwd: too many pixels");
    ConvArgs a;
    a.x = x;
    a.a = w;
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
    a.K = Cin * ksize * ksize;
    a.P = N * Ho * Wo;
    a.pad = pad;
    a.relu = relu;
    a.accumulate = 0;
    a.div_hw = make_fastdiv((uint32_t)(Ho * Wo));
    a.div_w = make_fastdiv((uint32_t)Wo);
    const long xb = ((long)(N - 1) * x_img_stride + (long)Cin * H * W) * 4, ab = (long)Cout * a.K * 4;
    SSN_CHECK_ARG(xb < (1l << 31) && ab < (1l << 31), "conv fwd: operand larger than 2 GiB (buffer addressing)");
    a.x_bytes = (uint32_t)xb;
    a.a_bytes = (uint32_t)ab;
    const int cfg = tile_cfg >= 0 ? tile_cfg : pick_tile(Cout, a.P);
    if (ksize == 1 && stride == 1) return launch_tile<1, 1, MODE_FWD>(a, cfg, stream);
    if (ksize == 3 && stride == 1) return launch_tile<3, 1, MODE_FWD>(a, cfg, stream);
    if (ksize == 3 && stride == 2) return launch_tile<3, 2, MODE_FWD>(a, cfg, stream);
    if (ksize == 7 && stride == 2) return launch_tile<7, 2, MODE_FWD>(a, cfg, stream);
    ssn_set_error("conv fwd: (k=%d, s=%d) has no kernel", ksize, stride);
    return SSN_ERR_ARG;
}

// dx[n][ci][hi][wi] (+)= sum_{co,r,s} wt[ci][(co,r,s)] * dy[n][co][(hi+pad-r)/S][(wi+pad-s)/S]
extern "C" int ssn_conv_dgrad(const float* dy, const float* wt, float* dx, int N, int Cout, int Ho, int Wo,
                              long dy_img_stride, int Cin, int H, int W, long dx_img_stride, int ksize,
                              int stride, int pad, int accumulate, int tile_cfg, hipStream_t stream) {
    SSN_CHECK_ARG(dy && wt && dx, "conv dgrad: null pointer");
    SSN_CHECK_ARG(ksize == 1 || ksize == 3, "conv dgrad: ksize %d unsupported", ksize);
    SSN_CHECK_ARG(stride == 1 || stride == 2, "conv dgrad: stride %d unsupported", stride);
    ConvArgs a;
    a.x = dy;
    a.a = wt;
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
    a.K = Cout * ksize * ksize;
    a.P = N * H * W;
    a.pad = pad;
    a.relu = 0;
    a.accumulate = accumulate;
    a.div_hw = make_fastdiv((uint32_t)(H * W));
    a.div_w = make_fastdiv((uint32_t)W);
    const long xb = ((long)(N - 1) * dy_img_stride + (long)Cout * Ho * Wo) * 4, ab = (long)Cin * a.K * 4;
    SSN_CHECK_ARG(xb < (1l << 31) && ab < (1l << 31), "conv dgrad: operand larger than 2 GiB (buffer addressing)");
    a.x_bytes = (uint32_t)xb;
    a.a_bytes = (uint32_t)ab;
    const int cfg = tile_cfg >= 0 ? tile_cfg : pick_tile(Cin, a.P);
    if (ksize == 1 && stride == 1) return launch_tile<1, 1, MODE_DGRAD>(a, cfg, stream);
    if (ksize == 3 && stride == 1) return launch_tile<3, 1, MODE_DGRAD>(a, cfg, stream);
    if (ksize == 3 && stride == 2) return launch_tile<3, 2, MODE_DGRAD>(a, cfg, stream);
    ssn_set_error("conv dgrad: (k=%d, s=%d) has no kernel", ksize, stride);
    return SSN_ERR_ARG;
}

extern "C" int ssn_conv_pick_tile(int M, long P) { return pick_tile(M, P); }
