// Weight (and bias) gradient of the backbone convolutions on planes tensors (planes.h), gfx950: the cuDNN wgrad behind
// loss.backward() (/root/reference/ssn_train.py:236) with fp32-class accuracy on v_mfma_f32_32x32x16_f16.
//
//   dW[co][ci][tap] = sum_p dY[co][p] * X[ci][p + tap]     p = (n, ho, wo) = the reduction index of the GEMM
//   db[co]          = sum_p dY[co][p]
//
// Both operands arrive split (two f16 planes, channel-blocked NC8HW8), so -- unlike conv_wgrad_x6.hip, which re-scales and
// re-splits every element in every K-tile that touches it (13 VALU per MFMA) -- nothing is converted here.  What the layout
// does NOT give is the k-contiguity an MFMA operand wants: the reduction index is the PIXEL, the layout keeps 8 CHANNELS of
// a pixel together.  The LDS transpose read of gfx950 closes that gap:
//   * one k-step = 16 pixel slots.  A fragment-plane (32 channels x 16 slots of one plane) is ONE LDS-DMA instruction: lane
//     l fetches the 16 bytes (8 channels) of slot l / 4, channel group l % 4, landing as a [16 slots][32 channels] f16
//     matrix with 64-byte rows -- whatever the stride, padding or tap of the layer (per-lane gather; pixels whose tap falls
//     outside the image and slots past the end carry an out-of-range offset and deposit zeros);
//   * two ds_read_b64_tr_b16 turn it into the MFMA operand (lane (channel, k-half) <- 8 slots of its channel),
//     conflict-free: a 32-lane service group reads 4 rows = 256 contiguous bytes;
//   * waves 0/1 fetch the two planes of dY, waves 2/3 those of X; the DMA runs two k-steps ahead in a 3-slot ring, the
//     transposed reads of k-step t+1 are dealt out between the MFMAs of k-step t (two register sets), one barrier per k-step.
// A workgroup = (tile of output channels) x (tile of input channels) x ONE tap x (share of the pixel range); partial slabs
// [split][co][ci * KK + tap (+ bias column)] are reduced in a fixed order by ssn_wgrad_reduce (deterministic).  The bias
// column is the product of the dY fragments with a fragment of ones (workgroups of the first input tile and tap only).
#include "planes.h"

namespace {

using namespace pl;

struct WgPlArgs {
    const void* g_hi;   // dY planes at the slice's first channel group
    const void* g_lo;
    const void* x_hi;   // X planes at the slice's first channel group
    const void* x_lo;
    float* part;        // [splits][M][ldp]
    const float* g_scale;
    const float* x_scale;
    int g_row_split, g_row_gap;   // rows m >= g_row_split of dY sit g_row_gap channels further up its tensor
    int N, Cin, H, W;             // X dims (Cin = real input channels: columns ci >= Cin are not stored)
    int M;                        // output channels
    int Ho, Wo;
    int kh, kw, stride, pad_h, pad_w;
    int K, ldp;                   // K = Cin * kh * kw, ldp = K + 1
    int P;                        // N * Ho * Wo
    uint32_t g_img_bytes, g_grp_bytes, x_img_bytes, x_grp_bytes;
    uint32_t g_bytes, x_bytes;    // per plane
    int splits, ksteps_per_split;
    int n_mtiles, n_ctiles;
    FastDiv div_hw, div_w, div_tiles, div_ct, div_kk;
};

#if defined(__HIP_DEVICE_COMPILE__)
#define WG_DMA_B128(rsrc_, dst_, voff_, soff_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_, SSN_LDS_PTR(dst_), 16, voff_, soff_, 0, 0)
#else
#define WG_DMA_B128(rsrc_, dst_, voff_, soff_) ((void)(dst_), (void)(voff_), (void)(soff_))
#endif

template <int WM, int WC, int TM, int TC>
__global__ __launch_bounds__(256, (TM * TC >= 6) ? 1 : 2) void wgrad_pl_kernel(WgPlArgs p) {
    constexpr int BM = WM * TM * 32;
    constexpr int BC = WC * TC * 32;
    constexpr int FA = BM / 32, FB = BC / 32;     // fragments per k-step of dY / X
    constexpr int NPW = FA > FB ? FA : FB;        // DMA instructions per wave and k-step (the surplus ones are dummies)
    constexpr int PIECE = 256;                    // dwords of one fragment-plane
    constexpr int STAGE = 2 * (FA + FB) * PIECE;  // [dY: frag][plane] then [X: frag][plane]
    constexpr int NSTAGE = 3;
    static_assert(WM * WC == 4, "4 waves");

    __shared__ __attribute__((aligned(1024))) uint32_t lds[NSTAGE * STAGE + PIECE];   // + the dummy piece

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int wm = wave / WC, wc = wave % WC;
    const int li = lane & 31, lh = lane >> 5;

    const uint32_t tiles = (uint32_t)p.n_mtiles * (uint32_t)p.n_ctiles * (uint32_t)(p.kh * p.kw);
    const uint32_t logical = xcd_remap(blockIdx.x, tiles * (uint32_t)p.splits);
    uint32_t z, tile, mct, tap, mt, ct;
    fd_divmod(logical, p.div_tiles, z, tile);
    fd_divmod(tile, p.div_kk, mct, tap);
    fd_divmod(mct, p.div_ct, mt, ct);
    const int m0 = (int)mt * BM, c0 = (int)ct * BC;
    const int tr = (int)tap / p.kw, ts = (int)tap - tr * p.kw;

    // ---- DMA role of this wave: operand (0 = dY, 1 = X) and plane ----
    const int op = wave >> 1, plane = wave & 1;
    const int myF = op ? FB : FA;
    const __amdgpu_buffer_rsrc_t rsrc = op ? pl_rsrc(plane ? p.x_lo : p.x_hi, p.x_bytes) : pl_rsrc(plane ? p.g_lo : p.g_hi, p.g_bytes);
    const uint32_t grp_bytes = op ? p.x_grp_bytes : p.g_grp_bytes;
    const uint32_t img_bytes = op ? p.x_img_bytes : p.g_img_bytes;
    const int dsl = lane >> 2;                                       // slot of the k-step this lane fetches
    const uint32_t lane_grp = (uint32_t)(lane & 3) * grp_bytes;      // its channel group inside a fragment
    // scalar byte offset of fragment f of this wave's operand (dY: rows behind the split sit g_row_gap channels further up)
    uint32_t frag_so[NPW];
#pragma unroll
    for (int f = 0; f < NPW; ++f) {
        const int ch = (op ? c0 : m0) + f * 32;
        const int gap = (!op && ch >= p.g_row_split) ? p.g_row_gap : 0;
        frag_so[f] = (uint32_t)((ch + gap) / 8) * grp_bytes;
    }
    const int ks_begin = (int)z * p.ksteps_per_split;
    const int total_ks = (p.P + 15) / 16;
    int ks_end = ks_begin + p.ksteps_per_split;
    if (ks_end > total_ks) ks_end = total_ks;
    const int nks = ks_end - ks_begin;

    auto issue = [&](int ks, uint32_t st_off) {   // fetch k-step ks (past the end: zeros) into the ring slot at st_off
        const int slot = ks * 16 + dsl;
        const bool live = ks < ks_end && slot < p.P;
        uint32_t n, q, ho, wo;
        fd_divmod((uint32_t)(live ? slot : 0), p.div_hw, n, q);
        uint32_t vo;
        if (op == 0) {
            vo = live ? n * img_bytes + q * 16u + lane_grp : PL_OOB;
        } else {
            fd_divmod(q, p.div_w, ho, wo);
            const int hi = (int)ho * p.stride + tr - p.pad_h, wi = (int)wo * p.stride + ts - p.pad_w;
            const bool ok = live && ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W);
            vo = ok ? n * img_bytes + (uint32_t)(hi * p.W + wi) * 16u + lane_grp : PL_OOB;
        }
        uint32_t* base = lds + st_off + (op ? 2 * FA * PIECE : 0) + plane * PIECE;
#pragma unroll
        for (int f = 0; f < NPW; ++f) {
            const bool real = f < myF;      // wave-uniform
            WG_DMA_B128(rsrc, real ? base + f * 2 * PIECE : lds + NSTAGE * STAGE, real ? vo : PL_OOB, frag_so[f]);
        }
    };

    f32x16 acc[TM][TC];
    f32x16 accb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    const bool do_bias = wave_uniform((ct == 0 && tap == 0 && wc == 0) ? 1 : 0) != 0;

    issue(ks_begin, 0);
    issue(ks_begin + 1, STAGE);
    issue(ks_begin + 2, 2 * STAGE);

    // transposed fragment reads: lane (channel li, k-half lh) <- slots 8 lh + 0..7 of its channel, as two reads of 4 slots:
    // inside a 16-lane group, lane 4 j + q supplies channels 4 q .. 4 q + 3 of slot j
    const int l16 = lane & 15, sg = (lane >> 4) & 1;
    const int lane_rd = ((8 * lh + (l16 >> 2)) * 64 + sg * 32 + (l16 & 3) * 8) / 4;   // dwords
    struct Frags {
        f16x8 a[2][TM], b[2][TC];
    };
    Frags fr0, fr1;
    constexpr int NREAD = 2 * (TM + TC);
    const uint32_t* rd_base;
    auto read_begin = [&](uint32_t st_off) { rd_base = lds + st_off + lane_rd; };
    auto read_step = [&](Frags& f, int k) {   // k is a compile-time constant at every call site
        const bool isb = k >= 2 * TM;
        const int kk = isb ? k - 2 * TM : k;
        const int pn = kk & 1, i = kk >> 1;
        const uint32_t* src = rd_base + (isb ? (2 * FA + (wc * TC + i) * 2 + pn) : ((wm * TM + i) * 2 + pn)) * PIECE;
        const u32x2 r0 = SSN_DS_READ_TR16_B64(src);
        const u32x2 r1 = SSN_DS_READ_TR16_B64(src + 64);
        const f16x8 v = __builtin_bit_cast(f16x8, u32x4{r0[0], r0[1], r1[0], r1[1]});
        if (isb)
            f.b[pn][i] = v;
        else
            f.a[pn][i] = v;
    };
    const f16x8 ones = __builtin_bit_cast(f16x8, u32x4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u});
    auto mfma = [&](const Frags& f, int next_ks, uint32_t dma_stage, Frags& nxt) {
        constexpr int PA[3] = {1, 0, 0};   // g_lo x_hi + g_hi x_hi + g_hi x_lo
        constexpr int PB[3] = {0, 0, 1};
        constexpr int NM = 3 * TM * TC;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TC; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[PA[c]][i], f.b[PB[c]][j], acc[i][j], 0, 0, 0);
                    const int idx = (c * TM + i) * TC + j;
#pragma unroll
                    for (int k = idx * NREAD / NM; k < (idx + 1) * NREAD / NM; ++k) read_step(nxt, k);
                    if (idx == NM / 2) issue(next_ks, dma_stage);
                    __builtin_amdgcn_sched_barrier(0);
                }
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[0][i], ones, accb[i], 0, 0, 0);
                accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[1][i], ones, accb[i], 0, 0, 0);
            }
        }
    };

    SSN_WAIT_VMCNT(2 * NPW);
    __builtin_amdgcn_s_barrier();
    read_begin(0);
#pragma unroll
    for (int k = 0; k < NREAD; ++k) read_step(fr0, k);
    uint32_t s_cur = 0, s_n1 = STAGE, s_n2 = 2 * STAGE;
    int ks = ks_begin;
    auto half = [&](Frags& cur, Frags& nxt) {
        SSN_WAIT_VMCNT(NPW);
        SSN_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_begin(s_n1);
        mfma(cur, ks + 3, s_cur, nxt);
        ++ks;
        const uint32_t o = s_cur;
        s_cur = s_n1;
        s_n1 = s_n2;
        s_n2 = o;
    };
    for (int t = 0; t < nks; t += 2) {
        half(fr0, fr1);
        half(fr1, fr0);
    }
    SSN_WAIT_LGKM0();
    SSN_WAIT_VMCNT(0);

    // ---- partial slab store: part[z][m][ci * KK + tap] ----
    const float inv = 1.f / (*p.g_scale * *p.x_scale);
    const int KK = p.kh * p.kw;
    float* out = p.part + (long)z * p.M * p.ldp;
#pragma unroll
    for (int j = 0; j < TC; ++j) {
        const int ci = c0 + (wc * TC + j) * 32 + li;
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.M) out[(long)m * p.ldp + ci * KK + (int)tap] = acc[i][j][r] * inv;
            }
        }
    }
    if (do_bias && li == 0) {
        const float ginv = 1.f / *p.g_scale;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.M) out[(long)m * p.ldp + p.K] = accb[i][r] * ginv;
            }
    }
}
#undef WG_DMA_B128

template <int WM, int WC, int TM, int TC>
int launch_wgpl(WgPlArgs& a, hipStream_t stream) {
    constexpr int BM = WM * TM * 32;
    constexpr int BC = WC * TC * 32;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.n_ctiles = (a.Cin + BC - 1) / BC;
    const unsigned tiles = (unsigned)a.n_mtiles * (unsigned)a.n_ctiles * (unsigned)(a.kh * a.kw);
    a.div_tiles = make_fastdiv(tiles);
    a.div_ct = make_fastdiv((uint32_t)a.n_ctiles);
    a.div_kk = make_fastdiv((uint32_t)(a.kh * a.kw));
    hipLaunchKernelGGL((wgrad_pl_kernel<WM, WC, TM, TC>), dim3(tiles * (unsigned)a.splits), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("wgrad_pl");
    return SSN_OK;
}

// tile configs (output channels x input channels):
//   0: 64 x 64   1: 128 x 64   2: 64 x 128   3: 128 x 128   4: 32 x 128   5: 128 x 32   6: 256 x 128   7: 128 x 256
//   8: 96 x 128 (1 x 4 waves)   9: 192 x 64   10: 192 x 128
constexpr int NCFG = 11;
const int kBM[NCFG] = {64, 128, 64, 128, 32, 128, 256, 128, 96, 192, 192};
const int kBC[NCFG] = {64, 64, 128, 128, 128, 32, 128, 256, 128, 64, 128};
const int kOcc[NCFG] = {3, 3, 3, 2, 3, 3, 1, 1, 2, 2, 1};

int launch_wgpl_tile(WgPlArgs& a, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_wgpl<2, 2, 1, 1>(a, stream);
        case 1: return launch_wgpl<2, 2, 2, 1>(a, stream);
        case 2: return launch_wgpl<2, 2, 1, 2>(a, stream);
        case 3: return launch_wgpl<2, 2, 2, 2>(a, stream);
        case 4: return launch_wgpl<1, 4, 1, 1>(a, stream);
        case 5: return launch_wgpl<4, 1, 1, 1>(a, stream);
        case 6: return launch_wgpl<2, 2, 4, 2>(a, stream);
        case 7: return launch_wgpl<2, 2, 2, 4>(a, stream);
        case 8: return launch_wgpl<1, 4, 3, 1>(a, stream);
        case 9: return launch_wgpl<2, 2, 3, 1>(a, stream);
        case 10: return launch_wgpl<2, 2, 3, 2>(a, stream);
    }
    ssn_set_error("conv_wgrad_pl: unknown tile config %d", cfg);
    return SSN_ERR_ARG;
}

int pick_tile(int M, int Cin) {
    double best = 1e300;
    int bc = 0;
    for (int c = 0; c < 6; ++c) {
        const double padded = (double)((M + kBM[c] - 1) / kBM[c]) * kBM[c] * (double)((Cin + kBC[c] - 1) / kBC[c]) * kBC[c];
        const double small = (kBM[c] * kBC[c] >= 128 * 128) ? 1.0 : (kBM[c] * kBC[c] >= 128 * 64 ? 1.08 : 1.25);
        if (padded * small < best) {
            best = padded * small;
            bc = c;
        }
    }
    return bc;
}

void plan(int M, int Cin, int KK, long P, int cfg, int* splits, int* ksteps_per_split) {
    const long tiles = (long)((M + kBM[cfg] - 1) / kBM[cfg]) * ((Cin + kBC[cfg] - 1) / kBC[cfg]) * KK;
    const long ksteps = (P + 15) / 16;
    plan_split_k(tiles, ksteps, kOcc[cfg], 32, 16, 0.02 + (double)M * Cin * KK * 6.7e-6, splits, ksteps_per_split);
    if (*ksteps_per_split & 1) {   // the pipeline runs two k-steps per trip
        ++*ksteps_per_split;
        *splits = (int)((ksteps + *ksteps_per_split - 1) / *ksteps_per_split);
    }
}

int fix_cfg(int tile_cfg, int M, int Cin) {
    return (tile_cfg >= 0 && tile_cfg < NCFG) ? tile_cfg : pick_tile(M, Cin);
}

}  // namespace

extern "C" int ssn_wgrad_reduce(const float* part, float* dw, float* db, int M, int K, int splits, hipStream_t stream);

extern "C" int ssn_conv_wgrad_pl_tiles(void) { return NCFG; }

extern "C" long ssn_conv_wgrad_pl_workspace_bytes(int N, int Cin, int Cout, int Ho, int Wo, int kh, int kw, int tile_cfg) {
    const int cfg = fix_cfg(tile_cfg, Cout, Cin);
    int splits, kps;
    plan(Cout, Cin, kh * kw, (long)N * Ho * Wo, cfg, &splits, &kps);
    return (long)splits * Cout * ((long)Cin * kh * kw + 1) * (long)sizeof(float);
}

// Weight + bias gradient on planes slices: g = dY [N, Cout, Ho, Wo] (final: its ReLU / BN backward applied), x = the layer's
// input [N, Cin (padded to 8), H, W]; any stride / padding / kh x kw taps.  dw [Cout][Cin][kh][kw] fp32, db [Cout] or null.
// *_img_groups: channel groups of the whole tensors.  g_row_gap > 0: rows >= g_row_split of g sit g_row_gap channels further
// up its tensor (fused block-input launch).
extern "C" int ssn_conv_wgrad_pl(const void* g_hi, const void* g_lo, const void* x_hi, const void* x_lo, float* dw, float* db,
                                 int N, int Cin, int H, int W, long x_img_groups, int Cout, int Ho, int Wo, long g_img_groups,
                                 int kh, int kw, int stride, int pad_h, int pad_w, void* workspace, long ws_bytes, int tile_cfg,
                                 const float* g_scale, const float* x_scale, int g_row_split, int g_row_gap,
                                 hipStream_t stream) {
    SSN_CHECK_ARG(g_hi && g_lo && x_hi && x_lo && dw && workspace && g_scale && x_scale, "conv wgrad pl: null pointer");
    SSN_CHECK_ARG(Cout > 0 && Cin > 0 && kh >= 1 && kw >= 1 && (stride == 1 || stride == 2), "conv wgrad pl: bad shape");
    SSN_CHECK_ARG(g_row_gap >= 0 && (g_row_gap == 0 || (g_row_split > 0 && g_row_split < Cout && g_row_split % 32 == 0 && g_row_gap % 8 == 0)),
                  "conv wgrad pl: bad row split");
    WgPlArgs a;
    a.g_hi = g_hi;
    a.g_lo = g_lo;
    a.x_hi = x_hi;
    a.x_lo = x_lo;
    a.part = (float*)workspace;
    a.g_scale = g_scale;
    a.x_scale = x_scale;
    a.g_row_split = g_row_gap ? g_row_split : 0x7fffffff;
    a.g_row_gap = g_row_gap;
    a.N = N;
    a.Cin = Cin;
    a.H = H;
    a.W = W;
    a.M = Cout;
    a.Ho = Ho;
    a.Wo = Wo;
    a.kh = kh;
    a.kw = kw;
    a.stride = stride;
    a.pad_h = pad_h;
    a.pad_w = pad_w;
    a.K = Cin * kh * kw;
    a.ldp = a.K + 1;
    a.P = N * Ho * Wo;
    const long gg = (long)Ho * Wo * 16, xg = (long)H * W * 16;
    const long gb = (long)N * g_img_groups * gg, xb = (long)N * x_img_groups * xg;
    SSN_CHECK_ARG(gb < (1l << 31) && xb < (1l << 31), "conv wgrad pl: operand plane larger than 2 GiB (buffer addressing)");
    a.g_grp_bytes = (uint32_t)gg;
    a.x_grp_bytes = (uint32_t)xg;
    a.g_img_bytes = (uint32_t)(g_img_groups * gg);
    a.x_img_bytes = (uint32_t)(x_img_groups * xg);
    // the descriptors end with the TENSORS (a fragment may reach past the slice: rows / columns that are never stored)
    a.g_bytes = (uint32_t)gb;
    a.x_bytes = (uint32_t)xb;
    a.div_hw = make_fastdiv((uint32_t)(Ho * Wo));
    a.div_w = make_fastdiv((uint32_t)Wo);
    const int cfg = fix_cfg(tile_cfg, Cout, Cin);
    plan(Cout, Cin, kh * kw, a.P, cfg, &a.splits, &a.ksteps_per_split);
    const long need = (long)a.splits * Cout * a.ldp * (long)sizeof(float);
    if (ws_bytes < need) {
        ssn_set_error("conv wgrad pl: workspace %ld < %ld bytes", ws_bytes, need);
        return SSN_ERR_WORKSPACE;
    }
    const int rc = launch_wgpl_tile(a, cfg, stream);
    if (rc != SSN_OK) return rc;
    return ssn_wgrad_reduce(a.part, dw, db, Cout, a.K, a.splits, stream);
}
