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
// [split][co][ci * KK + tap (+ bias column)] (nine-tap kernel: [tap * Cin + ci]) are reduced in a fixed order by ssn_wgrad_reduce (deterministic).  The bias
// column is the product of the dY fragments with a fragment of ones (workgroups of the first input tile and tap only).
#include "planes.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace {

using namespace pl;

struct WgPlArgs {
    const void* g_hi;   // dY planes at the slice's first channel group
    const void* g_lo;
    const void* x_hi;   // X planes at the slice's first channel group
    const void* x_lo;
    float* part;        // [splits][M][ldp]
    unsigned long long* trace;   // tooling only (tools/trace_wgrad_pl.py): per-block phase timestamps, normally null
    int dbg;                     // tooling only (-DPL_ABLATE builds): 1: fetch nothing, 2: X fragments of every third step only, 4: no fetch instructions
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
    uint32_t magic_wp;            // nine-tap / chunked kernels: ceil(2^32 / padded row length)
    int n_mtiles, n_ctiles;
    FastDiv div_hw, div_w, div_tiles, div_ct, div_kk;
};


#ifdef PL_ABLATE
#define WG_DBG(bit) (p.dbg & (bit))
#else
#define WG_DBG(bit) 0
#endif

// tooling: phase stamps of a block (cycle counter at start / loop start / loop end / end, HW_ID, XCC_ID, 100 MHz real time at start / end)
struct WgTrace {
    unsigned long long t0 = 0, t1 = 0, t2 = 0, r0 = 0;
    __device__ void begin(const WgPlArgs& p) {
        if (p.trace) {
            t0 = __builtin_readcyclecounter();
            r0 = __builtin_amdgcn_s_memrealtime();
        }
    }
    __device__ void mark1(const WgPlArgs& p) { if (p.trace) t1 = __builtin_readcyclecounter(); }
    __device__ void mark2(const WgPlArgs& p) { if (p.trace) t2 = __builtin_readcyclecounter(); }
    __device__ void end(const WgPlArgs& p) {
        if (p.trace && threadIdx.x == 0) {
            unsigned long long* t = p.trace + (size_t)blockIdx.x * 8;
            t[0] = t0; t[1] = t1; t[2] = t2;
            t[3] = __builtin_readcyclecounter();
            t[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
            t[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
            t[6] = r0;
            t[7] = __builtin_amdgcn_s_memrealtime();
        }
    }
};

#if defined(__HIP_DEVICE_COMPILE__)
#define WG_DMA_B128(rsrc_, dst_, voff_, soff_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_, SSN_LDS_PTR(dst_), 16, voff_, soff_, 0, 0)
#else
#define WG_DMA_B128(rsrc_, dst_, voff_, soff_) ((void)(dst_), (void)(voff_), (void)(soff_))
#endif

// LDS dwords of the one-tap body: a 3-slot ring of k-steps ([dY: frag][plane] then [X: frag][plane], 256 dwords per fragment-plane)
// + the dummy piece the surplus DMA instructions of the narrower operand land in
template <int WM, int WC, int TM, int TC>
constexpr int wgpl_lds_dwords() {
    return 3 * 2 * (WM * TM + WC * TC) * 256 + 256;
}

// The body of the one-tap kernel as a device function: `bid` = block id INSIDE the problem (the grid of a single-problem launch, or
// the block's offset from its problem's first block in a grouped launch), `lds` = wgpl_lds_dwords() dwords, 1 KiB aligned.
template <int WM, int WC, int TM, int TC>
__device__ __forceinline__ void wgrad_pl_body(const WgPlArgs& p, const uint32_t bid, uint32_t* lds) {
    constexpr int BM = WM * TM * 32;
    constexpr int BC = WC * TC * 32;
    constexpr int FA = BM / 32, FB = BC / 32;     // fragments per k-step of dY / X
    constexpr int NPW = FA > FB ? FA : FB;        // DMA instructions per wave and k-step (the surplus ones are dummies)
    constexpr int PIECE = 256;                    // dwords of one fragment-plane
    constexpr int STAGE = 2 * (FA + FB) * PIECE;  // [dY: frag][plane] then [X: frag][plane]
    // ring depth 3 (the k-step in registers + two in flight).  Deeper rings (4 - 6 slots, as many as the LDS share of a workgroup
    // holds) were measured 8 - 10 % SLOWER on the 1x1 block-input layers, with the fetch switched off too: what the fetch costs
    // this loop is not latency (tools/trace_wgrad_pl.py)
    constexpr int NSTAGE = 3;
    static_assert(WM * WC == 4, "4 waves");
    static_assert(NSTAGE * STAGE + PIECE == wgpl_lds_dwords<WM, WC, TM, TC>(), "LDS size");

    const int tid = threadIdx.x;
    WgTrace trc;
    trc.begin(p);
    const int lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int wm = wave / WC, wc = wave % WC;
    const int li = lane & 31, lh = lane >> 5;

    const uint32_t tiles = (uint32_t)p.n_mtiles * (uint32_t)p.n_ctiles * (uint32_t)(p.kh * p.kw);
    const uint32_t logical = xcd_remap(bid, tiles * (uint32_t)p.splits);
    uint32_t z, tile, mct, tap, mt, ct;
    fd_divmod(logical, p.div_tiles, z, tile);
    fd_divmod(tile, p.div_kk, mct, tap);
    fd_divmod(mct, p.div_ct, mt, ct);
    const int m0 = (int)mt * BM, c0 = (int)ct * BC;
    const int tr = (int)tap / p.kw, ts = (int)tap - tr * p.kw;

    // ---- DMA role of this wave: operand (0 = dY, 1 = X) and plane ----
    const int op = wave >> 1, plane = wave & 1;
    const int myF = op ? FB : FA;
    // (fields read into values BEFORE the selects: `c ? p.a : p.b` on two fields is a select of ADDRESSES, which pins a problem that
    // was loaded from a table -- the grouped kernels -- to scratch memory instead of registers)
    const void *xh = p.x_hi, *xl = p.x_lo, *gh = p.g_hi, *gl = p.g_lo;
    const uint32_t xbytes = p.x_bytes, gbytes = p.g_bytes, xgrp = p.x_grp_bytes, ggrp = p.g_grp_bytes, ximg = p.x_img_bytes, gimg = p.g_img_bytes;
    const __amdgpu_buffer_rsrc_t rsrc = op ? pl_rsrc(plane ? xl : xh, xbytes) : pl_rsrc(plane ? gl : gh, gbytes);
    const uint32_t grp_bytes = op ? xgrp : ggrp;
    const uint32_t img_bytes = op ? ximg : gimg;
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
            WG_DMA_B128(rsrc, real ? base + f * 2 * PIECE : lds + NSTAGE * STAGE, (real && !WG_DBG(1)) ? vo : PL_OOB, frag_so[f]);
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

#pragma unroll
    for (int i = 0; i < NSTAGE; ++i) issue(ks_begin + i, i * STAGE);

    // transposed fragment reads: lane (channel li, k-half lh) <- slots 8 lh + 0..7 of its channel, as two reads of 4 slots:
    // inside a 16-lane group, lane 4 j + q supplies channels 4 q .. 4 q + 3 of slot j
    const int l16 = lane & 15, sg = (lane >> 4) & 1;
    const int lane_rd = ((8 * lh + (l16 >> 2)) * 64 + sg * 32 + (l16 & 3) * 8) / 4;   // dwords
    struct Frags {
        f16x8 a[2][TM], b[2][TC];
    };
    Frags fr0, fr1;
    constexpr int NREAD = 2 * (TM + TC);
    // (asm reads, see planes.h: nothing the compiler orders or waits for -- the SSN_WAIT_LGKM0 at the head of every k-step does)
    const uint32_t *rd_a, *rd_b;
    auto read_begin = [&](uint32_t st_off) {
        rd_a = lds + st_off + lane_rd + (wm * TM * 2) * PIECE;
        rd_b = lds + st_off + lane_rd + (2 * FA + wc * TC * 2) * PIECE;
    };
    auto read_step = [&](Frags& f, int k) {   // k is a compile-time constant at every call site
        const bool isb = k >= 2 * TM;
        const int kk = isb ? k - 2 * TM : k;
        const int pn = kk & 1, i = kk >> 1;
        u32x2 r0, r1;
        if (isb) {
            SSN_DS_READ_TR16_B64_AT(r0, rd_b, (i * 2 + pn) * PIECE * 4);
            SSN_DS_READ_TR16_B64_AT(r1, rd_b, (i * 2 + pn) * PIECE * 4 + 256);
        } else {
            SSN_DS_READ_TR16_B64_AT(r0, rd_a, (i * 2 + pn) * PIECE * 4);
            SSN_DS_READ_TR16_B64_AT(r1, rd_a, (i * 2 + pn) * PIECE * 4 + 256);
        }
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

    static_assert((NSTAGE - 1) * NPW <= 63, "vmcnt range");
    SSN_WAIT_VMCNT((NSTAGE - 1) * NPW);
    __builtin_amdgcn_s_barrier();
    read_begin(0);
#pragma unroll
    for (int k = 0; k < NREAD; ++k) read_step(fr0, k);
    uint32_t st[NSTAGE];      // ring slots, oldest first: st[0] holds the k-step now in registers (free for the next fetch)
#pragma unroll
    for (int i = 0; i < NSTAGE; ++i) st[i] = (uint32_t)(i * STAGE);
    int ks = ks_begin;
    auto half = [&](Frags& cur, Frags& nxt) {
        SSN_WAIT_VMCNT((NSTAGE - 2) * NPW);
        SSN_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_begin(st[1]);
        mfma(cur, ks + NSTAGE, st[0], nxt);
        ++ks;
        const uint32_t o = st[0];
#pragma unroll
        for (int i = 0; i + 1 < NSTAGE; ++i) st[i] = st[i + 1];
        st[NSTAGE - 1] = o;
    };
    trc.mark1(p);
    for (int t = 0; t < nks; t += 2) {
        half(fr0, fr1);
        half(fr1, fr0);
    }
    SSN_WAIT_LGKM0();
    SSN_WAIT_VMCNT(0);
    trc.mark2(p);

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
    trc.end(p);
}

template <int WM, int WC, int TM, int TC>
__global__ __launch_bounds__(256, (TM * TC >= 6) ? 1 : 2) void wgrad_pl_kernel(WgPlArgs p) {
    __shared__ __attribute__((aligned(1024))) uint32_t lds[wgpl_lds_dwords<WM, WC, TM, TC>()];
    wgrad_pl_body<WM, WC, TM, TC>(p, blockIdx.x, lds);
}

// ---------------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 layers (two thirds of the weight-gradient work): ALL NINE TAPS in one workgroup.
//
// The one-tap kernel above streams dY and X once per (tile, tap): nine times the L2 -> LDS traffic of the layer, which is what
// bounds it (measured: 2.1 GB per launch on the 192 -> 192 layer of the 14 x 14 stage = the whole 0.2 ms).  Here a workgroup
// keeps nine accumulator sets and feeds them from ONE copy of the operands: the reduction index runs over PADDED pixel slots
//        slot = n * SP + hp * Wp + wp,   Wp = W + 1, SP = (H + 1) * Wp     (row 0 / column 0 of every image = zero border,
//                                                                          shared with the previous row / image)
// so that the input pixel of tap (r, s) for the output pixel at `slot` is simply slot + (r - 1) * Wp + (s - 1) -- a constant
// displacement of the transposed LDS read, no masks: whatever falls off the image lands on a border slot, and border slots
// hold zeros on both operands (their DMA offsets are out of range).  Price: (H + 1)(W + 1) / HW more k-steps (1.15 at 14 x 14).
// A chunk = 64 slots of dY and the 64 + 2 (Wp + 1) slots of X around them, double-buffered; 4 k-steps x 9 taps x 3 products per
// chunk and barrier; waves 0/1 fetch the planes of dY, waves 2/3 those of X.
// KK = 1 (1x1 layers): the same chunked pipeline without borders or halo -- 64 plain pixel slots per chunk, 4 k-steps per barrier
// instead of the one-tap kernel's single one, register tiles up to 64 x 64 per wave.
// KG = 2: EIGHT waves -- two groups of four, each with its own accumulators over the same output tile, splitting the four
// k-steps of every chunk between them (partial slabs z * 2 + group).  For the layers whose LDS footprint leaves room for one
// workgroup per CU only: a lone wave per SIMD keeps the matrix pipe 45 % busy in this loop (nobody covers its LDS round
// trips and DMA address arithmetic), two waves per SIMD 80 % (tools/trace_wgrad_pl.py).
// LDS bytes of the chunked body: two buffers of [dY: frag][plane][64 slots][32 ch] + [X: frag][plane][XP * 16 slots][32 ch]
template <int TM, int TC, int XP>
constexpr int wgpl9_lds_bytes() {
    return 2 * ((2 * TM) * 2 * 4 * 1024 + (2 * TC) * 2 * XP * 1024);
}

// (`bid`, `lds`: see wgrad_pl_body; wgpl9_lds_bytes() bytes, 1 KiB aligned)
// KH x KW taps with PH / PW padding pixels in front (same-size output): 3 x 3 / pad 1 (the default), and -- Inception-v3's 17 x 17 stage --
// 1 x 7 / pad (0, 3) and 7 x 1 / pad (3, 0): the padded enumeration keeps max(PW, KW - 1 - PW) zero columns in front of every row and
// max(PH, KH - 1 - PH) zero rows in front of every image, a tap (r, s) is the read displacement (r - PH) Wp + (s - PW).
template <int KK, int TM, int TC, int XP, int KG = 1, int KH = 3, int KW = 3, int PH = 1, int PW = 1>
__device__ __forceinline__ void wgrad_pl9_body(const WgPlArgs& p, const uint32_t bid, unsigned char* lds) {
    constexpr int NS = 64, KSC = NS / 16;
    constexpr int KSG = KSC / KG;                      // k-steps of a chunk per wave group
    static_assert(KK == 1 || KK == KH * KW, "1x1, or all taps of the window");
    static_assert((KSG * KK) % 2 == 0, "steps come in pairs");
    static_assert(KG == 1 || KG == 2, "one or two wave groups");
    static_assert(XP % KG == 0, "X pieces split evenly between the wave groups");
    constexpr int BM = 2 * TM * 32, BC = 2 * TC * 32;
    constexpr int FA = BM / 32, FB = BC / 32;
    constexpr int A_BYTES = FA * 2 * KSC * 1024;     // [frag][plane][64 slots][32 ch]
    constexpr int X_BYTES = FB * 2 * XP * 1024;      // [frag][plane][XP * 16 slots][32 ch]
    constexpr int STAGE = A_BYTES + X_BYTES;
    static_assert(2 * STAGE == wgpl9_lds_bytes<TM, TC, XP>(), "LDS size");

    const int tid = threadIdx.x;
    WgTrace trc;
    trc.begin(p);
    const int lane = tid & 63, wave8 = wave_uniform(tid >> 6);
    const int wave = wave8 & 3, grp = wave8 >> 2;     // grp: which k-steps of a chunk (and which share of the DMA pieces)
    const int wm = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const uint32_t tiles = (uint32_t)p.n_mtiles * (uint32_t)p.n_ctiles;
    const uint32_t logical = xcd_remap(bid, tiles * (uint32_t)p.splits);
    uint32_t z, tile, mt, ct;
    fd_divmod(logical, p.div_tiles, z, tile);
    fd_divmod(tile, p.div_ct, mt, ct);
    const int m0 = (int)mt * BM, c0 = (int)ct * BC;
    // zero rows in front of every image / zero columns in front of every row (shared with the previous image / row); none for 1x1
    constexpr int BH = KK > 1 ? (PH > KH - 1 - PH ? PH : KH - 1 - PH) : 0, BW = KK > 1 ? (PW > KW - 1 - PW ? PW : KW - 1 - PW) : 0;
    const int Wp = p.W + BW, SP = (p.H + BH) * Wp;
    const int D = KK > 1 ? PH * Wp + PW : 0;           // slots of X in front of the chunk's first dY slot
    const uint32_t T = (uint32_t)p.N * (uint32_t)SP;   // padded slots that can hold data

    // ---- DMA role: operand (0 = dY, 1 = X) and plane ----
    const int op = wave >> 1, plane = wave & 1;
    // (fields read into values BEFORE the selects: `c ? p.a : p.b` on two fields is a select of ADDRESSES, which pins a problem that
    // was loaded from a table -- the grouped kernels -- to scratch memory instead of registers)
    const void *xh = p.x_hi, *xl = p.x_lo, *gh = p.g_hi, *gl = p.g_lo;
    const uint32_t xbytes = p.x_bytes, gbytes = p.g_bytes, xgrp = p.x_grp_bytes, ggrp = p.g_grp_bytes, ximg = p.x_img_bytes, gimg = p.g_img_bytes;
    const __amdgpu_buffer_rsrc_t rsrc = op ? pl_rsrc(plane ? xl : xh, xbytes) : pl_rsrc(plane ? gl : gh, gbytes);
    const uint32_t grp_bytes = op ? xgrp : ggrp;
    const uint32_t img_bytes = op ? ximg : gimg;
    const int dsl = lane >> 2;
    const uint32_t lane_grp = (uint32_t)(lane & 3) * grp_bytes;
    constexpr int NF = FA > FB ? FA : FB;
    uint32_t frag_so[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int ch = (op ? c0 : m0) + f * 32;
        const int gap = (!op && ch >= p.g_row_split) ? p.g_row_gap : 0;
        frag_so[f] = (uint32_t)((ch + gap) / 8) * grp_bytes;
    }
    const int ck_begin = (int)z * p.ksteps_per_split;          // (chunks per split)
    const int total_ck = (int)((T + NS - 1) / NS);
    int ck_end = ck_begin + p.ksteps_per_split;
    if (ck_end > total_ck) ck_end = total_ck;

    // ---- operand fetch: piece q of this wave = 16 slots x 32 channels x FA (FB) fragments of its plane.  Each lane walks ONE slot
    // per piece through the padded pixel space, 64 slots further per chunk: (position inside the padded image, byte offset of
    // the image) are carried in registers and advanced with a compare-and-wrap, the row / column split is one mulhi with
    // p.magic_wp = ceil(2^32 / Wp) (exact below 2^16) -- a dozen VALU per piece where two 64-bit fast divisions took forty.
    constexpr int NPMAX = (XP > KSC ? XP : KSC) / KG;
    const int npieces = wave_uniform(op ? XP / KG : KSC / KG);
    const int lane_slot0 = (op ? -D : 0) + dsl;                   // slot of this lane in piece sg of chunk 0: lane_slot0 + 16 sg
    const uint32_t adv_img = (uint32_t)(NS / SP) * img_bytes, adv_u = (uint32_t)(NS % SP);
    uint32_t pu[NPMAX], pn[NPMAX];
#pragma unroll
    for (int q = 0; q < NPMAX; ++q) {
        const int sl = ck_begin * NS + lane_slot0 + (q * KG + grp) * 16;
        uint32_t n, u;
        fd_divmod((uint32_t)(sl < 0 ? sl + SP : sl), p.div_hw, n, u);       // div_hw = SP;  D <= SP: one image back at most
        pu[q] = u;
        pn[q] = (n - (sl < 0 ? 1u : 0u)) * img_bytes;                      // (wraps for the slots in front of image 0: never used)
    }
    auto issue_piece = [&](int ck, int buf, int q) {    // piece q of chunk ck -> buffer buf; the lane's state moves on to chunk ck + 1
        if (q >= npieces || (WG_DBG(4) && ck > ck_begin)) return;
        const int sg = q * KG + grp;
        const int sl = ck * NS + lane_slot0 + sg * 16;
        const uint32_t hp = __umulhi(pu[q], p.magic_wp), wp = pu[q] - hp * (uint32_t)Wp;
        const bool real = ck < ck_end && (uint32_t)sl < T && hp >= (uint32_t)BH && wp >= (uint32_t)BW && !WG_DBG(1);
        const uint32_t vo = real ? pn[q] + ((hp - BH) * (uint32_t)p.W + (wp - BW)) * 16u + lane_grp : PL_OOB;
        unsigned char* base = lds + buf * STAGE;
        if (op == 0) {
#pragma unroll
            for (int f = 0; f < FA; ++f)
                WG_DMA_B128(rsrc, reinterpret_cast<uint32_t*>(base + ((f * 2 + plane) * KSC + sg) * 1024), vo, frag_so[f]);
        } else {
#pragma unroll
            for (int f = 0; f < FB; ++f)
                WG_DMA_B128(rsrc, reinterpret_cast<uint32_t*>(base + A_BYTES + ((f * 2 + plane) * XP + sg) * 1024), vo, frag_so[f]);
        }
        uint32_t u = pu[q] + adv_u, im = pn[q] + adv_img;
        const bool wrap = u >= (uint32_t)SP;
        pu[q] = wrap ? u - (uint32_t)SP : u;
        pn[q] = wrap ? im + img_bytes : im;
    };

    f32x16 acc[KK][TM][TC];
    f32x16 accb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
#pragma unroll
        for (int t = 0; t < KK; ++t)
#pragma unroll
            for (int j = 0; j < TC; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;
    }
    const bool do_bias = wave_uniform((ct == 0 && wc == 0) ? 1 : 0) != 0;

    // transposed reads (see the one-tap kernel): byte offset of this lane inside a [16 slots][32 ch] k-step block
    const int l16 = lane & 15, sg16 = (lane >> 4) & 1;
    const int lane_rd = (8 * lh + (l16 >> 2)) * 64 + sg16 * 32 + (l16 & 3) * 8;
    int tapoff[KK];      // byte displacement of tap t inside the X rows: (D + (r - 1) Wp + (s - 1)) * 64
#pragma unroll
    for (int t = 0; t < KK; ++t) tapoff[t] = KK > 1 ? (D + (t / KW - PH) * Wp + (t % KW - PW)) * 64 : 0;
    const f16x8 ones = __builtin_bit_cast(f16x8, u32x4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u});

    // (asm reads, see planes.h: the SSN_WAIT_LGKM0 at the head of every step pair is what waits for them)
#define WG_RD_FRAG(dst_, base_, imm_)                                                \
    do {                                                                             \
        u32x2 r0_, r1_;                                                              \
        SSN_DS_READ_TR16_B64_AT(r0_, base_, imm_);                                   \
        SSN_DS_READ_TR16_B64_AT(r1_, base_, (imm_) + 256);                           \
        dst_ = __builtin_bit_cast(f16x8, u32x4{r0_[0], r0_[1], r1_[0], r1_[1]});     \
    } while (0)

#pragma unroll
    for (int q = 0; q < NPMAX; ++q) issue_piece(ck_begin, 0, q);
    int buf = 0;
    trc.mark1(p);
    for (int ck = ck_begin; ck < ck_end; ++ck) {
        SSN_WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();          // chunk ck is complete in `buf`; everybody is done reading the other buffer
        // this wave's fragment rows of this group's k-steps of the chunk
        const unsigned char* ab = lds + buf * STAGE + lane_rd + grp * (KSG * 1024) + wm * (TM * 2 * KSC * 1024);
        const unsigned char* xb = lds + buf * STAGE + lane_rd + grp * (KSG * 1024) + A_BYTES + wc * (TC * 2 * XP * 1024);
        // KSG x KK (k-step, tap) steps per chunk, taken in PAIRS: the three products of a step accumulate into the same registers,
        // and a matrix instruction that waits for its predecessor's result runs at half the pipe's rate -- alternating the
        // products of two steps (two taps = two accumulator sets) keeps consecutive instructions independent.  The X fragments
        // run one pair ahead of the MFMAs (four rotating register sets), the dY fragments one k-step ahead; the fetch of the
        // next chunk goes out one piece per pair, its address arithmetic in the shadow of the MFMAs.
        constexpr int NSTEP = KSG * KK, NPAIR = NSTEP / 2;
        static_assert(NSTEP % 2 == 0, "steps come in pairs");
        constexpr int NA = KK == 1 ? 4 : 2;    // 1x1: every step is its own k-step
        f16x8 af[NA][2][TM], bf[4][2][TC];     // af[set][plane][i], bf[set][plane][j]
        auto read_b = [&](int step, int set) {
            const int ks = step / KK, t = step % KK;
            const unsigned char* xt = xb + tapoff[t];
#pragma unroll
            for (int pn_ = 0; pn_ < 2; ++pn_)
#pragma unroll
                for (int j = 0; j < TC; ++j) WG_RD_FRAG(bf[set][pn_][j], xt, (j * 2 + pn_) * XP * 1024 + ks * 1024);
        };
        auto read_a = [&](int ks, int set) {
#pragma unroll
            for (int pn_ = 0; pn_ < 2; ++pn_)
#pragma unroll
                for (int i = 0; i < TM; ++i) WG_RD_FRAG(af[set][pn_][i], ab, ((i * 2 + pn_) * KSC + ks) * 1024);
        };
        read_a(0, 0);
        if (KK == 1) read_a(1, 1);
        read_b(0, 0);
        read_b(1, 1);
#pragma unroll
        for (int pr = 0; pr < NPAIR; ++pr) {
            const int s0 = 2 * pr, s1 = s0 + 1;
            const int ks0 = s0 / KK, t0 = s0 % KK, ks1 = s1 / KK, t1 = s1 % KK;
            SSN_WAIT_LGKM0();                       // the fragments of this pair (read during the previous one) are in
            __builtin_amdgcn_sched_barrier(0);
            if (KK == 1) {
                if (ks0 + 2 < KSG) read_a(ks0 + 2, (ks0 + 2) % NA);
                if (ks1 + 2 < KSG) read_a(ks1 + 2, (ks1 + 2) % NA);
            } else {
                if (t0 == 1 && ks0 + 1 < KSG) read_a(ks0 + 1, (ks0 + 1) % NA);
                if (t1 == 1 && ks1 + 1 < KSG) read_a(ks1 + 1, (ks1 + 1) % NA);
            }
            if (s0 + 2 < NSTEP && !(WG_DBG(2) && (s0 + 2) % 3)) read_b(s0 + 2, (s0 + 2) % 4);
            if (s1 + 2 < NSTEP && !(WG_DBG(2) && (s1 + 2) % 3)) read_b(s1 + 2, (s1 + 2) % 4);
            // (all pieces within the first third of the chunk: what bounds the lone-workgroup layers is the fetch itself)
            constexpr int PPP = (NPMAX + 5) / 6;
#pragma unroll
            for (int e = 0; e < PPP; ++e)
                if (pr * PPP + e < NPMAX) issue_piece(ck + 1, buf ^ 1, pr * PPP + e);
            constexpr int PA[3] = {1, 0, 0};   // g_lo x_hi + g_hi x_hi + g_hi x_lo
            constexpr int PB[3] = {0, 0, 1};
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TC; ++j) {
                        acc[t0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks0 % NA][PA[c]][i], bf[s0 % 4][PB[c]][j], acc[t0][i][j],
                                                                               0, 0, 0);
                        acc[t1][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks1 % NA][PA[c]][i], bf[s1 % 4][PB[c]][j], acc[t1][i][j],
                                                                               0, 0, 0);
                    }
            if (do_bias) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int ks = e ? ks1 : ks0, t = e ? t1 : t0;
                    if (t != KK - 1) continue;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks % NA][0][i], ones, accb[i], 0, 0, 0);
                        accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks % NA][1][i], ones, accb[i], 0, 0, 0);
                    }
                }
            }
        }
        if (NPMAX > NPAIR * ((NPMAX + 5) / 6)) {
#pragma unroll
            for (int q = NPAIR * ((NPMAX + 5) / 6); q < NPMAX; ++q) issue_piece(ck + 1, buf ^ 1, q);
        }
        buf ^= 1;
    }
    SSN_WAIT_VMCNT(0);
    trc.mark2(p);

    const float inv = 1.f / (*p.g_scale * *p.x_scale);
    float* out = p.part + ((long)z * KG + grp) * p.M * p.ldp;
#pragma unroll
    for (int j = 0; j < TC; ++j) {
        const int ci = c0 + (wc * TC + j) * 32 + li;
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.M) {
#pragma unroll
                    for (int t = 0; t < KK; ++t) out[(long)m * p.ldp + t * p.Cin + ci] = acc[t][i][j][r] * inv;   // tap-major: lanes = consecutive ci
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
    trc.end(p);
}

template <int KK, int TM, int TC, int XP, int KG = 1>
__global__ __launch_bounds__(256 * KG, (KG > 1 || KK * TM * TC >= 18 || TM * TC >= 4) ? 1 : 2) void wgrad_pl9_kernel(WgPlArgs p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[wgpl9_lds_bytes<TM, TC, XP>()];
    wgrad_pl9_body<KK, TM, TC, XP, KG>(p, blockIdx.x, lds);
}


// ---------------------------------------------------------------------------------------------------------------------------
// GROUPED launches: every weight gradient of a backward pass in <= 5 launches (BN-Inception; 7 kernel families in all) + one reduction.
//
// The weight gradients of a pass are mutually independent, and launched one by one each of them has to fill 256 CUs on its own:
// split-K factors of 20 - 130, i.e. workgroups that run 8 - 12 chunks and then store a 150 KB partial slab (2.2 GB of slabs per
// step at the bench batch, half of the family's time in ramps, prologues, epilogues and the reduction: profiles/r4_layer_efficiency.txt).
// Here the problems of a pass are gathered in a DEVICE-RESIDENT TABLE and each kernel family runs ONE grid over all of its problems:
//   * an item = (problem, output tile, share of the reduction range); the items of a problem are consecutive blocks, the problems
//     are ordered longest item first, so the hardware's in-order block dispatch is a longest-first list schedule: slots freed by
//     short items are refilled by the next problem's -- no ramp, no partly empty last round per layer;
//   * the reduction range of a problem is split only as far as the WHOLE group needs to fill the slots a few times over
//     (plan_group: item length ~ sqrt(2 x fixed cost x work per slot)): 3 - 5 x fewer partial slabs, the prologue / epilogue of a
//     workgroup amortised over 3 - 5 x more chunks;
//   * the bodies are the single-launch kernels' (wgrad_pl9_body / wgrad_pl_body), selected per problem by a wave-uniform switch;
//     the partial slabs are reduced in the fixed order of ssn_wgrad_reduce_multi: deterministic, bit-identical from call to call.
// Families (one launch each, only if it has problems): nine-tap 64 x 64 tiles with XP = 6 / 8 / 12 (rows <= 14 / <= 30 / <= 56
// pixels; 80 / 96 / 128 KiB of LDS), everything else (1x1, stride-2, rectangular taps) on the one-tap / chunked 1x1 bodies, and the
// space-to-depth stem (4 x 4 taps on 16-channel sub-blocks: wgrad_stem_body); Inception-v3 adds the seven-tap forms of the nine-tap body
// for its 1 x 7 / 7 x 1 layers.
constexpr int WGG_MAX = 96;             // problems per grouped launch (block -> problem search table travels by value)
constexpr int WGG_WRITE = 12;           // table entries written per plan-write launch (by-value kernel arguments: < 4 KiB)
struct WgGroupEntry {
    WgPlArgs a;
    int variant;                        // body selector inside the family
    uint32_t nblk;                      // blocks of the problem (the blocks up to the next problem's first one exit)
};
struct WgGroupIndex {
    int count;
    int blk0[WGG_MAX + 1];              // first block of problem i (multiples of 8: the XCD remap of a problem keeps its meaning)
};
struct WgGroupChunk {
    WgGroupEntry e[WGG_WRITE];
    int count;
};
__global__ __launch_bounds__(64) void wgrad_group_write_kernel(WgGroupChunk c, WgGroupEntry* table, int first) {
    if ((int)threadIdx.x < c.count) table[first + threadIdx.x] = c.e[threadIdx.x];
}
__device__ __forceinline__ int group_problem_of_block(const WgGroupIndex& ix, int b) {
    int lo = 0, hi = ix.count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ix.blk0[mid] <= b) lo = mid; else hi = mid - 1;
    }
    return wave_uniform(lo);
}

// nine-tap family: 64 x 64 tiles; KG = 2 where the LDS footprint leaves one workgroup per CU
template <int XP, int KG, int KH = 3, int KW = 3>
__global__ __launch_bounds__(256 * KG, KG > 1 ? 1 : 2) void wgrad_group9_kernel(const WgGroupEntry* __restrict__ table, WgGroupIndex ix) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[wgpl9_lds_bytes<1, 1, XP>()];
    const int e = group_problem_of_block(ix, (int)blockIdx.x);
    const uint32_t bid = blockIdx.x - (uint32_t)ix.blk0[e];
    const auto* ent = SSN_CONST_PTR(WgGroupEntry, table) + e;      // (scalar loads: the problem lives in SGPRs like a kernel argument)
    if (bid >= ent->nblk) return;
    const WgPlArgs p = ent->a;
    wgrad_pl9_body<KH * KW, 1, 1, XP, KG, KH, KW, KH / 2, KW / 2>(p, bid, lds);
}

// everything else: one-tap bodies 128 x 128 / 96 x 128 / 64 x 64 (any taps, stride, padding) and the chunked 1x1 body 64 x 64
constexpr int WGG1_LDS = 65536;
static_assert(wgpl_lds_dwords<2, 2, 2, 2>() * 4 <= WGG1_LDS && wgpl_lds_dwords<1, 4, 3, 1>() * 4 <= WGG1_LDS &&
              wgpl_lds_dwords<2, 2, 1, 1>() * 4 <= WGG1_LDS && wgpl9_lds_bytes<1, 1, 4>() <= WGG1_LDS, "group LDS");
__global__ __launch_bounds__(256, 2) void wgrad_group1_kernel(const WgGroupEntry* __restrict__ table, WgGroupIndex ix) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[WGG1_LDS];
    const int e = group_problem_of_block(ix, (int)blockIdx.x);
    const uint32_t bid = blockIdx.x - (uint32_t)ix.blk0[e];
    const auto* ent = SSN_CONST_PTR(WgGroupEntry, table) + e;
    if (bid >= ent->nblk) return;
    const WgPlArgs p = ent->a;
    switch (wave_uniform(ent->variant)) {
        case 0: wgrad_pl_body<2, 2, 2, 2>(p, bid, reinterpret_cast<uint32_t*>(lds)); break;
        case 1: wgrad_pl_body<1, 4, 3, 1>(p, bid, reinterpret_cast<uint32_t*>(lds)); break;
        case 2: wgrad_pl_body<2, 2, 1, 1>(p, bid, reinterpret_cast<uint32_t*>(lds)); break;
        default: wgrad_pl9_body<1, 1, 1, 4>(p, bid, lds); break;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The STEM: 4 x 4 taps / stride 1 / two padding pixels in front, one behind, on the <= 16-channel sub-blocks of a space-to-depth
// input (conv1 7x7 / 2 of the backbone manifest in its space-to-depth form, planes_exec.py) -- the one layer whose input has too few
// channels for the bodies above: 12 (RGB) or 40 (flow) real channels against 32-channel fragments, sixteen taps.  Rounds 2 - 4 ran it
// on the fp32-layout kernel (conv_wgrad_x6.hip: 0.62 ms, 23.7 VALU per MFMA, fed by a 925 MB fp32 copy of the output gradient and an
// fp32 space-to-depth copy of the frames).  Here a fragment's 32 "channels" are 16 real channels x TWO TAPS: the transposed LDS read
// takes a per-lane address, so the lanes of channels 0-15 read tap A's pixels and the lanes of channels 16-31 tap B's from ONE staged
// copy of X ([padded slot][16 channels], 32-byte rows) -- the nine-tap body's trick (one operand copy, taps = read displacements over a
// padded pixel enumeration: TWO zero columns in front of every row, two zero rows in front of every image, shared with the previous
// row / image) with the tap pairs (r, s) + (r + 2, s): their displacement differs by 2 (W + 2) slots = 4 mod 8 for W = 0 mod 4, so the
// two 128-byte row groups of a 32-lane service group fall into disjoint bank halves.  8 tap pairs x 2 output fragments = 16 fragment
// products per k-step, spread over 4 waves (output fragment x half of the pairs); two wave groups split the k-steps of a chunk.
// A workgroup = (64 output channels) x (one 16-channel sub-block of X) x (share of the padded slots).  Slabs: tap-major columns
// t * Cin + ci (ssn_wgrad_reduce_taps with 16 taps puts them back as dW[m][ci][r][s]).
// X lives in a RING of 1024 padded slots per plane: consecutive chunks of 128 output slots need windows of X that overlap in all but
// 128 slots, so a chunk fetches only its 128 new ones (the first version re-staged the whole 409-slot window per 64-slot chunk: 26 KB
// of X per 16 KB of dY, 27 B / clk / CU -- the L2 -> LDS ceiling, profiles/r4_clock_control.txt -- and ran at 0.52 ms).
constexpr int STEM_NS = 128;       // output slots per chunk (8 k-steps: four per wave group)
constexpr int STEM_DA = 256;       // slots of X kept in front of a chunk (>= 2 (W + 2) + 2, a multiple of the 32-slot DMA piece)
constexpr int STEM_EA = 128;       // ... and behind it (>= W + 3)
constexpr int STEM_RING = 1024;    // >= STEM_DA + 2 STEM_NS + STEM_EA
constexpr int STEM_MAX_W = 114;
constexpr int STEM_LDS = 2 * (2 * 2 * 8 * 1024) + 2 * STEM_RING * 32;
inline bool stem_layer(int kh, int kw, int stride, int pad_h, int pad_w, int H, int W, int Ho, int Wo) {
    return kh == 4 && kw == 4 && stride == 1 && pad_h == 2 && pad_w == 2 && Ho == H && Wo == W && W <= STEM_MAX_W;
}
__device__ __forceinline__ void wgrad_stem_body(const WgPlArgs& p, const uint32_t bid, unsigned char* lds) {
    constexpr int NS = STEM_NS, KSC = NS / 16, KG = 2, KSG = KSC / KG;
    constexpr int A_BYTES = 2 * 2 * KSC * 1024;           // one dY buffer: [frag][plane][128 slots][32 ch]
    constexpr int RING_BYTES = STEM_RING * 32;            // one plane of the X ring: [1024 slots][16 ch]
    constexpr uint32_t RMASK = STEM_RING - 1;
    static_assert(2 * A_BYTES + 2 * RING_BYTES == STEM_LDS, "LDS size");
    static_assert(STEM_DA >= 2 * (STEM_MAX_W + 2) + 2 && STEM_EA >= STEM_MAX_W + 3 && STEM_DA + 2 * NS + STEM_EA <= STEM_RING, "window");
    unsigned char* const xring = lds + 2 * A_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave8 = wave_uniform(tid >> 6);
    const int wave = wave8 & 3, grp = wave8 >> 2;
    const int wm = wave & 1, wc = wave >> 1;              // output fragment, half of the tap pairs
    const int li = lane & 31, lh = lane >> 5;

    const uint32_t tiles = (uint32_t)p.n_mtiles * (uint32_t)p.n_ctiles;
    const uint32_t logical = xcd_remap(bid, tiles * (uint32_t)p.splits);
    uint32_t z, tile, mt, sub;
    fd_divmod(logical, p.div_tiles, z, tile);
    fd_divmod(tile, p.div_ct, mt, sub);
    const int m0 = (int)mt * 64, c0 = (int)sub * 16;
    const int Wp = p.W + 2, SP = (p.H + 2) * Wp;
    const int T = p.N * SP;                               // padded slots that can hold data

    // ---- DMA role: operand (0 = dY, 1 = X) and plane; the pieces of a chunk are split between the two wave groups ----
    const int op = wave >> 1, plane = wave & 1;
    const void *xh = p.x_hi, *xl = p.x_lo, *gh = p.g_hi, *gl = p.g_lo;
    const uint32_t xbytes = p.x_bytes, gbytes = p.g_bytes, xgrp = p.x_grp_bytes, ggrp = p.g_grp_bytes, ximg = p.x_img_bytes, gimg = p.g_img_bytes;
    const __amdgpu_buffer_rsrc_t rsrc = op ? pl_rsrc(plane ? xl : xh, xbytes) : pl_rsrc(plane ? gl : gh, gbytes);
    const uint32_t grp_bytes = op ? xgrp : ggrp;
    const uint32_t img_bytes = op ? ximg : gimg;
    // dY: a piece = 16 slots x 4 channel groups (lane -> slot lane / 4, group lane % 4) of each of the two output fragments;
    // X: a piece = 32 slots x 2 channel groups (lane -> slot lane / 2, group lane % 2) of this sub-block
    const int dsl = op ? (lane >> 1) : (lane >> 2);
    const uint32_t lane_grp = (uint32_t)(op ? (lane & 1) : (lane & 3)) * grp_bytes;
    const uint32_t frag_so0 = (uint32_t)((op ? c0 : m0) / 8) * grp_bytes;
    const uint32_t frag_so1 = (uint32_t)((m0 + 32) / 8) * grp_bytes;   // (dY only)
    const int ck_begin = (int)z * p.ksteps_per_split;
    const int total_ck = (T + NS - 1) / NS;
    int ck_end = ck_begin + p.ksteps_per_split;
    if (ck_end > total_ck) ck_end = total_ck;

    // byte offset of padded slot `sl` (this lane's) inside a plane, or out of range: borders, slots outside [0, T), chunks past the end
    auto slot_offset = [&](int sl, bool live) -> uint32_t {
        const bool in = live && sl >= 0 && sl < T;
        uint32_t n, u;
        fd_divmod((uint32_t)(in ? sl : 0), p.div_hw, n, u);                  // div_hw = SP
        const uint32_t hp = __umulhi(u, p.magic_wp), wp = u - hp * (uint32_t)Wp;
        return (in && hp >= 2u && wp >= 2u) ? n * img_bytes + ((hp - 2u) * (uint32_t)p.W + (wp - 2u)) * 16u + lane_grp : PL_OOB;
    };
    // dY pieces sg = 0 .. 7 of chunk ck -> buffer buf (this wave: its plane, both fragments)
    auto issue_dy = [&](int ck, int buf, int sg) {
        const uint32_t vo = slot_offset(ck * NS + sg * 16 + dsl, ck < ck_end);
        unsigned char* base = lds + buf * A_BYTES;
        WG_DMA_B128(rsrc, reinterpret_cast<uint32_t*>(base + ((0 * 2 + plane) * KSC + sg) * 1024), vo, frag_so0);
        WG_DMA_B128(rsrc, reinterpret_cast<uint32_t*>(base + ((1 * 2 + plane) * KSC + sg) * 1024), vo, frag_so1);
    };
    // the X piece that starts at padded slot s0 (a multiple of 32 from -STEM_DA) -> its place in the ring
    auto issue_x = [&](int s0, bool live) {
        const uint32_t vo = slot_offset(s0 + dsl, live);
        const uint32_t row = (uint32_t)(s0 + STEM_DA) & RMASK;             // (s0 + STEM_DA >= 0, a multiple of 32: no piece wraps)
        WG_DMA_B128(rsrc, reinterpret_cast<uint32_t*>(xring + plane * RING_BYTES + row * 32), vo, frag_so0);
    };

    f32x16 acc[4];
    f32x16 accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const bool do_bias = wave_uniform((sub == 0 && wc == 0) ? 1 : 0) != 0;

    // transposed reads: dY as in the bodies above (64-byte rows); X: 32-byte rows of the ring, the lanes of channels 16-31 displaced
    // to tap B; ring row of this lane's first row piece for output slot 0 (+ chunk and k-step offsets, masked, per read)
    const int l16 = lane & 15, sg16 = (lane >> 4) & 1;
    const int a_rd = (8 * lh + (l16 >> 2)) * 64 + sg16 * 32 + (l16 & 3) * 8;
    int x_row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pi = wc * 4 + i;                                     // pair: column tap pi & 3, row taps (pi >> 2) and (pi >> 2) + 2
        const int tr = (pi >> 2) + 2 * sg16, ts = pi & 3;
        x_row[i] = STEM_DA + (tr - 2) * Wp + (ts - 2) + 8 * lh + (l16 >> 2);
    }
    const unsigned char* const x_lane = xring + (l16 & 3) * 8;
    const f16x8 ones = __builtin_bit_cast(f16x8, u32x4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u});

    // first chunk: its dY pieces and the whole X window [-DA, NS + EA) around it (16 pieces per plane, 8 per wave group)
    if (op == 0) {
#pragma unroll
        for (int q = 0; q < KSC / KG; ++q) issue_dy(ck_begin, 0, q * KG + grp);
    } else {
#pragma unroll
        for (int q = 0; q < (STEM_DA + NS + STEM_EA) / 32 / KG; ++q) issue_x(ck_begin * NS - STEM_DA + (q * KG + grp) * 32, true);
    }
    int buf = 0;
    for (int ck = ck_begin; ck < ck_end; ++ck) {
        SSN_WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();          // chunk ck (dY buffer `buf`, X window) is complete; everybody is done with chunk ck - 1
        const unsigned char* ab = lds + buf * A_BYTES + a_rd + (wm * 2 * KSC + grp * KSG) * 1024;
        const int xs0 = ck * NS + grp * KSG * 16;      // first output slot of this group's k-steps
        f16x8 af[2][2], bf[2][4][2];           // [register set][plane] / [set][pair][plane]
        auto read_set = [&](int ks, int set) {
#pragma unroll
            for (int pn_ = 0; pn_ < 2; ++pn_) WG_RD_FRAG(af[set][pn_], ab, (pn_ * KSC + ks) * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t r0 = (uint32_t)(x_row[i] + xs0 + ks * 16) & RMASK, r1 = (r0 + 4u) & RMASK;
                const unsigned char *q0 = x_lane + r0 * 32, *q1 = x_lane + r1 * 32;
#pragma unroll
                for (int pn_ = 0; pn_ < 2; ++pn_) {
                    u32x2 v0, v1;
                    SSN_DS_READ_TR16_B64_AT(v0, q0, pn_ * RING_BYTES);
                    SSN_DS_READ_TR16_B64_AT(v1, q1, pn_ * RING_BYTES);
                    bf[set][i][pn_] = __builtin_bit_cast(f16x8, u32x4{v0[0], v0[1], v1[0], v1[1]});
                }
            }
        };
        read_set(0, 0);
#pragma unroll
        for (int ks = 0; ks < KSG; ++ks) {
            SSN_WAIT_LGKM0();                  // the fragments of this k-step (read during the previous one) are in
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < KSG) read_set(ks + 1, (ks + 1) & 1);
            // the fetch of the next chunk, spread over the k-steps: dY wave: piece ks of its four (x two fragments); X wave: one of
            // its two new pieces in each of the first two k-steps
            if (op == 0) {
                issue_dy(ck + 1, buf ^ 1, ks * KG + grp);
            } else if (ks < NS / 32 / KG) {
                issue_x((ck + 1) * NS + STEM_EA + (ks * KG + grp) * 32, ck + 1 < ck_end);
            }
            constexpr int PA[3] = {1, 0, 0};   // g_lo x_hi + g_hi x_hi + g_hi x_lo
            constexpr int PB[3] = {0, 0, 1};
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][PA[c]], bf[ks & 1][i][PB[c]], acc[i], 0, 0, 0);
            if (do_bias) {
                accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][0], ones, accb, 0, 0, 0);
                accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][1], ones, accb, 0, 0, 0);
            }
        }
        buf ^= 1;
    }
    SSN_WAIT_VMCNT(0);

    const float inv = 1.f / (*p.g_scale * *p.x_scale);
    float* out = p.part + ((long)z * KG + grp) * p.M * p.ldp;
    const int ci = c0 + (li & 15);
    if (ci < p.Cin) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pi = wc * 4 + i;
            const int t = ((pi >> 2) + 2 * (li >> 4)) * 4 + (pi & 3);       // tap r * 4 + s of this lane's half of the fragment
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.M) out[(long)m * p.ldp + t * p.Cin + ci] = acc[i][r] * inv;
            }
        }
    }
    if (do_bias && li == 0) {
        const float ginv = 1.f / *p.g_scale;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < p.M) out[(long)m * p.ldp + p.K] = accb[r] * ginv;
        }
    }
}

__global__ __launch_bounds__(512, 1) void wgrad_group_stem_kernel(const WgGroupEntry* __restrict__ table, WgGroupIndex ix) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[STEM_LDS];
    const int e = group_problem_of_block(ix, (int)blockIdx.x);
    const uint32_t bid = blockIdx.x - (uint32_t)ix.blk0[e];
    const auto* ent = SSN_CONST_PTR(WgGroupEntry, table) + e;
    if (bid >= ent->nblk) return;
    const WgPlArgs p = ent->a;
    wgrad_stem_body(p, bid, lds);
}

#undef WG_DMA_B128
#undef WG_RD_FRAG

template <int KK, int TM, int TC, int XP, int KG = 1>
int launch_wgpl9(WgPlArgs& a, hipStream_t stream) {
    constexpr int BM = 2 * TM * 32, BC = 2 * TC * 32;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.n_ctiles = (a.Cin + BC - 1) / BC;
    const unsigned tiles = (unsigned)a.n_mtiles * (unsigned)a.n_ctiles;
    a.div_tiles = make_fastdiv(tiles);
    a.div_ct = make_fastdiv((uint32_t)a.n_ctiles);
    a.magic_wp = 0xFFFFFFFFu / (uint32_t)(a.W + (KK > 1 ? 1 : 0)) + 1u;
    hipLaunchKernelGGL((wgrad_pl9_kernel<KK, TM, TC, XP, KG>), dim3(tiles * (unsigned)a.splits), dim3(256 * KG), 0, stream, a);
    SSN_CHECK_LAUNCH("wgrad_pl9");
    return SSN_OK;
}

// nine-tap tile configs (output channels x input channels): 0: 64 x 64   1: 128 x 64   2: 64 x 128
// 3: 64 x 64 with eight waves (two groups splitting the k-steps of each chunk: twice the partial slabs)
constexpr int N9 = 4;
const int k9BM[N9] = {64, 128, 64, 64};
const int k9BC[N9] = {64, 64, 128, 64};
const int k9Occ[N9] = {1, 1, 1, 1};
const int k9KG[N9] = {1, 1, 1, 2};

template <int XP>
int launch_wgpl9_xp(WgPlArgs& a, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_wgpl9<9, 1, 1, XP>(a, stream);
        case 1: return launch_wgpl9<9, 2, 1, XP>(a, stream);
        case 2: return launch_wgpl9<9, 1, 2, XP>(a, stream);
        case 3: return launch_wgpl9<9, 1, 1, XP, 2>(a, stream);
    }
    ssn_set_error("conv_wgrad_pl (nine taps): unknown tile config %d", cfg);
    return SSN_ERR_ARG;
}
// X pieces (16 slots each) per chunk: 64 + 2 (W + 2) slots
int xp_for(int W) { return (64 + 2 * (W + 2) + 15) / 16; }
int launch_wgpl9_tile(WgPlArgs& a, int cfg, hipStream_t stream) {
    const int xp = xp_for(a.W);
    if (xp <= 6) return launch_wgpl9_xp<6>(a, cfg, stream);
    if (xp <= 8) return launch_wgpl9_xp<8>(a, cfg, stream);
    // wide rows: only the 64 x 64 tiles fit two buffers into the LDS
    return cfg == 3 ? launch_wgpl9<9, 1, 1, 12, 2>(a, stream) : launch_wgpl9<9, 1, 1, 12>(a, stream);
}
// chunked 1x1 tile configs (tile_cfg 200 + i): 0: 64 x 64   1: 128 x 64   2: 64 x 128   3: 128 x 128
constexpr int N1 = 4;
const int k1BM[N1] = {64, 128, 64, 128};
const int k1BC[N1] = {64, 64, 128, 128};
const int k1Occ[N1] = {2, 1, 1, 1};
int launch_wgpl1_tile(WgPlArgs& a, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_wgpl9<1, 1, 1, 4>(a, stream);
        case 1: return launch_wgpl9<1, 2, 1, 4>(a, stream);
        case 2: return launch_wgpl9<1, 1, 2, 4>(a, stream);
        case 3: return launch_wgpl9<1, 2, 2, 4>(a, stream);
    }
    ssn_set_error("conv_wgrad_pl (chunked 1x1): unknown tile config %d", cfg);
    return SSN_ERR_ARG;
}
void plan1(int M, int Cin, long slots, int cfg, int* splits, int* chunks_per_split) {
    const long tiles = (long)((M + k1BM[cfg] - 1) / k1BM[cfg]) * ((Cin + k1BC[cfg] - 1) / k1BC[cfg]);
    const long chunks = (slots + 63) / 64;
    plan_split_k(tiles, chunks, k1Occ[cfg], 4, 2, 0.005 + (double)M * Cin * 1.7e-6, splits, chunks_per_split);
}
bool chunked_1x1_layer(int kh, int kw, int stride, int pad_h, int pad_w) {
    return kh == 1 && kw == 1 && stride == 1 && pad_h == 0 && pad_w == 0;
}

int pick_tile9(int M, int Cin, int W) {
    if (xp_for(W) > 8) return 0;
    double best = 1e300;
    int bc = 0;
    for (int c = 0; c < 3; ++c) {
        const double padded = (double)((M + k9BM[c] - 1) / k9BM[c]) * k9BM[c] * (double)((Cin + k9BC[c] - 1) / k9BC[c]) * k9BC[c];
        const double small = c == 0 ? 1.1 : 1.0;
        if (padded * small < best) {
            best = padded * small;
            bc = c;
        }
    }
    return bc;
}
void plan9(int M, int Cin, long slots, int cfg, int W, int* splits, int* chunks_per_split) {
    const long tiles = (long)((M + k9BM[cfg] - 1) / k9BM[cfg]) * ((Cin + k9BC[cfg] - 1) / k9BC[cfg]);
    const long chunks = (slots + 63) / 64;
    // the 64 x 64 tile on rows of <= 14 pixels: 80 KiB of LDS, two workgroups per CU
    const int occ = (cfg == 0 && xp_for(W) <= 6) ? 2 : k9Occ[cfg];     // (config 3: one workgroup of eight waves)
    plan_split_k(tiles, chunks, occ, 4, 2, 0.005 + (double)M * Cin * 9 * 1.7e-6, splits, chunks_per_split);
}
bool nine_tap_layer(int kh, int kw, int stride, int pad_h, int pad_w, int H, int W, int Ho, int Wo) {
    return kh == 3 && kw == 3 && stride == 1 && pad_h == 1 && pad_w == 1 && Ho == H && Wo == W && W + 2 <= 58;
}

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

unsigned long long* g_wg_trace = nullptr;
int g_wg_dbg = 0;

// the shape / operand fields of a problem (everything but the tile and split plan)
int fill_wgpl(WgPlArgs& a, const void* g_hi, const void* g_lo, const void* x_hi, const void* x_lo, int N, int Cin, int H, int W,
              long x_img_groups, int Cout, int Ho, int Wo, long g_img_groups, int kh, int kw, int stride, int pad_h, int pad_w,
              const float* g_scale, const float* x_scale, int g_row_split, int g_row_gap) {
    SSN_CHECK_ARG(g_hi && g_lo && x_hi && x_lo && g_scale && x_scale, "conv wgrad pl: null pointer");
    SSN_CHECK_ARG(Cout > 0 && Cin > 0 && kh >= 1 && kw >= 1 && (stride == 1 || stride == 2), "conv wgrad pl: bad shape");
    SSN_CHECK_ARG(g_row_gap >= 0 && (g_row_gap == 0 || (g_row_split > 0 && g_row_split < Cout && g_row_split % 32 == 0 && g_row_gap % 8 == 0)),
                  "conv wgrad pl: bad row split");
    a.g_hi = g_hi;
    a.g_lo = g_lo;
    a.x_hi = x_hi;
    a.x_lo = x_lo;
    a.part = nullptr;
    a.trace = g_wg_trace;
    a.dbg = g_wg_dbg;
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
    // the descriptors end with the SLICES in the last image (a fragment may reach past the slice -- rows / columns that are never
    // stored: in earlier images it reads the neighbouring channels, behind the last image it must read nothing: a slice at the end
    // of its tensor would otherwise be over-read past the allocation)
    a.g_bytes = (uint32_t)((long)(N - 1) * a.g_img_bytes + (long)((Cout + g_row_gap + 7) / 8) * gg);
    a.x_bytes = (uint32_t)((long)(N - 1) * a.x_img_bytes + (long)((Cin + 7) / 8) * xg);
    a.div_hw = make_fastdiv((uint32_t)(Ho * Wo));
    a.div_w = make_fastdiv((uint32_t)Wo);
    a.magic_wp = 0;
    a.splits = 1;
    a.ksteps_per_split = 0;
    a.n_mtiles = a.n_ctiles = 0;
    a.div_tiles = a.div_ct = a.div_kk = make_fastdiv(1u);
    return SSN_OK;
}

// ---- grouped launches (see wgrad_group9_kernel): host side ----
// planner constants {nine-tap families, one-tap / chunked family}: fixed cost of an item in matrix-instruction slots of a wave
// (prologue + first fetch + partial-slab store + its share of the reduction), fewest reduction units an item may have
double g_group_fixed[2] = {450.0, 150.0};
int g_group_min_units[2] = {4, 32};
enum { WGF_9_XP6 = 0, WGF_9_XP8 = 1, WGF_9_XP12 = 2, WGF_1 = 3, WGF_STEM = 4, WGF_7_ROW = 5, WGF_7_COL = 6, WGF_COUNT = 7 };
// one-tap / chunked variants of the WGF_1 family (3 = the chunked 1x1 body): tile = output x input channels.  (A big-tile family --
// 256 x 128 / 128 x 256 one-tap, 128 x 128 chunked, one workgroup per CU -- was measured in the group and lost on every layer:
// profiles/r5_wgrad_group_variants.txt.)
const int kG1BM[4] = {128, 96, 64, 64};
const int kG1BC[4] = {128, 128, 64, 64};
struct WgGroupItem {
    WgPlArgs a;
    int family, variant;
    long units;            // reduction range in the body's own units: k-steps (one-tap) or 64-slot chunks (chunked bodies)
    double unit_cost;      // matrix instructions per wave and unit (what an item's length is measured in)
    long tiles;            // output tiles (x taps for the one-tap bodies)
    bool chunked;          // WGF_1: the chunked 1x1 body (units = 64-slot chunks, any count per share)
    int kg;                // partial slabs per split
    int taps;              // tap-major slabs (nine-tap bodies): 9, else 1
    long ws_off;           // byte offset of the problem's slabs in the workspace
    float* dw;
    float* db;
};
// family + variant of a problem; hint: a single-launch tile_cfg (0, 3, 8: one-tap 64 x 64 / 128 x 128 / 96 x 128; 200: chunked 1x1) or -1
int classify_group(WgGroupItem& it, int hint) {
    WgPlArgs& a = it.a;
    it.kg = 1;
    it.taps = 1;
    it.chunked = false;
    if ((hint < 0 || hint == 300) && stem_layer(a.kh, a.kw, a.stride, a.pad_h, a.pad_w, a.H, a.W, a.Ho, a.Wo)) {
        // the space-to-depth stem: (64 output channels) x (16-channel sub-block of X) tiles, sixteen taps as eight tap pairs
        it.family = WGF_STEM;
        it.variant = 0;
        it.kg = 2;
        it.taps = 16;
        const int Wp = a.W + 2, SP = (a.H + 2) * Wp;
        it.units = ((long)a.N * SP + STEM_NS - 1) / STEM_NS;
        it.unit_cost = 4 * 4 * 3;
        a.n_mtiles = (a.M + 63) / 64;
        a.n_ctiles = (a.Cin + 15) / 16;
        it.tiles = (long)a.n_mtiles * a.n_ctiles;
        a.div_tiles = make_fastdiv((uint32_t)it.tiles);
        a.div_ct = make_fastdiv((uint32_t)a.n_ctiles);
        a.magic_wp = 0xFFFFFFFFu / (uint32_t)Wp + 1u;
        a.div_hw = make_fastdiv((uint32_t)SP);
        a.div_w = make_fastdiv((uint32_t)Wp);
        return SSN_OK;
    }
    SSN_CHECK_ARG(hint != 300, "conv wgrad pl group: the stem body takes 4x4 / stride 1 / pad 2 layers on rows of <= %d pixels only", STEM_MAX_W);
    if (hint < 0 || (hint >= 100 && hint < 200)) {
        if (nine_tap_layer(a.kh, a.kw, a.stride, a.pad_h, a.pad_w, a.H, a.W, a.Ho, a.Wo)) {
            const int xp = xp_for(a.W);
            it.family = xp <= 6 ? WGF_9_XP6 : (xp <= 8 ? WGF_9_XP8 : WGF_9_XP12);
            it.variant = 0;
            it.kg = it.family == WGF_9_XP6 ? 1 : 2;
            it.taps = 9;
            const long T = (long)a.N * (a.H + 1) * (a.W + 1);
            it.units = (T + 63) / 64;
            it.unit_cost = 4 * 9 * 3 / (double)it.kg;
            a.n_mtiles = (a.M + 63) / 64;
            a.n_ctiles = (a.Cin + 63) / 64;
            it.tiles = (long)a.n_mtiles * a.n_ctiles;
            a.div_tiles = make_fastdiv((uint32_t)it.tiles);
            a.div_ct = make_fastdiv((uint32_t)a.n_ctiles);
            a.magic_wp = 0xFFFFFFFFu / (uint32_t)(a.W + 1) + 1u;
            a.div_hw = make_fastdiv((uint32_t)((a.H + 1) * (a.W + 1)));
            a.div_w = make_fastdiv((uint32_t)(a.W + 1));
            return SSN_OK;
        }
        // Inception-v3's 17 x 17 stage: 1 x 7 / pad (0, 3) and 7 x 1 / pad (3, 0) on the same body with seven taps
        const bool same = a.stride == 1 && a.Ho == a.H && a.Wo == a.W;
        const bool row7 = same && a.kh == 1 && a.kw == 7 && a.pad_h == 0 && a.pad_w == 3;
        const bool col7 = same && a.kh == 7 && a.kw == 1 && a.pad_h == 3 && a.pad_w == 0 && 64 + 6 * a.W <= 12 * 16;
        if (row7 || col7) {
            it.family = row7 ? WGF_7_ROW : WGF_7_COL;
            it.variant = 0;
            it.kg = row7 ? 1 : 2;
            it.taps = 7;
            const int Wp = a.W + (row7 ? 3 : 0), Hp = a.H + (row7 ? 0 : 3);
            it.units = ((long)a.N * Hp * Wp + 63) / 64;
            it.unit_cost = 4 * 7 * 3 / (double)it.kg;
            a.n_mtiles = (a.M + 63) / 64;
            a.n_ctiles = (a.Cin + 63) / 64;
            it.tiles = (long)a.n_mtiles * a.n_ctiles;
            a.div_tiles = make_fastdiv((uint32_t)it.tiles);
            a.div_ct = make_fastdiv((uint32_t)a.n_ctiles);
            a.magic_wp = 0xFFFFFFFFu / (uint32_t)Wp + 1u;
            a.div_hw = make_fastdiv((uint32_t)(Hp * Wp));
            a.div_w = make_fastdiv((uint32_t)Wp);
            return SSN_OK;
        }
        SSN_CHECK_ARG(hint < 0, "conv wgrad pl group: the multi-tap bodies take 3x3 / pad 1, 1x7 / pad (0,3) and 7x1 / pad (3,0) stride-1 layers only");
    }
    it.family = WGF_1;
    const bool can_chunk = chunked_1x1_layer(a.kh, a.kw, a.stride, a.pad_h, a.pad_w);
    int v;      // index into kG1BM / kG1BC
    if (hint >= 200) {
        SSN_CHECK_ARG(can_chunk, "conv wgrad pl group: the chunked body takes 1x1 / stride-1 layers only");
        SSN_CHECK_ARG(hint == 200, "conv wgrad pl group: chunked tile %d is not compiled into the grouped kernels", hint);
        v = 3;
    } else if (hint == 3) {
        v = 0;
    } else if (hint == 8) {
        v = 1;
    } else if (hint == 0) {
        v = 2;
    } else {
        SSN_CHECK_ARG(hint < 0, "conv wgrad pl group: tile %d is not compiled into the grouped kernels", hint);
        if (can_chunk && a.M <= 128) {
            v = 3;      // (what the per-layer table picks for the narrow 1x1 layers: 64 -> 64 at 56 x 56, 1024 -> 128 at 7 x 7)
        } else {
            double best = 1e300;
            v = 0;
            for (int c = 0; c < 3; ++c) {
                const double padded = (double)((a.M + kG1BM[c] - 1) / kG1BM[c]) * kG1BM[c] * (double)((a.Cin + kG1BC[c] - 1) / kG1BC[c]) * kG1BC[c];
                const double small = c == 0 ? 1.0 : (c == 1 ? 1.04 : 1.5);      // (operand fetch per MAC grows as the tile shrinks)
                if (padded * small < best) {
                    best = padded * small;
                    v = c;
                }
            }
        }
    }
    it.variant = v;
    a.n_mtiles = (a.M + kG1BM[v] - 1) / kG1BM[v];
    a.n_ctiles = (a.Cin + kG1BC[v] - 1) / kG1BC[v];
    a.div_ct = make_fastdiv((uint32_t)a.n_ctiles);
    it.chunked = v == 3;
    if (it.chunked) {
        it.tiles = (long)a.n_mtiles * a.n_ctiles;
        it.units = ((long)a.P + 63) / 64;
        it.unit_cost = 4 * 3.0 * (kG1BM[v] / 64) * (kG1BC[v] / 64);
        a.magic_wp = 0xFFFFFFFFu / (uint32_t)a.W + 1u;
        a.div_hw = make_fastdiv((uint32_t)(a.H * a.W));
        a.div_w = make_fastdiv((uint32_t)a.W);
    } else {
        const int KK = a.kh * a.kw;
        it.tiles = (long)a.n_mtiles * a.n_ctiles * KK;
        it.units = ((long)a.P + 15) / 16;
        it.unit_cost = 3.0 * (kG1BM[v] / 32) * (kG1BC[v] / 32) / 4;
        a.div_kk = make_fastdiv((uint32_t)KK);
    }
    a.div_tiles = make_fastdiv((uint32_t)it.tiles);
    return SSN_OK;
}
// Split plan of a family's problems.  Model: an item costs (its units x unit_cost + F) matrix-instruction slots of a wave, F = the
// fixed part (prologue, first fetch, partial-slab store, its share of the reduction); `slots` items run at a time; list scheduling
// of n items of length L on S slots ends within L of the mean load.  Total ~ W / S x (1 + F / L) + L / 2, minimal at L = sqrt(2 F W / S).
void plan_group(std::vector<WgGroupItem*>& fam, int slots, double fixed_cost, int min_units, bool even_units) {
    double work = 0;
    for (WgGroupItem* it : fam) work += (double)it->tiles * (double)it->units * it->unit_cost;
    double target = std::sqrt(2.0 * fixed_cost * work / slots);
    if (target < fixed_cost) target = fixed_cost;
    for (WgGroupItem* it : fam) {
        long per = (long)(target / it->unit_cost + 0.5);
        const long floor_units = (it->family >= WGF_1 && it->chunked) ? (min_units + 3) / 4 : min_units;   // (a chunk = 4 k-steps)
        if (per < floor_units) per = floor_units;
        if (per > it->units) per = it->units;
        long splits = (it->units + per - 1) / per;
        per = (it->units + splits - 1) / splits;           // (equal shares)
        if (even_units && !it->chunked && (per & 1)) ++per;       // the one-tap pipeline runs two k-steps per trip
        splits = (it->units + per - 1) / per;
        it->a.splits = (int)splits;
        it->a.ksteps_per_split = (int)per;
    }
}

bool group1_interleave() {
    static const bool on = [] {
        const char* e = std::getenv("SSN_WGRAD_INTERLEAVE");      // (tooling: 0 = the round-5 order, longest item first)
        return !(e && e[0] == '0');
    }();
    return on;
}

}  // namespace

// tooling (tools/trace_wgrad_pl.py): per-block phase stamps of the next launches into buf (8 qwords per block), null = off
extern "C" void ssn_conv_wgrad_pl_debug_trace(unsigned long long* buf) { g_wg_trace = buf; }
extern "C" void ssn_conv_wgrad_pl_debug_flags(int flags) { g_wg_dbg = flags; }

extern "C" int ssn_wgrad_reduce(const float* part, float* dw, float* db, int M, int K, int splits, hipStream_t stream);
extern "C" int ssn_wgrad_reduce_taps(const float* part, float* dw, float* db, int M, int K, int splits, int taps, hipStream_t stream);

extern "C" int ssn_conv_wgrad_pl_tiles(void) { return NCFG; }

extern "C" long ssn_conv_wgrad_pl_workspace_bytes(int N, int Cin, int Cout, int Ho, int Wo, int kh, int kw, int tile_cfg) {
    if (tile_cfg >= 200) {
        int s1, c1;
        plan1(Cout, Cin, (long)N * Ho * Wo, (tile_cfg - 200) < N1 ? tile_cfg - 200 : 0, &s1, &c1);
        return (long)s1 * Cout * ((long)Cin + 1) * (long)sizeof(float);
    }
    if (tile_cfg >= 100 || (tile_cfg < 0 && kh == 3 && kw == 3)) {
        // (nine-tap kernel; a 3x3 layer that turns out not to qualify at launch needs at most the one-tap kernel's slabs below)
        const int c9 = tile_cfg >= 100 ? tile_cfg - 100 : pick_tile9(Cout, Cin, Wo);
        int splits9, cps9;
        plan9(Cout, Cin, (long)N * (Ho + 1) * (Wo + 1), c9 < N9 ? c9 : 0, Wo, &splits9, &cps9);
        const long need9 = (long)splits9 * k9KG[c9 < N9 ? c9 : 0] * Cout * ((long)Cin * 9 + 1) * (long)sizeof(float);
        if (tile_cfg >= 100) return need9;
        const int cfg1 = fix_cfg(-1, Cout, Cin);
        int s1, k1;
        plan(Cout, Cin, 9, (long)N * Ho * Wo, cfg1, &s1, &k1);
        const long need1 = (long)s1 * Cout * ((long)Cin * 9 + 1) * (long)sizeof(float);
        return need9 > need1 ? need9 : need1;
    }
    const int cfg = fix_cfg(tile_cfg, Cout, Cin);
    int splits, kps;
    plan(Cout, Cin, kh * kw, (long)N * Ho * Wo, cfg, &splits, &kps);
    return (long)splits * Cout * ((long)Cin * kh * kw + 1) * (long)sizeof(float);
}

// ---- grouped weight gradients: ALL weight (+ bias) gradients of a backward pass in <= 5 launches + one reduction -------------------
// (replaces the per-layer cuDNN wgrad calls behind loss.backward(), /root/reference/ssn_train.py:236).  Problem i is described like
// the arguments of ssn_conv_wgrad_pl: plane pointers g_hi/g_lo/x_hi/x_lo[i], dw[i], db[i] (may be null), g_scale / x_scale[i],
// shape[i * 16 + ...] = {N, Cin, H, W, Cout, Ho, Wo, kh, kw, stride, pad_h, pad_w, g_row_split, g_row_gap, tile hint (-1: choose;
// 0 / 3 / 8: one-tap 64 x 64 / 128 x 128 / 96 x 128; 100: nine taps; 200: chunked 1x1; 300: the space-to-depth stem), 0}, groups[i * 2 + ...] = {x_img_groups,
// g_img_groups}.  workspace: ssn_conv_wgrad_pl_group_workspace_bytes() bytes (partial slabs of every problem, each in its own
// region); table: ssn_conv_wgrad_pl_group_table_bytes(count) bytes of device memory the launches read their problems from (written
// by this call, every call: operand addresses change from pass to pass).  Nothing is synchronised; capturable.
namespace {
int plan_group_all(int count, const int* shape, const long* groups, const void* const* g_hi, const void* const* g_lo,
                   const void* const* x_hi, const void* const* x_lo, float* const* dw, float* const* db, const float* const* g_scale,
                   const float* const* x_scale, std::vector<WgGroupItem>& items, long* ws_total) {
    items.resize((size_t)count);
    static const float one = 1.f;
    for (int i = 0; i < count; ++i) {
        const int* sh = shape + (size_t)i * 16;
        WgGroupItem& it = items[(size_t)i];
        // (planning only: no pointers given -- any non-null value passes the checks)
        const void* dummy = &one;
        int rc = fill_wgpl(it.a, g_hi ? g_hi[i] : dummy, g_lo ? g_lo[i] : dummy, x_hi ? x_hi[i] : dummy, x_lo ? x_lo[i] : dummy, sh[0], sh[1],
                           sh[2], sh[3], groups[i * 2], sh[4], sh[5], sh[6], groups[i * 2 + 1], sh[7], sh[8], sh[9], sh[10], sh[11],
                           g_scale ? g_scale[i] : &one, x_scale ? x_scale[i] : &one, sh[12], sh[13]);
        if (rc != SSN_OK) return rc;
        it.dw = dw ? dw[i] : nullptr;
        it.db = db ? db[i] : nullptr;
        rc = classify_group(it, sh[14]);
        if (rc != SSN_OK) return rc;
    }
    for (int f = 0; f < WGF_COUNT; ++f) {
        std::vector<WgGroupItem*> fam;
        for (WgGroupItem& it : items)
            if (it.family == f) fam.push_back(&it);
        if (fam.empty()) continue;
        if (f == WGF_1)
            plan_group(fam, 512, g_group_fixed[1], g_group_min_units[1], true);
        else
            plan_group(fam, (f == WGF_9_XP6 || f == WGF_7_ROW) ? 512 : 256, g_group_fixed[0], g_group_min_units[0], false);   // (the stem too)
    }
    long off = 0;
    for (WgGroupItem& it : items) {
        it.ws_off = off;
        off += ((long)it.a.splits * it.kg * it.a.M * it.a.ldp * (long)sizeof(float) + 1023) / 1024 * 1024;
    }
    *ws_total = off;
    return SSN_OK;
}
}  // namespace

// tooling / tests: planner constants of the grouped launches (values <= 0 keep the current one); see plan_group
extern "C" void ssn_conv_wgrad_pl_group_tuning(double fixed_nine, double fixed_one, int min_units_nine, int min_units_one) {
    if (fixed_nine > 0) g_group_fixed[0] = fixed_nine;
    if (fixed_one > 0) g_group_fixed[1] = fixed_one;
    if (min_units_nine > 0) g_group_min_units[0] = min_units_nine;
    if (min_units_one > 0) g_group_min_units[1] = min_units_one;
}
extern "C" long ssn_conv_wgrad_pl_group_table_bytes(int count) { return (long)(count > 0 ? count : 1) * (long)sizeof(WgGroupEntry); }

// workspace bytes of a group (the same planner as the launch); plan_out (optional): [count][4] = {family, variant, splits, units per split}
extern "C" long ssn_conv_wgrad_pl_group_workspace_bytes(int count, const int* shape, const long* groups, int* plan_out) {
    if (count <= 0 || !shape || !groups) return 0;
    std::vector<WgGroupItem> items;
    long total = 0;
    if (plan_group_all(count, shape, groups, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, items, &total) != SSN_OK)
        return -1;
    if (plan_out)
        for (int i = 0; i < count; ++i) {
            plan_out[i * 4] = items[(size_t)i].family;
            plan_out[i * 4 + 1] = items[(size_t)i].variant;
            plan_out[i * 4 + 2] = items[(size_t)i].a.splits;
            plan_out[i * 4 + 3] = items[(size_t)i].a.ksteps_per_split;
        }
    return total;
}

extern "C" int ssn_wgrad_reduce_multi(int count, const float* const* part, float* const* dw, float* const* db, const int* M,
                                      const int* K, const int* splits, const int* taps, hipStream_t stream);

extern "C" int ssn_conv_wgrad_pl_group(int count, const void* const* g_hi, const void* const* g_lo, const void* const* x_hi,
                                       const void* const* x_lo, float* const* dw, float* const* db, const int* shape,
                                       const long* groups, const float* const* g_scale, const float* const* x_scale, void* workspace,
                                       long ws_bytes, void* table, long table_bytes, hipStream_t stream) {
    if (count == 0) return SSN_OK;
    SSN_CHECK_ARG(count > 0 && g_hi && g_lo && x_hi && x_lo && dw && db && shape && groups && g_scale && x_scale && workspace && table,
                  "conv wgrad pl group: null pointer");
    for (int i = 0; i < count; ++i) SSN_CHECK_ARG(dw[i], "conv wgrad pl group: problem %d has no dw", i);
    std::vector<WgGroupItem> items;
    long need = 0;
    int rc = plan_group_all(count, shape, groups, g_hi, g_lo, x_hi, x_lo, dw, db, g_scale, x_scale, items, &need);
    if (rc != SSN_OK) return rc;
    if (ws_bytes < need) {
        ssn_set_error("conv wgrad pl group: workspace %ld < %ld bytes", ws_bytes, need);
        return SSN_ERR_WORKSPACE;
    }
    if (table_bytes < ssn_conv_wgrad_pl_group_table_bytes(count)) {
        ssn_set_error("conv wgrad pl group: table %ld < %ld bytes", table_bytes, ssn_conv_wgrad_pl_group_table_bytes(count));
        return SSN_ERR_WORKSPACE;
    }
    WgGroupEntry* dev = static_cast<WgGroupEntry*>(table);
    int first = 0;      // table position of the next launch's problems
    for (int f = 0; f < WGF_COUNT; ++f) {
        std::vector<WgGroupItem*> fam;
        for (WgGroupItem& it : items)
            if (it.family == f) fam.push_back(&it);
        if (fam.empty()) continue;
        // longest item first (ties: the caller's order)
        std::stable_sort(fam.begin(), fam.end(), [](const WgGroupItem* x, const WgGroupItem* y) {
            return x->a.ksteps_per_split * x->unit_cost > y->a.ksteps_per_split * y->unit_cost;
        });
        if (f == WGF_1 && group1_interleave()) {
            // One-tap / chunked family: its problems are of two kinds -- the 28 x 28 / 56 x 56 layers sit at their operand floor
            // (0.5 GB for 20 - 30 GFLOP), the 14 x 14 / 7 x 7 ones are matrix work -- and their items are all about equally long, so
            // "longest first" says nothing and happened to put the memory-bound half in front.  Dealt alternately from both ends of
            // the bytes-per-MAC order, a memory-bound problem always runs beside a matrix-bound one (the in-order dispatch keeps
            // ~ 1.5 problems in flight).  The order of the problems does not touch any problem's own summation order.
            auto bytes_per_mac = [](const WgGroupItem* x) {
                const double s2 = (double)x->a.stride * x->a.stride;
                return ((double)x->a.M + (double)x->a.Cin * s2) / ((double)x->a.M * x->a.Cin * x->a.kh * x->a.kw);
            };
            std::stable_sort(fam.begin(), fam.end(), [&](const WgGroupItem* x, const WgGroupItem* y) { return bytes_per_mac(x) > bytes_per_mac(y); });
            std::vector<WgGroupItem*> dealt;
            for (size_t lo = 0, hi = fam.size(); lo < hi;) {
                dealt.push_back(fam[lo++]);
                if (lo < hi) dealt.push_back(fam[--hi]);
            }
            fam.swap(dealt);
        }
        for (size_t base = 0; base < fam.size(); base += WGG_MAX) {
            const int n = (int)std::min<size_t>(WGG_MAX, fam.size() - base);
            WgGroupIndex ix;
            ix.count = n;
            long blocks = 0;
            for (int c0 = 0; c0 < n; c0 += WGG_WRITE) {
                WgGroupChunk ch;
                ch.count = std::min(WGG_WRITE, n - c0);
                for (int j = 0; j < ch.count; ++j) {
                    WgGroupItem& it = *fam[base + (size_t)(c0 + j)];
                    it.a.part = reinterpret_cast<float*>(static_cast<char*>(workspace) + it.ws_off);
                    ch.e[j].a = it.a;
                    ch.e[j].variant = it.variant;
                    ch.e[j].nblk = (uint32_t)(it.tiles * it.a.splits);
                    ix.blk0[c0 + j] = (int)blocks;
                    blocks += (it.tiles * it.a.splits + 7) / 8 * 8;
                }
                hipLaunchKernelGGL(wgrad_group_write_kernel, dim3(1), dim3(64), 0, stream, ch, dev, first + c0);
            }
            SSN_CHECK_ARG(blocks < (1l << 31), "conv wgrad pl group: too many blocks");
            for (int j = n; j <= WGG_MAX; ++j) ix.blk0[j] = (int)blocks;
            const WgGroupEntry* tab = dev + first;
            switch (f) {
                case WGF_9_XP6: hipLaunchKernelGGL((wgrad_group9_kernel<6, 1>), dim3((unsigned)blocks), dim3(256), 0, stream, tab, ix); break;
                case WGF_9_XP8: hipLaunchKernelGGL((wgrad_group9_kernel<8, 2>), dim3((unsigned)blocks), dim3(512), 0, stream, tab, ix); break;
                case WGF_9_XP12: hipLaunchKernelGGL((wgrad_group9_kernel<12, 2>), dim3((unsigned)blocks), dim3(512), 0, stream, tab, ix); break;
                case WGF_STEM: hipLaunchKernelGGL(wgrad_group_stem_kernel, dim3((unsigned)blocks), dim3(512), 0, stream, tab, ix); break;
                case WGF_7_ROW: hipLaunchKernelGGL((wgrad_group9_kernel<6, 1, 1, 7>), dim3((unsigned)blocks), dim3(256), 0, stream, tab, ix); break;
                case WGF_7_COL: hipLaunchKernelGGL((wgrad_group9_kernel<12, 2, 7, 1>), dim3((unsigned)blocks), dim3(512), 0, stream, tab, ix); break;
                default: hipLaunchKernelGGL(wgrad_group1_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, tab, ix); break;
            }
            first += n;
        }
    }
    SSN_CHECK_LAUNCH("conv_wgrad_pl_group");
    // the reduction of every problem's slabs, in the fixed order of the single launches
    std::vector<const float*> parts((size_t)count);
    std::vector<float*> dws((size_t)count), dbs((size_t)count);
    std::vector<int> Ms((size_t)count), Ks((size_t)count), sp((size_t)count), tp((size_t)count);
    for (int i = 0; i < count; ++i) {
        const WgGroupItem& it = items[(size_t)i];
        parts[(size_t)i] = reinterpret_cast<const float*>(static_cast<char*>(workspace) + it.ws_off);
        dws[(size_t)i] = it.dw;
        dbs[(size_t)i] = it.db;
        Ms[(size_t)i] = it.a.M;
        Ks[(size_t)i] = it.a.K;
        sp[(size_t)i] = it.a.splits * it.kg;
        tp[(size_t)i] = it.taps;
    }
    return ssn_wgrad_reduce_multi(count, parts.data(), dws.data(), dbs.data(), Ms.data(), Ks.data(), sp.data(), tp.data(), stream);
}

// Weight + bias gradient on planes slices: g = dY [N, Cout, Ho, Wo] (final: its ReLU / BN backward applied), x = the layer's
// input [N, Cin (padded to 8), H, W]; any stride / padding / kh x kw taps.  dw [Cout][Cin][kh][kw] fp32, db [Cout] or null.
// *_img_groups: channel groups of the whole tensors.  g_row_gap > 0: rows >= g_row_split of g sit g_row_gap channels further
// up its tensor (fused block-input launch).
extern "C" int ssn_conv_wgrad_pl(const void* g_hi, const void* g_lo, const void* x_hi, const void* x_lo, float* dw, float* db,
                                 int N, int Cin, int H, int W, long x_img_groups, int Cout, int Ho, int Wo, long g_img_groups,
                                 int kh, int kw, int stride, int pad_h, int pad_w, void* workspace, long ws_bytes, int tile_cfg,
                                 const float* g_scale, const float* x_scale, int g_row_split, int g_row_gap,
                                 int* deferred_reduce, hipStream_t stream) {
    SSN_CHECK_ARG(dw && workspace, "conv wgrad pl: null pointer");
    WgPlArgs a;
    const int rcf = fill_wgpl(a, g_hi, g_lo, x_hi, x_lo, N, Cin, H, W, x_img_groups, Cout, Ho, Wo, g_img_groups, kh, kw, stride, pad_h, pad_w,
                              g_scale, x_scale, g_row_split, g_row_gap);
    if (rcf != SSN_OK) return rcf;
    a.part = (float*)workspace;
    // tile_cfg >= 200: the chunked 1x1 kernel with tile tile_cfg - 200
    if (tile_cfg >= 200) {
        SSN_CHECK_ARG(chunked_1x1_layer(kh, kw, stride, pad_h, pad_w) && tile_cfg - 200 < N1,
                      "conv wgrad pl: the chunked kernel (tile %d) takes 1x1 / stride-1 layers only", tile_cfg);
        const int c1 = tile_cfg - 200;
        plan1(Cout, Cin, a.P, c1, &a.splits, &a.ksteps_per_split);
        const long need1 = (long)a.splits * Cout * a.ldp * (long)sizeof(float);
        if (ws_bytes < need1) {
            ssn_set_error("conv wgrad pl: workspace %ld < %ld bytes", ws_bytes, need1);
            return SSN_ERR_WORKSPACE;
        }
        a.div_hw = make_fastdiv((uint32_t)(H * W));
        a.div_w = make_fastdiv((uint32_t)W);
        const int rc1 = launch_wgpl1_tile(a, c1, stream);
        if (rc1 != SSN_OK) return rc1;
        if (deferred_reduce) {
            deferred_reduce[0] = a.splits;
            deferred_reduce[1] = 1;
            return SSN_OK;
        }
        return ssn_wgrad_reduce(a.part, dw, db, Cout, a.K, a.splits, stream);
    }
    // tile_cfg >= 100: the nine-tap kernel with tile tile_cfg - 100; < 0: it for every layer that qualifies
    if ((tile_cfg >= 100 || tile_cfg < 0) && nine_tap_layer(kh, kw, stride, pad_h, pad_w, H, W, Ho, Wo)) {
        int c9 = tile_cfg >= 100 ? tile_cfg - 100 : pick_tile9(Cout, Cin, W);
        SSN_CHECK_ARG(c9 < N9, "conv wgrad pl: unknown nine-tap tile %d", c9);
        if (xp_for(W) > 8 && c9 != 3) c9 = 0;
        plan9(Cout, Cin, (long)N * (H + 1) * (W + 1), c9, W, &a.splits, &a.ksteps_per_split);
        const long need9 = (long)a.splits * k9KG[c9] * Cout * a.ldp * (long)sizeof(float);
        if (ws_bytes < need9) {
            ssn_set_error("conv wgrad pl: workspace %ld < %ld bytes", ws_bytes, need9);
            return SSN_ERR_WORKSPACE;
        }
        a.div_hw = make_fastdiv((uint32_t)((H + 1) * (W + 1)));
        a.div_w = make_fastdiv((uint32_t)(W + 1));
        const int rc9 = launch_wgpl9_tile(a, c9, stream);
        if (rc9 != SSN_OK) return rc9;
        if (deferred_reduce) {
            deferred_reduce[0] = a.splits * k9KG[c9];
            deferred_reduce[1] = 9;
            return SSN_OK;
        }
        return ssn_wgrad_reduce_taps(a.part, dw, db, Cout, a.K, a.splits * k9KG[c9], 9, stream);
    }
    SSN_CHECK_ARG(tile_cfg < 100, "conv wgrad pl: the nine-tap kernel takes 3x3 / stride 1 / pad 1 layers only");
    const int cfg = fix_cfg(tile_cfg, Cout, Cin);
    plan(Cout, Cin, kh * kw, a.P, cfg, &a.splits, &a.ksteps_per_split);
    const long need = (long)a.splits * Cout * a.ldp * (long)sizeof(float);
    if (ws_bytes < need) {
        ssn_set_error("conv wgrad pl: workspace %ld < %ld bytes", ws_bytes, need);
        return SSN_ERR_WORKSPACE;
    }
    const int rc = launch_wgpl_tile(a, cfg, stream);
    if (rc != SSN_OK) return rc;
    if (deferred_reduce) {
        deferred_reduce[0] = a.splits;
        deferred_reduce[1] = 1;
        return SSN_OK;
    }
    return ssn_wgrad_reduce(a.part, dw, db, Cout, a.K, a.splits, stream);
}
