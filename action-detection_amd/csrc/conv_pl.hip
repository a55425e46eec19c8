// Implicit-GEMM convolution on planes tensors (planes.h): forward conv + frozen-BN + ReLU and the data gradient of the
// backbone layers behind /root/reference/ssn_models.py:266,298 with fp32-class accuracy on v_mfma_f32_32x32x16_f16,
// gfx950.  Same arithmetic as conv_x6.hip (two f16 terms per operand, three partial products per k16 step, fp32
// accumulation, power-of-two operand scales) -- but the activation operand ARRIVES split: the K loop of this kernel
// contains no VALU work at all.
//
//   * K runs in slabs of 16 channels x one tap (taps innermost), i.e. one MFMA k-step.  A slab's B tile is
//     [plane][k-half][BN pixels][16 B] in LDS, filled by LDS-DMA: one instruction = 64 pixels x 8 channels of one
//     plane (1 KiB), gathered per lane (a lane = one output pixel; taps outside the image and pixels past the end
//     carry an out-of-range offset and deposit zeros), so stride-2 layers, unpadded layers and any kh x kw taps (runtime)
//     take the same path -- no guard bytes in front of tensors, no padded pixel enumeration, no zero rows.
//   * A (weights): the packed image of conv_x6.hip (ssn_conv_x6_pack_* : row = 2 planes x 2 k-halves x 16 B,
//     chunk-swizzled, amax behind the rows), copied 1 KiB per instruction.
//   * Fragments: ONE ds_read_b128 per (tile, plane) for either operand, conflict-free; a wave keeps two register sets
//     and reads slab t+1 while it multiplies slab t; the DMA runs two slabs ahead in a 3-slot ring; one s_barrier per
//     slab; no branches in the loop (slabs past the end are issued with every offset out of range).
//   * Epilogue: affine + ReLU (or accumulate + the fused ReLU / frozen-BN backward of the producer), multiplied by the
//     OUTPUT tensor's scale, clamped to the f16 range, split into the two planes and stored 8 bytes (4 channels) per
//     lane, plane and 8-row group; the largest magnitude goes to the output tensor's amax slot.
#include "planes.h"

namespace {

using namespace pl;

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

constexpr int APITCH = 16;    // dwords per packed weight row (conv_x6_kernel.h)

struct PlConvArgs {
    const void* x_hi;     // planes of the gather source at the slice's first channel group
    const void* x_lo;
    const uint32_t* ap;   // packed split weights [nslab][M][APITCH], then [amax, 0, 0, 0]
    const float* x_scale; // scale slot of the source tensor
    const float* y_scale; // scale slot of the output tensor
    float* y_amax;        // amax slot of the output tensor (nullptr: not recorded)
    void* y_hi;
    void* y_lo;
    const float* scale;   // per output channel (forward): folded BN scale / shift, or nullptr
    const float* shift;
    int N, C, H, W;       // gather-source dims (C = channels read, multiple of 8)
    uint32_t x_img_bytes; // bytes per image of one source plane (all channel groups of the tensor)
    uint32_t x_grp_bytes; // bytes per channel group of one image: H * W * 16
    int M;                // output channels written (multiple of 8)
    int Ho, Wo;           // enumerated pixel grid
    uint32_t y_img_bytes, y_grp_bytes;
    int P;                // N * Ho * Wo
    int kh, kw, stride, pad_h, pad_w;
    int relu, accumulate;
    int raw_from;             // output rows >= raw_from take no affine and no ReLU
    int row_split, row_gap;   // forward: output rows >= row_split are stored row_gap channels further up (multiples of 32 / 8)
    int k_split, k_gap;       // source channels >= k_split sit k_gap channels further up its tensor (multiples of 16 / 8)
    const void* mask_hi;      // dgrad: hi plane of the forward activation whose ReLU / frozen-BN backward is fused (or null)
    const float* mask_scale;  // per dx channel; NaN = not a ReLU output
    uint32_t mask_img_bytes;
    int n_ptiles, n_mtiles, ngroups;   // ngroups = ceil(C / 16)
    uint32_t x_bytes, a_bytes, y_bytes, mask_bytes;   // per plane
    // sub-sampled output (stride-2 dgrad as four stride-1 problems, one per parity class of the input pixel): the
    // enumerated pixel (u, v) is stored at (2u + sub_a, 2v + sub_b) of planes sub_W wide.  sub_W == 0: dense output.
    int sub_a, sub_b, sub_W;
    // haloed 3x3 kernel (conv_pl9_kernel): ceil(2^32 / (W + 2)) for the row / column split of a padded slot, FastDiv of the padded
    // image (H + 2)(W + 2)
    uint32_t magic_wp;
    FastDiv div_sp;
    unsigned long long* trace;   // tooling only: per-block phase timestamps (tools/ablate_conv_pl.py), normally null
    int dbg;                     // tooling only: ablation switches (1: no B fetch, 2: no A fetch, 4: no fragment reads, 8: no stores)
    FastDiv div_hw, div_w, div_mt;
    // conv_pl9_kernel with per-image tiles (the last member: the offsets of everything above are those of the verified kernels)
    FastDiv div_tpi;             // tiles per image = ceil(H W / BN)
};

// tooling build only (tools/build_trace_lib.sh: -DPL_ABLATE): runtime ablation switches; compiled out of the product
#ifdef PL_ABLATE
#define PL_DBG(bit) (p.dbg & (bit))
#else
#define PL_DBG(bit) 0
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define PL_DMA_B128(rsrc_, dst_, voff_, soff_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_, SSN_LDS_PTR(dst_), 16, voff_, soff_, 0, 0)
#else
#define PL_DMA_B128(rsrc_, dst_, voff_, soff_) ((void)(dst_), (void)(voff_), (void)(soff_))
#endif

// (Round 4 measured variants of the 128 x 128 / 96 x 128 / 192 x 64 / 64 x 128 tiles compiled for THREE / FOUR workgroups per CU --
// __launch_bounds__(256, 3 | 4) with a 3-slot ring; no spills: 149 / 129 / 129 / 110 VGPRs -- to cover one workgroup's prologue /
// epilogue with the loops of two others: the autotuner picked them for 39 of ~100 entries, end to end they changed nothing (17.55 vs
// 17.55 ms, profiles/r4_occupancy_tiles_and_deferred_reduce_ab.txt) -- under load the matrix pipe is clock-limited, not occupancy-
// limited (r4_clock_control.txt) -- and they were removed.)
template <int MODE, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256, (TM * TN >= 8) ? 1 : 2) void conv_pl_kernel(PlConvArgs p) {
    constexpr int NW = 4;
    constexpr int NT = 256;
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(BN % 64 == 0 && BN <= 256, "tile width 64 / 128 / 256 pixels");
    constexpr int SEGS = BN / 64;              // 64-pixel DMA pieces per (plane, k-half) row
    constexpr int A_PIECES = BM / 16;          // 1 KiB pieces of the BM x 64 B weight tile
    constexpr int NA = (A_PIECES + NW - 1) / NW;
    constexpr int A_STAGE = NA * NW * 256;     // dwords
    constexpr int PK = BN * 4;                 // dwords of one (plane, k-half) row of the B tile
    constexpr int B_STAGE = 4 * PK;
    constexpr int STAGE = A_STAGE + B_STAGE;
    // ring depth: 4 slots (the DMA runs three slabs ahead) wherever two co-resident workgroups still fit the 160 KiB of LDS --
    // one slab of MFMAs (0.4-1.2k cycles) covers less of a loaded L2 / HBM round trip than two did for the fp32-layout kernels,
    // whose slabs carried twice the VALU work -- else 3
    constexpr bool ONE_WAVE = (TM * TN >= 8);
    constexpr int NSTAGE = (STAGE * 4 * 4 * (ONE_WAVE ? 1 : 2) <= 163840) ? 4 : 3;
    constexpr int NB = 4 * SEGS / NW;          // B pieces per wave and slab
    constexpr int NLOAD = NA + NB;

    __shared__ __attribute__((aligned(1024))) uint32_t lds[NSTAGE * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const uint32_t nblk = (uint32_t)p.n_ptiles * (uint32_t)p.n_mtiles;
    const uint32_t logical = xcd_remap(blockIdx.x, nblk);
    uint32_t ptile, mtile;
    fd_divmod(logical, p.div_mt, ptile, mtile);
    const int m0 = (int)mtile * BM;
    const int p0 = (int)ptile * BN;
    const int KK = p.kh * p.kw;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, trr = 0;
    if (p.trace) {
        tr0 = __builtin_readcyclecounter();
        trr = __builtin_amdgcn_s_memrealtime();     // the constant 100 MHz counter: calibrates the tick of the cycle counter
    }

    // ---- B gather state: this lane fetches pixel seg * 64 + lane of the tile (one lane = one pixel, 16 B = 8 channels) ----
    const int seg = wave % SEGS;
    uint32_t gbase;      // byte offset of the tap-(0,0) input pixel inside a plane (may wrap below zero)
    uint32_t gmask = 0;  // bit t: tap t reads inside the image
    {
        const int gp = p0 + seg * 64 + lane;
        const bool gvalid = gp < p.P;
        uint32_t n, hw, ho, wo;
        fd_divmod((uint32_t)(gvalid ? gp : 0), p.div_hw, n, hw);
        fd_divmod(hw, p.div_w, ho, wo);
        const int h0 = (MODE == MODE_FWD) ? (int)ho * p.stride - p.pad_h : (int)ho + p.pad_h;
        const int w0 = (MODE == MODE_FWD) ? (int)wo * p.stride - p.pad_w : (int)wo + p.pad_w;
        int t = 0;
        for (int r = 0; r < p.kh; ++r)
            for (int s = 0; s < p.kw; ++s, ++t) {
                const int hi = (MODE == MODE_FWD) ? h0 + r : h0 - r;
                const int wi = (MODE == MODE_FWD) ? w0 + s : w0 - s;
                if (gvalid && ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W)) gmask |= 1u << t;
            }
        gbase = n * p.x_img_bytes + (uint32_t)((h0 * p.W + w0) * 16);
        if (PL_DBG(1)) gmask = 0;
    }

    // ---- A copy: wave w moves 1 KiB pieces (q * NW + w) of the tile ----
    uint32_t aoff[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        const int f = (q * NW + wave) * 64 + lane;   // 16-byte chunk of the tile
        aoff[q] = (f < BM * 4 && m0 + f / 4 < p.M && !PL_DBG(2)) ? (uint32_t)(m0 * APITCH) * 4u + (uint32_t)f * 16u : PL_OOB;
    }
    const __amdgpu_buffer_rsrc_t xrsrc[2] = {pl_rsrc(p.x_hi, p.x_bytes), pl_rsrc(p.x_lo, p.x_bytes)};
    const __amdgpu_buffer_rsrc_t arsrc = pl_rsrc(p.ap, p.a_bytes);
    const uint32_t a_step = (uint32_t)(p.M * APITCH) * 4u;
    const int nslab = p.ngroups * KK;
    const bool c_half = (p.C & 8) != 0;   // the last slab holds 8 channels only

    // Slab order: taps OUTERMOST, channel groups innermost (slab u = tap * ngroups + g reads packed weight slab g * KK + tap):
    // the tap displacement, its validity bit and hence the per-lane gather offset only change once per ngroups slabs, so the
    // producer's per-slab state is three scalar additions.  Producer state = the slab that is fetched next.
    int pf_g = 0, pf_tap = 0, pf_col = 0;
    uint32_t pf_d = 0;      // (r * W + s) * 16 of the tap
    uint32_t pf_x = 0;      // + 2 channel groups per slab
    uint32_t pf_a = 0;      // + KK weight slabs per slab
    const uint32_t row_wrap = (uint32_t)(p.W - p.kw) * 16u, group_step = 2u * p.x_grp_bytes, a_gstep = a_step * (uint32_t)KK;
    const int gap_g = p.k_gap ? p.k_split / 16 : -1;   // channel group in front of which the source has a gap
    const uint32_t gap_bytes = (uint32_t)(p.k_gap / 8) * p.x_grp_bytes;
    uint32_t cur_vo = (gmask & 1u) ? gbase : PL_OOB;   // this lane's gather offset for the producer's tap
    uint32_t cur_dead = 0u;
    uint32_t d_vo, d_so, d_aso, d_dead;
    uint32_t *d_b, *d_a;
    bool d_tail;
    auto issue_begin = [&](uint32_t st_off) {   // st_off = dword offset of the destination ring slot
        d_dead = cur_dead;
        d_vo = cur_vo;
        d_tail = (pf_g == p.ngroups - 1) && c_half;
        d_so = pf_x;
        d_b = lds + st_off + A_STAGE + seg * 256;
        d_aso = pf_a;
        d_a = lds + st_off + wave * 256;
        pf_a += a_gstep;
        pf_x += group_step;
        ++pf_g;
        if (pf_g == gap_g) pf_x += gap_bytes;
        if (pf_g == p.ngroups) {      // next tap (past the last one: every offset out of range -- the ring runs ahead of the loop)
            pf_g = 0;
            pf_x = 0;
            ++pf_tap;
            pf_a = a_step * (uint32_t)pf_tap;
            pf_d += 16u;
            if (++pf_col == p.kw) {
                pf_col = 0;
                pf_d += row_wrap;
            }
            const uint32_t cand = (MODE == MODE_FWD) ? gbase + pf_d : gbase - pf_d;
            const bool live = pf_tap < KK;
            cur_vo = (live && ((gmask >> pf_tap) & 1u)) ? cand : PL_OOB;
            cur_dead = live ? 0u : PL_OOB;
        }
    };
    auto issue_piece = [&](int k) {   // k is a compile-time constant at every call site
        if (k < NB) {
            const int pk = (wave + k * NW) / SEGS;   // wave-uniform: 2 * plane + k-half
            const int plane = pk >> 1, khalf = pk & 1;
            uint32_t v = d_vo;
            if (d_tail && khalf) v = PL_OOB;
            PL_DMA_B128(xrsrc[plane], d_b + pk * PK, v, d_so + (khalf ? p.x_grp_bytes : 0u));
        } else {
            PL_DMA_B128(arsrc, d_a + (k - NB) * NW * 256, aoff[k - NB] | d_dead, d_aso);
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

#pragma unroll
    for (int st = 0; st < NSTAGE; ++st) issue(st * STAGE);

    // operand scales (powers of two, exact)
    const float sa = f16_scale_of(__builtin_bit_cast(float, p.ap[p.a_bytes >> 2]));
    const float sb = *p.x_scale;
    const float so = *p.y_scale;
    const float inv = so / (sa * sb);     // accumulator -> scaled output units
    // the epilogue's per-row parameters, requested NOW (their round trip hides behind the first fetch of the K loop): row tid of the
    // tile -- multiplier, shift (scaled), mask scale (NaN: pass through), floor.  Fetched in the epilogue they cost every workgroup a
    // memory round trip between its last MFMA and its first store, and the two workgroups of a CU reach that point together.
    float ep_mul = inv, ep_add = 0.f, ep_msc = __builtin_nanf(""), ep_flo = -__builtin_inff();
    if (tid < BM) {
        const int m = m0 + tid;
        const bool ok = m < p.M;
        const bool aff = ok && p.scale && m < p.raw_from;
        if (aff) {
            ep_mul = p.scale[m + (m >= p.row_split ? p.row_gap : 0)] * inv;
            ep_add = p.shift[m] * so;
        }
        if (ok && p.mask_scale) ep_msc = p.mask_scale[m];
        if (p.relu && m < p.raw_from) ep_flo = 0.f;
    }

    // fragment addressing: A row (wm*TM+i)*32 + li, 16-byte chunk (2*plane + lh) ^ ((row >> 2) & 3)
    const int swz = (li >> 2) & 3;
    int achunk[2];
#pragma unroll
    for (int pn = 0; pn < 2; ++pn) achunk[pn] = ((2 * pn + lh) ^ swz) * 4;
    const int arow = (wm * TM * 32 + li) * APITCH;
    const int bcol = A_STAGE + lh * PK + (wn * TN * 32 + li) * 4;

    struct Frags {
        f16x8 af[2][TM];
        f16x8 bf[2][TN];
    };
    Frags fr0, fr1;
    constexpr int NREAD = 2 * TN + 2 * TM;
    const uint32_t* rd_base;
    auto read_begin = [&](uint32_t st_off) { rd_base = lds + st_off; };
    auto read_step = [&](Frags& f, int k) {   // k is a compile-time constant at every call site
        if (k < 2 * TN) {
            const int pn = k / TN, j = k % TN;
            f.bf[pn][j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(rd_base + bcol + pn * 2 * PK + j * 128));
        } else {
            const int q = k - 2 * TN, pn = q / TM, i = q % TM;
            f.af[pn][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(rd_base + arow + i * 32 * APITCH + achunk[pn]));
        }
    };
    // Three partial products per accumulator tile: w_lo x_hi, w_hi x_hi, w_hi x_lo; between the MFMAs of slab t: the LDS
    // reads of slab t+1 (other register set) and the DMA pieces of slab t+3.
    auto mfma = [&](const Frags& f, uint32_t dma_stage, Frags& nxt) {
        constexpr int PA[3] = {1, 0, 0};
        constexpr int PB[3] = {0, 0, 1};
        constexpr int NM = 3 * TM * TN;
        constexpr int EVERY = NM / NLOAD > 0 ? NM / NLOAD : 1;
        issue_begin(dma_stage);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.af[PA[c]][i], f.bf[PB[c]][j], acc[i][j], 0, 0, 0);
                    const int idx = (c * TM + i) * TN + j;
#pragma unroll
                    for (int k = idx * NREAD / NM; k < (idx + 1) * NREAD / NM; ++k)
                        if (!PL_DBG(4)) read_step(nxt, k);
                    if ((idx + 1) % EVERY == 0 && (idx + 1) / EVERY <= NLOAD) issue_piece((idx + 1) / EVERY - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
        for (int k = NM / EVERY; k < NLOAD; ++k) issue_piece(k);
    };

    if (p.trace) tr1 = __builtin_readcyclecounter();
    SSN_WAIT_VMCNT((NSTAGE - 1) * NLOAD);
    __builtin_amdgcn_s_barrier();
    read_begin(0);
#pragma unroll
    for (int k = 0; k < NREAD; ++k) read_step(fr0, k);
    uint32_t s_cur = 0, s_n1 = STAGE, s_n2 = 2 * STAGE, s_n3 = 3 * STAGE;   // ring slots of slabs t, t+1, t+2 (, t+3)
    auto half = [&](Frags& cur, Frags& nxt) {
        SSN_WAIT_VMCNT((NSTAGE - 2) * NLOAD);   // this wave's pieces of slab t+1 (the later slabs' may still be in flight)
        SSN_WAIT_LGKM0();        // ... and its reads of slab t are back
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_begin(s_n1);
        mfma(cur, s_cur, nxt);   // (refills the slot of slab t, whose fragments are in registers, with slab t + NSTAGE)
        const uint32_t o = s_cur;
        s_cur = s_n1;
        s_n1 = s_n2;
        if (NSTAGE == 4) {
            s_n2 = s_n3;
            s_n3 = o;
        } else {
            s_n2 = o;
        }
    };
    // two slabs per trip (the register sets swap roles); an odd slab count runs one all-zero slab at the end
    for (int t = 0; t < nslab; t += 2) {
        half(fr0, fr1);
        half(fr1, fr0);
    }
    SSN_WAIT_LGKM0();
    SSN_WAIT_VMCNT(0);   // the out-of-range tail pieces still write (zeros) into the ring the epilogue is about to reuse
    if (p.trace) tr2 = __builtin_readcyclecounter();

#define PL_PEND p.P
#include "conv_pl_epilogue.inc"
#undef PL_PEND
    if (p.trace && tid == 0) {
        unsigned long long* t = p.trace + (size_t)blockIdx.x * 8;
        t[0] = tr0;
        t[1] = tr1;
        t[2] = tr2;
        t[3] = __builtin_readcyclecounter();
        t[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
        t[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
        t[6] = trr;
        t[7] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 layers (69 % of the backbone's MACs), forward and dgrad: ONE haloed copy of a 16-channel input slab in LDS
// serves all nine taps.
//
// conv_pl_kernel above fetches the B tile of every (tap, channel group) slab separately: the same input pixels travel L2 -> LDS nine
// times (8 KiB per slab for a 128-pixel tile, as much as the weights).  Here the tile's input pixels are enumerated in PADDED slots
//        U = n SP + (h + 1) Wp + (w + 1),   Wp = W + 2,  SP = (H + 2) Wp
// so that the input pixel of tap (r, s) of an output pixel is a CONSTANT slot displacement (r - 1) Wp + (s - 1) from the pixel's own
// slot, and the border slots hold zeros (their DMA offsets are out of range): per channel group ONE copy of the slots
// [U(first pixel) - Wp - 1, U(last pixel) + Wp + 1] of the tile (HS slots x 4 (plane, k-half) rows, <= 20 KiB) is fetched, the nine
// taps read their B fragments from it at displaced addresses (one ds_read_b128 per fragment and plane, as before).  B-operand DMA per
// group: HS / 64 x 4 instructions instead of 9 x BN / 64 x 4 (128-pixel tile: 20 instead of 72).
//   * slab order: channel groups OUTERMOST, taps innermost (= the packed weight order g * 9 + tap: A advances by one slab per slab);
//   * the halo of group g + 1 is fetched during the first HS / 64 slabs of group g into the other of two halo buffers: wave w owns row
//     (plane w >> 1, k-half w & 1) and issues one 64-slot piece per slab between its MFMAs (a dummy piece in the remaining slabs keeps the
//     per-slab vmcnt constant); the halo of a group that does not exist is fetched as zeros (the dead slab of an odd slab count);
//   * everything else -- A ring, two fragment register sets, one barrier per slab, epilogue -- is conv_pl_kernel's.
// dgrad of such a layer is the same kernel with the tap displacement mirrored (MODE_DGRAD) on the transposed packed operand.
// PI ("per image"): pixel tiles do not cross images -- tile t of image n covers pixels [n H W + t BN, min(+ BN, (n + 1) H W)) -- so that a
// tile's slot span stays BN + a few rows whatever the image size (a 128-pixel tile across two 56 x 56 images spans 371 slots; inside
// one image 253): the last tile of an image is partly empty (56 x 56: 2 % of the pixels enumerated).
template <int MODE, int WM, int WN, int TM, int TN, bool PI = false>
__global__ __launch_bounds__(256, (TM * TN >= 8) ? 1 : 2) void conv_pl9_kernel(PlConvArgs p) {
    constexpr int NW = 4;
    constexpr int NT = 256;
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(BN == 64 || BN == 128, "tile width 64 / 128 pixels");
    constexpr int HS = BN == 128 ? 320 : 192;          // halo slots per (plane, k-half) row (the host checks every tile's span)
    constexpr int NSEG = HS / 64;                       // 64-slot DMA pieces per row
    constexpr int A_PIECES = BM / 16;
    constexpr int NA = (A_PIECES + NW - 1) / NW;
    constexpr int A_STAGE = NA * NW * 256;              // dwords
    constexpr int HROW = HS * 4;                        // dwords of one halo row
    constexpr int HBUF = 4 * HROW;                      // one halo buffer: rows (plane, k-half)
    constexpr int NSTAGE = ((4 * A_STAGE + 2 * HBUF + 256) * 4 * 2 <= 163840) ? 4 : 3;      // A ring: 4 slots where two workgroups fit
    constexpr int NLOAD = NA + 1;                       // per wave and slab: its A pieces + one halo (or dummy) piece
    constexpr int HALO0 = NSTAGE * A_STAGE;             // dword offset of the two halo buffers
    constexpr int DUMMY = HALO0 + 2 * HBUF;             // 1 KiB nobody reads
    static_assert(NSEG <= 9 - (NSTAGE - 1), "the halo of the next group must have landed before its first read");
    static_assert((DUMMY + 256) * 4 * 2 <= 163840, "two workgroups per CU");

    __shared__ __attribute__((aligned(1024))) uint32_t lds[DUMMY + 256];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const uint32_t nblk = (uint32_t)p.n_ptiles * (uint32_t)p.n_mtiles;
    const uint32_t logical = xcd_remap(blockIdx.x, nblk);
    uint32_t ptile, mtile;
    fd_divmod(logical, p.div_mt, ptile, mtile);
    const int m0 = (int)mtile * BM;
    int p0 = (int)ptile * BN;
    [[maybe_unused]] int pl_pend = 0;      // PI: first pixel this tile must not touch (otherwise p.P is read where it is needed, as before)
    if constexpr (PI) {
        uint32_t img, t;
        fd_divmod(ptile, p.div_tpi, img, t);
        const int hw = p.H * p.W;
        p0 = (int)img * hw + (int)t * BN;
        pl_pend = ((int)img + 1) * hw < p.P ? ((int)img + 1) * hw : p.P;
    }
    const int Wp = p.W + 2;
    const int SPs = (p.H + 2) * Wp;

    // padded slot of an output pixel
    auto slot_of = [&](int pp) -> int {
        uint32_t n, hw, h, w;
        fd_divmod((uint32_t)pp, p.div_hw, n, hw);
        fd_divmod(hw, p.div_w, h, w);
        return (int)n * SPs + ((int)h + 1) * Wp + (int)w + 1;
    };
    const int ubase = slot_of(p0) - (Wp + 1);          // first slot of the halo (wave-uniform; >= 0)

    // ---- halo fetch state: this wave's row, this lane's source offset of every piece ----
    const int hplane = wave >> 1, hkhalf = wave & 1;
    uint32_t hoff[NSEG];
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
        const uint32_t U = (uint32_t)(ubase + sg * 64 + lane);
        uint32_t n, u;
        fd_divmod(U, p.div_sp, n, u);
        const uint32_t hp = __umulhi(u, p.magic_wp), wp = u - hp * (uint32_t)Wp;
        const bool ok = n < (uint32_t)p.N && hp >= 1u && hp <= (uint32_t)p.H && wp >= 1u && wp <= (uint32_t)p.W && !PL_DBG(1);
        hoff[sg] = ok ? n * p.x_img_bytes + ((hp - 1u) * (uint32_t)p.W + (wp - 1u)) * 16u : PL_OOB;
    }
    const __amdgpu_buffer_rsrc_t hrsrc = pl_rsrc(hplane ? p.x_lo : p.x_hi, p.x_bytes);
    const bool c_half = (p.C & 8) != 0;                // the last group holds 8 channels only: its second k-half is zeros

    // ---- A copy: wave w moves 1 KiB pieces (q * NW + w) of the tile ----
    uint32_t aoff[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        const int f = (q * NW + wave) * 64 + lane;
        aoff[q] = (f < BM * 4 && m0 + f / 4 < p.M && !PL_DBG(2)) ? (uint32_t)(m0 * APITCH) * 4u + (uint32_t)f * 16u : PL_OOB;
    }
    const __amdgpu_buffer_rsrc_t arsrc = pl_rsrc(p.ap, p.a_bytes);
    const uint32_t a_step = (uint32_t)(p.M * APITCH) * 4u;

    // A producer: byte offset of the packed weight slab fetched next (one slab further per slab; slabs past the end are requested with
    // every offset out of range -- `dead` -- and deposit zeros)
    uint32_t pf_a = 0;
    auto issue_a = [&](uint32_t st_off, uint32_t dead, int q) {     // piece q of this wave into the ring slot at dword st_off
        PL_DMA_B128(arsrc, lds + st_off + wave * 256 + q * NW * 256, aoff[q] | dead, pf_a);
    };
    // halo producer: piece `sg` (compile-time; >= NSEG: the dummy piece that keeps the per-slab vmcnt constant) of a channel group into
    // halo buffer `hb` (compile-time); `live` / `so`: the group exists (and this wave's k-half of it) / its scalar source offset
    auto issue_halo = [&](bool live, uint32_t so, int hb, int sg) {
        if (sg < NSEG) {
            PL_DMA_B128(hrsrc, lds + HALO0 + hb * HBUF + (2 * hplane + hkhalf) * HROW + sg * 256, live ? hoff[sg] : PL_OOB, so);
        } else {
            PL_DMA_B128(hrsrc, lds + DUMMY, PL_OOB, 0u);
        }
    };
    const uint32_t h_khalf = hkhalf ? p.x_grp_bytes : 0u;
    auto halo_live = [&](int g) { return g < p.ngroups && !(hkhalf && c_half && g == p.ngroups - 1); };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: the whole halo of group 0, then the first NSTAGE weight slabs (each with its dummy piece: constant count per slab)
    {
        const bool l0 = halo_live(0);
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) issue_halo(l0, h_khalf, 0, sg);
    }
#pragma unroll
    for (int st = 0; st < NSTAGE; ++st) {
#pragma unroll
        for (int q = 0; q < NA; ++q) issue_a(st * A_STAGE, 0u, q);
        pf_a += a_step;
        issue_halo(false, 0u, 0, NSEG);
    }

    const float sa = f16_scale_of(__builtin_bit_cast(float, p.ap[p.a_bytes >> 2]));
    const float sb = *p.x_scale;
    const float so = *p.y_scale;
    const float inv = so / (sa * sb);
    // the epilogue's per-row parameters, requested NOW (their round trip hides behind the first fetch of the K loop): row tid of the
    // tile -- multiplier, shift (scaled), mask scale (NaN: pass through), floor.  Fetched in the epilogue they cost every workgroup a
    // memory round trip between its last MFMA and its first store, and the two workgroups of a CU reach that point together.
    float ep_mul = inv, ep_add = 0.f, ep_msc = __builtin_nanf(""), ep_flo = -__builtin_inff();
    if (tid < BM) {
        const int m = m0 + tid;
        const bool ok = m < p.M;
        const bool aff = ok && p.scale && m < p.raw_from;
        if (aff) {
            ep_mul = p.scale[m + (m >= p.row_split ? p.row_gap : 0)] * inv;
            ep_add = p.shift[m] * so;
        }
        if (ok && p.mask_scale) ep_msc = p.mask_scale[m];
        if (p.relu && m < p.raw_from) ep_flo = 0.f;
    }

    // fragment addressing.  A as in conv_pl_kernel; B: the lane's pixel -> the slot one row and one column in front of its centre slot
    // inside the halo, so that the window row / column of a tap is a non-negative displacement: row * Wp (a scalar) + column (an immediate)
    const int swz = (li >> 2) & 3;
    int achunk[2];
#pragma unroll
    for (int pn = 0; pn < 2; ++pn) achunk[pn] = ((2 * pn + lh) ^ swz) * 4;
    const int arow = (wm * TM * 32 + li) * APITCH;
    int bcen[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pp = p0 + (wn * TN + j) * 32 + li;
        bcen[j] = HALO0 + lh * HROW + ((pp < (PI ? pl_pend : p.P) ? slot_of(pp) - ubase : Wp + 1) - (Wp + 1)) * 4;   // (dwords; lh = the k-half row)
    }
    const int wrow[3] = {0, Wp * 4, 2 * Wp * 4};     // dwords

    struct Frags {
        f16x8 af[2][TM];
        f16x8 bf[2][TN];
    };
    Frags fr0, fr1;
    constexpr int NREAD = 2 * TN + 2 * TM;
    // Fragment reads of the slab (tap `tap` -- compile-time -- of the group in halo buffer `hb`, weights in the ring slot at `st_off`):
    // forward reads window position (r, s), dgrad the mirrored one
    auto read_step = [&](Frags& f, int k, uint32_t st_off, int hb, int tap) {
        if (k < 2 * TN) {
            const int pn = k / TN, j = k % TN;
            const int r = (MODE == MODE_FWD) ? tap / 3 : 2 - tap / 3, c = (MODE == MODE_FWD) ? tap % 3 : 2 - tap % 3;
            f.bf[pn][j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(lds + bcen[j] + wrow[r] + (hb * HBUF + pn * 2 * HROW + c * 4)));
        } else {
            const int q = k - 2 * TN, pn = q / TM, i = q % TM;
            f.af[pn][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(lds + st_off + arow + i * 32 * APITCH + achunk[pn]));
        }
    };

    SSN_WAIT_VMCNT((NSTAGE - 1) * NLOAD);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int k = 0; k < NREAD; ++k) read_step(fr0, k, 0u, 0, 0);
    uint32_t s_cur = 0, s_n1 = A_STAGE, s_n2 = 2 * A_STAGE, s_n3 = 3 * A_STAGE;    // ring slots of slabs t, t+1, t+2 (, t+3)
    // Per-group scalars of the trip: the halo of the NEXT group (fetched during the first NSEG slabs of this one) and whether the weight
    // slabs requested from this group's last NSTAGE slabs on exist
    bool nx_live;
    uint32_t nx_so, tail_dead;
    auto group_begin = [&](int g) {
        nx_live = halo_live(g + 1);
        nx_so = (uint32_t)(g + 1) * 2u * p.x_grp_bytes + h_khalf;
        tail_dead = g + 1 < p.ngroups ? 0u : PL_OOB;
    };
    // One slab = tap `tap` of the group in halo buffer `hb` (both compile-time at every call site: the nine taps of a group are unrolled,
    // so the tap displacement, the halo piece and its destination, the read addresses are immediates -- as a runtime state machine they
    // were 65 scalar instructions per slab, 16 per MFMA on the 64 x 128 tile: more than two waves per SIMD can hide).  Three partial
    // products per accumulator tile; between the MFMAs: the LDS reads of the next slab (other register set), the halo piece `tap` of the
    // next group, and the weight pieces of slab t + NSTAGE into the slot of slab t.
    auto slab = [&](const Frags& f, Frags& nxt, int tap, int hb) {
        constexpr int PA[3] = {1, 0, 0};
        constexpr int PB[3] = {0, 0, 1};
        constexpr int NM = 3 * TM * TN;
        constexpr int EVERY = NM / NLOAD > 0 ? NM / NLOAD : 1;
        SSN_WAIT_VMCNT((NSTAGE - 2) * NLOAD);   // this wave's pieces of slab t+1 (the later slabs' may still be in flight)
        SSN_WAIT_LGKM0();                       // ... and its reads of slab t are back
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int ntap = tap == 8 ? 0 : tap + 1, nhb = tap == 8 ? hb ^ 1 : hb;
        const uint32_t dead = (tap + NSTAGE >= 9) ? tail_dead : 0u;
        auto piece = [&](int k) {          // k compile-time: 0 = the halo / dummy piece, 1.. = the A pieces
            if (k == 0)
                issue_halo(nx_live, nx_so, hb ^ 1, tap);
            else
                issue_a(s_cur, dead, k - 1);
        };
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.af[PA[c]][i], f.bf[PB[c]][j], acc[i][j], 0, 0, 0);
                    const int idx = (c * TM + i) * TN + j;
#pragma unroll
                    for (int k = idx * NREAD / NM; k < (idx + 1) * NREAD / NM; ++k)
                        if (!PL_DBG(4)) read_step(nxt, k, s_n1, nhb, ntap);
                    if ((idx + 1) % EVERY == 0 && (idx + 1) / EVERY <= NLOAD) piece((idx + 1) / EVERY - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
        for (int k = NM / EVERY; k < NLOAD; ++k) piece(k);
        pf_a += a_step;
        const uint32_t o = s_cur;
        s_cur = s_n1;
        s_n1 = s_n2;
        if (NSTAGE == 4) {
            s_n2 = s_n3;
            s_n3 = o;
        } else {
            s_n2 = o;
        }
    };
    // two groups per trip (nine is odd: the register sets swap roles from one group to the next); the last group of an odd count runs
    // behind the loop (a `break` between the two halves of the trip cost 50 - 80 registers and spills on the two largest tiles: the
    // allocator then has to reconcile two exits)
    auto group = [&](int g, Frags& fa, Frags& fb, int hb) {
        group_begin(g);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap & 1)
                slab(fb, fa, tap, hb);
            else
                slab(fa, fb, tap, hb);
        }
    };
    int g = 0;
    for (; g + 1 < p.ngroups; g += 2) {
        group(g, fr0, fr1, 0);
        group(g + 1, fr1, fr0, 1);
    }
    if (g < p.ngroups) group(g, fr0, fr1, 0);
    SSN_WAIT_LGKM0();
    SSN_WAIT_VMCNT(0);

#define PL_PEND (PI ? pl_pend : p.P)
#include "conv_pl_epilogue.inc"
#undef PL_PEND
}
#undef PL_DMA_B128

int g_pl_default_tile = -1;
int g_pl_dbg = 0;
unsigned long long* g_pl_trace = nullptr;

template <int MODE, int WM, int WN, int TM, int TN>
int launch_cfg(PlConvArgs& a, hipStream_t stream) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.div_mt = make_fastdiv((uint32_t)a.n_mtiles);
    a.n_ptiles = (a.P + BN - 1) / BN;
    const unsigned nblk = (unsigned)a.n_ptiles * (unsigned)a.n_mtiles;
    hipLaunchKernelGGL((conv_pl_kernel<MODE, WM, WN, TM, TN>), dim3(nblk), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("conv_pl");
    return SSN_OK;
}

// Tile ids (rows x pixels), 4 waves each:
//   0 128x128 (2x2 waves of 64x64)   1 64x128   2 128x64   3 64x64   4 192x128   5 256x128   6 128x256
//   7 96x128 (1x4 waves of 96x32)    8 160x128 (1x4)    9 32x128 (1x4)   10 64x256 (2x2 of 32x128)   11 192x64
constexpr int PL_NCFG = 12;
const int kPlBM[PL_NCFG] = {128, 64, 128, 64, 192, 256, 128, 96, 160, 32, 64, 192};
const int kPlBN[PL_NCFG] = {128, 128, 64, 64, 128, 128, 256, 128, 128, 128, 256, 64};

template <int MODE>
int launch_tile(PlConvArgs& a, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_cfg<MODE, 2, 2, 2, 2>(a, stream);
        case 1: return launch_cfg<MODE, 2, 2, 1, 2>(a, stream);
        case 2: return launch_cfg<MODE, 2, 2, 2, 1>(a, stream);
        case 3: return launch_cfg<MODE, 2, 2, 1, 1>(a, stream);
        case 4: return launch_cfg<MODE, 2, 2, 3, 2>(a, stream);
        case 5: return launch_cfg<MODE, 2, 2, 4, 2>(a, stream);
        case 6: return launch_cfg<MODE, 2, 2, 2, 4>(a, stream);
        case 7: return launch_cfg<MODE, 1, 4, 3, 1>(a, stream);
        case 8: return launch_cfg<MODE, 1, 4, 5, 1>(a, stream);
        case 9: return launch_cfg<MODE, 1, 4, 1, 1>(a, stream);
        case 10: return launch_cfg<MODE, 2, 2, 1, 4>(a, stream);
        case 11: return launch_cfg<MODE, 2, 2, 3, 1>(a, stream);
    }
    ssn_set_error("conv_pl: unknown tile config %d", cfg);
    return SSN_ERR_ARG;
}

template <int MODE, int WM, int WN, int TM, int TN, bool PI = false>
int launch_cfg9(PlConvArgs& a, hipStream_t stream) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.div_mt = make_fastdiv((uint32_t)a.n_mtiles);
    if (PI) {
        const int tpi = (a.H * a.W + BN - 1) / BN;
        a.div_tpi = make_fastdiv((uint32_t)tpi);
        a.n_ptiles = a.N * tpi;
    } else {
        a.div_tpi = make_fastdiv(1u);
        a.n_ptiles = (a.P + BN - 1) / BN;
    }
    const unsigned nblk = (unsigned)a.n_ptiles * (unsigned)a.n_mtiles;
    hipLaunchKernelGGL((conv_pl9_kernel<MODE, WM, WN, TM, TN, PI>), dim3(nblk), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("conv_pl9");
    return SSN_OK;
}
// per-image tiles: the 128-pixel tiles that run two workgroups per CU
bool halo_tile_pi(int cfg) { return cfg == 0 || cfg == 1 || cfg == 4 || cfg == 7 || cfg == 8 || cfg == 9; }
template <int MODE>
int launch_tile9_pi(PlConvArgs& a, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_cfg9<MODE, 2, 2, 2, 2, true>(a, stream);
        case 1: return launch_cfg9<MODE, 2, 2, 1, 2, true>(a, stream);
        case 4: return launch_cfg9<MODE, 2, 2, 3, 2, true>(a, stream);
        case 7: return launch_cfg9<MODE, 1, 4, 3, 1, true>(a, stream);
        case 8: return launch_cfg9<MODE, 1, 4, 5, 1, true>(a, stream);
        case 9: return launch_cfg9<MODE, 1, 4, 1, 1, true>(a, stream);
    }
    ssn_set_error("conv_pl9: tile config %d has no per-image haloed variant", cfg);
    return SSN_ERR_ARG;
}
// does every BN-pixel tile INSIDE an H x W image span at most `hs` padded slots?
bool halo_fits_pi(int H, int W, int bn, int hs) {
    const int Wp = W + 2;
    auto slot = [&](int q) { return (q / W + 1) * Wp + (q % W) + 1; };
    for (int p0 = 0; p0 < H * W; p0 += bn) {
        const int p1 = p0 + bn - 1 < H * W - 1 ? p0 + bn - 1 : H * W - 1;
        if (slot(p1) - slot(p0) + 2 * (Wp + 1) + 1 > hs) return false;
    }
    return true;
}
// haloed variants exist for the tiles of 64 / 128 pixels that run two workgroups per CU
bool halo_tile(int cfg) { return cfg == 0 || cfg == 1 || cfg == 2 || cfg == 3 || cfg == 4 || cfg == 7 || cfg == 8 || cfg == 9 || cfg == 11; }
template <int MODE>
int launch_tile9(PlConvArgs& a, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_cfg9<MODE, 2, 2, 2, 2>(a, stream);
        case 1: return launch_cfg9<MODE, 2, 2, 1, 2>(a, stream);
        case 2: return launch_cfg9<MODE, 2, 2, 2, 1>(a, stream);
        case 3: return launch_cfg9<MODE, 2, 2, 1, 1>(a, stream);
        case 4: return launch_cfg9<MODE, 2, 2, 3, 2>(a, stream);
        case 7: return launch_cfg9<MODE, 1, 4, 3, 1>(a, stream);
        case 8: return launch_cfg9<MODE, 1, 4, 5, 1>(a, stream);
        case 9: return launch_cfg9<MODE, 1, 4, 1, 1>(a, stream);
        case 11: return launch_cfg9<MODE, 2, 2, 3, 1>(a, stream);
    }
    ssn_set_error("conv_pl9: tile config %d has no haloed variant", cfg);
    return SSN_ERR_ARG;
}
// does every BN-pixel tile of an N x H x W grid span at most `hs` padded slots (incl. the halo of Wp + 1 slots on either side)?
bool halo_fits(int N, int H, int W, int bn, int hs) {
    const long P = (long)N * H * W;
    const int Wp = W + 2, SP = (H + 2) * Wp;
    auto slot = [&](long pp) { return (pp / (H * W)) * SP + ((pp % (H * W)) / W + 1) * Wp + (pp % W) + 1; };
    for (long t = 0; t * bn < P; ++t) {
        const long p0 = t * bn, p1 = (p0 + bn - 1 < P - 1) ? p0 + bn - 1 : P - 1;
        if (slot(p1) - slot(p0) + 2 * (Wp + 1) + 1 > hs) return false;
        if (t > 4096) break;      // (tiles repeat with period lcm(bn, H W) / bn; the worst case shows up long before)
    }
    return true;
}

// tile_cfg >= 32 asks for the haloed kernel with tile tile_cfg - 32; layers it does not take (not 3x3 / stride 1 / pad 1 / same size,
// a tile without a haloed variant, a tile whose slot span exceeds the LDS budget) run the plain kernel with that tile
constexpr int PL_HALO_BASE = 32;
constexpr int PL_HALO_PI_BASE = 48;      // tile_cfg 48 + c: the haloed kernel with per-image tiles of shape c
inline int plain_tile(int cfg) { return cfg >= PL_HALO_PI_BASE ? cfg - PL_HALO_PI_BASE : (cfg >= PL_HALO_BASE ? cfg - PL_HALO_BASE : cfg); }
template <int MODE>
int launch_any(PlConvArgs& a, int cfg, bool halo_layer, hipStream_t stream) {
    if (cfg >= PL_HALO_PI_BASE) {
        cfg -= PL_HALO_PI_BASE;
        if (halo_layer && cfg < PL_NCFG && halo_tile_pi(cfg) && halo_fits_pi(a.H, a.W, 128, 320)) {
            a.magic_wp = (uint32_t)((0x100000000ull + (unsigned)(a.W + 2) - 1) / (unsigned)(a.W + 2));
            a.div_sp = make_fastdiv((uint32_t)((a.H + 2) * (a.W + 2)));
            return launch_tile9_pi<MODE>(a, cfg, stream);
        }
        return launch_tile<MODE>(a, cfg < PL_NCFG ? cfg : 0, stream);
    }
    if (cfg >= PL_HALO_BASE) {
        cfg -= PL_HALO_BASE;
        if (halo_layer && cfg < PL_NCFG && halo_tile(cfg) && halo_fits(a.N, a.H, a.W, kPlBN[cfg], kPlBN[cfg] == 128 ? 320 : 192)) {
            a.magic_wp = (uint32_t)((0x100000000ull + (unsigned)(a.W + 2) - 1) / (unsigned)(a.W + 2));
            a.div_sp = make_fastdiv((uint32_t)((a.H + 2) * (a.W + 2)));
            return launch_tile9<MODE>(a, cfg, stream);
        }
    }
    return launch_tile<MODE>(a, cfg, stream);
}

// fewest padded rows x columns, then the larger tile (slots = 512 workgroup slots: prefer a grid that fills them)
int default_tile(int M, long P) {
    double best = 1e300;
    int bc = 0;
    for (int c = 0; c < PL_NCFG; ++c) {
        if (c == 5 || c == 6 || c == 10) continue;      // the register-heavy tiles: autotuner only
        const long mt = (M + kPlBM[c] - 1) / kPlBM[c], pt = (P + kPlBN[c] - 1) / kPlBN[c];
        const double padded = (double)(mt * kPlBM[c]) * (double)(pt * kPlBN[c]);
        const double small = (kPlBM[c] * kPlBN[c] >= 128 * 128) ? 1.0 : (kPlBM[c] * kPlBN[c] >= 64 * 128 ? 1.1 : 1.3);
        const long blocks = mt * pt;
        const double fill = blocks < 256 ? 256.0 / (double)blocks : 1.0;      // fewer workgroups than CUs: pay for the idle ones
        const double cost = padded * small * fill;
        if (cost < best) {
            best = cost;
            bc = c;
        }
    }
    return bc;
}

int fill_common(PlConvArgs& a, const void* x_hi, const void* x_lo, const uint32_t* ap, void* y_hi, void* y_lo, int N, int C,
                int H, int W, long x_img_groups, int M, int Ho, int Wo, long y_img_groups, int kh, int kw,
                const float* x_scale, const float* y_scale, float* y_amax, int k_gap, int row_gap, const char* what) {
    SSN_CHECK_ARG(x_hi && x_lo && ap && y_hi && y_lo && x_scale && y_scale, "%s: null pointer", what);
    SSN_CHECK_ARG(C > 0 && C % 8 == 0 && M > 0 && M % 8 == 0, "%s: channel counts must be multiples of 8 (C %d, M %d)", what, C, M);
    SSN_CHECK_ARG(kh >= 1 && kw >= 1 && kh * kw <= 32, "%s: %dx%d taps unsupported", what, kh, kw);
    a.x_hi = x_hi;
    a.x_lo = x_lo;
    a.ap = ap;
    a.y_hi = y_hi;
    a.y_lo = y_lo;
    a.x_scale = x_scale;
    a.y_scale = y_scale;
    a.y_amax = y_amax;
    a.N = N;
    a.C = C;
    a.H = H;
    a.W = W;
    a.M = M;
    a.Ho = Ho;
    a.Wo = Wo;
    a.kh = kh;
    a.kw = kw;
    a.P = N * Ho * Wo;
    a.ngroups = (C + 15) / 16;
    const long xg = (long)H * W * 16, yg = (long)Ho * Wo * 16;
    const long xb = (long)N * x_img_groups * xg, yb = (long)N * y_img_groups * yg;
    const long ab = (long)a.ngroups * kh * kw * M * APITCH * 4;
    SSN_CHECK_ARG(x_img_groups >= (C + k_gap) / 8 && y_img_groups >= (M + row_gap) / 8, "%s: slice wider than its tensor", what);
    SSN_CHECK_ARG(xb < (1l << 31) && yb < (1l << 31) && ab < (1l << 31) && (long)N * Ho * Wo < (1l << 31),
                  "%s: operand plane larger than 2 GiB (buffer addressing)", what);
    a.x_grp_bytes = (uint32_t)xg;
    a.y_grp_bytes = (uint32_t)yg;
    a.x_img_bytes = (uint32_t)(x_img_groups * xg);
    a.y_img_bytes = (uint32_t)(y_img_groups * yg);
    // (the descriptors start at the slice's first group: the bytes behind the last image's slice are never addressed)
    a.x_bytes = (uint32_t)((long)(N - 1) * a.x_img_bytes + (long)((C + k_gap) / 8) * xg);
    a.y_bytes = (uint32_t)((long)(N - 1) * a.y_img_bytes + (long)((M + row_gap) / 8) * yg);
    a.a_bytes = (uint32_t)ab;
    a.div_hw = make_fastdiv((uint32_t)(Ho * Wo));
    a.div_w = make_fastdiv((uint32_t)Wo);
    a.sub_a = a.sub_b = a.sub_W = 0;
    a.mask_hi = nullptr;
    a.mask_scale = nullptr;
    a.mask_img_bytes = 0;
    a.mask_bytes = 0;
    a.k_split = 0;
    a.k_gap = 0;
    a.row_split = 0x7fffffff;
    a.row_gap = 0;
    a.raw_from = 0x7fffffff;
    a.scale = a.shift = nullptr;
    a.relu = a.accumulate = 0;
    a.dbg = g_pl_dbg;
    a.trace = g_pl_trace;
    return SSN_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int ssn_conv_pl_tiles(void) { return PL_NCFG; }
// 1 when tile_cfg (>= 32) runs the haloed kernel on a 3x3 / stride 1 / pad 1 layer of N x H x W pixels, 0 when it falls back
extern "C" int ssn_conv_pl_halo_taken(int N, int H, int W, int tile_cfg) {
    if (tile_cfg >= PL_HALO_PI_BASE) {
        const int c = tile_cfg - PL_HALO_PI_BASE;
        return c < PL_NCFG && halo_tile_pi(c) && halo_fits_pi(H, W, 128, 320);
    }
    const int cfg = tile_cfg - PL_HALO_BASE;
    return cfg >= 0 && cfg < PL_NCFG && halo_tile(cfg) && halo_fits(N, H, W, kPlBN[cfg], kPlBN[cfg] == 128 ? 320 : 192);
}
extern "C" void ssn_conv_pl_debug_flags(int flags) { g_pl_dbg = flags; }
extern "C" void ssn_conv_pl_debug_trace(unsigned long long* buf) { g_pl_trace = buf; }
extern "C" int ssn_conv_pl_tile_shape(int cfg, int* bm, int* bn) {
    if (cfg < 0 || cfg >= PL_NCFG) return SSN_ERR_ARG;
    *bm = kPlBM[cfg];
    *bn = kPlBN[cfg];
    return SSN_OK;
}

// Forward convolution + frozen-BN affine + ReLU on planes tensors (replaces cuDNN conv + BN(eval) + ReLU behind
// /root/reference/ssn_models.py:266).  x_*/y_*: plane pointers at the first channel group of the source / destination
// slice; *_img_groups: channel groups (C / 8) of the whole tensors.  w_packed: ssn_conv_x6_pack_* forward operand.
extern "C" int ssn_conv_pl_fwd(const void* x_hi, const void* x_lo, const float* w_packed, const float* scale,
                               const float* shift, void* y_hi, void* y_lo, int N, int Cin, int H, int W,
                               long x_img_groups, int Cout, int Ho, int Wo, long y_img_groups, int kh, int kw, int stride,
                               int pad_h, int pad_w, int relu, int tile_cfg, const float* x_scale, const float* y_scale,
                               float* y_amax, int raw_from, int row_split, int row_gap, hipStream_t stream) {
    PlConvArgs a;
    int rc = fill_common(a, x_hi, x_lo, (const uint32_t*)w_packed, y_hi, y_lo, N, Cin, H, W, x_img_groups, Cout, Ho, Wo,
                         y_img_groups, kh, kw, x_scale, y_scale, y_amax, 0, row_gap, "conv pl fwd");
    if (rc != SSN_OK) return rc;
    SSN_CHECK_ARG(stride == 1 || stride == 2, "conv pl fwd: stride %d unsupported", stride);
    SSN_CHECK_ARG(!row_gap || (row_split % 32 == 0 && row_gap % 8 == 0 && row_split > 0 && row_split < Cout),
                  "conv pl fwd: a row split must be a multiple of 32 inside (0, M), the gap a multiple of 8");
    a.stride = stride;
    a.pad_h = pad_h;
    a.pad_w = pad_w;
    a.scale = scale;
    a.shift = shift;
    a.relu = relu;
    a.raw_from = raw_from > 0 ? raw_from : 0x7fffffff;
    if (row_gap) {
        a.row_split = row_split;
        a.row_gap = row_gap;
    }
    const int cfg = tile_cfg >= 0 ? tile_cfg : (g_pl_default_tile >= 0 ? g_pl_default_tile : default_tile(Cout, a.P));
    const bool halo_layer = kh == 3 && kw == 3 && stride == 1 && pad_h == 1 && pad_w == 1 && Ho == H && Wo == W && !row_gap;
    return launch_any<MODE_FWD>(a, cfg, halo_layer, stream);
}

// Data gradient of a stride-1 convolution (any kh x kw taps): dx = sum over taps of w^T dy, as a gather over dy with
// padding (pad_h, pad_w) of the FORWARD layer.  wt_packed: the transposed operand (pack mode 1).  accumulate: dx += ;
// mask_hi / mask_scale: fuse dx <- dx * (x > 0) * mask_scale[c] (ReLU + frozen-BN backward of the layer that produced x;
// mask_hi = hi plane of x at dx's slice, mask_img_groups = groups of x's tensor).
// taps_reversed: wt_packed is the transposed, tap-reversed operand (pack mode 2: rectangular-tap layers).
extern "C" int ssn_conv_pl_dgrad(const void* dy_hi, const void* dy_lo, const float* wt_packed, void* dx_hi, void* dx_lo,
                                 int N, int Cout, int Ho, int Wo, long dy_img_groups, int Cin, int H, int W,
                                 long dx_img_groups, int kh, int kw, int pad_h, int pad_w, int accumulate,
                                 const void* mask_hi, long mask_img_groups, const float* mask_scale, int tile_cfg,
                                 const float* dy_scale, const float* dx_scale, float* dx_amax, int k_split, int k_gap,
                                 int taps_reversed, hipStream_t stream) {
    PlConvArgs a;
    // dgrad runs on the grid of dx: "source" = dy (C = Cout channels, Ho x Wo), "output" = dx (M = Cin, H x W)
    int rc = fill_common(a, dy_hi, dy_lo, (const uint32_t*)wt_packed, dx_hi, dx_lo, N, Cout, Ho, Wo, dy_img_groups, Cin, H, W,
                         dx_img_groups, kh, kw, dy_scale, dx_scale, dx_amax, k_gap, 0, "conv pl dgrad");
    if (rc != SSN_OK) return rc;
    SSN_CHECK_ARG(!k_gap || (k_split % 16 == 0 && k_gap % 8 == 0 && k_split > 0 && k_split < Cout && Cout % 16 == 0),
                  "conv pl dgrad: a channel split must be a multiple of 16 inside (0, C)");
    a.stride = 1;
    a.pad_h = pad_h;
    a.pad_w = pad_w;
    a.accumulate = accumulate;
    a.k_split = k_split;
    a.k_gap = k_gap;
    if (mask_hi && mask_scale) {
        a.mask_hi = mask_hi;
        a.mask_scale = mask_scale;
        const long mg = (long)H * W * 16;
        SSN_CHECK_ARG(mask_img_groups >= Cin / 8 && (long)N * mask_img_groups * mg < (1l << 31), "conv pl dgrad: bad mask tensor");
        a.mask_img_bytes = (uint32_t)(mask_img_groups * mg);
        a.mask_bytes = (uint32_t)((long)(N - 1) * a.mask_img_bytes + (long)(Cin / 8) * mg);
    }
    const int cfg = tile_cfg >= 0 ? tile_cfg : (g_pl_default_tile >= 0 ? g_pl_default_tile : default_tile(Cin, a.P));
    if (taps_reversed) {
        // the operand was packed transposed AND tap-reversed (ssn_conv_x6_pack_dgrad_rect / pack_rect_multi mode 2): the data
        // gradient is then the forward correlation of dy with it, padding k - 1 - pad
        a.pad_h = kh - 1 - pad_h;
        a.pad_w = kw - 1 - pad_w;
        return launch_tile<MODE_FWD>(a, plain_tile(cfg), stream);
    }
    const bool halo_layer = kh == 3 && kw == 3 && pad_h == 1 && pad_w == 1 && Ho == H && Wo == W && !k_gap;
    return launch_any<MODE_DGRAD>(a, cfg, halo_layer, stream);
}

// Data gradient of a 3x3 / stride-2 convolution (pad 1 on an even input, or pad 0) on planes slices: four stride-1 launches,
// one per parity class of the input pixel, each a (1 + a) x (1 + b)-tap gather over dy stored at the class's pixels of dx
// (no tap is multiplied that does not contribute).  wt_packed: ssn_conv_x6_pack_dgrad_s2 (four sections).
extern "C" int ssn_conv_pl_dgrad_s2(const void* dy_hi, const void* dy_lo, const float* wt_packed, void* dx_hi, void* dx_lo, int N,
                                    int Cout, int Ho, int Wo, long dy_img_groups, int Cin, int H, int W, long dx_img_groups,
                                    int pad, int accumulate, const void* mask_hi, long mask_img_groups, const float* mask_scale,
                                    int tile_cfg, const float* dy_scale, const float* dx_scale, float* dx_amax,
                                    hipStream_t stream) {
    SSN_CHECK_ARG((pad == 1 && H % 2 == 0 && W % 2 == 0 && Ho == H / 2 && Wo == W / 2) ||
                      (pad == 0 && H >= 3 && W >= 3 && Ho == (H - 3) / 2 + 1 && Wo == (W - 3) / 2 + 1),
                  "conv pl dgrad s2: needs a 3x3 / stride-2 convolution with pad 1 and even input size, or pad 0 (%dx%d -> %dx%d, pad %d)",
                  H, W, Ho, Wo, pad);
    long off = 0;
    for (int cls = 0; cls < 4; ++cls) {
        const int ca = cls >> 1, cb = cls & 1, kh = 1 + ca, kw = 1 + cb;
        const int sub_a = pad ? ca : 1 - ca, sub_b = pad ? cb : 1 - cb;     // parity of the class's input rows / columns
        const int gh = pad ? Ho : (H - sub_a + 1) / 2, gw = pad ? Wo : (W - sub_b + 1) / 2;
        PlConvArgs a;
        // source = dy, enumerated grid = the class's pixels (u, v); the output geometry is patched to dx's dense planes below
        int rc = fill_common(a, dy_hi, dy_lo, (const uint32_t*)wt_packed + off, dx_hi, dx_lo, N, Cout, Ho, Wo, dy_img_groups, Cin, gh,
                             gw, dx_img_groups, kh, kw, dy_scale, dx_scale, dx_amax, 0, 0, "conv pl dgrad s2");
        if (rc != SSN_OK) return rc;
        const long yg = (long)H * W * 16;
        SSN_CHECK_ARG((long)N * dx_img_groups * yg < (1l << 31), "conv pl dgrad s2: dx plane larger than 2 GiB");
        a.y_grp_bytes = (uint32_t)yg;
        a.y_img_bytes = (uint32_t)(dx_img_groups * yg);
        a.y_bytes = (uint32_t)((long)(N - 1) * a.y_img_bytes + (long)(Cin / 8) * yg);
        a.stride = 1;
        a.pad_h = pad ? 0 : ca;
        a.pad_w = pad ? 0 : cb;
        a.accumulate = accumulate;
        a.sub_a = sub_a;
        a.sub_b = sub_b;
        a.sub_W = W;
        if (mask_hi && mask_scale) {
            a.mask_hi = mask_hi;
            a.mask_scale = mask_scale;
            a.mask_img_bytes = (uint32_t)(mask_img_groups * yg);
            a.mask_bytes = (uint32_t)((long)(N - 1) * a.mask_img_bytes + (long)(Cin / 8) * yg);
        }
        int cfg = tile_cfg >= 0 ? tile_cfg : (g_pl_default_tile >= 0 ? g_pl_default_tile : default_tile(Cin, a.P));
        cfg = plain_tile(cfg);
        rc = launch_tile<MODE_FWD>(a, cfg, stream);
        if (rc != SSN_OK) return rc;
        off += (long)a.ngroups * kh * kw * Cin * APITCH + 4;
    }
    return SSN_OK;
}
