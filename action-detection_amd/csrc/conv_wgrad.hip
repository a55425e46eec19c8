// Weight (and bias) gradient of the backbone convolutions on the exact-f32 MFMA, gfx950.
//
// Replaces the cuDNN wgrad the reference reaches through loss.backward()
// (/root/reference/ssn_train.py:236) for every Conv2d of model_zoo.BNInception.
//
//   dW[co][kk] = sum_p G[co][p] * X[kk][p]        kk = (ci, r, s),  p = (n, ho, wo)
//   db[co]     = sum_p G[co][p]
//
// G is the gradient w.r.t. the conv output (already multiplied by the ReLU mask and the folded
// BN scale, see ssn_relu_bn_bwd).  Both operands are contiguous along p in NCHW, so both LDS
// tiles are filled with coalesced loads; the reduction dimension of the MFMA is the pixel
// index.  The pixel range is split over `splits` workgroups (split-K); each writes its partial
// [Cout][K(+1)] slab, and ssn_conv_wgrad's second kernel sums the slabs in a fixed order, so the
// result is deterministic (no float atomics).
#include <type_traits>

#include "ssn_common.h"

namespace {

struct WgradArgs {
    const float* g;  // [N][..Cout..][Ho][Wo] channel-slice base
    const float* x;  // [N][..Cin..][H][W] channel-slice base
    float* part;     // [splits][M][ldp]
    int N, Cin, H, W;
    long x_img_stride;
    int M, Ho, Wo;
    long g_img_stride;
    int K;    // Cin*KS*KS
    int ldp;  // K + 1 (last column = bias gradient)
    int P;    // N*Ho*Wo
    int pad;
    int splits, chunks_per_split;  // chunk = 32 pixels
    int n_mtiles, n_ktiles;
    uint32_t g_bytes, x_bytes;     // extents for the buffer descriptors
    FastDiv div_hw, div_w, div_tiles, div_kt;
};

constexpr int BP = 32;      // pixels per LDS slab
constexpr int PITCH = 36;   // floats; rows of 144 B keep ds_read_b128 column reads conflict-free
constexpr uint32_t OOB = 0x80000000u;  // out-of-range marker (extents < 2^31, row constants < 2^31)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// Loop structure: a workgroup owns a fixed [BM x BN] tile of dW and walks its pixel range in chunks of
// 32.  Everything that depends on the ROW a thread loads (output channel for G; (c, r, s) for the im2col
// of X) is loop-invariant and computed once: a byte-offset constant per row and the tap index.  Per
// chunk a thread decodes its ONE pixel (2 magic divisions), builds a KS*KS-bit border-validity mask, and
// then every element costs one add (+ a bit test for X) before a branch-free buffer load -- invalid
// elements get an out-of-range offset and read as 0.
template <int KS, int S, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs p) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int NA = BM / 8;
    constexpr int NBR = BN / 8;
    constexpr int KK = KS * KS;
    static_assert(WM * WN == 4, "4 waves per workgroup");

    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * PITCH];
    float* As0 = lds;
    float* Bs0 = lds + 2 * BM * PITCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
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

    const int pix = tid & 31;
    const int rgrp = tid >> 5;  // 0..7
    const int HW = p.H * p.W;
    const int howo = p.Ho * p.Wo;

    // ---- loop-invariant row constants ----
    uint32_t a_const[NA];   // byte offset of row m inside an image of G, or OOB
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + rgrp + 8 * i;
        a_const[i] = (m < p.M) ? (uint32_t)(m * howo) * 4u : OOB;
    }
    uint32_t b_const[NBR];  // byte offset of (c, r, s) relative to the pixel's window origin, or OOB
    int b_tap[NBR];         // r * KS + s
#pragma unroll
    for (int i = 0; i < NBR; ++i) {
        const int kk = kk0 + rgrp + 8 * i;
        int c, tap;
        if (KS == 1) {
            c = kk;
            tap = 0;
        } else {
            c = kk / KK;
            tap = kk - c * KK;
        }
        const int r = tap / KS, s = tap - r * KS;
        b_const[i] = (kk < p.K) ? (uint32_t)(c * HW + r * p.W + s) * 4u : OOB;
        b_tap[i] = tap;
    }
    const __amdgpu_buffer_rsrc_t grsrc = make_rsrc(p.g, p.g_bytes);
    const __amdgpu_buffer_rsrc_t xrsrc = make_rsrc(p.x, p.x_bytes);

    float areg[NA], breg[NBR];

    auto load_slab = [&](int chunk) {
        const int pp = chunk * BP + pix;
        const bool valid = pp < p.P;
        uint32_t n, hw, ho, wo;
        fd_divmod((uint32_t)(valid ? pp : 0), p.div_hw, n, hw);
        fd_divmod(hw, p.div_w, ho, wo);
        // G: one add per element
        const uint32_t abase = valid ? (uint32_t)((long)n * p.g_img_stride * 4) + hw * 4u : OOB;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            areg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grsrc, abase + a_const[i], 0, 0));
        // X: window origin + border-validity mask over the KS*KS taps (bit r*KS+s)
        const int h0 = (int)ho * S - p.pad;
        const int w0 = (int)wo * S - p.pad;
        const uint32_t bbase = (uint32_t)((long)n * p.x_img_stride * 4) + (uint32_t)(h0 * p.W + w0) * 4u;
        using mask_t = typename std::conditional<(KK <= 32), uint32_t, uint64_t>::type;
        mask_t mask = 0;
        if (valid) {
            uint32_t rows = 0, cols = 0;
#pragma unroll
            for (int r = 0; r < KS; ++r) {
                rows |= ((unsigned)(h0 + r) < (unsigned)p.H) ? (1u << r) : 0u;
                cols |= ((unsigned)(w0 + r) < (unsigned)p.W) ? (1u << r) : 0u;
            }
#pragma unroll
            for (int r = 0; r < KS; ++r)
                if (rows & (1u << r)) mask |= (mask_t)cols << (r * KS);
        }
#pragma unroll
        for (int i = 0; i < NBR; ++i) {
            const bool ok = (mask >> b_tap[i]) & 1;
            const uint32_t off = ok ? bbase + b_const[i] : OOB;
            breg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, off, 0, 0));
        }
    };
    auto store_slab = [&](int buf) {
        float* As = As0 + buf * BM * PITCH + rgrp * PITCH + pix;
        float* Bs = Bs0 + buf * BN * PITCH + rgrp * PITCH + pix;
#pragma unroll
        for (int i = 0; i < NA; ++i) As[8 * i * PITCH] = areg[i];
#pragma unroll
        for (int i = 0; i < NBR; ++i) Bs[8 * i * PITCH] = breg[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float rowsum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) rowsum[i] = 0.f;
    const bool do_bias = (kt == 0) && (wn == 0);

    const int total_chunks = (p.P + BP - 1) / BP;
    const int c_begin = (int)z * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > total_chunks) c_end = total_chunks;
    const int nch = c_end - c_begin;

    if (nch > 0) {
        load_slab(c_begin);
        store_slab(0);
    }
    __syncthreads();
    for (int t = 0; t < nch; ++t) {
        const int buf = t & 1;
        if (t + 1 < nch) load_slab(c_begin + t + 1);
        const float* As = As0 + buf * BM * PITCH + (wm * TM * 32 + li) * PITCH + 4 * lh;
        const float* Bs = Bs0 + buf * BN * PITCH + (wn * TN * 32 + li) * PITCH + 4 * lh;
        // fragment groups (8 pixels = 4 MFMA steps) software-pipelined one group ahead of the MFMAs
        f32x4 af[2][TM], bf[2][TN];
        auto load_group = [&](int u, int slot) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[slot][i] = *reinterpret_cast<const f32x4*>(As + i * 32 * PITCH + 8 * u);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[slot][j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * PITCH + 8 * u);
        };
        load_group(0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cur = u & 1;
            if (u + 1 < 4) load_group(u + 1, cur ^ 1);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    rowsum[i] += (af[cur][i].x + af[cur][i].y) + (af[cur][i].z + af[cur][i].w);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][s], bf[cur][j][s], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nch) store_slab(buf ^ 1);
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
                if (m < p.M) out[(long)m * p.ldp + kk] = acc[i][j][r];
            }
        }
    }
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float tot = rowsum[i] + __shfl_xor(rowsum[i], 32, 64);
            const int m = m0 + (wm * TM + i) * 32 + li;
            if (lh == 0 && m < p.M) out[(long)m * p.ldp + p.K] = tot;
        }
    }
}

// dst[m*K + kk] = sum_z part[z][m][kk];  db[m] = sum_z part[z][m][K].  taps > 1: the slabs' columns are TAP-MAJOR
// (column t * (K / taps) + ci holds dW[m][ci][t]: what lets the nine-tap kernel store a tap's columns lane-contiguously)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part, float* dw, float* db, int M, int K,
                                                           int ldp, int splits, int taps) {
    const long total = (long)M * ldp;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        float s = 0.f;
        int z = 0;
        for (; z + 8 <= splits; z += 8) {   // eight independent loads in flight, fixed summation order
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = part[(long)(z + q) * total + idx];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; z < splits; ++z) s += part[(long)z * total + idx];
        const int m = (int)(idx / ldp);
        const int kk = (int)(idx - (long)m * ldp);
        if (kk < K) {
            int col = kk;
            if (taps > 1) {
                const int cin = K / taps, t = kk / cin;
                col = (kk - t * cin) * taps + t;
            }
            dw[(long)m * K + col] = s;
        } else if (db) {
            db[m] = s;
        }
    }
}

// The split-K slabs of MANY layers in one launch (the table travels by value: capturable, nothing staged): block b belongs to tensor
// t with blk0[t] <= b < blk0[t + 1] and strides over that tensor's M x (K + 1) outputs.  Same arithmetic and summation order as
// wgrad_reduce_kernel -- the executor defers the reductions of a backward pass and issues them together (44 launches -> 1).
constexpr int RM_MAX = 48;
struct ReduceTable {
    const float* part[RM_MAX];
    float* dw[RM_MAX];
    float* db[RM_MAX];
    int M[RM_MAX], K[RM_MAX], splits[RM_MAX], taps[RM_MAX];
    int blk0[RM_MAX + 1];
    int count;
};
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(ReduceTable t) {
    int ti = 0;
    while (ti + 1 < t.count && (int)blockIdx.x >= t.blk0[ti + 1]) ++ti;
    const float* part = t.part[ti];
    float* dw = t.dw[ti];
    float* db = t.db[ti];
    const int M = t.M[ti], K = t.K[ti], ldp = K + 1, splits = t.splits[ti], taps = t.taps[ti];
    const long total = (long)M * ldp;
    const uint32_t cin_t = (uint32_t)(K / (taps > 0 ? taps : 1));
    const long nb = t.blk0[ti + 1] - t.blk0[ti];
    for (long idx = (long)((int)blockIdx.x - t.blk0[ti]) * 256 + threadIdx.x; idx < total; idx += nb * 256) {
        float s = 0.f;
        int z = 0;
        for (; z + 8 <= splits; z += 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = part[(long)(z + q) * total + idx];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; z < splits; ++z) s += part[(long)z * total + idx];
        const int m = (int)((uint32_t)idx / (uint32_t)ldp);      // (32-bit: a slab holds < 2^31 elements -- checked by the host)
        const int kk = (int)((uint32_t)idx - (uint32_t)m * (uint32_t)ldp);
        if (kk < K) {
            int col = kk;
            if (taps > 1) {
                const int tt = (int)((uint32_t)kk / cin_t);
                col = (kk - tt * (int)cin_t) * taps + tt;
            }
            dw[(long)m * K + col] = s;
        } else if (db) {
            db[m] = s;
        }
    }
}

template <int KS, int S, int WM, int WN, int TM, int TN>
int launch_wgrad(WgradArgs& a, hipStream_t stream) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    a.n_mtiles = (a.M + BM - 1) / BM;
    a.n_ktiles = (a.K + BN - 1) / BN;
    const unsigned tiles = (unsigned)a.n_mtiles * (unsigned)a.n_ktiles;
    a.div_tiles = make_fastdiv(tiles);
    a.div_kt = make_fastdiv((uint32_t)a.n_ktiles);
    hipLaunchKernelGGL((conv_wgrad_kernel<KS, S, WM, WN, TM, TN>), dim3(tiles * (unsigned)a.splits), dim3(256), 0,
                       stream, a);
    SSN_CHECK_LAUNCH("conv_wgrad");
    return SSN_OK;
}

// tile configs: 0: 2,2,1,1 -> 64(co) x 64(kk)   1: 1,4,1,1 -> 32 x 128   2: 2,2,2,2 -> 128 x 128
//               3: 2,2,1,2 -> 64 x 128          4: 1,4,3,1 -> 96 x 128   5: 1,4,2,1 -> 64 x 128 (waves along kk)
//               6: 2,2,2,1 -> 128 x 64
const int kWgBM[7] = {64, 32, 128, 64, 96, 64, 128};
const int kWgBN[7] = {64, 128, 128, 128, 128, 128, 64};

template <int KS, int S>
int launch_wgrad_tile(WgradArgs& a, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_wgrad<KS, S, 2, 2, 1, 1>(a, stream);
        case 1: return launch_wgrad<KS, S, 1, 4, 1, 1>(a, stream);
        case 2: return launch_wgrad<KS, S, 2, 2, 2, 2>(a, stream);
        case 3: return launch_wgrad<KS, S, 2, 2, 1, 2>(a, stream);
        case 4: return launch_wgrad<KS, S, 1, 4, 3, 1>(a, stream);
        case 5: return launch_wgrad<KS, S, 1, 4, 2, 1>(a, stream);
        case 6: return launch_wgrad<KS, S, 2, 2, 2, 1>(a, stream);
    }
    ssn_set_error("conv_wgrad: unknown tile config %d", cfg);
    return SSN_ERR_ARG;
}

int pick_wgrad_tile(int M, int K) {
    double best = 1e300;
    int bc = 0;
    for (int c = 0; c < 7; ++c) {
        const double padded = (double)((M + kWgBM[c] - 1) / kWgBM[c]) * kWgBM[c] *
                              (double)((K + kWgBN[c] - 1) / kWgBN[c]) * kWgBN[c];
        const double reuse = (kWgBM[c] * kWgBN[c] >= 128 * 64) ? 1.0 : 1.12;
        if (padded * reuse < best) {
            best = padded * reuse;
            bc = c;
        }
    }
    return bc;
}

// co-resident workgroups per CU of each tile config (LDS 2 x (BM + BN) x 144 B; VGPRs as compiled)
const int kWgOcc[7] = {4, 3, 2, 2, 2, 2, 2};

void plan_wgrad(int M, int K, long P, int cfg, int* splits, int* chunks_per_split) {
    const long tiles = (long)((M + kWgBM[cfg] - 1) / kWgBM[cfg]) * ((K + kWgBN[cfg] - 1) / kWgBN[cfg]);
    // reduce pass: ~8 bytes per output element and split at ~2 TB/s, in units of a ~2.5 us (32-pixel, f32 MFMA) chunk
    plan_split_k(tiles, (P + BP - 1) / BP, kWgOcc[cfg], 8, 6, 0.02 + (double)M * K * 1.6e-6, splits, chunks_per_split);
}

}  // namespace

// dw[m][kk] = sum_z part[z][m][kk], db[m] = sum_z part[z][m][K] (fixed order); shared with conv_wgrad_x6.hip
extern "C" int ssn_wgrad_reduce_taps(const float* part, float* dw, float* db, int M, int K, int splits, int taps,
                                     hipStream_t stream) {
    SSN_CHECK_ARG(part && dw && M > 0 && K > 0 && splits > 0 && taps >= 1 && K % taps == 0, "wgrad reduce: bad arguments");
    const long total = (long)M * (K + 1);
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, part, dw, db, M, K, K + 1,
                       splits, taps);
    SSN_CHECK_LAUNCH("wgrad_reduce");
    return SSN_OK;
}
extern "C" int ssn_wgrad_reduce(const float* part, float* dw, float* db, int M, int K, int splits, hipStream_t stream) {
    return ssn_wgrad_reduce_taps(part, dw, db, M, K, splits, 1, stream);
}
// ssn_wgrad_reduce_taps for `count` layers in ceil(count / 48) launches (host arrays, one entry per layer; db entries may be null).
extern "C" int ssn_wgrad_reduce_multi(int count, const float* const* part, float* const* dw, float* const* db, const int* M,
                                      const int* K, const int* splits, const int* taps, hipStream_t stream) {
    SSN_CHECK_ARG(count >= 0 && (count == 0 || (part && dw && db && M && K && splits && taps)), "wgrad reduce multi: bad arguments");
    for (int base = 0; base < count; base += RM_MAX) {
        ReduceTable t;
        t.count = count - base < RM_MAX ? count - base : RM_MAX;
        int blocks = 0;
        for (int i = 0; i < t.count; ++i) {
            const int j = base + i;
            SSN_CHECK_ARG(part[j] && dw[j] && M[j] > 0 && K[j] > 0 && splits[j] > 0 && taps[j] >= 1 && K[j] % taps[j] == 0 &&
                              (long)M[j] * (K[j] + 1) < (1l << 31),
                          "wgrad reduce multi: bad entry %d", j);
            t.part[i] = part[j];
            t.dw[i] = dw[j];
            t.db[i] = db[j];
            t.M[i] = M[j];
            t.K[i] = K[j];
            t.splits[i] = splits[j];
            t.taps[i] = taps[j];
            t.blk0[i] = blocks;
            long nb = ((long)M[j] * (K[j] + 1) + 255) / 256;
            if (nb > 1024) nb = 1024;
            blocks += (int)nb;
        }
        t.blk0[t.count] = blocks;
        hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, t);
    }
    SSN_CHECK_LAUNCH("wgrad_reduce_multi");
    return SSN_OK;
}

extern "C" long ssn_conv_wgrad_workspace_bytes(int N, int Cin, int Cout, int Ho, int Wo, int ksize, int tile_cfg) {
    const int K = Cin * ksize * ksize;
    const int cfg = tile_cfg >= 0 ? tile_cfg : pick_wgrad_tile(Cout, K);
    int splits, cps;
    plan_wgrad(Cout, K, (long)N * Ho * Wo, cfg, &splits, &cps);
    return (long)splits * Cout * (K + 1) * (long)sizeof(float);
}

extern "C" int ssn_conv_wgrad(const float* g, const float* x, float* dw, float* db, int N, int Cin, int H, int W,
                              long x_img_stride, int Cout, int Ho, int Wo, long g_img_stride, int ksize, int stride,
                              int pad, void* workspace, long ws_bytes, int tile_cfg, hipStream_t stream) {
    SSN_CHECK_ARG(g && x && dw && workspace, "conv wgrad: null pointer");
    SSN_CHECK_ARG(ksize == 1 || ksize == 3 || ksize == 7, "conv wgrad: ksize %d unsupported", ksize);
    WgradArgs a;
    a.g = g;
    a.x = x;
    a.part = (float*)workspace;
    a.N = N;
    a.Cin = Cin;
    a.H = H;
    a.W = W;
    a.x_img_stride = x_img_stride;
    a.M = Cout;
    a.Ho = Ho;
    a.Wo = Wo;
    a.g_img_stride = g_img_stride;
    a.K = Cin * ksize * ksize;
    a.ldp = a.K + 1;
    a.P = N * Ho * Wo;
    a.pad = pad;
    a.div_hw = make_fastdiv((uint32_t)(Ho * Wo));
    a.div_w = make_fastdiv((uint32_t)Wo);
    const long gb = ((long)(N - 1) * g_img_stride + (long)Cout * Ho * Wo) * 4;
    const long xb = ((long)(N - 1) * x_img_stride + (long)Cin * H * W) * 4;
    SSN_CHECK_ARG(gb < (1l << 31) && xb < (1l << 31), "conv wgrad: operand larger than 2 GiB (buffer addressing)");
    a.g_bytes = (uint32_t)gb;
    a.x_bytes = (uint32_t)xb;
    const int cfg = tile_cfg >= 0 ? tile_cfg : pick_wgrad_tile(Cout, a.K);
    plan_wgrad(Cout, a.K, a.P, cfg, &a.splits, &a.chunks_per_split);
    const long need = (long)a.splits * Cout * a.ldp * (long)sizeof(float);
    if (ws_bytes < need) {
        ssn_set_error("conv wgrad: workspace %ld < %ld bytes", ws_bytes, need);
        return SSN_ERR_WORKSPACE;
    }
    int rc;
    if (ksize == 1 && stride == 1)
        rc = launch_wgrad_tile<1, 1>(a, cfg, stream);
    else if (ksize == 3 && stride == 1)
        rc = launch_wgrad_tile<3, 1>(a, cfg, stream);
    else if (ksize == 3 && stride == 2)
        rc = launch_wgrad_tile<3, 2>(a, cfg, stream);
    else if (ksize == 7 && stride == 2)
        rc = launch_wgrad_tile<7, 2>(a, cfg, stream);
    else {
        ssn_set_error("conv wgrad: (k=%d, s=%d) has no kernel", ksize, stride);
        return SSN_ERR_ARG;
    }
    if (rc != SSN_OK) return rc;
    return ssn_wgrad_reduce(a.part, dw, db, Cout, a.K, a.splits, stream);
}
