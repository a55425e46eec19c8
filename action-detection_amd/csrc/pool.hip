// Pooling layers of BN-Inception: 3x3 max/avg pools with Caffe-style
// ceil_mode output sizes (the host computes Ho/Wo; windows are clipped here), average pools
// with count_include_pad=True, and the final global average pool.  The generic kernels are one thread per
// output (fwd) or input (bwd) element with W-contiguous lanes; the large layers take the vectorised variants
// further down (strips of 4 / 2 elements per thread, wide loads).
//
// Max pooling stores the argmax as a 0..k*k-1 window-local index (uint8) using torch's
// tie rule (first maximum in row-major window order, strict >), so the backward pass routes
// gradients exactly like the reference's autograd even on the many exact-zero ties that
// follow a ReLU.
#include "ssn_common.h"

namespace {

// All taps of a window are fetched with branch-free raw buffer loads: an out-of-window / out-of-tensor tap
// gets an out-of-range offset and reads as 0, so the 9 (fwd) or up-to-9 (bwd) loads of an element are all
// in flight together instead of being serialised behind exec-mask branches.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pool_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
constexpr uint32_t POOL_OOB = 0x80000000u;

struct PoolArgs {
    const float* x;
    float* y;
    uint8_t* idx;  // max: argmax out (may be null); unused for avg
    int N, C, H, W, Ho, Wo;
    long x_img_stride, y_img_stride;
    int k, stride, pad;
    long total;
    uint32_t x_bytes;
    // optional per-channel affine + ReLU on the pooled value (average pools only): the Inception "pool projection"
    // branch computed as  relu(scale * avgpool(conv1x1(x)) + shift)  -- see ssn_avgpool_affine_fwd
    const float* aff_scale;
    const float* aff_shift;
    int aff_relu;
    float* amax;   // amax slot of y's tensor (nullptr: not tracked; ssn_common.h: amax_emit)
    FastDiv div_chw, div_hw, div_w;
};

// KC/SC/PC: compile-time window (BN-Inception uses 3/2/0 and 3/1/1); other shapes instantiate on demand.
template <bool MAX, int KC, int SC, int PC>
__global__ __launch_bounds__(256) void pool_fwd_kernel(PoolArgs p) {
    const int howo = p.Ho * p.Wo;
    const __amdgpu_buffer_rsrc_t xr = pool_rsrc(p.x, p.x_bytes);
    float vmax = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long)gridDim.x * 256) {
        uint32_t n, rem, c, hw, ho, wo;
        fd_divmod((uint32_t)i, p.div_chw, n, rem);
        fd_divmod(rem, p.div_hw, c, hw);
        fd_divmod(hw, p.div_w, ho, wo);
        const uint32_t base = (uint32_t)(((long)n * p.x_img_stride + (long)c * p.H * p.W) * 4);
        const int h0 = (int)ho * SC - PC, w0 = (int)wo * SC - PC;
        float v[KC * KC];
        bool ok[KC * KC];
#pragma unroll
        for (int r = 0; r < KC; ++r)
#pragma unroll
            for (int s = 0; s < KC; ++s) {
                const int hi = h0 + r, wi = w0 + s;
                const bool in = ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W);
                ok[r * KC + s] = in;
                const uint32_t off = in ? base + (uint32_t)(hi * p.W + wi) * 4u : POOL_OOB;
                v[r * KC + s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, 0, 0));
            }
        float out;
        if (MAX) {
            // torch rule: start at the first in-window element, move on (v > best) or NaN
            float best = -INFINITY;
            int bi = 0;
            bool first = true;
#pragma unroll
            for (int t = 0; t < KC * KC; ++t) {
                const bool take = ok[t] && (first || v[t] > best || v[t] != v[t]);
                bi = (ok[t] && first) ? t : bi;
                first = first && !ok[t];
                bi = take ? t : bi;
                best = take ? v[t] : best;
            }
            out = best;
            if (p.idx) p.idx[(long)n * p.C * howo + rem] = (uint8_t)bi;
        } else {
            float s_ = 0.f;
#pragma unroll
            for (int t = 0; t < KC * KC; ++t) s_ += v[t];
            // count_include_pad=True: divisor = window clipped to the PADDED extent (torch avg_pool2d)
            int he = h0 + KC, we = w0 + KC;
            if (he > p.H + PC) he = p.H + PC;
            if (we > p.W + PC) we = p.W + PC;
            out = s_ / (float)((he - h0) * (we - w0));
            if (p.aff_scale) {
                out = out * p.aff_scale[c] + p.aff_shift[c];
                if (p.aff_relu) out = fmaxf(out, 0.f);
            }
        }
        p.y[(long)n * p.y_img_stride + rem] = out;
        vmax = fmaxf(vmax, fabsf(out));
    }
    amax_emit(p.amax, vmax);
}

struct PoolBwdArgs {
    const float* dy;
    const uint8_t* idx;
    float* dx;
    int N, C, H, W, Ho, Wo;
    long dy_img_stride, dx_img_stride;
    int k, stride, pad, accumulate;
    const float* mask_y;      // optional fused ReLU+BN backward of the tensor dx is the gradient of
    const float* mask_scale;
    long mask_img_stride;
    long total;  // N*C*H*W
    uint32_t dy_bytes, idx_bytes;
    float* amax;   // amax slot of dx's tensor (nullptr: not tracked)
    FastDiv div_chw, div_hw, div_w;
};

template <bool MAX, int KC, int SC, int PC>
__global__ __launch_bounds__(256) void pool_bwd_kernel(PoolBwdArgs p) {
    const int howo = p.Ho * p.Wo;
    constexpr int NWIN = (KC + SC - 1) / SC;  // max windows per axis covering one input pixel
    const __amdgpu_buffer_rsrc_t dyr = pool_rsrc(p.dy, p.dy_bytes);
    const __amdgpu_buffer_rsrc_t ixr = pool_rsrc(MAX ? (const void*)p.idx : (const void*)p.dy, MAX ? p.idx_bytes : 4u);
    float vmax = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long)gridDim.x * 256) {
        uint32_t n, rem, c, hw, hi, wi;
        fd_divmod((uint32_t)i, p.div_chw, n, rem);
        fd_divmod(rem, p.div_hw, c, hw);
        fd_divmod(hw, p.div_w, hi, wi);
        const uint32_t dybase = (uint32_t)(((long)n * p.dy_img_stride + (long)c * howo) * 4);
        const uint32_t ixbase = (uint32_t)(((long)n * p.C + c) * howo);
        // output windows that contain (hi, wi): ho in [ceil((hi+pad-k+1)/s), floor((hi+pad)/s)]
        const int hp = (int)hi + PC, wp = (int)wi + PC;
        int ho_lo = hp - KC + 1;
        ho_lo = ho_lo <= 0 ? 0 : (ho_lo + SC - 1) / SC;
        int wo_lo = wp - KC + 1;
        wo_lo = wo_lo <= 0 ? 0 : (wo_lo + SC - 1) / SC;
        int ho_hi = hp / SC, wo_hi = wp / SC;
        if (ho_hi > p.Ho - 1) ho_hi = p.Ho - 1;
        if (wo_hi > p.Wo - 1) wo_hi = p.Wo - 1;
        float gv[NWIN * NWIN];
        unsigned iv[NWIN * NWIN];
#pragma unroll
        for (int a = 0; a < NWIN; ++a)
#pragma unroll
            for (int b = 0; b < NWIN; ++b) {
                const int ho = ho_lo + a, wo = wo_lo + b;
                const bool in = (ho <= ho_hi) && (wo <= wo_hi);
                const uint32_t e = (uint32_t)(ho * p.Wo + wo);
                gv[a * NWIN + b] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(dyr, in ? dybase + e * 4u : POOL_OOB, 0, 0));
                if (MAX)
                    iv[a * NWIN + b] = __builtin_amdgcn_raw_buffer_load_b8(ixr, in ? ixbase + e : POOL_OOB, 0, 0);
            }
        float g = 0.f;
#pragma unroll
        for (int a = 0; a < NWIN; ++a)
#pragma unroll
            for (int b = 0; b < NWIN; ++b) {
                const int ho = ho_lo + a, wo = wo_lo + b;
                const bool in = (ho <= ho_hi) && (wo <= wo_hi);
                const int r = hp - ho * SC, s = wp - wo * SC;
                if (MAX) {
                    g += (in && iv[a * NWIN + b] == (unsigned)(r * KC + s)) ? gv[a * NWIN + b] : 0.f;
                } else {
                    const int h0 = ho * SC - PC, w0 = wo * SC - PC;
                    int he = h0 + KC, we = w0 + KC;
                    if (he > p.H + PC) he = p.H + PC;
                    if (we > p.W + PC) we = p.W + PC;
                    g += in ? gv[a * NWIN + b] / (float)((he - h0) * (we - w0)) : 0.f;
                }
            }
        float* dst = p.dx + (long)n * p.dx_img_stride + rem;
        if (p.accumulate) g += *dst;
        if (p.mask_y) {
            const float sc = p.mask_scale[c];
            g = (sc != sc) ? g : (p.mask_y[(long)n * p.mask_img_stride + rem] > 0.f ? g * sc : 0.f);
        }
        *dst = g;
        vmax = fmaxf(vmax, fabsf(g));
    }
    amax_emit(p.amax, vmax);
}

// one wave per (n, c): mean over the HW plane
// ---------------------------------------------------------------------------------------------------------------
// Vectorised variants: a thread produces V (4 or 2) horizontally adjacent elements.  The texture path takes one
// wave instruction per ~16 cycles whatever its width, so the one-dword-per-tap kernels above (9 loads per element)
// are bound by the load ISSUE rate, not by HBM; here a row of the window is one wide load plus at most two border
// dwords (2.25 load instructions per element instead of 9), and results leave as one 16/8-byte store.
// ---------------------------------------------------------------------------------------------------------------
typedef uint32_t pu32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t pu32x2 __attribute__((ext_vector_type(2)));

struct PoolVecArgs {
    const float* x;   // stencil source (forward: input; avg backward: dy)
    float* y;         // destination (forward: output; avg backward: dx)
    uint8_t* idx;     // max forward: argmax out (may be null)
    int C, H, W, Ho, Wo;
    long x_img_stride, y_img_stride;
    long total;       // N * C * Ho * (Wo / V)
    uint32_t x_bytes;
    int accumulate;           // avg backward only
    const float* mask_y;      // avg backward only: fused ReLU + frozen-BN backward (see PoolBwdArgs)
    const float* mask_scale;
    long mask_img_stride;
    const float* aff_scale;   // average forward only: per-channel affine + ReLU on the pooled value (see PoolArgs)
    const float* aff_shift;
    int aff_relu;
    float* amax;              // amax slot of y's tensor (nullptr: not tracked)
    FastDiv div_chq, div_hq, div_q;   // C*Ho*Wq, Ho*Wq, Wq  (Wq = Wo / V)
};

template <int V>
__device__ __forceinline__ void pool_load_wide(const __amdgpu_buffer_rsrc_t& r, uint32_t off, float* dst) {
    if (V == 4) {
        const pu32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t u = t[e];
            dst[e] = __builtin_bit_cast(float, u);
        }
    } else {
        const pu32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t u = t[e];
            dst[e] = __builtin_bit_cast(float, u);
        }
    }
}
template <int V>
__device__ __forceinline__ void pool_store_wide(float* dst, const float* v) {
    if (V == 4)
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    else
        *reinterpret_cast<float2*>(dst) = float2{v[0], v[1]};
}

// 3x3 window.  SC = 2 (PC = 0, V = 4, W == 2 Wo): max pooling with argmax.  SC = 1 (PC = 1, W == Wo): max or average
// (count_include_pad: the divisor is always 9); with BWD the same stencil runs on dy and the store is the
// read-modify-write epilogue of pool_bwd_kernel (average backward = average forward of the gradient for s1/p1).
template <bool MAX, int SC, int PC, int V, bool BWD>
__global__ __launch_bounds__(256) void pool3_vec_kernel(PoolVecArgs p) {
    constexpr int NCOL = (V - 1) * SC + 3;
    const int Wq = p.Wo / V;
    const int howo = p.Ho * p.Wo;
    const __amdgpu_buffer_rsrc_t xr = pool_rsrc(p.x, p.x_bytes);
    float vmax = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long)gridDim.x * 256) {
        uint32_t n, rem, c, hq, ho, wq;
        fd_divmod((uint32_t)i, p.div_chq, n, rem);
        fd_divmod(rem, p.div_hq, c, hq);
        fd_divmod(hq, p.div_q, ho, wq);
        const int wo0 = (int)wq * V;
        const uint32_t base = (uint32_t)(((long)n * p.x_img_stride + (long)c * p.H * p.W) * 4);
        float v[3][NCOL];
        bool rok[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = (int)ho * SC - PC + r;
            rok[r] = (unsigned)hi < (unsigned)p.H;
            const uint32_t row = rok[r] ? base + (uint32_t)(hi * p.W) * 4u : POOL_OOB;
            if (SC == 2) {
                const uint32_t c0 = row + (uint32_t)(2 * wo0) * 4u;
                pool_load_wide<4>(xr, c0, &v[r][0]);
                pool_load_wide<4>(xr, c0 + 16u, &v[r][4]);
                v[r][8] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                        xr, (2 * wo0 + 8 < p.W) ? c0 + 32u : POOL_OOB, 0, 0));
            } else {
                const uint32_t c0 = row + (uint32_t)wo0 * 4u;
                v[r][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, wo0 > 0 ? c0 - 4u : POOL_OOB, 0, 0));
                pool_load_wide<V>(xr, c0, &v[r][1]);
                v[r][V + 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                            xr, (wo0 + V < p.W) ? c0 + 4u * V : POOL_OOB, 0, 0));
            }
        }
        const bool cok_first = (SC == 2) || (wo0 > 0);                       // column 0 of the strip exists
        const bool cok_last = (SC == 2) ? (2 * wo0 + 8 < p.W) : (wo0 + V < p.W);   // column NCOL-1 exists
        float out[V];
        uint32_t arg = 0;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            if (MAX) {
                // torch rule: start at the first in-window element, move on (v > best) or NaN
                float best = -INFINITY;
                int bi = 0;
                bool first = true;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int r = t / 3, col = e * SC + t % 3;
                    const bool ok = rok[r] && (col != 0 || cok_first) && (col != NCOL - 1 || cok_last);
                    const float x = v[r][col];
                    const bool take = ok && (first || x > best || x != x);
                    bi = (ok && first) ? t : bi;
                    first = first && !ok;
                    bi = take ? t : bi;
                    best = take ? x : best;
                }
                out[e] = best;
                arg |= (uint32_t)bi << (8 * e);
            } else {
                float s_ = 0.f;   // out-of-image taps were loaded as 0
#pragma unroll
                for (int t = 0; t < 9; ++t) s_ += v[t / 3][e * SC + t % 3];
                out[e] = s_ / 9.f;   // count_include_pad: a 3x3 / s1 / p1 window is never clipped by the padded extent
            }
        }
        if (!MAX && !BWD && p.aff_scale) {
            const float sc = p.aff_scale[c], sh = p.aff_shift[c];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                out[e] = out[e] * sc + sh;
                if (p.aff_relu) out[e] = fmaxf(out[e], 0.f);
            }
        }
        const long o = (long)n * p.y_img_stride + (long)c * howo + (long)ho * p.Wo + wo0;
        if (BWD) {
            float old[V], mk[V];
            if (p.accumulate) {
                if (V == 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(p.y + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) old[e] = t[e];
                } else {
                    const float2 t = *reinterpret_cast<const float2*>(p.y + o);
                    old[0] = t.x;
                    old[1] = t.y;
                }
#pragma unroll
                for (int e = 0; e < V; ++e) out[e] += old[e];
            }
            if (p.mask_y) {
                const long mo = (long)n * p.mask_img_stride + (long)c * howo + (long)ho * p.Wo + wo0;
                if (V == 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(p.mask_y + mo);
#pragma unroll
                    for (int e = 0; e < 4; ++e) mk[e] = t[e];
                } else {
                    const float2 t = *reinterpret_cast<const float2*>(p.mask_y + mo);
                    mk[0] = t.x;
                    mk[1] = t.y;
                }
                const float sc = p.mask_scale[c];
#pragma unroll
                for (int e = 0; e < V; ++e) out[e] = (sc != sc) ? out[e] : (mk[e] > 0.f ? out[e] * sc : 0.f);
            }
        }
        pool_store_wide<V>(p.y + o, out);
#pragma unroll
        for (int e = 0; e < V; ++e) vmax = fmaxf(vmax, fabsf(out[e]));
        if (MAX && p.idx) {
            uint8_t* ip = p.idx + ((long)n * p.C + c) * howo + (long)ho * p.Wo + wo0;
            if (V == 4)
                *reinterpret_cast<uint32_t*>(ip) = arg;
            else
                *reinterpret_cast<uint16_t*>(ip) = (uint16_t)arg;
        }
    }
    amax_emit(p.amax, vmax);
}

// 3x3 / stride 2 / pad 0 max-pool backward with W == 2 Wo (and Wo % 2 == 0, so a strip of 4 input pixels starts on an
// even output column): a thread owns input pixels (hi, wi0 .. wi0 + 3).  They are covered by output columns
// wq - 1, wq, wq + 1 (wq = wi0 / 2) and by one (odd hi) or two (even hi) output rows: 2 x (1 + 2) gradient dwords
// and 2 x (1 + 2) argmax bytes instead of 4 x 8 scalar loads.
__global__ __launch_bounds__(256) void pool_max2_bwd_vec_kernel(PoolBwdArgs p, FastDiv div_chq, FastDiv div_hq,
                                                                FastDiv div_q, long total_q) {
    const int howo = p.Ho * p.Wo;
    const __amdgpu_buffer_rsrc_t dyr = pool_rsrc(p.dy, p.dy_bytes);
    const __amdgpu_buffer_rsrc_t ixr = pool_rsrc(p.idx, p.idx_bytes);
    float vmax = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_q; i += (long)gridDim.x * 256) {
        uint32_t n, rem, c, hq, hi, q;
        fd_divmod((uint32_t)i, div_chq, n, rem);
        fd_divmod(rem, div_hq, c, hq);
        fd_divmod(hq, div_q, hi, q);
        const int wi0 = (int)q * 4, wq = (int)q * 2;
        const uint32_t dybase = (uint32_t)(((long)n * p.dy_img_stride + (long)c * howo) * 4);
        const uint32_t ixbase = (uint32_t)(((long)n * p.C + c) * howo);
        // output rows: a = 0 -> ho = hi / 2 (window row r = hi - 2 ho = 0 or 1); a = 1 -> ho = hi / 2 - 1 (r = 2, even hi)
        float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int ho = (int)hi / 2 - a;
            const int r = (int)hi - 2 * ho;
            const bool rowin = (ho >= 0) && (ho < p.Ho) && (r <= 2);
            const uint32_t e0 = (uint32_t)(ho * p.Wo + wq);
            // columns wq - 1 | wq, wq + 1
            const bool left = rowin && wq > 0;
            const float gl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                           dyr, left ? dybase + (e0 - 1u) * 4u : POOL_OOB, 0, 0));
            const uint32_t il = __builtin_amdgcn_raw_buffer_load_b8(ixr, left ? ixbase + e0 - 1u : POOL_OOB, 0, 0);
            const pu32x2 gm = __builtin_amdgcn_raw_buffer_load_b64(dyr, rowin ? dybase + e0 * 4u : POOL_OOB, 0, 0);
            const uint32_t im = __builtin_amdgcn_raw_buffer_load_b16(ixr, rowin ? ixbase + e0 : POOL_OOB, 0, 0);
            const uint32_t g0u = gm[0], g1u = gm[1];
            const float g0 = __builtin_bit_cast(float, g0u), g1 = __builtin_bit_cast(float, g1u);
            const uint32_t i0 = im & 0xFFu, i1 = (im >> 8) & 0xFFu;
            const uint32_t rb = (uint32_t)(r * 3);
            // pixel wi0 + d is tap s = wi0 + d - 2 wo of window wo
            g[0] += (left && il == rb + 2u) ? gl : 0.f;        // wo = wq - 1, s = 2
            g[0] += (rowin && i0 == rb + 0u) ? g0 : 0.f;       // wo = wq,     s = 0
            g[1] += (rowin && i0 == rb + 1u) ? g0 : 0.f;       //              s = 1
            g[2] += (rowin && i0 == rb + 2u) ? g0 : 0.f;       //              s = 2
            g[2] += (rowin && i1 == rb + 0u) ? g1 : 0.f;       // wo = wq + 1, s = 0
            g[3] += (rowin && i1 == rb + 1u) ? g1 : 0.f;       //              s = 1
        }
        const long o = (long)n * p.dx_img_stride + (long)c * p.H * p.W + (long)hi * p.W + wi0;
        if (p.accumulate) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p.dx + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] += t[e];
        }
        if (p.mask_y) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p.mask_y + (long)n * p.mask_img_stride + (long)c * p.H * p.W +
                                                            (long)hi * p.W + wi0);
            const float sc = p.mask_scale[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = (sc != sc) ? g[e] : (t[e] > 0.f ? g[e] * sc : 0.f);
        }
        *reinterpret_cast<f32x4*>(p.dx + o) = f32x4{g[0], g[1], g[2], g[3]};
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(g[0]), fabsf(g[1]))), fmaxf(fabsf(g[2]), fabsf(g[3])));
    }
    amax_emit(p.amax, vmax);
}

// The same for a vertical PAIR of input rows (2k, 2k + 1; H even): they are covered by the output rows k - 1 (window row 2, the
// even input row only) and k (window rows 0 and 1), so one thread fetches 2 x (1 + 2) gradient dwords and 2 x (1 + 2) argmax
// bytes for EIGHT input pixels -- half the load instructions per pixel of the one-row kernel, which is what bounds it.
__global__ __launch_bounds__(256) void pool_max2_bwd_vec2_kernel(PoolBwdArgs p, FastDiv div_chq, FastDiv div_hq,
                                                                 FastDiv div_q, long total_q) {
    const int howo = p.Ho * p.Wo;
    const __amdgpu_buffer_rsrc_t dyr = pool_rsrc(p.dy, p.dy_bytes);
    const __amdgpu_buffer_rsrc_t ixr = pool_rsrc(p.idx, p.idx_bytes);
    float vmax = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_q; i += (long)gridDim.x * 256) {
        uint32_t n, rem, c, hq, k, q;
        fd_divmod((uint32_t)i, div_chq, n, rem);     // div_chq = C * (H / 2) * Wq
        fd_divmod(rem, div_hq, c, hq);               // div_hq = (H / 2) * Wq
        fd_divmod(hq, div_q, k, q);
        const int wi0 = (int)q * 4, wq = (int)q * 2;
        const uint32_t dybase = (uint32_t)(((long)n * p.dy_img_stride + (long)c * howo) * 4);
        const uint32_t ixbase = (uint32_t)(((long)n * p.C + c) * howo);
        float g[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};     // [input row 2k / 2k + 1][pixel]
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int ho = (int)k - a;                        // a = 0: window rows 0 (even row) and 1 (odd row); a = 1: row 2 (even)
            const bool rowin = (ho >= 0) && (ho < p.Ho);
            const uint32_t e0 = (uint32_t)(ho * p.Wo + wq);
            const bool left = rowin && wq > 0;
            const float gl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                           dyr, left ? dybase + (e0 - 1u) * 4u : POOL_OOB, 0, 0));
            const uint32_t il = __builtin_amdgcn_raw_buffer_load_b8(ixr, left ? ixbase + e0 - 1u : POOL_OOB, 0, 0);
            const pu32x2 gm = __builtin_amdgcn_raw_buffer_load_b64(dyr, rowin ? dybase + e0 * 4u : POOL_OOB, 0, 0);
            const uint32_t im = __builtin_amdgcn_raw_buffer_load_b16(ixr, rowin ? ixbase + e0 : POOL_OOB, 0, 0);
            const uint32_t g0u = gm[0], g1u = gm[1];
            const float g0 = __builtin_bit_cast(float, g0u), g1 = __builtin_bit_cast(float, g1u);
            const uint32_t i0 = im & 0xFFu, i1 = (im >> 8) & 0xFFu;
#pragma unroll
            for (int row = 0; row < 2; ++row) {               // input row 2k + row is window row r of output row ho
                const int r = 2 * a + row;                    // a = 0: r = 0 / 1;  a = 1: r = 2 (even row only)
                if (r > 2) continue;
                const uint32_t rb = (uint32_t)(r * 3);
                g[row][0] += (left && il == rb + 2u) ? gl : 0.f;        // wo = wq - 1, s = 2
                g[row][0] += (rowin && i0 == rb + 0u) ? g0 : 0.f;       // wo = wq,     s = 0
                g[row][1] += (rowin && i0 == rb + 1u) ? g0 : 0.f;       //              s = 1
                g[row][2] += (rowin && i0 == rb + 2u) ? g0 : 0.f;       //              s = 2
                g[row][2] += (rowin && i1 == rb + 0u) ? g1 : 0.f;       // wo = wq + 1, s = 0
                g[row][3] += (rowin && i1 == rb + 1u) ? g1 : 0.f;       //              s = 1
            }
        }
        const float sc = p.mask_y ? p.mask_scale[c] : 0.f;
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const long o = (long)n * p.dx_img_stride + (long)c * p.H * p.W + (long)(2 * k + row) * p.W + wi0;
            if (p.accumulate) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(p.dx + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[row][e] += t[e];
            }
            if (p.mask_y) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(p.mask_y + (long)n * p.mask_img_stride + (long)c * p.H * p.W +
                                                                (long)(2 * k + row) * p.W + wi0);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[row][e] = (sc != sc) ? g[row][e] : (t[e] > 0.f ? g[row][e] * sc : 0.f);
            }
            *reinterpret_cast<f32x4*>(p.dx + o) = f32x4{g[row][0], g[row][1], g[row][2], g[row][3]};
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(g[row][0]), fabsf(g[row][1]))), fmaxf(fabsf(g[row][2]), fabsf(g[row][3])));
        }
    }
    amax_emit(p.amax, vmax);
}

__global__ __launch_bounds__(256) void gap_fwd_kernel(const float* x, float* y, int NC, int C, int HW,
                                                      long x_img_stride) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= NC) return;
    const int n = wave / C, c = wave - n * C;
    const float* xp = x + (long)n * x_img_stride + (long)c * HW;
    float s = 0.f;
    for (int i = lane; i < HW; i += 64) s += xp[i];
    s = wave_sum(s);
    if (lane == 0) y[wave] = s / (float)HW;
}
// dx[n][c][:] (+)= dy[n][c] / HW: one wave per (n, c) plane, lanes along the contiguous pixels (no divisions per element)
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* dy, float* dx, int NC, int C, int HW,
                                                      long dx_img_stride, int accumulate, float* amax) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    float vmax = 0.f;
    if (wave < NC) {
        const int n = wave / C, c = wave - n * C;
        const float g = dy[wave] / (float)HW;
        float* dst = dx + (long)n * dx_img_stride + (long)c * HW;
        for (int i = lane; i < HW; i += 64) {
            const float o = accumulate ? dst[i] + g : g;
            dst[i] = o;
            vmax = fmaxf(vmax, fabsf(o));
        }
    }
    amax_emit(amax, vmax);
}

inline unsigned grid_for(long total, int cap = 65536) {
    long b = (total + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

static int pool_fwd_impl(int is_max, const float* x, float* y, unsigned char* argmax, int N, int C, int H, int W,
                         long x_img_stride, int Ho, int Wo, long y_img_stride, int ksize, int stride, int pad,
                         const float* aff_scale, const float* aff_shift, int aff_relu, float* y_amax,
                         hipStream_t stream) {
    SSN_CHECK_ARG(x && y, "pool_fwd: null pointer");
    PoolArgs a;
    a.amax = y_amax;
    a.aff_scale = aff_scale;
    a.aff_shift = aff_shift;
    a.aff_relu = aff_relu;
    a.x = x;
    a.y = y;
    a.idx = (uint8_t*)argmax;
    a.N = N;
    a.C = C;
    a.H = H;
    a.W = W;
    a.Ho = Ho;
    a.Wo = Wo;
    a.x_img_stride = x_img_stride;
    a.y_img_stride = y_img_stride;
    a.k = ksize;
    a.stride = stride;
    a.pad = pad;
    a.total = (long)N * C * Ho * Wo;
    SSN_CHECK_ARG(a.total < (1l << 31), "pool_fwd: tensor too large");
    const long xb = ((long)(N - 1) * x_img_stride + (long)C * H * W) * 4;
    SSN_CHECK_ARG(xb < (1l << 31), "pool_fwd: operand larger than 2 GiB (buffer addressing)");
    a.x_bytes = (uint32_t)xb;
    a.div_chw = make_fastdiv((uint32_t)(C * Ho * Wo));
    a.div_hw = make_fastdiv((uint32_t)(Ho * Wo));
    a.div_w = make_fastdiv((uint32_t)Wo);
    // vectorised paths (strips of 4 or 2 outputs per thread) where the rows line up; everything is 16/8-byte aligned
    // then because all channel planes are multiples of 4 (2) floats
    const bool al16 = ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && x_img_stride % 4 == 0 && y_img_stride % 4 == 0;
    const int vec = (ksize == 3 && al16 && (H * W) % 4 == 0)
                        ? ((is_max && stride == 2 && pad == 0 && W == 2 * Wo && Wo % 4 == 0) ? 4
                           : (!is_max && stride == 1 && pad == 1 && W == Wo && H == Ho) ? (Wo % 4 == 0 ? 4 : (Wo % 2 == 0 ? 2 : 0))
                                                                                    : 0)
                        : 0;
    if (vec) {
        PoolVecArgs v;
        v.amax = y_amax;
        v.x = x;
        v.y = y;
        v.idx = (uint8_t*)argmax;
        v.C = C;
        v.H = H;
        v.W = W;
        v.Ho = Ho;
        v.Wo = Wo;
        v.x_img_stride = x_img_stride;
        v.y_img_stride = y_img_stride;
        v.total = (long)N * C * Ho * (Wo / vec);
        v.x_bytes = a.x_bytes;
        v.accumulate = 0;
        v.mask_y = nullptr;
        v.mask_scale = nullptr;
        v.mask_img_stride = 0;
        v.aff_scale = aff_scale;
        v.aff_shift = aff_shift;
        v.aff_relu = aff_relu;
        v.div_chq = make_fastdiv((uint32_t)(C * Ho * (Wo / vec)));
        v.div_hq = make_fastdiv((uint32_t)(Ho * (Wo / vec)));
        v.div_q = make_fastdiv((uint32_t)(Wo / vec));
        const dim3 vgrid(grid_for(v.total));
        if (is_max)
            hipLaunchKernelGGL((pool3_vec_kernel<true, 2, 0, 4, false>), vgrid, dim3(256), 0, stream, v);
        else if (vec == 4)
            hipLaunchKernelGGL((pool3_vec_kernel<false, 1, 1, 4, false>), vgrid, dim3(256), 0, stream, v);
        else
            hipLaunchKernelGGL((pool3_vec_kernel<false, 1, 1, 2, false>), vgrid, dim3(256), 0, stream, v);
        SSN_CHECK_LAUNCH("pool_fwd (vec)");
        return SSN_OK;
    }
    const dim3 grid(grid_for(a.total));
    if (ksize == 3 && stride == 2 && pad == 0 && is_max)
        hipLaunchKernelGGL((pool_fwd_kernel<true, 3, 2, 0>), grid, dim3(256), 0, stream, a);
    else if (ksize == 3 && stride == 1 && pad == 1 && is_max)
        hipLaunchKernelGGL((pool_fwd_kernel<true, 3, 1, 1>), grid, dim3(256), 0, stream, a);
    else if (ksize == 3 && stride == 1 && pad == 1)
        hipLaunchKernelGGL((pool_fwd_kernel<false, 3, 1, 1>), grid, dim3(256), 0, stream, a);
    else if (ksize == 3 && stride == 2 && pad == 0)
        hipLaunchKernelGGL((pool_fwd_kernel<false, 3, 2, 0>), grid, dim3(256), 0, stream, a);
    else {
        ssn_set_error("pool_fwd: window (k=%d, s=%d, p=%d) is not instantiated (BN-Inception uses 3/2/0 and 3/1/1)",
                      ksize, stride, pad);
        return SSN_ERR_ARG;
    }
    SSN_CHECK_LAUNCH("pool_fwd");
    return SSN_OK;
}

extern "C" int ssn_pool_fwd(int is_max, const float* x, float* y, unsigned char* argmax, int N, int C, int H, int W,
                            long x_img_stride, int Ho, int Wo, long y_img_stride, int ksize, int stride, int pad,
                            float* y_amax, hipStream_t stream) {
    return pool_fwd_impl(is_max, x, y, argmax, N, C, H, W, x_img_stride, Ho, Wo, y_img_stride, ksize, stride, pad, nullptr,
                         nullptr, 0, y_amax, stream);
}

// y = relu?(scale[c] * avgpool(x) + shift[c]): the pool-projection branch of an Inception block with the pool moved
// BEHIND its (linear, bias-free) 1x1 convolution -- avgpool(conv1x1(x)) == conv1x1(avgpool(x)) for zero padding with
// count_include_pad -- so the pool runs on the projection's few output channels instead of the block's many input
// channels, and the folded BN affine + ReLU of the projection are applied here.
extern "C" int ssn_avgpool_affine_fwd(const float* x, float* y, const float* scale, const float* shift, int relu, int N,
                                      int C, int H, int W, long x_img_stride, int Ho, int Wo, long y_img_stride,
                                      int ksize, int stride, int pad, float* y_amax, hipStream_t stream) {
    SSN_CHECK_ARG(scale && shift, "avgpool_affine_fwd: null pointer");
    return pool_fwd_impl(0, x, y, nullptr, N, C, H, W, x_img_stride, Ho, Wo, y_img_stride, ksize, stride, pad, scale, shift,
                         relu, y_amax, stream);
}

extern "C" int ssn_pool_bwd(int is_max, const float* dy, const unsigned char* argmax, float* dx, int N, int C, int H,
                            int W, long dx_img_stride, int Ho, int Wo, long dy_img_stride, int ksize, int stride,
                            int pad, int accumulate, const float* mask_y, long mask_img_stride,
                            const float* mask_scale, float* dx_amax, hipStream_t stream) {
    SSN_CHECK_ARG(dy && dx && (!is_max || argmax), "pool_bwd: null pointer");
    PoolBwdArgs a;
    a.amax = dx_amax;
    a.dy = dy;
    a.idx = (const uint8_t*)argmax;
    a.dx = dx;
    a.N = N;
    a.C = C;
    a.H = H;
    a.W = W;
    a.Ho = Ho;
    a.Wo = Wo;
    a.dy_img_stride = dy_img_stride;
    a.dx_img_stride = dx_img_stride;
    a.k = ksize;
    a.stride = stride;
    a.pad = pad;
    a.accumulate = accumulate;
    a.mask_y = mask_scale ? mask_y : nullptr;
    a.mask_scale = mask_y ? mask_scale : nullptr;
    a.mask_img_stride = mask_img_stride;
    a.total = (long)N * C * H * W;
    SSN_CHECK_ARG(a.total < (1l << 31), "pool_bwd: tensor too large");
    const long db = ((long)(N - 1) * dy_img_stride + (long)C * Ho * Wo) * 4;
    SSN_CHECK_ARG(db < (1l << 31), "pool_bwd: operand larger than 2 GiB (buffer addressing)");
    a.dy_bytes = (uint32_t)db;
    a.idx_bytes = (uint32_t)((long)N * C * Ho * Wo);
    a.div_chw = make_fastdiv((uint32_t)(C * H * W));
    a.div_hw = make_fastdiv((uint32_t)(H * W));
    a.div_w = make_fastdiv((uint32_t)W);
    const bool al16 = ((uintptr_t)dx % 16 == 0) && dx_img_stride % 4 == 0 && (H * W) % 4 == 0 &&
                      (!a.mask_y || (((uintptr_t)a.mask_y % 16 == 0) && mask_img_stride % 4 == 0));
    if (ksize == 3 && al16 && is_max && stride == 2 && pad == 0 && W == 2 * Wo && Wo % 2 == 0) {
        const int Wq = W / 4;
        if (H % 2 == 0) {      // vertical pairs of input rows per thread
            const long total_p = (long)N * C * (H / 2) * Wq;
            hipLaunchKernelGGL(pool_max2_bwd_vec2_kernel, dim3(grid_for(total_p)), dim3(256), 0, stream, a,
                               make_fastdiv((uint32_t)(C * (H / 2) * Wq)), make_fastdiv((uint32_t)((H / 2) * Wq)),
                               make_fastdiv((uint32_t)Wq), total_p);
            SSN_CHECK_LAUNCH("pool_bwd (vec max, row pairs)");
            return SSN_OK;
        }
        const long total_q = (long)N * C * H * Wq;
        hipLaunchKernelGGL(pool_max2_bwd_vec_kernel, dim3(grid_for(total_q)), dim3(256), 0, stream, a,
                           make_fastdiv((uint32_t)(C * H * Wq)), make_fastdiv((uint32_t)(H * Wq)),
                           make_fastdiv((uint32_t)Wq), total_q);
        SSN_CHECK_LAUNCH("pool_bwd (vec max)");
        return SSN_OK;
    }
    if (ksize == 3 && al16 && !is_max && stride == 1 && pad == 1 && W == Wo && H == Ho && W % 2 == 0 &&
        ((uintptr_t)dy % 16 == 0) && dy_img_stride % 4 == 0) {
        const int vec = W % 4 == 0 ? 4 : 2;
        PoolVecArgs v;
        v.amax = dx_amax;
        v.x = dy;
        v.y = dx;
        v.idx = nullptr;
        v.C = C;
        v.H = Ho;
        v.W = Wo;
        v.Ho = H;
        v.Wo = W;
        v.x_img_stride = dy_img_stride;
        v.y_img_stride = dx_img_stride;
        v.total = (long)N * C * H * (W / vec);
        v.x_bytes = a.dy_bytes;
        v.accumulate = accumulate;
        v.mask_y = a.mask_y;
        v.mask_scale = a.mask_scale;
        v.mask_img_stride = mask_img_stride;
        v.aff_scale = nullptr;
        v.aff_shift = nullptr;
        v.aff_relu = 0;
        v.div_chq = make_fastdiv((uint32_t)(C * H * (W / vec)));
        v.div_hq = make_fastdiv((uint32_t)(H * (W / vec)));
        v.div_q = make_fastdiv((uint32_t)(W / vec));
        const dim3 vgrid(grid_for(v.total));
        if (vec == 4)
            hipLaunchKernelGGL((pool3_vec_kernel<false, 1, 1, 4, true>), vgrid, dim3(256), 0, stream, v);
        else
            hipLaunchKernelGGL((pool3_vec_kernel<false, 1, 1, 2, true>), vgrid, dim3(256), 0, stream, v);
        SSN_CHECK_LAUNCH("pool_bwd (vec avg)");
        return SSN_OK;
    }
    const dim3 grid(grid_for(a.total));
    if (ksize == 3 && stride == 2 && pad == 0 && is_max)
        hipLaunchKernelGGL((pool_bwd_kernel<true, 3, 2, 0>), grid, dim3(256), 0, stream, a);
    else if (ksize == 3 && stride == 1 && pad == 1 && is_max)
        hipLaunchKernelGGL((pool_bwd_kernel<true, 3, 1, 1>), grid, dim3(256), 0, stream, a);
    else if (ksize == 3 && stride == 1 && pad == 1)
        hipLaunchKernelGGL((pool_bwd_kernel<false, 3, 1, 1>), grid, dim3(256), 0, stream, a);
    else if (ksize == 3 && stride == 2 && pad == 0)
        hipLaunchKernelGGL((pool_bwd_kernel<false, 3, 2, 0>), grid, dim3(256), 0, stream, a);
    else {
        ssn_set_error("pool_bwd: window (k=%d, s=%d, p=%d) is not instantiated (BN-Inception uses 3/2/0 and 3/1/1)",
                      ksize, stride, pad);
        return SSN_ERR_ARG;
    }
    SSN_CHECK_LAUNCH("pool_bwd");
    return SSN_OK;
}

extern "C" int ssn_global_avgpool_fwd(const float* x, float* y, int N, int C, int HW, long x_img_stride,
                                      hipStream_t stream) {
    SSN_CHECK_ARG(x && y, "gap_fwd: null pointer");
    const int NC = N * C;
    hipLaunchKernelGGL(gap_fwd_kernel, dim3((NC + 3) / 4), dim3(256), 0, stream, x, y, NC, C, HW, x_img_stride);
    SSN_CHECK_LAUNCH("gap_fwd");
    return SSN_OK;
}
extern "C" int ssn_global_avgpool_bwd(const float* dy, float* dx, int N, int C, int HW, long dx_img_stride,
                                      int accumulate, float* dx_amax, hipStream_t stream) {
    SSN_CHECK_ARG(dy && dx, "gap_bwd: null pointer");
    const int NC = N * C;
    hipLaunchKernelGGL(gap_bwd_kernel, dim3((NC + 3) / 4), dim3(256), 0, stream, dy, dx, NC, C, HW, dx_img_stride, accumulate,
                       dx_amax);
    SSN_CHECK_LAUNCH("gap_bwd");
    return SSN_OK;
}
