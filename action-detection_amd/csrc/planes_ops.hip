// Planes tensors (planes.h): conversion from / to fp32 NCHW, scale bookkeeping, and the HBM-bound layers of the backbone
// in the planes layout (pools, global pool, channel sums) -- the elementwise side of the path behind
// /root/reference/ssn_models.py:266, gfx950.  One thread = one pixel x 8 channels: 16-byte loads / stores per plane.
#include "planes.h"

namespace {

using namespace pl;

// ---- scale bookkeeping ----
// scale[i] <- scale for the next step from amax[i] (kept when the tensor was not written), amax[i] <- 0.
// flag[0] |= 1 when a tensor outgrew the scale it was stored with (values were clamped: the host re-calibrates);
// flag[1] counts the slots whose scale changed (calibration runs until this stays 0).
// Slots with an ODD global index (slot0 + i: the executor's gradient tensors) take `odd_bits` more bits of head-room: the largest
// element of a gradient tensor moves far more from step to step than an activation's does (dropout masks, OHEM selections: > 5x
// between consecutive steps on the same batch was measured, round 4) -- with 2 bits the guard fired every ~10th step.
__global__ __launch_bounds__(256) void scales_update_kernel(float* amax, float* scale, int* flag, int n, int exact, int slot0,
                                                            int odd_bits) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = amax[i], s = scale[i];
    const float room = ((slot0 + i) & 1) ? 1.f / (float)(1 << odd_bits) : 1.f;     // extra head-room as a factor on the targets
    // (exact: the tensor is measured BEFORE it is stored with the scale derived here -- nothing was clamped, whatever the old scale)
    if (!exact && (a * s >= PL_F16_MAX || a != a)) atomicOr(flag, 1);
    float ns = pl_scale_from_amax(a, s);
    if (exact) ns *= (float)(1 << PL_HEADROOM_BITS);   // a tensor measured before it is stored needs no head-room
    else if (ns != s) ns *= room;                      // (ns == s: amax was 0 / not finite and the old scale stands)
    // hysteresis: keep the old scale while the stored maximum stays inside [2^10, 2^14) (x room) -- at least 2 bits of head-room, at
    // most 3 bits of the low plane's range given away -- so that scales do not flap between steps
    if (!exact && a * s < 16384.f * room && a * s >= 1024.f * room) ns = s;
    if (ns != s) atomicAdd(flag + 1, 1);
    scale[i] = ns;
    amax[i] = 0.f;
}

// The guard of the delayed scales, one launch at the END of a pass (forward: activation slots, backward: gradient slots; slots
// nobody wrote hold 0 and are skipped): flag[0] |= 1 when a tensor's recorded (pre-clamp) maximum did not fit the scale it was stored
// with -- its values were clamped --, |= 2 when it sat more than PL_UNDERFLOW_BITS below the target (the low plane has run out of
// exponent range: precision is draining).  Reads only: the slots stay as they are for the update at the head of the next pass.
// The host (planes_exec.py) polls the word -- synchronously after an eager pass, asynchronously behind a graph replay -- and repeats
// the pass with fresh scales; ssn_sgd_step_multi skips its update while the word is set, so a flagged step is retryable.
__global__ __launch_bounds__(256) void range_check_kernel(const float* amax, const float* scale, int* flag, int n, int odd_bits) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = amax[i], s = scale[i];
    const float room = (i & 1) ? 1.f / (float)(1 << odd_bits) : 1.f;
    if (a * s >= PL_F16_MAX || a != a)
        atomicOr(flag, 1);
    else if (a > 0.f && a * s < PL_UNDERFLOW_FLOOR * room)
        atomicOr(flag, 2);
}

struct CvtArgs {
    const float* x;
    void* hi;
    void* lo;
    const float* scale;
    float* amax;
    long x_img_stride;       // floats between images of x
    int N, C, H, W;          // x dims (C real channels)
    int G;                   // channel groups written (ceil(C' / 8), C' = C or 4C)
    long img_groups;         // groups of the whole planes tensor
    int s2d;                 // space-to-depth: planes channel (c*2+a)*2+b at (h', w') = x[c][2h'+a][2w'+b]
    FastDiv dv_hw, dv_g, dv_wo;   // output pixels per plane, channel groups, output row length (s2d)
};

// Index arithmetic of the kernels below is 32-bit with host-computed magic divisions: a plane holds < 2^27 16-byte groups (2 GiB
// buffer addressing), and three or four 64-bit `idx % W` per element cost a few hundred instructions (round 5: -4 % on the stem pools'
// backward; what bounds those kernels is their selection arithmetic and the redundant window loads: profiles/r5_pool_lab.txt).
struct PoolIdx {
    uint32_t n, g, h, w;
};
__device__ __forceinline__ PoolIdx pool_decode(uint32_t idx, const FastDiv& dw, const FastDiv& dh, const FastDiv& dg) {
    PoolIdx r;
    uint32_t t, ng;
    fd_divmod(idx, dw, t, r.w);
    fd_divmod(t, dh, ng, r.h);
    fd_divmod(ng, dg, r.n, r.g);
    return r;
}

__global__ __launch_bounds__(256) void pl_from_f32_kernel(CvtArgs p) {
    const int HWo = p.s2d ? (p.H / 2) * (p.W / 2) : p.H * p.W;
    const uint32_t total = (uint32_t)p.N * (uint32_t)p.G * (uint32_t)HWo;
    const float s = *p.scale;
    float vmax = 0.f;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        uint32_t qq, ng, gg, nn;
        fd_divmod(idx, p.dv_hw, ng, qq);
        fd_divmod(ng, p.dv_g, nn, gg);
        const int q = (int)qq, g = (int)gg, n = (int)nn;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 8 * g + e;
            float x = 0.f;
            if (!p.s2d) {
                if (c < p.C) x = p.x[(long)n * p.x_img_stride + (long)c * HWo + q];
            } else if (c < 4 * p.C) {
                const int cc = c >> 2, a = (c >> 1) & 1, b = c & 1;
                uint32_t hq, wq;
                fd_divmod((uint32_t)q, p.dv_wo, hq, wq);
                const int ho = (int)hq, wo = (int)wq;
                x = p.x[(long)n * p.x_img_stride + ((long)cc * p.H + 2 * ho + a) * p.W + 2 * wo + b];
            }
            vmax = fmaxf(vmax, fabsf(x));
            v[e] = pl_clamp(x * s);
        }
        u32x4 hi, lo;
        pl_split8(v, hi, lo);
        const long o = (((long)n * p.img_groups + g) * HWo + q);
        reinterpret_cast<u32x4*>(p.hi)[o] = hi;
        reinterpret_cast<u32x4*>(p.lo)[o] = lo;
    }
    amax_emit_block(p.amax, vmax);
}

__global__ __launch_bounds__(256) void pl_to_f32_kernel(const void* hi, const void* lo, long img_groups, float* y,
                                                       long y_img_stride, int N, int C, int HW, const float* scale, FastDiv dv_hw,
                                                       FastDiv dv_g) {
    const int G = (C + 7) / 8;
    const uint32_t total = (uint32_t)N * (uint32_t)G * (uint32_t)HW;
    const float inv = 1.f / *scale;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        uint32_t qq, ng, gg, nn;
        fd_divmod(idx, dv_hw, ng, qq);
        fd_divmod(ng, dv_g, nn, gg);
        const int q = (int)qq, g = (int)gg, n = (int)nn;
        const long o = (((long)n * img_groups + g) * HW + q);
        float v[8];
        pl_join8(reinterpret_cast<const u32x4*>(hi)[o], reinterpret_cast<const u32x4*>(lo)[o], v);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (8 * g + e < C) y[(long)n * y_img_stride + (long)(8 * g + e) * HW + q] = v[e] * inv;
    }
}

// im2col of a planes tensor with FEW channels (the first convolution of a backbone that is not run in space-to-depth form: 3 or 10
// channels in one or two groups): y[n][c * kh * kw + r * kw + s][ho][wo] = x[n][c][ho * stride + r - pad_h][wo * stride + s - pad_w] (zero
// outside), both planes copied as they are -- no arithmetic, y shares x's scale.  With it the weight gradient of that layer is a 1x1
// problem on K = C kh kw channels (dW[m][c][r][s] IS dW'[m][c kh kw + r kw + s]) instead of kh kw taps of a 3-channel operand that each
// re-read the whole output gradient (Inception-v3's 3 -> 32 3x3 / 2 layer at 299 x 299: 1.9 ms on the one-tap body, 3 TF).
// One thread = one output pixel and one output channel group (8 of the K channels).
struct Im2colArgs {
    const void* x_hi;
    const void* x_lo;
    void* y_hi;
    void* y_lo;
    long x_img_groups, y_img_groups;
    int N, C, H, W, Ho, Wo, kh, kw, stride, pad_h, pad_w;
    int K, KG;               // C * kh * kw and its channel groups
    FastDiv dv_hw, dv_kg, dv_wo, dv_kk, dv_kw;
};
__global__ __launch_bounds__(256) void pl_im2col_kernel(Im2colArgs p) {
    const uint32_t HoWo = (uint32_t)p.Ho * (uint32_t)p.Wo;
    const uint32_t total = (uint32_t)p.N * (uint32_t)p.KG * HoWo;
    const uint32_t HW = (uint32_t)p.H * (uint32_t)p.W;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        uint32_t q, ng, kg, n, ho, wo;
        fd_divmod(idx, p.dv_hw, ng, q);
        fd_divmod(ng, p.dv_kg, n, kg);
        fd_divmod(q, p.dv_wo, ho, wo);
        uint32_t hi[4] = {0u, 0u, 0u, 0u}, lo[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t k = kg * 8u + (uint32_t)e;
            uint32_t c, t, r, s_;
            fd_divmod(k, p.dv_kk, c, t);
            fd_divmod(t, p.dv_kw, r, s_);
            const int h = (int)ho * p.stride + (int)r - p.pad_h, w = (int)wo * p.stride + (int)s_ - p.pad_w;
            const bool ok = k < (uint32_t)p.K && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            uint32_t vh = 0u, vl = 0u;
            if (ok) {
                const long o = (((long)n * p.x_img_groups + (c >> 3)) * HW + (uint32_t)(h * p.W + w)) * 8 + (c & 7u);      // f16 index
                vh = reinterpret_cast<const unsigned short*>(p.x_hi)[o];
                vl = reinterpret_cast<const unsigned short*>(p.x_lo)[o];
            }
            hi[e >> 1] |= vh << (16 * (e & 1));
            lo[e >> 1] |= vl << (16 * (e & 1));
        }
        const long o = ((long)n * p.y_img_groups + kg) * HoWo + q;
        reinterpret_cast<u32x4*>(p.y_hi)[o] = u32x4{hi[0], hi[1], hi[2], hi[3]};
        reinterpret_cast<u32x4*>(p.y_lo)[o] = u32x4{lo[0], lo[1], lo[2], lo[3]};
    }
}

// ---- pooling on planes: one thread = one (image, channel group, pixel) = 8 channels, 16-byte accesses per plane ----
struct PoolArgs {
    const void* x_hi;     // input slice (forward: activation; backward: output gradient)
    const void* x_lo;
    void* y_hi;           // output slice
    void* y_lo;
    unsigned char* argmax;   // max pool: window-local index per output element, [N][G][Ho Wo][8] bytes (compact, slice-local)
    const float* x_scale;
    const float* y_scale;
    float* y_amax;
    const float* aff_scale;  // avgpool_affine: per channel (of the slice) scale / shift; backward: mask scale (NaN = pass through)
    const float* aff_shift;
    const void* mask_hi;     // backward: hi plane of the forward activation whose ReLU / frozen-BN backward is fused (or null)
    long x_img_groups, y_img_groups, mask_img_groups;
    int N, G;                // images, channel groups of the slice
    int H, W;                // spatial size of the pool's INPUT tensor
    int Ho, Wo;              // ... and of its OUTPUT tensor
    int k, s, pad;
    int relu, accumulate;
    float* y_f32;            // backward kernels: store fp32 NCHW here instead of planes (a gradient only a fp32-layout kernel reads)
    long y_f32_img_stride;
    FastDiv dv_w, dv_h, dv_wo, dv_ho, dv_g, dv_hw;   // W, H, Wo, Ho, G, H * W (index decoding: pool_decode)
    FastDiv dv_wb, dv_hb;                            // ceil(W / 2), ceil(H / 2): the 2 x 2 input blocks of the k3s2 backward
};

__device__ __forceinline__ void load8(const void* hi, const void* lo, long o, float (&v)[8]) {
    pl_join8(reinterpret_cast<const u32x4*>(hi)[o], reinterpret_cast<const u32x4*>(lo)[o], v);
}

// max pool forward (ceil_mode windows computed by the host; taps outside the image are skipped; first maximum in scan
// order wins, as torch: gradients then route identically through the exact-zero ties behind every ReLU)
template <int KT, int ST>
__global__ __launch_bounds__(256) void pl_maxpool_fwd_kernel(PoolArgs p) {
    const int pk = KT ? KT : p.k, ps = ST ? ST : p.s;
    const uint32_t total = (uint32_t)p.N * (uint32_t)p.G * (uint32_t)p.Ho * (uint32_t)p.Wo;
    const float r = *p.y_scale / *p.x_scale;
    float vmax = 0.f;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const PoolIdx ix = pool_decode(idx, p.dv_wo, p.dv_ho, p.dv_g);
        const int wo = (int)ix.w, ho = (int)ix.h, g = (int)ix.g, n = (int)ix.n;
        const long ibase = ((long)n * p.x_img_groups + g) * p.H * p.W;
        float best[8];
        unsigned char arg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            best[e] = -__builtin_inff();
            arg[e] = 0;
        }
        if constexpr (KT > 0) {
            // all taps of the window in flight at once (taps outside the image re-read the window's first valid pixel and are
            // kept out of the comparison)
            u32x4 thi[KT * KT], tlo[KT * KT];
            bool ok[KT * KT];
#pragma unroll
            for (int t = 0; t < KT * KT; ++t) {
                const int h = ho * ps - p.pad + t / KT, w = wo * ps - p.pad + t % KT;
                ok[t] = ((unsigned)h < (unsigned)p.H) && ((unsigned)w < (unsigned)p.W);
                const long o = ibase + (ok[t] ? (long)h * p.W + w : 0);
                thi[t] = reinterpret_cast<const u32x4*>(p.x_hi)[o];
                tlo[t] = reinterpret_cast<const u32x4*>(p.x_lo)[o];
            }
#pragma unroll
            for (int t = 0; t < KT * KT; ++t) {
                float v[8];
                pl_join8(thi[t], tlo[t], v);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (ok[t] && v[e] > best[e]) {
                        best[e] = v[e];
                        arg[e] = (unsigned char)t;
                    }
            }
        } else {
            for (int dr = 0; dr < pk; ++dr) {
                const int h = ho * ps - p.pad + dr;
                if ((unsigned)h >= (unsigned)p.H) continue;
                for (int ds = 0; ds < pk; ++ds) {
                    const int w = wo * ps - p.pad + ds;
                    if ((unsigned)w >= (unsigned)p.W) continue;
                    float v[8];
                    load8(p.x_hi, p.x_lo, ibase + (long)h * p.W + w, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (v[e] > best[e]) {
                            best[e] = v[e];
                            arg[e] = (unsigned char)(dr * pk + ds);
                        }
                }
            }
        }
        float out[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sv = best[e] * r;
            vmax = fmaxf(vmax, fabsf(sv));      // the TRUE magnitude (before the clamp): what the next scale is derived from
            out[e] = pl_clamp(sv);
        }
        u32x4 hi, lo;
        pl_split8(out, hi, lo);
        const long o = ((long)n * p.y_img_groups + g) * p.Ho * p.Wo + (long)ho * p.Wo + wo;
        reinterpret_cast<u32x4*>(p.y_hi)[o] = hi;
        reinterpret_cast<u32x4*>(p.y_lo)[o] = lo;
        if (p.argmax) {
            uint32_t a0 = arg[0] | (arg[1] << 8) | (arg[2] << 16) | ((uint32_t)arg[3] << 24);
            uint32_t a1 = arg[4] | (arg[5] << 8) | (arg[6] << 16) | ((uint32_t)arg[7] << 24);
            reinterpret_cast<u32x2*>(p.argmax)[idx] = u32x2{a0, a1};
        }
    }
    amax_emit_block(p.y_amax, vmax / *p.y_scale);
}

// 3x3 max pool forward written for instruction count (as pl_maxpool_bwd_k3s2_fast_kernel below: buffer loads with 32-bit offsets, no
// predicate state carried across the nine taps -- a tap outside the image is loaded from nowhere (zeros) and turned into -inf by
// eight selects on the packed words --, the argmax kept in one register per channel).  The general kernel above spills its nine tap
// predicates through lane writes (213 of its 1100 vector instructions).  Same results: first maximum in scan order.
template <int ST>
__global__ __launch_bounds__(256) void pl_maxpool_fwd_k3_fast_kernel(PoolArgs p) {
    const uint32_t total = (uint32_t)p.N * (uint32_t)p.G * (uint32_t)p.Ho * (uint32_t)p.Wo;
    const float r = *p.y_scale / *p.x_scale;
    const uint32_t HW = (uint32_t)p.H * (uint32_t)p.W;
    const uint32_t x_bytes = (uint32_t)(((long)(p.N - 1) * p.x_img_groups + p.G) * HW * 16);
    const __amdgpu_buffer_rsrc_t r_hi = pl_rsrc(p.x_hi, x_bytes), r_lo = pl_rsrc(p.x_lo, x_bytes);
    float vmax = 0.f;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const PoolIdx ix = pool_decode(idx, p.dv_wo, p.dv_ho, p.dv_g);
        const int wo = (int)ix.w, ho = (int)ix.h;
        const uint32_t xb = (ix.n * (uint32_t)p.x_img_groups + ix.g) * HW;
        u32x4 thi[9], tlo[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int h = ho * ST - p.pad + t / 3, w = wo * ST - p.pad + t % 3;
            const bool ok = ((unsigned)h < (unsigned)p.H) && ((unsigned)w < (unsigned)p.W);
            const uint32_t off = ok ? (xb + (uint32_t)(h * p.W + w)) * 16u : PL_OOB;
            thi[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r_hi, off, 0, 0));
            tlo[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r_lo, off, 0, 0));
#pragma unroll
            for (int d = 0; d < 4; ++d) thi[t][d] = ok ? thi[t][d] : 0xFC00FC00u;      // -inf: never the maximum (its lo word is 0)
        }
        float best[8];
        uint32_t arg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            best[e] = -__builtin_inff();
            arg[e] = 0u;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float v[8];
            pl_join8(thi[t], tlo[t], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool gt = v[e] > best[e];
                best[e] = gt ? v[e] : best[e];
                arg[e] = gt ? (uint32_t)t : arg[e];
            }
            __builtin_amdgcn_sched_barrier(0);      // (a tap's eight compare masks die here: hoisted together they spill the scalar file)
        }
        float out[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sv = best[e] * r;
            vmax = fmaxf(vmax, fabsf(sv));      // the TRUE magnitude (before the clamp)
            out[e] = pl_clamp(sv);
        }
        u32x4 hi, lo;
        pl_split8(out, hi, lo);
        const long o = ((long)ix.n * p.y_img_groups + ix.g) * p.Ho * p.Wo + (long)ho * p.Wo + wo;
        reinterpret_cast<u32x4*>(p.y_hi)[o] = hi;
        reinterpret_cast<u32x4*>(p.y_lo)[o] = lo;
        if (p.argmax) {
            const uint32_t a0 = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
            const uint32_t a1 = arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24);
            reinterpret_cast<u32x2*>(p.argmax)[idx] = u32x2{a0, a1};
        }
    }
    amax_emit_block(p.y_amax, vmax / *p.y_scale);
}

// finish a gradient element group: (+ old), fused ReLU / frozen-BN backward of the producer ...
__device__ __forceinline__ void prep_grad8(float (&v)[8], const PoolArgs& p, long o, long mo, int c0) {
    if (p.accumulate) {
        float old[8];
        load8(p.y_hi, p.y_lo, o, old);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += old[e];
    }
    if (p.mask_hi) {
        const u32x4 mk = reinterpret_cast<const u32x4*>(p.mask_hi)[mo];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sc = p.aff_scale[c0 + e];
            const float m = (e & 1) ? f16_pair_hi(mk[e >> 1]) : f16_pair_lo(mk[e >> 1]);
            v[e] = (sc != sc) ? v[e] : (m > 0.f ? v[e] * sc : 0.f);
        }
    }
}
// ... clamp, amax, split, store (planes), or the fp32 NCHW store (y_scale is 1): o = (n * y_img_groups + g) * HW + q, y_img_groups = G
__device__ __forceinline__ float store_grad8(float (&v)[8], const PoolArgs& p, long o, int c0) {
    float vmax = 0.f;
    if (p.y_f32) {
        const long hw = (long)p.H * p.W;
        uint32_t qq, ng, nn, gg;
        fd_divmod((uint32_t)o, p.dv_hw, ng, qq);      // (fp32 output: y_img_groups == G)
        fd_divmod(ng, p.dv_g, nn, gg);
        const long q = qq, n = nn;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            vmax = fmaxf(vmax, fabsf(v[e]));
            p.y_f32[n * p.y_f32_img_stride + (long)(c0 + e) * hw + q] = v[e];
        }
        return vmax;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        vmax = fmaxf(vmax, fabsf(v[e]));        // before the clamp (planes.h: the true magnitude is recorded)
        v[e] = pl_clamp(v[e]);
    }
    u32x4 hi, lo;
    pl_split8(v, hi, lo);
    reinterpret_cast<u32x4*>(p.y_hi)[o] = hi;
    reinterpret_cast<u32x4*>(p.y_lo)[o] = lo;
    return vmax;
}
__device__ __forceinline__ float finish_grad8(float (&v)[8], const PoolArgs& p, long o, long mo, int c0) {
    prep_grad8(v, p, o, mo, c0);
    return store_grad8(v, p, o, c0);
}

// max pool backward in gather form: input pixel (h, w) collects the output gradients of the windows whose argmax it is
// (x = output gradient [Ho x Wo], y = input gradient [H x W]); deterministic, no atomics
template <int KT, int ST>
__global__ __launch_bounds__(256) void pl_maxpool_bwd_kernel(PoolArgs p) {
    const uint32_t total = (uint32_t)p.N * (uint32_t)p.G * (uint32_t)p.H * (uint32_t)p.W;
    const float r = *p.y_scale / *p.x_scale;
    float vmax = 0.f;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const PoolIdx ix = pool_decode(idx, p.dv_w, p.dv_h, p.dv_g);
        const int w = (int)ix.w, h = (int)ix.h, g = (int)ix.g, n = (int)ix.n;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        // windows (ho, wo) with ho * s - pad <= h <= ho * s - pad + k - 1
        const int pk = KT ? KT : p.k, ps = ST ? ST : p.s;      // (compile-time in the instantiated cases: no runtime division)
        int ho_lo = (h + p.pad - pk + ps) / ps;   // ceil((h + pad - k + 1) / s) for non-negative numerators
        if (h + p.pad - pk + 1 <= 0) ho_lo = 0;
        int wo_lo = (w + p.pad - pk + ps) / ps;
        if (w + p.pad - pk + 1 <= 0) wo_lo = 0;
        int ho_hi = (h + p.pad) / ps, wo_hi = (w + p.pad) / ps;
        if (ho_hi > p.Ho - 1) ho_hi = p.Ho - 1;
        if (wo_hi > p.Wo - 1) wo_hi = p.Wo - 1;
        if constexpr (KT > 0) {
            // at most NW x NW windows contain a pixel: fetch all of them at once (windows that do not exist are masked out)
            constexpr int NWN = (KT + ST - 1) / ST;
            u32x2 am[NWN * NWN];
            u32x4 dhi[NWN * NWN], dlo[NWN * NWN];
            int local[NWN * NWN];
#pragma unroll
            for (int t = 0; t < NWN * NWN; ++t) {
                const int ho = ho_lo + t / NWN, wo = wo_lo + t % NWN;
                const bool ok = ho <= ho_hi && wo <= wo_hi;
                local[t] = ok ? (h - (ho * ST - p.pad)) * KT + (w - (wo * ST - p.pad)) : 255;
                const long oo = ok ? (long)ho * p.Wo + wo : 0;
                am[t] = reinterpret_cast<const u32x2*>(p.argmax)[((long)n * p.G + g) * p.Ho * p.Wo + oo];
                const long o = ((long)n * p.x_img_groups + g) * p.Ho * p.Wo + oo;
                dhi[t] = reinterpret_cast<const u32x4*>(p.x_hi)[o];
                dlo[t] = reinterpret_cast<const u32x4*>(p.x_lo)[o];
            }
#pragma unroll
            for (int t = 0; t < NWN * NWN; ++t) {
                float d[8];
                pl_join8(dhi[t], dlo[t], d);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int a = (int)((am[t][e >> 2] >> (8 * (e & 3))) & 0xFFu);
                    v[e] += (a == local[t]) ? d[e] : 0.f;
                }
            }
        } else {
            for (int ho = ho_lo; ho <= ho_hi; ++ho)
                for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                    const int local = (h - (ho * p.s - p.pad)) * p.k + (w - (wo * p.s - p.pad));
                    const long oo = (long)ho * p.Wo + wo;
                    const u32x2 am = reinterpret_cast<const u32x2*>(p.argmax)[((long)n * p.G + g) * p.Ho * p.Wo + oo];
                    float d[8];
                    load8(p.x_hi, p.x_lo, ((long)n * p.x_img_groups + g) * p.Ho * p.Wo + oo, d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int a = (int)((am[e >> 2] >> (8 * (e & 3))) & 0xFFu);
                        v[e] += (a == local) ? d[e] : 0.f;
                    }
                }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= r;
        const long o = ((long)n * p.y_img_groups + g) * p.H * p.W + (long)h * p.W + w;
        const long mo = ((long)n * p.mask_img_groups + g) * p.H * p.W + (long)h * p.W + w;
        vmax = fmaxf(vmax, finish_grad8(v, p, o, mo, 8 * g));
    }
    amax_emit_block(p.y_amax, vmax / *p.y_scale);
}

// 3x3 / stride-2 max pool backward with one thread per 2 x 2 block of INPUT pixels: the four pixels of a block can only belong to
// the same 2 x 2 windows, so those are fetched once (argmax + both planes) instead of once per pixel -- 2.25x fewer loads than the
// gather form above, which matters for the two stem pools (a third of the step's pooling traffic)
template <int PAD>
__global__ __launch_bounds__(256) void pl_maxpool_bwd_k3s2_kernel(PoolArgs p) {
    const int Hb = (p.H + 1) / 2, Wb = (p.W + 1) / 2;
    const uint32_t total = (uint32_t)p.N * (uint32_t)p.G * (uint32_t)Hb * (uint32_t)Wb;
    const float r = *p.y_scale / *p.x_scale;
    constexpr int base_off = (PAD + 1) / 2 - 1;      // first candidate window of block i is i + base_off (pad 0: i - 1, pad 1: i)
    float vmax = 0.f;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const PoolIdx ix = pool_decode(idx, p.dv_wb, p.dv_hb, p.dv_g);
        const int j = (int)ix.w, i = (int)ix.h, g = (int)ix.g, n = (int)ix.n;
        u32x2 am[4];
        float d[4][8];
        bool wok[4];
        // p.relu ("pooled mask"): mask_hi is the hi plane of the pool's OUTPUT -- the ReLU decision of the element a window routes its
        // gradient to is the sign of that window's pooled value (it IS that element), so the backward reads the pooled tensor (1/4 of
        // the pixels) instead of the full-resolution activation
        const bool pooled_mask = p.mask_hi && p.relu;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ho = i + base_off + (t >> 1), wo = j + base_off + (t & 1);
            wok[t] = (unsigned)ho < (unsigned)p.Ho && (unsigned)wo < (unsigned)p.Wo;
            const long oo = wok[t] ? (long)ho * p.Wo + wo : 0;
            am[t] = reinterpret_cast<const u32x2*>(p.argmax)[((long)n * p.G + g) * p.Ho * p.Wo + oo];
            load8(p.x_hi, p.x_lo, ((long)n * p.x_img_groups + g) * p.Ho * p.Wo + oo, d[t]);
            if (!wok[t]) am[t] = u32x2{0xFFFFFFFFu, 0xFFFFFFFFu};          // (a window that does not exist matches no local index)
            if (pooled_mask) {
                const u32x4 pm = reinterpret_cast<const u32x4*>(p.mask_hi)[((long)n * p.mask_img_groups + g) * p.Ho * p.Wo + oo];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sc = p.aff_scale[8 * g + e];
                    const float m = (e & 1) ? f16_pair_hi(pm[e >> 1]) : f16_pair_lo(pm[e >> 1]);
                    d[t][e] = (sc != sc) ? d[t][e] : (m > 0.f ? d[t][e] * sc : 0.f);
                }
            }
        }
        float v[4][8];
        bool live[4];
        long oq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int h = 2 * i + (q >> 1), w = 2 * j + (q & 1);
            live[q] = h < p.H && w < p.W;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[q][e] = 0.f;
            if (!live[q]) continue;
            // window t = (a, b) of the block's 2 x 2 candidates covers pixel q = (qr, qc) at the window-local position
            // rr = qr + 2 (1 - a) - (1 - PAD) ... a COMPILE-TIME table: of the 16 (pixel, window) pairs only 9 can match
            // (an odd row / column lies in one window only), the others are never compared
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int rr = (q >> 1) - 2 * (base_off + (t >> 1)) + PAD, ss = (q & 1) - 2 * (base_off + (t & 1)) + PAD;   // constants after unrolling
                if (rr < 0 || rr > 2 || ss < 0 || ss > 2) continue;
                const unsigned local = (unsigned)(rr * 3 + ss);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned a = (am[t][e >> 2] >> (8 * (e & 3))) & 0xFFu;
                    v[q][e] += (a == local) ? d[t][e] : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[q][e] *= r;
            oq[q] = ((long)n * p.y_img_groups + g) * p.H * p.W + (long)h * p.W + w;
            const long mo = ((long)n * p.mask_img_groups + g) * p.H * p.W + (long)h * p.W + w;
            if (!pooled_mask) prep_grad8(v[q], p, oq[q], mo, 8 * g);
        }
        if (p.y_f32 && (p.W & 1) == 0) {
            // fp32 NCHW output (the stem's pool: 925 MB per step): the two pixels of a block row are neighbours in every channel
            // plane -- one 8-byte store per channel and row, contiguous across the lanes of a wave
            const long hw = (long)p.H * p.W;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                if (!live[2 * rr]) continue;          // (W even: the pair is live or dead together)
                float* dst = p.y_f32 + (long)n * p.y_f32_img_stride + (long)(8 * g) * hw + (long)(2 * i + rr) * p.W + 2 * j;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    vmax = fmaxf(vmax, fmaxf(fabsf(v[2 * rr][e]), fabsf(v[2 * rr + 1][e])));
                    *reinterpret_cast<float2*>(dst + (long)e * hw) = float2{v[2 * rr][e], v[2 * rr + 1][e]};
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (live[q]) vmax = fmaxf(vmax, store_grad8(v[q], p, oq[q], 8 * g));
        }
    }
    amax_emit_block(p.y_amax, vmax / *p.y_scale);
}

// The same backward for the cases the training step runs (planes output, no accumulation; MASK: the pooled mask), written for
// instruction count: the general kernel above spends ~1500 vector + ~850 scalar instructions per 2 x 2 block -- divergent branches
// around every masked select, per-element scale loads, 64-bit address arithmetic per access -- and runs at 3.1 TB/s where a plain
// stream of its bytes reaches 6.2 (profiles/r5_pool_lab.txt: 0.11 of the stem pool's 0.41 ms is that arithmetic).  Here: buffer loads
// with 32-bit offsets (a window that does not exist reads zeros: no predicate reaches the arithmetic), the eight channel factors
// (scale x r, or r for a pass-through channel) loaded once per thread as two 16-byte loads, selects instead of branches.
// Same values as the general kernel (r is a power of two: folding it into the channel factor is exact).
template <int PAD, bool MASK>
__global__ __launch_bounds__(256) void pl_maxpool_bwd_k3s2_fast_kernel(PoolArgs p) {
    const int Hb = (p.H + 1) / 2, Wb = (p.W + 1) / 2;
    const uint32_t total = (uint32_t)p.N * (uint32_t)p.G * (uint32_t)Hb * (uint32_t)Wb;
    const float r = *p.y_scale / *p.x_scale;
    constexpr int base_off = (PAD + 1) / 2 - 1;
    const uint32_t HoWo = (uint32_t)p.Ho * (uint32_t)p.Wo;
    // descriptors end with the slice in the last image (see ssn_conv_wgrad_pl: nothing is read past a slice at the end of its tensor)
    const uint32_t x_bytes = (uint32_t)(((long)(p.N - 1) * p.x_img_groups + p.G) * HoWo * 16);
    const uint32_t m_bytes = MASK ? (uint32_t)(((long)(p.N - 1) * p.mask_img_groups + p.G) * HoWo * 16) : 0u;
    const __amdgpu_buffer_rsrc_t r_hi = pl_rsrc(p.x_hi, x_bytes), r_lo = pl_rsrc(p.x_lo, x_bytes);
    const __amdgpu_buffer_rsrc_t r_am = pl_rsrc(p.argmax, (uint32_t)p.N * (uint32_t)p.G * HoWo * 8u);
    const __amdgpu_buffer_rsrc_t r_mk = pl_rsrc(MASK ? p.mask_hi : p.x_hi, MASK ? m_bytes : 0u);
    float vmax = 0.f;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const PoolIdx ix = pool_decode(idx, p.dv_wb, p.dv_hb, p.dv_g);
        const int j = (int)ix.w, i = (int)ix.h;
        const uint32_t g = ix.g, n = ix.n;
        float kk[8];          // per channel: what a routed value is multiplied by when its ReLU passed
        bool pass[8];         // NaN scale: not a ReLU / frozen-BN output -- the gradient passes whatever the sign
        if (MASK) {
            const f32x4 s0 = reinterpret_cast<const f32x4*>(p.aff_scale + 8 * g)[0], s1 = reinterpret_cast<const f32x4*>(p.aff_scale + 8 * g)[1];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sc = e < 4 ? s0[e] : s1[e - 4];
                pass[e] = sc != sc;
                kk[e] = pass[e] ? r : sc * r;
            }
        }
        const uint32_t xb = (n * (uint32_t)p.x_img_groups + g) * HoWo, ab = (n * (uint32_t)p.G + g) * HoWo;
        const uint32_t mb = MASK ? (n * (uint32_t)p.mask_img_groups + g) * HoWo : 0u;
        u32x4 hi[4], lo[4], pm[4];
        u32x2 am[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ho = i + base_off + (t >> 1), wo = j + base_off + (t & 1);
            const bool ok = (unsigned)ho < (unsigned)p.Ho && (unsigned)wo < (unsigned)p.Wo;
            const uint32_t q = (uint32_t)(ho * p.Wo + wo);
            hi[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r_hi, ok ? (xb + q) * 16u : PL_OOB, 0, 0));
            lo[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r_lo, ok ? (xb + q) * 16u : PL_OOB, 0, 0));
            am[t] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r_am, ok ? (ab + q) * 8u : PL_OOB, 0, 0));
            if (MASK) pm[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r_mk, ok ? (mb + q) * 16u : PL_OOB, 0, 0));
        }
        float d[4][8];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = (e & 1) ? f16_pair_hi(hi[t][e >> 1]) + f16_pair_hi(lo[t][e >> 1])
                                        : f16_pair_lo(hi[t][e >> 1]) + f16_pair_lo(lo[t][e >> 1]);
                if (MASK) {
                    const float m = (e & 1) ? f16_pair_hi(pm[t][e >> 1]) : f16_pair_lo(pm[t][e >> 1]);
                    d[t][e] = v * ((m > 0.f || pass[e]) ? kk[e] : 0.f);
                } else {
                    d[t][e] = v * r;
                }
            }
        const long obase = (((long)n * p.y_img_groups + g) * p.H + 2 * i) * p.W + 2 * j;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int h = 2 * i + (q >> 1), w = 2 * j + (q & 1);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int rr = (q >> 1) - 2 * (base_off + (t >> 1)) + PAD, ss = (q & 1) - 2 * (base_off + (t & 1)) + PAD;   // constants
                if (rr < 0 || rr > 2 || ss < 0 || ss > 2) continue;
                const unsigned local = (unsigned)(rr * 3 + ss);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned a = (am[t][e >> 2] >> (8 * (e & 3))) & 0xFFu;
                    v[e] += (a == local) ? d[t][e] : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                vmax = fmaxf(vmax, fabsf(v[e]));        // before the clamp: the true magnitude is recorded (a pixel outside the image holds 0)
                v[e] = pl_clamp(v[e]);
            }
            u32x4 ohi, olo;
            pl_split8(v, ohi, olo);
            if (h < p.H && w < p.W) {
                const long o = obase + (long)(q >> 1) * p.W + (q & 1);
                reinterpret_cast<u32x4*>(p.y_hi)[o] = ohi;
                reinterpret_cast<u32x4*>(p.y_lo)[o] = olo;
            }
        }
    }
    amax_emit_block(p.y_amax, vmax / *p.y_scale);
}

// y = relu?(scale[c] * avgpool_kxk(x) + shift[c]), stride 1, zero padding counted (count_include_pad): the pool BEHIND its 1x1
// projection (a 1x1 convolution commutes with the zero-padded average)
__global__ __launch_bounds__(256) void pl_avgpool_affine_kernel(PoolArgs p) {
    const uint32_t total = (uint32_t)p.N * (uint32_t)p.G * (uint32_t)p.H * (uint32_t)p.W;
    const float inv = 1.f / ((float)(p.k * p.k) * *p.x_scale), so = *p.y_scale;
    float vmax = 0.f;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const PoolIdx ix = pool_decode(idx, p.dv_w, p.dv_h, p.dv_g);
        const int w = (int)ix.w, h = (int)ix.h, g = (int)ix.g, n = (int)ix.n;
        const long ibase = ((long)n * p.x_img_groups + g) * p.H * p.W;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int dr = 0; dr < p.k; ++dr) {
            const int hh = h - p.pad + dr;
            if ((unsigned)hh >= (unsigned)p.H) continue;
            for (int ds = 0; ds < p.k; ++ds) {
                const int ww = w - p.pad + ds;
                if ((unsigned)ww >= (unsigned)p.W) continue;
                float v[8];
                load8(p.x_hi, p.x_lo, ibase + (long)hh * p.W + ww, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e];
            }
        }
        float out[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = acc[e] * inv;
            if (p.aff_scale) v = v * p.aff_scale[8 * g + e] + p.aff_shift[8 * g + e];
            if (p.relu) v = fmaxf(v, 0.f);
            vmax = fmaxf(vmax, fabsf(v * so));
            out[e] = pl_clamp(v * so);
        }
        u32x4 hi, lo;
        pl_split8(out, hi, lo);
        const long o = ((long)n * p.y_img_groups + g) * p.H * p.W + (long)h * p.W + w;
        reinterpret_cast<u32x4*>(p.y_hi)[o] = hi;
        reinterpret_cast<u32x4*>(p.y_lo)[o] = lo;
    }
    amax_emit_block(p.y_amax, vmax / so);
}

// The 3x3 case (every average pool of the two backbones) with all nine taps in flight at once: buffer loads with 32-bit offsets, a tap
// outside the image reads zeros (zero padding is counted: the divisor stays 9).  The general kernel above walks a runtime k x k window
// tap by tap -- nine dependent round trips per thread: 70 % of its wave cycles wait (PMC), 2.3 TB/s.
__global__ __launch_bounds__(256) void pl_avgpool3_fast_kernel(PoolArgs p) {
    const uint32_t total = (uint32_t)p.N * (uint32_t)p.G * (uint32_t)p.H * (uint32_t)p.W;
    const float inv = 1.f / (9.f * *p.x_scale), so = *p.y_scale;
    const uint32_t HW = (uint32_t)p.H * (uint32_t)p.W;
    const uint32_t x_bytes = (uint32_t)(((long)(p.N - 1) * p.x_img_groups + p.G) * HW * 16);
    const __amdgpu_buffer_rsrc_t r_hi = pl_rsrc(p.x_hi, x_bytes), r_lo = pl_rsrc(p.x_lo, x_bytes);
    float vmax = 0.f;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const PoolIdx ix = pool_decode(idx, p.dv_w, p.dv_h, p.dv_g);
        const int w = (int)ix.w, h = (int)ix.h;
        const uint32_t xb = (ix.n * (uint32_t)p.x_img_groups + ix.g) * HW;
        u32x4 thi[9], tlo[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hh = h - 1 + t / 3, ww = w - 1 + t % 3;
            const bool ok = ((unsigned)hh < (unsigned)p.H) && ((unsigned)ww < (unsigned)p.W);
            const uint32_t off = ok ? (xb + (uint32_t)(hh * p.W + ww)) * 16u : PL_OOB;
            thi[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r_hi, off, 0, 0));
            tlo[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r_lo, off, 0, 0));
        }
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {      // (row-major tap order: the summation order of the general kernel)
            float v[8];
            pl_join8(thi[t], tlo[t], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
        float out[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = acc[e] * inv;
            if (p.aff_scale) v = v * p.aff_scale[8 * ix.g + e] + p.aff_shift[8 * ix.g + e];
            if (p.relu) v = fmaxf(v, 0.f);
            vmax = fmaxf(vmax, fabsf(v * so));
            out[e] = pl_clamp(v * so);
        }
        u32x4 hi, lo;
        pl_split8(out, hi, lo);
        const long o = ((long)ix.n * p.y_img_groups + ix.g) * HW + (long)h * p.W + w;
        reinterpret_cast<u32x4*>(p.y_hi)[o] = hi;
        reinterpret_cast<u32x4*>(p.y_lo)[o] = lo;
    }
    amax_emit_block(p.y_amax, vmax / so);
}

// in place: g <- g * (y > 0) * scale[c]  (NaN scale: channel passes through) -- the ReLU / frozen-BN backward of a slice whose
// last writer could not fuse it
__global__ __launch_bounds__(256) void pl_relu_bn_bwd_kernel(PoolArgs p) {
    const uint32_t total = (uint32_t)p.N * (uint32_t)p.G * (uint32_t)p.H * (uint32_t)p.W;
    float vmax = 0.f;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        uint32_t qq, ng, gg, nn;
        fd_divmod(idx, p.dv_hw, ng, qq);
        fd_divmod(ng, p.dv_g, nn, gg);
        const long q = qq;
        const int g = (int)gg, n = (int)nn;
        const long o = ((long)n * p.y_img_groups + g) * p.H * p.W + q;
        const long mo = ((long)n * p.mask_img_groups + g) * p.H * p.W + q;
        float v[8];
        load8(p.y_hi, p.y_lo, o, v);
        vmax = fmaxf(vmax, finish_grad8(v, p, o, mo, 8 * g));
    }
    amax_emit_block(p.y_amax, vmax / *p.y_scale);
}

// global average pool: planes [N][G][HW][8] -> fp32 [N][C]
__global__ __launch_bounds__(256) void pl_gap_fwd_kernel(const void* hi, const void* lo, long img_groups, float* out, int N, int G,
                                                        int HW, const float* scale) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * G) return;
    const int g = idx % G, n = idx / G;
    const long base = ((long)n * img_groups + g) * HW;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    // eight pixels' loads in flight, summed in pixel order (one memory round trip per pixel made this 50 us for 58 MB)
    constexpr int B = 8;
    for (int q0 = 0; q0 < HW; q0 += B) {
        u32x4 th[B], tl[B];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const int q = q0 + j < HW ? q0 + j : HW - 1;
            th[j] = reinterpret_cast<const u32x4*>(hi)[base + q];
            tl[j] = reinterpret_cast<const u32x4*>(lo)[base + q];
        }
#pragma unroll
        for (int j = 0; j < B; ++j) {
            float v[8];
            pl_join8(th[j], tl[j], v);
            if (q0 + j < HW) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e];
            }
        }
    }
    const float inv = 1.f / ((float)HW * *scale);
#pragma unroll
    for (int e = 0; e < 8; ++e) out[(long)n * G * 8 + 8 * g + e] = acc[e] * inv;
}

// ... the same sums (same order: bit-identical) with coalesced loads: a workgroup takes GAP_NG consecutive channel groups of one image --
// one contiguous range of each plane -- through the LDS, then one thread per channel adds its HW values in pixel order.  (The kernel
// above reads 16 bytes per lane from addresses HW * 16 bytes apart: 64 cache lines per load instruction, 40 - 50 us for 58 MB.)
constexpr int GAP_NG = 16, GAP_MAX = 1024;      // channel groups per workgroup; 16-byte entries per plane in the LDS
__global__ __launch_bounds__(256) void pl_gap_fwd_lds_kernel(const void* hi, const void* lo, long img_groups, float* out, int N, int G,
                                                            int HW, const float* scale, int gblocks) {
    __shared__ __attribute__((aligned(16))) u32x4 t_hi[GAP_MAX];
    __shared__ __attribute__((aligned(16))) u32x4 t_lo[GAP_MAX];
    const int n = blockIdx.x / gblocks, g0 = (blockIdx.x - n * gblocks) * GAP_NG;
    const int ng = G - g0 < GAP_NG ? G - g0 : GAP_NG;
    const long base = ((long)n * img_groups + g0) * HW;
    const int cnt = ng * HW;
#pragma unroll 2
    for (int i = threadIdx.x; i < cnt; i += 256) {
        t_hi[i] = reinterpret_cast<const u32x4*>(hi)[base + i];
        t_lo[i] = reinterpret_cast<const u32x4*>(lo)[base + i];
    }
    __syncthreads();
    const int j = threadIdx.x >> 3, e = threadIdx.x & 7;
    if (j >= ng) return;
    const uint32_t* ph = reinterpret_cast<const uint32_t*>(t_hi + j * HW) + (e >> 1);
    const uint32_t* pw = reinterpret_cast<const uint32_t*>(t_lo + j * HW) + (e >> 1);
    float acc = 0.f;
    for (int q = 0; q < HW; ++q) {
        const uint32_t h = ph[q * 4], l = pw[q * 4];
        acc += (e & 1) ? f16_pair_hi(h) + f16_pair_hi(l) : f16_pair_lo(h) + f16_pair_lo(l);
    }
    out[(long)n * G * 8 + 8 * (g0 + j) + e] = acc * (1.f / ((float)HW * *scale));
}

// global average pool backward: dx[n][c][q] = dy[n][c] / HW, with the fused ReLU / frozen-BN backward of the pooled tensor
// (PoolArgs: aff_shift = dy fp32 [N][C], y = dx planes, mask as usual)
__global__ __launch_bounds__(256) void pl_gap_bwd_kernel(PoolArgs p) {
    const int HW = p.H * p.W;
    const uint32_t total = (uint32_t)p.N * (uint32_t)p.G * (uint32_t)HW;
    const float k = *p.y_scale / (float)HW;
    float vmax = 0.f;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        uint32_t qq, ng, gg, nn;
        fd_divmod(idx, p.dv_hw, ng, qq);
        fd_divmod(ng, p.dv_g, nn, gg);
        const long q = qq;
        const int g = (int)gg, n = (int)nn;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p.aff_shift[(long)n * p.G * 8 + 8 * g + e] * k;
        const long o = ((long)n * p.y_img_groups + g) * HW + q;
        const long mo = ((long)n * p.mask_img_groups + g) * HW + q;
        vmax = fmaxf(vmax, finish_grad8(v, p, o, mo, 8 * g));
    }
    amax_emit_block(p.y_amax, vmax / *p.y_scale);
}

// per-channel sums of a planes slice (bias gradient of a projection in front of its pool): two passes, fixed order
constexpr int CS_SHARES = 32;
__global__ __launch_bounds__(256) void pl_channel_sum_kernel(const void* hi, const void* lo, long img_groups, int N, int G, int HW,
                                                            float* part) {
    __shared__ float red[256][9];
    const int g = blockIdx.x, share = blockIdx.y;
    const long total = (long)N * HW;
    const long per = (total + CS_SHARES - 1) / CS_SHARES;
    const long begin = (long)share * per;
    long end = begin + per;
    if (end > total) end = total;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    // (image, pixel) of this thread's first element by ONE division, then carried along: no division in the loop
    long n = (begin + threadIdx.x) / HW, q = (begin + threadIdx.x) - n * HW;
    for (long i = begin + threadIdx.x; i < end; i += 256) {
        float v[8];
        load8(hi, lo, (n * img_groups + g) * HW + q, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
        q += 256;
        while (q >= HW) {
            q -= HW;
            ++n;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[e];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[threadIdx.x][e] += red[threadIdx.x + st][e];
        __syncthreads();
    }
    if (threadIdx.x < 8) part[((long)share * G + g) * 8 + threadIdx.x] = red[0][threadIdx.x];
}
__global__ __launch_bounds__(256) void pl_channel_sum_final_kernel(const float* part, float* out, int C, int G, const float* scale) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int sh = 0; sh < CS_SHARES; ++sh) s += part[(long)sh * G * 8 + c];
    out[c] = s / *scale;
}

// several slices in one pair of launches (the bias gradients of all pool projections of a backward pass)
constexpr int CSM_MAX = 24;
struct ChannelSumTable {
    const void* hi[CSM_MAX];
    const void* lo[CSM_MAX];
    const float* scale[CSM_MAX];
    float* out[CSM_MAX];
    long img_groups[CSM_MAX];
    int HW[CSM_MAX];
    int g0[CSM_MAX + 1];       // first channel group of the entry in the launch (prefix sums of G)
    int N, count;
};
__global__ __launch_bounds__(256) void pl_channel_sum_multi_kernel(ChannelSumTable t, float* part) {
    __shared__ float red[256][9];
    int ti = 0;
    while (ti + 1 < t.count && (int)blockIdx.x >= t.g0[ti + 1]) ++ti;
    const int g = (int)blockIdx.x - t.g0[ti], share = blockIdx.y;
    const int HW = t.HW[ti];
    const long total = (long)t.N * HW;
    const long per = (total + CS_SHARES - 1) / CS_SHARES;
    const long begin = (long)share * per;
    long end = begin + per;
    if (end > total) end = total;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    long n = (begin + threadIdx.x) / HW, q = (begin + threadIdx.x) - n * HW;      // (see pl_channel_sum_kernel)
    for (long i = begin + threadIdx.x; i < end; i += 256) {
        float v[8];
        load8(t.hi[ti], t.lo[ti], (n * t.img_groups[ti] + g) * HW + q, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
        q += 256;
        while (q >= HW) {
            q -= HW;
            ++n;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[e];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[threadIdx.x][e] += red[threadIdx.x + st][e];
        __syncthreads();
    }
    const int GT = t.g0[t.count];
    if (threadIdx.x < 8) part[((long)share * GT + blockIdx.x) * 8 + threadIdx.x] = red[0][threadIdx.x];
}
__global__ __launch_bounds__(256) void pl_channel_sum_multi_final_kernel(ChannelSumTable t, const float* part) {
    const int GT = t.g0[t.count];
    const int c = blockIdx.x * 256 + threadIdx.x;       // channel of the launch (8 per group)
    if (c >= GT * 8) return;
    int ti = 0;
    while (ti + 1 < t.count && (c >> 3) >= t.g0[ti + 1]) ++ti;
    float s = 0.f;
    for (int sh = 0; sh < CS_SHARES; ++sh) s += part[(long)sh * GT * 8 + c];
    t.out[ti][c - 8 * t.g0[ti]] = s / *t.scale[ti];
}

// Grid of the elementwise kernels of this file (all of them grid-stride loops): at most 2048 workgroups = eight per CU, the rest
// of the elements by the loop.  [r6] The cap was 65536: one wave's worth of work per wave, and every wave ends in amax_emit's
// read-compare-atomicMax on ONE address -- tens of thousands of short-lived waves per launch.  Measured on the bench step
// (profiles/r6_grid_cap.txt): frames -> planes 142 -> 72 us, gap backward 50 - 74 -> 29, the stride-2 pools -7 %; step -0.22 ms;
// flat between 512 and 3072.  Results do not depend on the grid (element-wise work + a maximum).  SSN_PL_GRID_CAP: tooling.
int grid_for(long total) {
    static const long cap = [] {
        const char* e = std::getenv("SSN_PL_GRID_CAP");
        return e ? std::atol(e) : 2048l;
    }();
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
// amax / scale: `n` consecutive slots.  exact != 0: scales without head-room (tensors measured before they are stored: the
// caller's frames).  flag: int[2] = {overflow (sticky), slots whose scale changed (cumulative)}.
// slot0: global index of the first slot (amax / scale point at it); slots with an odd global index take odd_extra_bits more bits
// of head-room (the executor keeps its gradient tensors there).
extern "C" int ssn_pl_scales_update(float* amax, float* scale, int* flag, int n, int exact, int slot0, int odd_extra_bits,
                                    hipStream_t stream) {
    SSN_CHECK_ARG(amax && scale && flag && n >= 0 && slot0 >= 0 && odd_extra_bits >= 0 && odd_extra_bits <= 8,
                  "pl scales update: bad arguments");
    if (n == 0) return SSN_OK;
    hipLaunchKernelGGL(scales_update_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, amax, scale, flag, n, exact, slot0,
                       odd_extra_bits);
    SSN_CHECK_LAUNCH("pl_scales_update");
    return SSN_OK;
}

// flag[0] |= 1 (a tensor overflowed the scale it was stored with) / 2 (severe underflow) over `n` consecutive slots; no slot is
// modified.  One launch at the end of a forward / backward pass: what makes the delayed scales safe on changing data.
extern "C" int ssn_pl_range_check(const float* amax, const float* scale, int* flag, int n, int odd_extra_bits, hipStream_t stream) {
    SSN_CHECK_ARG(amax && scale && flag && n >= 0 && odd_extra_bits >= 0 && odd_extra_bits <= 8, "pl range check: bad arguments");
    if (n == 0) return SSN_OK;
    hipLaunchKernelGGL(range_check_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, amax, scale, flag, n, odd_extra_bits);
    SSN_CHECK_LAUNCH("pl_range_check");
    return SSN_OK;
}

// fp32 NCHW (channel slice: x = first channel, x_img_stride floats between images) -> planes at (hi, lo) = the first
// group of the destination slice.  s2d: the space-to-depth view of the stride-2 stem (4C channels at H/2 x W/2).
// Channels past C (resp. 4C) up to the next multiple of 8 are written as zeros.  amax may be null.
extern "C" int ssn_pl_from_f32(const float* x, long x_img_stride, void* hi, void* lo, int N, int C, int H, int W,
                               long img_groups, int s2d, const float* scale, float* amax, hipStream_t stream) {
    SSN_CHECK_ARG(x && hi && lo && scale && N > 0 && C > 0 && H > 0 && W > 0, "pl from f32: bad arguments");
    SSN_CHECK_ARG(!s2d || (H % 2 == 0 && W % 2 == 0), "pl from f32: space-to-depth needs even sizes");
    CvtArgs a;
    a.x = x;
    a.hi = hi;
    a.lo = lo;
    a.scale = scale;
    a.amax = amax;
    a.x_img_stride = x_img_stride;
    a.N = N;
    a.C = C;
    a.H = H;
    a.W = W;
    a.G = ((s2d ? 4 * C : C) + 7) / 8;
    a.img_groups = img_groups;
    a.s2d = s2d;
    SSN_CHECK_ARG(img_groups >= a.G, "pl from f32: slice wider than its tensor");
    const long total = (long)N * a.G * (s2d ? (H / 2) * (W / 2) : H * W);
    SSN_CHECK_ARG(total < (1l << 27), "pl from f32: more than 2^27 16-byte groups per plane");
    a.dv_hw = make_fastdiv((uint32_t)(s2d ? (H / 2) * (W / 2) : H * W));
    a.dv_g = make_fastdiv((uint32_t)a.G);
    a.dv_wo = make_fastdiv((uint32_t)(s2d ? W / 2 : W));
    hipLaunchKernelGGL(pl_from_f32_kernel, dim3(grid_for(total)), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("pl_from_f32");
    return SSN_OK;
}

extern "C" int ssn_pl_to_f32(const void* hi, const void* lo, long img_groups, float* y, long y_img_stride, int N, int C,
                             int HW, const float* scale, hipStream_t stream) {
    SSN_CHECK_ARG(hi && lo && y && scale && N > 0 && C > 0 && HW > 0, "pl to f32: bad arguments");
    const long total = (long)N * ((C + 7) / 8) * HW;
    SSN_CHECK_ARG(total < (1l << 27), "pl to f32: more than 2^27 16-byte groups per plane");
    hipLaunchKernelGGL(pl_to_f32_kernel, dim3(grid_for(total)), dim3(256), 0, stream, hi, lo, img_groups, y, y_img_stride, N, C,
                       HW, scale, make_fastdiv((uint32_t)HW), make_fastdiv((uint32_t)((C + 7) / 8)));
    SSN_CHECK_LAUNCH("pl_to_f32");
    return SSN_OK;
}

// im2col of a planes slice with few channels (see pl_im2col_kernel): x [N, C, H, W] -> y [N, C kh kw (padded to 8), Ho, Wo], both
// planes copied; y carries x's scale.  For the weight gradient of a first convolution as a 1x1 problem.
extern "C" int ssn_pl_im2col(const void* x_hi, const void* x_lo, long x_img_groups, void* y_hi, void* y_lo, long y_img_groups, int N,
                             int C, int H, int W, int Ho, int Wo, int kh, int kw, int stride, int pad_h, int pad_w, hipStream_t stream) {
    SSN_CHECK_ARG(x_hi && x_lo && y_hi && y_lo && N > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && kh > 0 && kw > 0 && stride > 0,
                  "pl im2col: bad arguments");
    Im2colArgs a;
    a.x_hi = x_hi;
    a.x_lo = x_lo;
    a.y_hi = y_hi;
    a.y_lo = y_lo;
    a.x_img_groups = x_img_groups;
    a.y_img_groups = y_img_groups;
    a.N = N; a.C = C; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.kh = kh; a.kw = kw; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w;
    a.K = C * kh * kw;
    a.KG = (a.K + 7) / 8;
    SSN_CHECK_ARG(y_img_groups >= a.KG && x_img_groups >= (C + 7) / 8, "pl im2col: slice wider than its tensor");
    const long total = (long)N * a.KG * Ho * Wo;
    SSN_CHECK_ARG(total < (1l << 27) && (long)N * x_img_groups * H * W < (1l << 27), "pl im2col: more than 2^27 16-byte groups per plane");
    a.dv_hw = make_fastdiv((uint32_t)(Ho * Wo));
    a.dv_kg = make_fastdiv((uint32_t)a.KG);
    a.dv_wo = make_fastdiv((uint32_t)Wo);
    a.dv_kk = make_fastdiv((uint32_t)(kh * kw));
    a.dv_kw = make_fastdiv((uint32_t)kw);
    hipLaunchKernelGGL(pl_im2col_kernel, dim3(grid_for(total)), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("pl_im2col");
    return SSN_OK;
}

static // the fast pool kernels address with 32-bit byte offsets and descriptor sizes computed from the WHOLE tensor's groups per image: a
// narrow slice of a wide, large tensor must not wrap -- such calls take the general (64-bit indexing) kernels
// (2 GiB, not 4: a lane outside the image is sent to offset 0x80000000, which must lie BEHIND the descriptor's end)
inline bool pool_fits32(int N, long img_groups, int G, long hw) {
    return ((long)(N - 1) * img_groups + G) * hw * 16 <= (1l << 31);
}
int fill_pool(PoolArgs& a, const void* x_hi, const void* x_lo, long x_img_groups, void* y_hi, void* y_lo, long y_img_groups,
                     int N, int C, int H, int W, int Ho, int Wo, int k, int s, int pad, const float* x_scale, const float* y_scale,
                     float* y_amax, const char* what) {
    SSN_CHECK_ARG(y_scale && N > 0 && C > 0 && C % 8 == 0 && H > 0 && W > 0, "%s: bad arguments", what);
    a.x_hi = x_hi;
    a.x_lo = x_lo;
    a.y_hi = y_hi;
    a.y_lo = y_lo;
    a.argmax = nullptr;
    a.x_scale = x_scale;
    a.y_scale = y_scale;
    a.y_amax = y_amax;
    a.aff_scale = a.aff_shift = nullptr;
    a.mask_hi = nullptr;
    a.x_img_groups = x_img_groups;
    a.y_img_groups = y_img_groups;
    a.mask_img_groups = 0;
    a.N = N;
    a.G = C / 8;
    a.H = H;
    a.W = W;
    a.Ho = Ho;
    a.Wo = Wo;
    a.k = k;
    a.s = s;
    a.pad = pad;
    a.relu = 0;
    a.accumulate = 0;
    a.y_f32 = nullptr;
    a.y_f32_img_stride = 0;
    SSN_CHECK_ARG((long)N * (C / 8) * H * W < (1l << 27) && (long)N * (C / 8) * Ho * Wo < (1l << 27),
                  "%s: more than 2^27 16-byte groups per plane (32-bit indexing, 2 GiB buffer addressing)", what);
    a.dv_w = make_fastdiv((uint32_t)W);
    a.dv_h = make_fastdiv((uint32_t)H);
    a.dv_wo = make_fastdiv((uint32_t)(Wo > 0 ? Wo : 1));
    a.dv_ho = make_fastdiv((uint32_t)(Ho > 0 ? Ho : 1));
    a.dv_g = make_fastdiv((uint32_t)(C / 8));
    a.dv_hw = make_fastdiv((uint32_t)(H * W));
    a.dv_wb = make_fastdiv((uint32_t)((W + 1) / 2));
    a.dv_hb = make_fastdiv((uint32_t)((H + 1) / 2));
    return SSN_OK;
}

// Max pool forward on planes slices (C channels, a multiple of 8): x [N, C, H, W] -> y [N, C, Ho, Wo] (ceil-mode sizes from the
// caller), argmax: uint8 [N][C/8][Ho*Wo][8] or null (inference).  Replaces nn.MaxPool2d of the backbone manifest.
extern "C" int ssn_pl_maxpool_fwd(const void* x_hi, const void* x_lo, long x_img_groups, void* y_hi, void* y_lo,
                                  long y_img_groups, unsigned char* argmax, int N, int C, int H, int W, int Ho, int Wo, int k,
                                  int s, int pad, const float* x_scale, const float* y_scale, float* y_amax, hipStream_t stream) {
    PoolArgs a;
    int rc = fill_pool(a, x_hi, x_lo, x_img_groups, y_hi, y_lo, y_img_groups, N, C, H, W, Ho, Wo, k, s, pad, x_scale, y_scale, y_amax,
                       "pl maxpool fwd");
    if (rc != SSN_OK) return rc;
    SSN_CHECK_ARG(x_hi && x_lo && x_scale && k >= 1 && k <= 15 && s >= 1, "pl maxpool fwd: bad arguments");
    a.argmax = argmax;
    const dim3 grid(grid_for((long)N * a.G * Ho * Wo));
    const bool fits = pool_fits32(N, x_img_groups, a.G, (long)H * W);
    if (k == 3 && s == 2 && fits)
        hipLaunchKernelGGL((pl_maxpool_fwd_k3_fast_kernel<2>), grid, dim3(256), 0, stream, a);
    else if (k == 3 && s == 1 && fits)
        hipLaunchKernelGGL((pl_maxpool_fwd_k3_fast_kernel<1>), grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((pl_maxpool_fwd_kernel<0, 0>), grid, dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("pl_maxpool_fwd");
    return SSN_OK;
}

// Max pool backward: dy [N, C, Ho, Wo] + argmax -> dx [N, C, H, W] (+= with accumulate); mask_hi / mask_scale [C]: the fused
// ReLU / frozen-BN backward of the layer that produced the pool's input (mask_hi = hi plane of that activation at dx's slice).
// dx_f32 != null: dx is written as fp32 NCHW there instead (dx_scale -> 1.0): the gradient of the stem's output, which only the
// fp32-layout weight-gradient kernel of the 3-channel first layer reads.
// mask_pooled != 0 (3x3 / stride 2, no accumulation): mask_hi is the hi plane of the pool's OUTPUT activation (Ho x Wo) instead of
// its input's -- the element a window's gradient goes to is that window's maximum, so its ReLU decision is the sign of the pooled
// value: the backward of the two stem pools then reads a quarter of the mask bytes.
extern "C" int ssn_pl_maxpool_bwd(const void* dy_hi, const void* dy_lo, long dy_img_groups, const unsigned char* argmax,
                                  void* dx_hi, void* dx_lo, long dx_img_groups, int N, int C, int H, int W, int Ho, int Wo, int k,
                                  int s, int pad, int accumulate, const void* mask_hi, long mask_img_groups,
                                  const float* mask_scale, int mask_pooled, const float* dy_scale, const float* dx_scale,
                                  float* dx_amax, float* dx_f32, long dx_f32_img_stride, hipStream_t stream) {
    PoolArgs a;
    int rc = fill_pool(a, dy_hi, dy_lo, dy_img_groups, dx_hi, dx_lo, dx_img_groups, N, C, H, W, Ho, Wo, k, s, pad, dy_scale, dx_scale,
                       dx_amax, "pl maxpool bwd");
    if (rc != SSN_OK) return rc;
    SSN_CHECK_ARG(dy_hi && dy_lo && dy_scale && argmax && ((dx_hi && dx_lo) || dx_f32), "pl maxpool bwd: bad arguments");
    if (dx_f32) {       // fp32 NCHW output: dx_scale must point at 1.0, no accumulation (nothing to read back in this layout)
        SSN_CHECK_ARG(!accumulate, "pl maxpool bwd: fp32 output cannot accumulate");
        a.y_f32 = dx_f32;
        a.y_f32_img_stride = dx_f32_img_stride;
        a.y_img_groups = C / 8;
    }
    a.argmax = const_cast<unsigned char*>(argmax);
    a.accumulate = accumulate;
    if (mask_hi && mask_scale) {
        a.mask_hi = mask_hi;
        a.aff_scale = mask_scale;
        a.mask_img_groups = mask_img_groups;
    }
    SSN_CHECK_ARG(!mask_pooled || (mask_hi && mask_scale && !accumulate && k == 3 && s == 2 && (pad == 0 || pad == 1)),
                  "pl maxpool bwd: a pooled mask needs a 3x3 / stride-2 pool (pad 0 / 1) that does not accumulate");
    a.relu = mask_pooled ? 1 : 0;
    const dim3 grid(grid_for((long)N * a.G * H * W));
    // the training step's cases (planes output, nothing to accumulate, no mask or the pooled one, 16-byte aligned scale vector) on
    // the low-instruction-count kernel
    const bool fast = k == 3 && s == 2 && (pad == 0 || pad == 1) && !dx_f32 && !accumulate && (!a.mask_hi || mask_pooled) &&
                      (!a.mask_hi || (reinterpret_cast<uintptr_t>(mask_scale) & 15) == 0) && (long)N * C / 8 * Ho * Wo * 16 < (1l << 31) &&
                      pool_fits32(N, dy_img_groups, a.G, (long)Ho * Wo) && pool_fits32(N, dx_img_groups, a.G, (long)H * W) &&
                      (!a.mask_hi || pool_fits32(N, mask_img_groups, a.G, (long)Ho * Wo));
    if (fast) {
        const dim3 gb(grid_for((long)N * a.G * ((H + 1) / 2) * ((W + 1) / 2)));
        if (pad == 0 && a.mask_hi) hipLaunchKernelGGL((pl_maxpool_bwd_k3s2_fast_kernel<0, true>), gb, dim3(256), 0, stream, a);
        else if (pad == 0) hipLaunchKernelGGL((pl_maxpool_bwd_k3s2_fast_kernel<0, false>), gb, dim3(256), 0, stream, a);
        else if (a.mask_hi) hipLaunchKernelGGL((pl_maxpool_bwd_k3s2_fast_kernel<1, true>), gb, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((pl_maxpool_bwd_k3s2_fast_kernel<1, false>), gb, dim3(256), 0, stream, a);
    } else if (k == 3 && s == 2 && (pad == 0 || pad == 1))
        if (pad == 0)
            hipLaunchKernelGGL(pl_maxpool_bwd_k3s2_kernel<0>, dim3(grid_for((long)N * a.G * ((H + 1) / 2) * ((W + 1) / 2))), dim3(256), 0,
                               stream, a);
        else
            hipLaunchKernelGGL(pl_maxpool_bwd_k3s2_kernel<1>, dim3(grid_for((long)N * a.G * ((H + 1) / 2) * ((W + 1) / 2))), dim3(256), 0,
                               stream, a);
    else if (k == 3 && s == 2)
        hipLaunchKernelGGL((pl_maxpool_bwd_kernel<3, 2>), grid, dim3(256), 0, stream, a);
    else if (k == 3 && s == 1)
        hipLaunchKernelGGL((pl_maxpool_bwd_kernel<3, 1>), grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((pl_maxpool_bwd_kernel<0, 0>), grid, dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("pl_maxpool_bwd");
    return SSN_OK;
}

// y = relu?(scale[c] * avgpool(x) + shift[c]) (k x k, stride 1, zero padding counted): forward of the pool behind its
// projection (scale / shift null: plain average) -- and, with dy as x and no affine, its backward (the same stencil).
extern "C" int ssn_pl_avgpool_affine(const void* x_hi, const void* x_lo, long x_img_groups, void* y_hi, void* y_lo,
                                     long y_img_groups, const float* scale, const float* shift, int relu, int N, int C, int H,
                                     int W, int k, int pad, const float* x_scale, const float* y_scale, float* y_amax,
                                     hipStream_t stream) {
    PoolArgs a;
    int rc = fill_pool(a, x_hi, x_lo, x_img_groups, y_hi, y_lo, y_img_groups, N, C, H, W, H, W, k, 1, pad, x_scale, y_scale, y_amax,
                       "pl avgpool affine");
    if (rc != SSN_OK) return rc;
    SSN_CHECK_ARG(x_hi && x_lo && x_scale && 2 * pad == k - 1 && (!scale == !shift), "pl avgpool affine: bad arguments");
    a.aff_scale = scale;
    a.aff_shift = shift;
    a.relu = relu;
    if (k == 3 && pad == 1 && pool_fits32(N, x_img_groups, a.G, (long)H * W) && pool_fits32(N, y_img_groups, a.G, (long)H * W))
        hipLaunchKernelGGL(pl_avgpool3_fast_kernel, dim3(grid_for((long)N * a.G * H * W)), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(pl_avgpool_affine_kernel, dim3(grid_for((long)N * a.G * H * W)), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("pl_avgpool_affine");
    return SSN_OK;
}

// in place: g <- g * (y > 0) * scale[c] on a planes slice (y_hi = hi plane of the activation at the same channels)
extern "C" int ssn_pl_relu_bn_bwd(void* g_hi, void* g_lo, long g_img_groups, const void* y_hi, long y_img_groups,
                                  const float* scale, int N, int C, int HW, const float* g_scale, float* g_amax,
                                  hipStream_t stream) {
    PoolArgs a;
    int rc = fill_pool(a, nullptr, nullptr, 0, g_hi, g_lo, g_img_groups, N, C, HW, 1, HW, 1, 1, 1, 0, g_scale, g_scale, g_amax,
                       "pl relu bn bwd");
    if (rc != SSN_OK) return rc;
    SSN_CHECK_ARG(y_hi && scale, "pl relu bn bwd: bad arguments");
    a.mask_hi = y_hi;
    a.aff_scale = scale;
    a.mask_img_groups = y_img_groups;
    hipLaunchKernelGGL(pl_relu_bn_bwd_kernel, dim3(grid_for((long)N * a.G * HW)), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("pl_relu_bn_bwd");
    return SSN_OK;
}

extern "C" int ssn_pl_gap_fwd(const void* x_hi, const void* x_lo, long x_img_groups, float* y, int N, int C, int HW,
                              const float* x_scale, hipStream_t stream) {
    SSN_CHECK_ARG(x_hi && x_lo && y && x_scale && C % 8 == 0, "pl gap fwd: bad arguments");
    if (GAP_NG * HW <= GAP_MAX) {
        const int gblocks = (C / 8 + GAP_NG - 1) / GAP_NG;
        hipLaunchKernelGGL(pl_gap_fwd_lds_kernel, dim3((unsigned)(N * gblocks)), dim3(256), 0, stream, x_hi, x_lo, x_img_groups, y, N, C / 8,
                           HW, x_scale, gblocks);
    } else {
        hipLaunchKernelGGL(pl_gap_fwd_kernel, dim3((N * (C / 8) + 255) / 256), dim3(256), 0, stream, x_hi, x_lo, x_img_groups, y, N, C / 8,
                           HW, x_scale);
    }
    SSN_CHECK_LAUNCH("pl_gap_fwd");
    return SSN_OK;
}

// dx = dy / HW broadcast over the pixels, times (x > 0) * mask_scale[c] when the mask is given
extern "C" int ssn_pl_gap_bwd(const float* dy, void* dx_hi, void* dx_lo, long dx_img_groups, int N, int C, int HW,
                              const void* mask_hi, long mask_img_groups, const float* mask_scale, const float* dx_scale,
                              float* dx_amax, hipStream_t stream) {
    PoolArgs a;
    int rc = fill_pool(a, nullptr, nullptr, 0, dx_hi, dx_lo, dx_img_groups, N, C, HW, 1, HW, 1, 1, 1, 0, dx_scale, dx_scale, dx_amax,
                       "pl gap bwd");
    if (rc != SSN_OK) return rc;
    SSN_CHECK_ARG(dy, "pl gap bwd: bad arguments");
    a.aff_shift = dy;
    if (mask_hi && mask_scale) {
        a.mask_hi = mask_hi;
        a.aff_scale = mask_scale;
        a.mask_img_groups = mask_img_groups;
    }
    hipLaunchKernelGGL(pl_gap_bwd_kernel, dim3(grid_for((long)N * a.G * HW)), dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("pl_gap_bwd");
    return SSN_OK;
}

extern "C" long ssn_pl_channel_sum_workspace_bytes(int C) { return (long)CS_SHARES * ((C + 7) / 8) * 8 * (long)sizeof(float); }

// out[c] = sum over images and pixels of a planes slice (fixed order: deterministic)
// `count` slices (HOST arrays, one entry per slice; the same N) in ONE pair of launches; workspace:
// ssn_pl_channel_sum_workspace_bytes(sum of C) bytes.  Same summation order per slice as ssn_pl_channel_sum: bit-identical results.
extern "C" int ssn_pl_channel_sum_multi(int count, const void* const* g_hi, const void* const* g_lo, const long* g_img_groups,
                                        float* const* out, int N, const int* C, const int* HW, const float* const* g_scale,
                                        void* workspace, long ws_bytes, hipStream_t stream) {
    SSN_CHECK_ARG(count >= 0 && (count == 0 || (g_hi && g_lo && g_img_groups && out && C && HW && g_scale && workspace)),
                  "pl channel sum multi: bad arguments");
    for (int base = 0; base < count; base += CSM_MAX) {
        ChannelSumTable t;
        t.count = count - base < CSM_MAX ? count - base : CSM_MAX;
        t.N = N;
        int groups = 0;
        for (int i = 0; i < t.count; ++i) {
            const int j = base + i;
            SSN_CHECK_ARG(g_hi[j] && g_lo[j] && out[j] && g_scale[j] && C[j] > 0 && C[j] % 8 == 0 && HW[j] > 0,
                          "pl channel sum multi: bad entry %d", j);
            t.hi[i] = g_hi[j];
            t.lo[i] = g_lo[j];
            t.scale[i] = g_scale[j];
            t.out[i] = out[j];
            t.img_groups[i] = g_img_groups[j];
            t.HW[i] = HW[j];
            t.g0[i] = groups;
            groups += C[j] / 8;
        }
        t.g0[t.count] = groups;
        if (ws_bytes < ssn_pl_channel_sum_workspace_bytes(groups * 8)) {
            ssn_set_error("pl channel sum multi: workspace too small");
            return SSN_ERR_WORKSPACE;
        }
        hipLaunchKernelGGL(pl_channel_sum_multi_kernel, dim3(groups, CS_SHARES), dim3(256), 0, stream, t, (float*)workspace);
        hipLaunchKernelGGL(pl_channel_sum_multi_final_kernel, dim3((groups * 8 + 255) / 256), dim3(256), 0, stream, t,
                           (const float*)workspace);
    }
    SSN_CHECK_LAUNCH("pl_channel_sum_multi");
    return SSN_OK;
}

extern "C" int ssn_pl_channel_sum(const void* g_hi, const void* g_lo, long g_img_groups, float* out, int N, int C, int HW,
                                  const float* g_scale, void* workspace, long ws_bytes, hipStream_t stream) {
    SSN_CHECK_ARG(g_hi && g_lo && out && g_scale && workspace && C % 8 == 0, "pl channel sum: bad arguments");
    if (ws_bytes < ssn_pl_channel_sum_workspace_bytes(C)) {
        ssn_set_error("pl channel sum: workspace too small");
        return SSN_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(pl_channel_sum_kernel, dim3(C / 8, CS_SHARES), dim3(256), 0, stream, g_hi, g_lo, g_img_groups, N, C / 8, HW,
                       (float*)workspace);
    hipLaunchKernelGGL(pl_channel_sum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, (const float*)workspace, out, C,
                       C / 8, g_scale);
    SSN_CHECK_LAUNCH("pl_channel_sum");
    return SSN_OK;
}
