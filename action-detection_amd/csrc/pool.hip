// Pooling layers of BN-Inception: 3x3 max/avg pools with Caffe-style
// ceil_mode output sizes (the host computes Ho/Wo; windows are clipped here), average pools
// with count_include_pad=True, and the final global average pool.  HBM-bound; every kernel is
// one thread per output (fwd) or per input (bwd) element with W-contiguous lanes.
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
    FastDiv div_chw, div_hw, div_w;
};

// KC/SC/PC: compile-time window (BN-Inception uses 3/2/0 and 3/1/1); other shapes instantiate on demand.
template <bool MAX, int KC, int SC, int PC>
__global__ __launch_bounds__(256) void pool_fwd_kernel(PoolArgs p) {
    const int howo = p.Ho * p.Wo;
    const __amdgpu_buffer_rsrc_t xr = pool_rsrc(p.x, p.x_bytes);
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
        }
        p.y[(long)n * p.y_img_stride + rem] = out;
    }
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
    FastDiv div_chw, div_hw, div_w;
};

template <bool MAX, int KC, int SC, int PC>
__global__ __launch_bounds__(256) void pool_bwd_kernel(PoolBwdArgs p) {
    const int howo = p.Ho * p.Wo;
    constexpr int NWIN = (KC + SC - 1) / SC;  // max windows per axis covering one input pixel
    const __amdgpu_buffer_rsrc_t dyr = pool_rsrc(p.dy, p.dy_bytes);
    const __amdgpu_buffer_rsrc_t ixr = pool_rsrc(MAX ? (const void*)p.idx : (const void*)p.dy, MAX ? p.idx_bytes : 4u);
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
            g = (sc < 0.f) ? g * -sc : (p.mask_y[(long)n * p.mask_img_stride + rem] > 0.f ? g * sc : 0.f);
        }
        *dst = g;
    }
}

// one wave per (n, c): mean over the HW plane
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
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* dy, float* dx, long total, int C, int HW,
                                                      long dx_img_stride, int accumulate, FastDiv div_chw,
                                                      FastDiv div_hw) {
    const float inv = 1.f / (float)HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        uint32_t n, rem, c, hw;
        fd_divmod((uint32_t)i, div_chw, n, rem);
        fd_divmod(rem, div_hw, c, hw);
        float* dst = dx + (long)n * dx_img_stride + rem;
        const float g = dy[(long)n * C + c] * inv;
        *dst = accumulate ? *dst + g : g;
    }
}

inline unsigned grid_for(long total, int cap = 65536) {
    long b = (total + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int ssn_pool_fwd(int is_max, const float* x, float* y, unsigned char* argmax, int N, int C, int H, int W,
                            long x_img_stride, int Ho, int Wo, long y_img_stride, int ksize, int stride, int pad,
                            hipStream_t stream) {
    SSN_CHECK_ARG(x && y, "pool_fwd: null pointer");
    PoolArgs a;
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

extern "C" int ssn_pool_bwd(int is_max, const float* dy, const unsigned char* argmax, float* dx, int N, int C, int H,
                            int W, long dx_img_stride, int Ho, int Wo, long dy_img_stride, int ksize, int stride,
                            int pad, int accumulate, const float* mask_y, long mask_img_stride,
                            const float* mask_scale, hipStream_t stream) {
    SSN_CHECK_ARG(dy && dx && (!is_max || argmax), "pool_bwd: null pointer");
    PoolBwdArgs a;
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
                                      int accumulate, hipStream_t stream) {
    SSN_CHECK_ARG(dy && dx, "gap_bwd: null pointer");
    const long total = (long)N * C * HW;
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dy, dx, total, C, HW,
                       dx_img_stride, accumulate, make_fastdiv((uint32_t)(C * HW)), make_fastdiv((uint32_t)HW));
    SSN_CHECK_LAUNCH("gap_bwd");
    return SSN_OK;
}
