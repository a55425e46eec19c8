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

struct PoolArgs {
    const float* x;
    float* y;
    uint8_t* idx;  // max: argmax out (may be null); unused for avg
    int N, C, H, W, Ho, Wo;
    long x_img_stride, y_img_stride;
    int k, stride, pad;
    long total;
    FastDiv div_chw, div_hw, div_w;
};

// KC/SC/PC > 0: compile-time window (the two shapes BN-Inception uses: 3/2/0 and 3/1/1) so the tap
// loops unroll; KC == 0: runtime fallback.
template <bool MAX, int KC, int SC, int PC>
__global__ __launch_bounds__(256) void pool_fwd_kernel(PoolArgs p) {
    const int howo = p.Ho * p.Wo;
    const int K_ = KC ? KC : p.k, S_ = KC ? SC : p.stride, P_ = KC ? PC : p.pad;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long)gridDim.x * 256) {
        uint32_t n, rem, c, hw, ho, wo;
        fd_divmod((uint32_t)i, p.div_chw, n, rem);
        fd_divmod(rem, p.div_hw, c, hw);
        fd_divmod(hw, p.div_w, ho, wo);
        const float* xp = p.x + (long)n * p.x_img_stride + (long)c * p.H * p.W;
        const int h0 = (int)ho * S_ - P_, w0 = (int)wo * S_ - P_;
        float out;
        if (MAX) {
            float best = -INFINITY;
            int bi = 0;
            bool first = true;
#pragma unroll
            for (int r = 0; r < K_; ++r) {
                const int hi = h0 + r;
                if ((unsigned)hi >= (unsigned)p.H) continue;
#pragma unroll
                for (int s = 0; s < K_; ++s) {
                    const int wi = w0 + s;
                    if ((unsigned)wi >= (unsigned)p.W) continue;
                    const float v = xp[hi * p.W + wi];
                    // torch rule: start at the first in-bounds element, move on (v > best) or NaN
                    if (first) {
                        bi = r * K_ + s;
                        first = false;
                    }
                    if (v > best || v != v) {
                        best = v;
                        bi = r * K_ + s;
                    }
                }
            }
            out = best;
            if (p.idx) p.idx[(long)n * p.C * howo + rem] = (uint8_t)bi;
        } else {
            float s_ = 0.f;
#pragma unroll
            for (int r = 0; r < K_; ++r) {
                const int hi = h0 + r;
                if ((unsigned)hi >= (unsigned)p.H) continue;
#pragma unroll
                for (int s = 0; s < K_; ++s) {
                    const int wi = w0 + s;
                    if ((unsigned)wi >= (unsigned)p.W) continue;
                    s_ += xp[hi * p.W + wi];
                }
            }
            // count_include_pad=True: divisor = window clipped to the PADDED extent (torch avg_pool2d)
            int he = h0 + K_, we = w0 + K_;
            if (he > p.H + P_) he = p.H + P_;
            if (we > p.W + P_) we = p.W + P_;
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
    long total;  // N*C*H*W
    FastDiv div_chw, div_hw, div_w;
};

template <bool MAX, int KC, int SC, int PC>
__global__ __launch_bounds__(256) void pool_bwd_kernel(PoolBwdArgs p) {
    const int howo = p.Ho * p.Wo;
    const int K_ = KC ? KC : p.k, S_ = KC ? SC : p.stride, P_ = KC ? PC : p.pad;
    constexpr int NWIN = KC ? (KC + SC - 1) / SC : 0;  // max windows per axis covering one input pixel
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long)gridDim.x * 256) {
        uint32_t n, rem, c, hw, hi, wi;
        fd_divmod((uint32_t)i, p.div_chw, n, rem);
        fd_divmod(rem, p.div_hw, c, hw);
        fd_divmod(hw, p.div_w, hi, wi);
        const float* dyp = p.dy + (long)n * p.dy_img_stride + (long)c * howo;
        const uint8_t* ip = MAX ? p.idx + ((long)n * p.C + c) * howo : nullptr;
        // output windows that contain (hi, wi): ho in [ceil((hi+pad-k+1)/s), floor((hi+pad)/s)]
        const int hp = (int)hi + P_, wp = (int)wi + P_;
        int ho_lo = hp - K_ + 1;
        ho_lo = ho_lo <= 0 ? 0 : (ho_lo + S_ - 1) / S_;
        int wo_lo = wp - K_ + 1;
        wo_lo = wo_lo <= 0 ? 0 : (wo_lo + S_ - 1) / S_;
        int ho_hi = hp / S_, wo_hi = wp / S_;
        if (ho_hi > p.Ho - 1) ho_hi = p.Ho - 1;
        if (wo_hi > p.Wo - 1) wo_hi = p.Wo - 1;
        float g = 0.f;
        auto visit = [&](int ho, int wo) {
            const int r = hp - ho * S_, s = wp - wo * S_;
            if (MAX) {
                if (ip[ho * p.Wo + wo] == (uint8_t)(r * K_ + s)) g += dyp[ho * p.Wo + wo];
            } else {
                const int h0 = ho * S_ - P_, w0 = wo * S_ - P_;
                int he = h0 + K_, we = w0 + K_;
                if (he > p.H + P_) he = p.H + P_;
                if (we > p.W + P_) we = p.W + P_;
                g += dyp[ho * p.Wo + wo] / (float)((he - h0) * (we - w0));
            }
        };
        if (KC) {
#pragma unroll
            for (int a = 0; a < NWIN; ++a)
#pragma unroll
                for (int b = 0; b < NWIN; ++b)
                    if (ho_lo + a <= ho_hi && wo_lo + b <= wo_hi) visit(ho_lo + a, wo_lo + b);
        } else {
            for (int ho = ho_lo; ho <= ho_hi; ++ho)
                for (int wo = wo_lo; wo <= wo_hi; ++wo) visit(ho, wo);
        }
        float* dst = p.dx + (long)n * p.dx_img_stride + rem;
        *dst = p.accumulate ? *dst + g : g;
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
    SSN_CHECK_ARG(ksize * ksize <= 255, "pool_fwd: window too large");
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
    else if (is_max)
        hipLaunchKernelGGL((pool_fwd_kernel<true, 0, 0, 0>), grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((pool_fwd_kernel<false, 0, 0, 0>), grid, dim3(256), 0, stream, a);
    SSN_CHECK_LAUNCH("pool_fwd");
    return SSN_OK;
}

extern "C" int ssn_pool_bwd(int is_max, const float* dy, const unsigned char* argmax, float* dx, int N, int C, int H,
                            int W, long dx_img_stride, int Ho, int Wo, long dy_img_stride, int ksize, int stride,
                            int pad, int accumulate, hipStream_t stream) {
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
    a.total = (long)N * C * H * W;
    SSN_CHECK_ARG(a.total < (1l << 31), "pool_bwd: tensor too large");
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
    else if (is_max)
        hipLaunchKernelGGL((pool_bwd_kernel<true, 0, 0, 0>), grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((pool_bwd_kernel<false, 0, 0, 0>), grid, dim3(256), 0, stream, a);
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
