// Training-mode BatchNorm2d (+ ReLU) on planes tensors: bn_mode 'partial' (the first BatchNorm2d) and 'full' (all of them) of
// /root/reference/ssn_models.py:95-105,156-174 for the planes executor -- the same mathematics as bn_train.hip (which serves the
// fp32-layout executor), on the NC8HW8 two-plane layout of planes.h.  gfx950.
//
//   forward :  z = conv(x) WITHOUT its bias (planes, scale s_z), batch mean / biased variance per channel over (N, H, W),
//              xhat = (z - mean) * invstd,  y = relu?(gamma * xhat + beta)  (planes, scale s_y)
//   backward:  g = dy * (y > 0)  (the sign of y's HIGH plane, as every fused backward epilogue of this layout reads it),
//              dbeta = sum g,  dgamma = sum g * xhat,  dz = gamma * invstd * (g - dbeta / n - xhat * dgamma / n)
//              (planes, scale s_dz -- or fp32 NCHW for the stem, whose weight gradient runs on the fp32-layout kernel)
//
// All passes are HBM-bound: a thread moves one 16-byte group of 8 channels per plane.  The reductions are two-level and atomic-free:
// workgroup (group g, share s) reduces share s of the (image, pixel) positions of 8 channels, a tiny second kernel combines the
// shares IN DOUBLE.  The variance is accumulated around the per-channel pivot K = z[0][c][0] in STORED units (sum (z - K),
// sum (z - K)^2), converted to real units once per channel.
#include "ssn_common.h"
#include "planes.h"

namespace {

using namespace pl;

constexpr int BNP_SHARES = 32;

struct BnPlArgs {
    const void* z_hi;      // the convolution output (no bias, no affine, no ReLU)
    const void* z_lo;
    long z_img_groups;
    const float* z_scale;
    const void* dy_hi;     // backward: gradient of y
    const void* dy_lo;
    long dy_img_groups;
    const float* dy_scale;
    const void* mask_hi;   // backward: HIGH plane of y (ReLU decision), or null
    long mask_img_groups;
    void* o_hi;            // apply: y;  backward: dz
    void* o_lo;
    long o_img_groups;
    const float* o_scale;
    float* o_amax;
    float* o_f32;          // backward: dz as fp32 NCHW instead of planes
    long o_f32_img_stride;
    const float* mean;
    const float* invstd;
    const float* gamma;
    const float* beta;
    const float* dgamma;
    const float* dbeta;
    float* part;           // [BNP_SHARES][G][16]
    int N, G, HW, relu;
};

__device__ __forceinline__ void bnp_load8(const void* hi, const void* lo, long o, float (&v)[8]) {
    pl_join8(reinterpret_cast<const u32x4*>(hi)[o], reinterpret_cast<const u32x4*>(lo)[o], v);
}

// 16 running sums per thread -> part[(share * G + g) * 16 + 0..15] in a fixed order
__device__ __forceinline__ void bnp_block_reduce16(float (&a)[16], float* part_row, float (*red)[17]) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red[threadIdx.x][e] = a[e];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[threadIdx.x][e] += red[threadIdx.x + st][e];
        __syncthreads();
    }
    if (threadIdx.x < 16) part_row[threadIdx.x] = red[0][threadIdx.x];
}

// sums of (z - K) and (z - K)^2 in stored units, K[e] = stored z[0][8 g + e][0]
__global__ __launch_bounds__(256) void pl_bn_stats_part_kernel(BnPlArgs p) {
    __shared__ float red[256][17];
    const int g = blockIdx.x, share = blockIdx.y;
    const long total = (long)p.N * p.HW;
    const long per = (total + BNP_SHARES - 1) / BNP_SHARES;
    const long begin = (long)share * per;
    long end = begin + per;
    if (end > total) end = total;
    float K[8];
    bnp_load8(p.z_hi, p.z_lo, (long)g * p.HW, K);
    float a[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] = 0.f;
    for (long i = begin + threadIdx.x; i < end; i += 256) {
        const long n = i / p.HW, q = i - n * p.HW;
        float v[8];
        bnp_load8(p.z_hi, p.z_lo, (n * p.z_img_groups + g) * p.HW + q, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = v[e] - K[e];
            a[e] += d;
            a[8 + e] += d * d;
        }
    }
    bnp_block_reduce16(a, p.part + ((long)share * p.G + g) * 16, red);
}
// mean / invstd per channel in real units; running statistics as torch.nn.BatchNorm2d updates them (the running mean sees the conv
// bias the convolution left out)
__global__ __launch_bounds__(256) void pl_bn_stats_final_kernel(BnPlArgs p, const float* conv_bias, float* mean, float* invstd,
                                                               float* running_mean, float* running_var, int C, float eps,
                                                               float momentum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int g = c >> 3, e = c & 7;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < BNP_SHARES; ++s) {
        s1 += (double)p.part[((long)s * p.G + g) * 16 + e];
        s2 += (double)p.part[((long)s * p.G + g) * 16 + 8 + e];
    }
    float K[8];
    bnp_load8(p.z_hi, p.z_lo, (long)g * p.HW, K);
    const double n = (double)p.N * (double)p.HW;
    const double inv_s = 1.0 / (double)*p.z_scale;
    const double m = ((double)K[e] + s1 / n) * inv_s;
    double var = (s2 - s1 * s1 / n) / n * inv_s * inv_s;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double mb = m + (conv_bias ? (double)conv_bias[c] : 0.0);
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mb);
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
}

// y = relu?(gamma * (z - mean) * invstd + beta)
__global__ __launch_bounds__(256) void pl_bn_apply_kernel(BnPlArgs p) {
    const long total = (long)p.N * p.G * p.HW;
    const float inv_sz = 1.f / *p.z_scale, so = *p.o_scale;
    float vmax = 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long q = idx % p.HW;
        const long ng = idx / p.HW;
        const int g = (int)(ng % p.G), n = (int)(ng / p.G);
        float v[8], out[8];
        bnp_load8(p.z_hi, p.z_lo, ((long)n * p.z_img_groups + g) * p.HW + q, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 8 * g + e;
            float y = p.gamma[c] * ((v[e] * inv_sz - p.mean[c]) * p.invstd[c]) + p.beta[c];
            if (p.relu) y = fmaxf(y, 0.f);
            vmax = fmaxf(vmax, fabsf(y * so));       // before the clamp (planes.h: the true magnitude is recorded)
            out[e] = pl_clamp(y * so);
        }
        u32x4 hi, lo;
        pl_split8(out, hi, lo);
        const long o = ((long)n * p.o_img_groups + g) * p.HW + q;
        reinterpret_cast<u32x4*>(p.o_hi)[o] = hi;
        reinterpret_cast<u32x4*>(p.o_lo)[o] = lo;
    }
    amax_emit_block(p.o_amax, vmax / so);
}

// g = dy * (y > 0 | 1) and xhat of one 8-channel group, real units
__device__ __forceinline__ void bnp_g_xhat(const BnPlArgs& p, int n, int g, long q, float inv_sz, float inv_sdy, float (&gr)[8],
                                           float (&xh)[8]) {
    float z[8];
    bnp_load8(p.z_hi, p.z_lo, ((long)n * p.z_img_groups + g) * p.HW + q, z);
    bnp_load8(p.dy_hi, p.dy_lo, ((long)n * p.dy_img_groups + g) * p.HW + q, gr);
    u32x4 mk = {0u, 0u, 0u, 0u};
    if (p.mask_hi) mk = reinterpret_cast<const u32x4*>(p.mask_hi)[((long)n * p.mask_img_groups + g) * p.HW + q];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = 8 * g + e;
        const float m = (e & 1) ? f16_pair_hi(mk[e >> 1]) : f16_pair_lo(mk[e >> 1]);
        gr[e] = (!p.mask_hi || m > 0.f) ? gr[e] * inv_sdy : 0.f;
        xh[e] = (z[e] * inv_sz - p.mean[c]) * p.invstd[c];
    }
}
// part: sum g, sum g * xhat
__global__ __launch_bounds__(256) void pl_bn_bwd_part_kernel(BnPlArgs p) {
    __shared__ float red[256][17];
    const int g = blockIdx.x, share = blockIdx.y;
    const long total = (long)p.N * p.HW;
    const long per = (total + BNP_SHARES - 1) / BNP_SHARES;
    const long begin = (long)share * per;
    long end = begin + per;
    if (end > total) end = total;
    const float inv_sz = 1.f / *p.z_scale, inv_sdy = 1.f / *p.dy_scale;
    float a[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] = 0.f;
    for (long i = begin + threadIdx.x; i < end; i += 256) {
        const long n = i / p.HW, q = i - n * p.HW;
        float gr[8], xh[8];
        bnp_g_xhat(p, (int)n, g, q, inv_sz, inv_sdy, gr, xh);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[e] += gr[e];
            a[8 + e] += gr[e] * xh[e];
        }
    }
    bnp_block_reduce16(a, p.part + ((long)share * p.G + g) * 16, red);
}
__global__ __launch_bounds__(256) void pl_bn_bwd_final_kernel(const float* part, float* dgamma, float* dbeta, int C, int G) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int g = c >> 3, e = c & 7;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < BNP_SHARES; ++s) {
        s1 += (double)part[((long)s * G + g) * 16 + e];
        s2 += (double)part[((long)s * G + g) * 16 + 8 + e];
    }
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
}
// dz = gamma * invstd * (g - dbeta / n - xhat * dgamma / n)
__global__ __launch_bounds__(256) void pl_bn_bwd_apply_kernel(BnPlArgs p) {
    const long total = (long)p.N * p.G * p.HW;
    const float inv_sz = 1.f / *p.z_scale, inv_sdy = 1.f / *p.dy_scale;
    const float so = p.o_f32 ? 1.f : *p.o_scale;
    const float inv_count = 1.f / ((float)p.N * (float)p.HW);
    float vmax = 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long q = idx % p.HW;
        const long ng = idx / p.HW;
        const int g = (int)(ng % p.G), n = (int)(ng / p.G);
        float gr[8], xh[8], out[8];
        bnp_g_xhat(p, n, g, q, inv_sz, inv_sdy, gr, xh);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 8 * g + e;
            const float v = p.gamma[c] * p.invstd[c] * (gr[e] - p.dbeta[c] * inv_count - xh[e] * (p.dgamma[c] * inv_count)) * so;
            vmax = fmaxf(vmax, fabsf(v));
            out[e] = v;
        }
        if (p.o_f32) {
#pragma unroll
            for (int e = 0; e < 8; ++e) p.o_f32[(long)n * p.o_f32_img_stride + (long)(8 * g + e) * p.HW + q] = out[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = pl_clamp(out[e]);
            u32x4 hi, lo;
            pl_split8(out, hi, lo);
            const long o = ((long)n * p.o_img_groups + g) * p.HW + q;
            reinterpret_cast<u32x4*>(p.o_hi)[o] = hi;
            reinterpret_cast<u32x4*>(p.o_lo)[o] = lo;
        }
    }
    amax_emit_block(p.o_amax, vmax / so);
}

inline unsigned bnp_grid(long total) {      // (grid-stride kernels: eight workgroups per CU, as planes_ops.hip's grid_for -- profiles/r6_grid_cap.txt)
    long b = (total + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
// bytes of workspace the two reductions below need for a slice of C channels
extern "C" long ssn_pl_bn_train_workspace_bytes(int C) { return (long)BNP_SHARES * ((C + 7) / 8) * 16 * (long)sizeof(float); }

// Batch statistics of the planes slice z [N][C][HW] (C a multiple of 8; scale *z_scale): mean[c], invstd[c] = 1 / sqrt(biased var +
// eps) in real units; running_mean / running_var (optional, together) are updated in place like torch.nn.BatchNorm2d(momentum) does,
// with conv_bias[c] (optional) added to the batch mean first -- z is the convolution WITHOUT its bias.
// Replaces the statistics half of F.batch_norm(training=True) behind model_zoo.BNInception.forward
// (/root/reference/ssn_models.py:266 with bn_mode 'partial' / 'full', :95-105,156-174) for the planes executor.
extern "C" int ssn_pl_bn_train_stats(const void* z_hi, const void* z_lo, long z_img_groups, const float* z_scale,
                                     const float* conv_bias, float* mean, float* invstd, float* running_mean, float* running_var,
                                     int N, int C, int HW, float eps, float momentum, void* workspace, long ws_bytes,
                                     hipStream_t stream) {
    SSN_CHECK_ARG(z_hi && z_lo && z_scale && mean && invstd && workspace && N >= 1 && C >= 8 && C % 8 == 0 && HW >= 1,
                  "pl_bn_train_stats: bad arguments (C must be a multiple of 8)");
    SSN_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "pl_bn_train_stats: running_mean / running_var go together");
    if (ws_bytes < ssn_pl_bn_train_workspace_bytes(C)) {
        ssn_set_error("pl_bn_train_stats: workspace %ld < %ld bytes", ws_bytes, ssn_pl_bn_train_workspace_bytes(C));
        return SSN_ERR_WORKSPACE;
    }
    BnPlArgs p = {};
    p.z_hi = z_hi;
    p.z_lo = z_lo;
    p.z_img_groups = z_img_groups;
    p.z_scale = z_scale;
    p.part = (float*)workspace;
    p.N = N;
    p.G = C / 8;
    p.HW = HW;
    hipLaunchKernelGGL(pl_bn_stats_part_kernel, dim3((unsigned)p.G, (unsigned)BNP_SHARES), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(pl_bn_stats_final_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream, p, conv_bias, mean, invstd,
                       running_mean, running_var, C, eps, momentum);
    SSN_CHECK_LAUNCH("pl_bn_train_stats");
    return SSN_OK;
}

// y = relu?(gamma * (z - mean) * invstd + beta) into the planes slice y (scale *y_scale; y_amax: its amax slot, pre-clamp maximum).
extern "C" int ssn_pl_bn_train_apply(const void* z_hi, const void* z_lo, long z_img_groups, const float* z_scale, void* y_hi,
                                     void* y_lo, long y_img_groups, const float* y_scale, float* y_amax, const float* mean,
                                     const float* invstd, const float* gamma, const float* beta, int relu, int N, int C, int HW,
                                     hipStream_t stream) {
    SSN_CHECK_ARG(z_hi && z_lo && z_scale && y_hi && y_lo && y_scale && mean && invstd && gamma && beta && N >= 1 && C >= 8 &&
                      C % 8 == 0 && HW >= 1,
                  "pl_bn_train_apply: bad arguments (C must be a multiple of 8)");
    BnPlArgs p = {};
    p.z_hi = z_hi;
    p.z_lo = z_lo;
    p.z_img_groups = z_img_groups;
    p.z_scale = z_scale;
    p.o_hi = y_hi;
    p.o_lo = y_lo;
    p.o_img_groups = y_img_groups;
    p.o_scale = y_scale;
    p.o_amax = y_amax;
    p.mean = mean;
    p.invstd = invstd;
    p.gamma = gamma;
    p.beta = beta;
    p.relu = relu;
    p.N = N;
    p.G = C / 8;
    p.HW = HW;
    hipLaunchKernelGGL(pl_bn_apply_kernel, dim3(bnp_grid((long)N * p.G * HW)), dim3(256), 0, stream, p);
    SSN_CHECK_LAUNCH("pl_bn_train_apply");
    return SSN_OK;
}

// Backward of the same layer (autograd of F.batch_norm(training=True) + ReLU, entered from loss.backward(),
// /root/reference/ssn_train.py:236): dgamma[c], dbeta[c], and dz -- into the planes slice dz (scale *dz_scale, amax slot dz_amax),
// or, with dz_f32, as fp32 NCHW (image stride dz_f32_img_stride floats; dz_amax then takes its plain maximum).  y_hi: the HIGH
// plane of the layer's output (the ReLU decision; null: no ReLU).
extern "C" int ssn_pl_bn_train_bwd(const void* dy_hi, const void* dy_lo, long dy_img_groups, const float* dy_scale, const void* y_hi,
                                   long y_img_groups, const void* z_hi, const void* z_lo, long z_img_groups, const float* z_scale,
                                   const float* mean, const float* invstd, const float* gamma, float* dgamma, float* dbeta,
                                   void* dz_hi, void* dz_lo, long dz_img_groups, const float* dz_scale, float* dz_amax,
                                   float* dz_f32, long dz_f32_img_stride, int N, int C, int HW, void* workspace, long ws_bytes,
                                   hipStream_t stream) {
    SSN_CHECK_ARG(dy_hi && dy_lo && dy_scale && z_hi && z_lo && z_scale && mean && invstd && gamma && dgamma && dbeta && workspace &&
                      N >= 1 && C >= 8 && C % 8 == 0 && HW >= 1,
                  "pl_bn_train_bwd: bad arguments (C must be a multiple of 8)");
    SSN_CHECK_ARG(dz_f32 || (dz_hi && dz_lo && dz_scale), "pl_bn_train_bwd: no destination");
    if (ws_bytes < ssn_pl_bn_train_workspace_bytes(C)) {
        ssn_set_error("pl_bn_train_bwd: workspace %ld < %ld bytes", ws_bytes, ssn_pl_bn_train_workspace_bytes(C));
        return SSN_ERR_WORKSPACE;
    }
    BnPlArgs p = {};
    p.z_hi = z_hi;
    p.z_lo = z_lo;
    p.z_img_groups = z_img_groups;
    p.z_scale = z_scale;
    p.dy_hi = dy_hi;
    p.dy_lo = dy_lo;
    p.dy_img_groups = dy_img_groups;
    p.dy_scale = dy_scale;
    p.mask_hi = y_hi;
    p.mask_img_groups = y_img_groups;
    p.o_hi = dz_hi;
    p.o_lo = dz_lo;
    p.o_img_groups = dz_img_groups;
    p.o_scale = dz_scale;
    p.o_amax = dz_amax;
    p.o_f32 = dz_f32;
    p.o_f32_img_stride = dz_f32_img_stride;
    p.mean = mean;
    p.invstd = invstd;
    p.gamma = gamma;
    p.dgamma = dgamma;
    p.dbeta = dbeta;
    p.part = (float*)workspace;
    p.N = N;
    p.G = C / 8;
    p.HW = HW;
    hipLaunchKernelGGL(pl_bn_bwd_part_kernel, dim3((unsigned)p.G, (unsigned)BNP_SHARES), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(pl_bn_bwd_final_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream, (const float*)workspace,
                       dgamma, dbeta, C, p.G);
    hipLaunchKernelGGL(pl_bn_bwd_apply_kernel, dim3(bnp_grid((long)N * p.G * HW)), dim3(256), 0, stream, p);
    SSN_CHECK_LAUNCH("pl_bn_train_bwd");
    return SSN_OK;
}
