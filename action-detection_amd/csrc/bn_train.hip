// Training-mode BatchNorm2d (+ ReLU) of the backbone: bn_mode 'partial' (the first BatchNorm2d) and 'full' (all of
// them) of /root/reference/ssn_models.py:95-105,156-174 -- the layers SSN.train() does NOT put into eval mode normalise
// with the statistics of the batch, update their running statistics (momentum 0.1, unbiased variance) and have the
// full batch-norm backward (the mean / variance terms of dx; dgamma, dbeta).  gfx950.
//
//   forward :  z = conv(x) WITHOUT its bias (a constant per channel cancels in z - mean(z); it only enters the running
//              mean), mean / biased var per channel over (N, H, W), xhat = (z - mean) * invstd,
//              y = relu?(gamma * xhat + beta)
//   backward:  g = dy * (y > 0),  dbeta = sum g,  dgamma = sum g * xhat,
//              dz = gamma * invstd * (g - dbeta / n - xhat * dgamma / n)
//
// All four kernels are HBM-bound passes over an NCHW channel slice.  The per-channel reductions are two-level and
// atomic-free (deterministic): workgroup (c, s) reduces the images of share s of channel c -- lanes along the
// contiguous pixels, no divisions -- and a second, tiny kernel combines the shares of a channel IN DOUBLE.  The
// variance is accumulated around a per-channel pivot K = z[0][c][0] (sum (z - K), sum (z - K)^2): with K within a few
// sigma of the mean nothing cancels, which a plain E[z^2] - E[z]^2 does not survive when |mean| >> sigma.
#include "ssn_common.h"

namespace {

constexpr int BN_SHARES_MAX = 32;

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// part[(c * S + s) * 2 + {0, 1}] = sum (z - K), sum (z - K)^2 over the images of share s
__global__ __launch_bounds__(256) void bn_stats_part_kernel(const float* z, float* part, int N, int HW, long img_stride,
                                                            int S) {
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    const int n0 = (int)((long)N * s / S), n1 = (int)((long)N * (s + 1) / S);
    const float K = z[(long)c * HW];
    float a1 = 0.f, a2 = 0.f;
    for (int n = n0; n < n1; ++n) {
        const float* row = z + (long)n * img_stride + (long)c * HW;
        for (int hw = tid; hw < HW; hw += 256) {
            const float d = row[hw] - K;
            a1 += d;
            a2 += d * d;
        }
    }
    const float s1 = block_sum_256(a1, red);
    const float s2 = block_sum_256(a2, red);
    if (tid == 0) {
        part[((long)c * S + s) * 2] = s1;
        part[((long)c * S + s) * 2 + 1] = s2;
    }
}
// mean / invstd of z per channel; running statistics as torch.nn.BatchNorm2d updates them (the running mean sees the
// conv bias the convolution left out)
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float* z, const float* part, const float* conv_bias,
                                                             float* mean, float* invstd, float* running_mean,
                                                             float* running_var, int C, int S, int HW, long count,
                                                             float eps, float momentum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < S; ++s) {
        s1 += (double)part[((long)c * S + s) * 2];
        s2 += (double)part[((long)c * S + s) * 2 + 1];
    }
    const double n = (double)count;
    const double m = (double)z[(long)c * HW] + s1 / n;
    double var = (s2 - s1 * s1 / n) / n;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double mb = m + (conv_bias ? (double)conv_bias[c] : 0.0);
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mb);
        const double unbiased = count > 1 ? var * n / (n - 1.0) : var;
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
}

// y = relu?(gamma * (z - mean) * invstd + beta), z and y channel slices
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* z, float* y, const float* mean, const float* invstd,
                                                       const float* gamma, const float* beta, int relu, long total,
                                                       long z_img_stride, long y_img_stride, FastDiv div_chw,
                                                       FastDiv div_hw, float* amax) {
    float vmax = 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        uint32_t n, rem, c, hw;
        fd_divmod((uint32_t)idx, div_chw, n, rem);
        fd_divmod(rem, div_hw, c, hw);
        const float xh = (z[(long)n * z_img_stride + rem] - mean[c]) * invstd[c];
        float v = gamma[c] * xh + beta[c];
        if (relu) v = fmaxf(v, 0.f);
        y[(long)n * y_img_stride + rem] = v;
        vmax = fmaxf(vmax, fabsf(v));
    }
    amax_emit(amax, vmax);
}

// part[(c * S + s) * 2 + {0, 1}] = sum g, sum g * xhat   (g = dy * (y > 0) when relu)
__global__ __launch_bounds__(256) void bn_bwd_part_kernel(const float* dy, const float* y, const float* z,
                                                          const float* mean, const float* invstd, float* part, int relu,
                                                          int N, int HW, long dy_img_stride, long y_img_stride,
                                                          long z_img_stride, int S) {
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    const int n0 = (int)((long)N * s / S), n1 = (int)((long)N * (s + 1) / S);
    const float m = mean[c], is = invstd[c];
    float a1 = 0.f, a2 = 0.f;
    for (int n = n0; n < n1; ++n) {
        const long o = (long)c * HW;
        const float* gr = dy + (long)n * dy_img_stride + o;
        const float* yr = y + (long)n * y_img_stride + o;
        const float* zr = z + (long)n * z_img_stride + o;
        for (int hw = tid; hw < HW; hw += 256) {
            const float g = (!relu || yr[hw] > 0.f) ? gr[hw] : 0.f;
            a1 += g;
            a2 += g * ((zr[hw] - m) * is);
        }
    }
    const float s1 = block_sum_256(a1, red);
    const float s2 = block_sum_256(a2, red);
    if (tid == 0) {
        part[((long)c * S + s) * 2] = s1;
        part[((long)c * S + s) * 2 + 1] = s2;
    }
}
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* part, float* dgamma, float* dbeta, int C, int S) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < S; ++s) {
        s1 += (double)part[((long)c * S + s) * 2];
        s2 += (double)part[((long)c * S + s) * 2 + 1];
    }
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
}
// dz = gamma * invstd * (g - dbeta / n - xhat * dgamma / n)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* dy, const float* y, const float* z,
                                                           const float* mean, const float* invstd, const float* gamma,
                                                           const float* dgamma, const float* dbeta, float* dz, int relu,
                                                           long total, float inv_count, long dy_img_stride,
                                                           long y_img_stride, long z_img_stride, long dz_img_stride,
                                                           FastDiv div_chw, FastDiv div_hw, float* amax) {
    float vmax = 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        uint32_t n, rem, c, hw;
        fd_divmod((uint32_t)idx, div_chw, n, rem);
        fd_divmod(rem, div_hw, c, hw);
        const float g = (!relu || y[(long)n * y_img_stride + rem] > 0.f) ? dy[(long)n * dy_img_stride + rem] : 0.f;
        const float xh = (z[(long)n * z_img_stride + rem] - mean[c]) * invstd[c];
        const float v = gamma[c] * invstd[c] * (g - dbeta[c] * inv_count - xh * (dgamma[c] * inv_count));
        dz[(long)n * dz_img_stride + rem] = v;
        vmax = fmaxf(vmax, fabsf(v));
    }
    amax_emit(amax, vmax);
}

inline unsigned bn_grid(long total, int cap = 8192) {
    long b = (total + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}
inline int bn_shares(int N) { return N < BN_SHARES_MAX ? (N < 1 ? 1 : N) : BN_SHARES_MAX; }

}  // namespace

// floats of workspace the two reductions below need (C channels, N images)
extern "C" long ssn_bn_train_workspace_floats(int N, int C) { return 2L * C * bn_shares(N); }

// Batch statistics of the channel slice z [N][C][HW] (image stride z_img_stride): mean[c], invstd[c] = 1 / sqrt(biased
// var + eps); running_mean / running_var (optional) are updated in place like torch.nn.BatchNorm2d(momentum) does, with
// conv_bias[c] (optional) added to the batch mean first -- z is the convolution WITHOUT its bias.
// Replaces the statistics half of F.batch_norm(training=True) behind model_zoo.BNInception.forward
// (/root/reference/ssn_models.py:266 with bn_mode 'partial' / 'full', :95-105,156-174).
extern "C" int ssn_bn_train_stats(const float* z, const float* conv_bias, float* mean, float* invstd, float* running_mean,
                                  float* running_var, int N, int C, int HW, long z_img_stride, float eps, float momentum,
                                  void* workspace, size_t ws_bytes, hipStream_t stream) {
    SSN_CHECK_ARG(z && mean && invstd && workspace && N >= 1 && C >= 1 && HW >= 1, "bn_train_stats: bad arguments");
    SSN_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_train_stats: running_mean / running_var go together");
    const int S = bn_shares(N);
    if (ws_bytes < (size_t)ssn_bn_train_workspace_floats(N, C) * sizeof(float)) {
        ssn_set_error("bn_train_stats: workspace %zu < %zu bytes", ws_bytes,
                      (size_t)ssn_bn_train_workspace_floats(N, C) * sizeof(float));
        return SSN_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(bn_stats_part_kernel, dim3((unsigned)C, (unsigned)S), dim3(256), 0, stream, z, (float*)workspace, N,
                       HW, z_img_stride, S);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream, z,
                       (const float*)workspace, conv_bias, mean, invstd, running_mean, running_var, C, S, HW,
                       (long)N * HW, eps, momentum);
    SSN_CHECK_LAUNCH("bn_train_stats");
    return SSN_OK;
}

// y = relu?(gamma * (z - mean) * invstd + beta); y_amax: amax slot of y's tensor or NULL.
extern "C" int ssn_bn_train_apply(const float* z, float* y, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, int relu, int N, int C, int HW, long z_img_stride, long y_img_stride,
                                  float* y_amax, hipStream_t stream) {
    SSN_CHECK_ARG(z && y && mean && invstd && gamma && beta, "bn_train_apply: null pointer");
    const long total = (long)N * C * HW;
    SSN_CHECK_ARG(total < (1l << 31), "bn_train_apply: tensor too large");
    hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_grid(total)), dim3(256), 0, stream, z, y, mean, invstd, gamma, beta, relu,
                       total, z_img_stride, y_img_stride, make_fastdiv((uint32_t)(C * HW)), make_fastdiv((uint32_t)HW),
                       y_amax);
    SSN_CHECK_LAUNCH("bn_train_apply");
    return SSN_OK;
}

// Backward of the same layer (autograd of F.batch_norm(training=True) + ReLU, entered from loss.backward(),
// /root/reference/ssn_train.py:236): dgamma[c], dbeta[c], and dz (may alias nothing of dy / y / z).
extern "C" int ssn_bn_train_bwd(const float* dy, const float* y, const float* z, const float* mean, const float* invstd,
                                const float* gamma, float* dgamma, float* dbeta, float* dz, int relu, int N, int C, int HW,
                                long dy_img_stride, long y_img_stride, long z_img_stride, long dz_img_stride,
                                void* workspace, size_t ws_bytes, float* dz_amax, hipStream_t stream) {
    SSN_CHECK_ARG(dy && y && z && mean && invstd && gamma && dgamma && dbeta && dz && workspace, "bn_train_bwd: null pointer");
    const long total = (long)N * C * HW;
    SSN_CHECK_ARG(total < (1l << 31), "bn_train_bwd: tensor too large");
    const int S = bn_shares(N);
    if (ws_bytes < (size_t)ssn_bn_train_workspace_floats(N, C) * sizeof(float)) {
        ssn_set_error("bn_train_bwd: workspace %zu < %zu bytes", ws_bytes,
                      (size_t)ssn_bn_train_workspace_floats(N, C) * sizeof(float));
        return SSN_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(bn_bwd_part_kernel, dim3((unsigned)C, (unsigned)S), dim3(256), 0, stream, dy, y, z, mean, invstd,
                       (float*)workspace, relu, N, HW, dy_img_stride, y_img_stride, z_img_stride, S);
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, stream,
                       (const float*)workspace, dgamma, dbeta, C, S);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bn_grid(total)), dim3(256), 0, stream, dy, y, z, mean, invstd, gamma,
                       (const float*)dgamma, (const float*)dbeta, dz, relu, total, 1.f / (float)((long)N * HW),
                       dy_img_stride, y_img_stride, z_img_stride, dz_img_stride, make_fastdiv((uint32_t)(C * HW)),
                       make_fastdiv((uint32_t)HW), dz_amax);
    SSN_CHECK_LAUNCH("bn_train_bwd");
    return SSN_OK;
}
